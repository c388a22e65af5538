"""AVS_F16P8 - the 3-byte storage format of the wide block outputs of ResNet layers 1-2 in the AVS_F16X2 trunk (fp16 hi
+ an 8-bit remainder in units of ulp(max(|hi|, 2^-6)) / 256; include/avsum_hip.h).  The format is restated here on the CPU
(``emu_p8_pack`` / ``emu_p8_unpack``) and the GPU conversions are checked against it bit for bit; the kernels that read
and write it are checked against the AVS_F16X2 kernels on operands both formats hold exactly:
  * avs_conv2d_nhwc_bnstats with an AVS_F16P8 input (A fragments fetched into registers, lo halves rebuilt there) gives
    the SAME BITS as the AVS_F16X2 input path - outputs and folded statistics;
  * avs_conv2d_nhwc_affine writing / reading AVS_F16P8 agrees with its AVS_F16X2 form to the format's resolution
    (2^-19 of a value's binade).
What the format does to the scores is measured in tools/h3_storage_study.py (nothing: 5.6e-6 either way) and asserted
end to end by tests/test_gpu_accuracy.py, whose f16x2 mode stores these tensors in it."""
import pytest
import torch

from test_gpu_f16x2 import emu_pack, emu_unpack

pytestmark = pytest.mark.gpu


def _ops():
    from avsum_amd import ops
    return ops


def _parts(x):
    xc = x.float().clamp(-65504.0, 65504.0)
    hi = xc.half()
    e = torch.frexp(hi.float())[1].clamp(min=-5)         # E + 1, floored: a step of 2^-24 (the fp16 grid) below 2^-6
    q = torch.round(torch.ldexp(xc - hi.float(), 19 - e)) + 128.0
    return hi, e, q.clamp(1, 255)


def emu_p8_pack(x):
    """fp32 [..., C] (C % 16 == 0) -> uint8 [rows, 3 C]: per 16 values the hi halves of 0-7, of 8-15, 16 remainder bytes."""
    c = x.shape[-1]
    assert c % 16 == 0
    hi, _, q = _parts(x.contiguous())
    hib = hi.reshape(-1, 16).view(torch.uint8).reshape(-1, 32)
    return torch.cat([hib, q.reshape(-1, 16).to(torch.uint8)], 1).reshape(-1, 3 * c).contiguous()


def emu_p8_unpack(b, shape):
    blk = b.reshape(-1, 48)
    hi = blk[:, :32].contiguous().view(torch.float16).float()
    e = torch.frexp(hi)[1].clamp(min=-5)
    e = torch.where(hi == 0, torch.full_like(e, -5), e)    # (zero counts as below 2^-6; the encoder stores u = 128 there)
    v = hi + torch.ldexp(blk[:, 32:].float() - 128.0, e - 19)
    return v.reshape(shape)


def test_pack_unpack_bit_exact(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    x = torch.cat([torch.randn(4096, generator=g) * s for s in (1.0, 1e-3, 1e-6, 300.0, 7e4)] +
                  [torch.tensor([0.0, -0.0, 65504.0, -65504.0, 1e9, -1e9, 6e-8, 2.0 ** -24, 1.0, 1.0 + 2.0 ** -11,
                                 1.0 - 2.0 ** -12, 2.0 ** -14, 3.0, -3.0, 0.5, 1e-8])]).reshape(-1, 16)
    ref = emu_p8_pack(x)
    got = ops.f16p8_pack(x.to(dev))
    assert got.data.shape == ref.shape and torch.equal(got.data.cpu(), ref)
    back = ops.f16p8_unpack(got).cpu()
    assert torch.equal(back, emu_p8_unpack(ref, x.shape))
    inside = x.abs() <= 65504
    # 19-20 significant bits: half a step = ulp(hi) / 512 <= |x| 2^-19 (a whole step where the remainder is clamped: a
    # near-tie of the fp16 rounding); the same absolute floor as AVS_F16X2 (2^-25)
    err = (back - x).abs()[inside]
    assert (err <= x.abs()[inside] * 2.0 ** -18 + 2.0 ** -25).all()
    assert (err <= x.abs()[inside] * 2.0 ** -19 + 2.0 ** -25).float().mean().item() > 0.99
    # a value the format holds is a fixed point, and AVS_F16X2 holds it too
    again = ops.f16p8_unpack(ops.f16p8_pack(back.to(dev))).cpu()
    assert torch.equal(again, back)
    assert torch.equal(emu_unpack(emu_pack(back)), back)


@pytest.mark.parametrize("frames,hw,cin,cout,gf", [(12, 56, 256, 64, 1), (9, 28, 512, 128, 1), (8, 28, 512, 128, 4),
                                                   (5, 14, 256, 64, 5), (3, 20, 64, 256, 1)])
@pytest.mark.parametrize("layout,tile", [(0, 2), (1, 2), (1, 1)])
def test_conv_bnstats_p8_input_same_bits(dev, frames, hw, cin, cout, gf, layout, tile):
    """conv1 of a bottleneck reading an AVS_F16P8 block output: the same outputs and statistics, bit for bit, as the
    AVS_F16X2 path on the same values (same products, same order; only the A operand's way into the registers differs)."""
    ops = _ops()
    from avsum_amd import _abi
    code = ops.dtype_code(torch.float32, "f16x2")
    g = torch.Generator().manual_seed(frames + hw + cin)
    x0 = torch.relu(torch.randn(frames, hw, hw, cin, generator=g) * 1.5 + 0.3)
    # (from 2^-6 up a stored value has ONE split into hi + remainder, the split AVS_F16X2 makes; below, where the step is
    #  the fp16 grid itself, a remainder of exactly half an ulp of hi - a tie of the fp16 rounding - holds the same VALUE
    #  as the other split: not the same operand bits, so those values are kept out of the bit-for-bit comparison)
    x0 = torch.where(x0 < 2.0 ** -6, torch.zeros_like(x0), x0)
    xb = emu_p8_pack(x0)
    xq = emu_p8_unpack(xb, x0.shape)                         # what both formats hold exactly
    assert torch.equal(emu_unpack(emu_pack(xq)), xq)
    wp = emu_pack(torch.randn(cout, cin, generator=g) / cin ** 0.5).to(dev)
    wsel = ops.weights_kstep32(wp) if layout else wp
    gamma, beta = (torch.rand(cout, generator=g) + 0.5).to(dev), torch.randn(cout, generator=g).to(dev)
    rpg = gf * hw * hw
    geom = (frames, hw, hw, cin, 1, 1, 1, 1, 0, 0, hw, hw, cout)
    xs = (hw * hw * cin, hw * cin, cin)
    x2 = emu_pack(xq).to(dev)
    x8 = ops.P8(xb.to(dev), x0.shape)
    y2 = torch.empty((frames, hw, hw, cout), device=dev)
    y8 = torch.empty_like(y2)
    a2 = ops.conv2d_raw(code, *geom, x2, *xs, wsel, cin, y2, cout, bnstats=(rpg, gamma, beta, 1e-5), w_layout=layout,
                        variant=tile)   # _abi.TILE_256 / TILE_128: the statistics are summed per tile
    a8 = ops.conv2d_raw(code, *geom, x8, *xs, wsel, cin, y8, cout, bnstats=(rpg, gamma, beta, 1e-5), w_layout=layout,
                        variant=tile)
    assert a2 is not None and a8 is not None
    assert torch.equal(y8.view(torch.int32), y2.view(torch.int32))
    assert torch.equal(a8[0], a2[0]) and torch.equal(a8[1], a2[1])
    # and it is the convolution: float64 on the stored values
    ref = xq.double().reshape(-1, cin) @ emu_unpack(wp.cpu()).double().t()
    got = ops.f16x2_unpack(y8).cpu().double().reshape(-1, cout)
    assert (got - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())


def test_p8_input_is_refused_where_no_kernel_reads_it(dev):
    ops = _ops()
    code = ops.dtype_code(torch.float32, "f16x2")
    x8 = ops.f16p8_pack(torch.randn(2, 8, 8, 48).to(dev))       # cin = 48: not a multiple of 32
    wp = emu_pack(torch.randn(64, 48)).to(dev)
    y = torch.empty((2, 8, 8, 64), device=dev)
    with pytest.raises(Exception):
        ops.conv2d_raw(code, 2, 8, 8, 48, 1, 1, 1, 1, 0, 0, 8, 8, 64, x8, 8 * 8 * 48, 8 * 48, 48, wp, 48, y, 64,
                       bnstats=(64, torch.ones(64, device=dev), torch.zeros(64, device=dev), 1e-5))
    with pytest.raises(ValueError):   # the plain convolution has no such form
        ops.conv2d_raw(code, 2, 8, 8, 48, 1, 1, 1, 1, 0, 0, 8, 8, 64, x8, 8 * 8 * 48, 8 * 48, 48, wp, 48, y, 64)


@pytest.mark.parametrize("rpg,k,n,res,variant", [(3136, 64, 256, "p8", 0), (784, 128, 512, "p8", 0), (784, 128, 512, "h2", 2),
                                                 (200, 64, 256, "p8", 1), (3136, 64, 256, None, 0), (100, 128, 128, "p8", 2)])
def test_affine_pass_p8_output_and_residual(dev, rpg, k, n, res, variant):
    """avs_conv2d_nhwc_affine storing its output as AVS_F16P8 and reading an AVS_F16P8 residual: the stored values agree
    with the AVS_F16X2 form's (same accumulators, same fp32 epilogue) to the format's resolution."""
    ops = _ops()
    code = ops.dtype_code(torch.float32, "f16x2")
    groups = 3
    rows = groups * rpg - 7                                   # a shorter last group, a ragged last tile
    g = torch.Generator().manual_seed(rpg + k + n)
    xp = emu_pack(torch.relu(torch.randn(rows, k, generator=g))).to(dev)
    wp = emu_pack(torch.randn(n, k, generator=g) / k ** 0.5).to(dev)
    sc = (torch.rand(groups, n, generator=g) + 0.5).to(dev)
    sh = torch.randn(groups, n, generator=g).to(dev)
    r0 = torch.relu(torch.randn(rows, n, generator=g) * 2)
    rb = emu_p8_pack(r0)
    rq = emu_p8_unpack(rb, r0.shape)
    res2 = emu_pack(rq).to(dev) if res else None
    res8 = None if res is None else (ops.P8(rb.to(dev), (rows, n)) if res == "p8" else res2)
    y2 = torch.empty((rows, n), device=dev)
    ops.conv2d_affine(code, rows, 1, 1, k, 1, 1, 1, 1, n, xp, k, k, k, wp, k, y2, n, rpg, sc, sh, res2, True, None,
                      variant=variant)
    outs = []
    for _ in range(2):
        y8 = ops.P8.empty((rows, n), dev)
        y8.data.fill_(0x5a)
        ops.conv2d_affine(code, rows, 1, 1, k, 1, 1, 1, 1, n, xp, k, k, k, wp, k, y8, n, rpg, sc, sh, res8, True, None,
                          variant=variant)
        outs.append(y8)
    assert torch.equal(outs[0].data, outs[1].data)
    v2 = ops.f16x2_unpack(y2).cpu()
    v8 = ops.f16p8_unpack(outs[0]).cpu()
    assert (v8 >= 0).all() and (v8 - v2).abs().max().item() > 0          # (it IS another rounding)
    assert ((v8 - v2).abs() <= v2.abs() * 2.0 ** -18 + 2.0 ** -24).all()
    # the stored values are fixed points of the format
    assert torch.equal(emu_p8_unpack(emu_p8_pack(v8), v8.shape), v8)


def test_trunk_stores_the_inner_block_outputs_in_p8(dev):
    """The AVS_F16X2 trunk with and without the 3-byte storage of the inner block outputs of layers 1-2: features within
    1e-4 of each other relative to the largest (the format keeps 19-20 bits), and the format is really in use."""
    import numpy as np
    from avsum_amd import ops
    from avsum_amd.cnn import ResNet50Runner, resnet50_trunk
    torch.manual_seed(3)
    trunk = resnet50_trunk().to(dev)
    frames = torch.from_numpy(np.random.default_rng(2).integers(0, 256, (6, 224, 224, 3), dtype=np.uint8)).to(dev)
    r8 = ResNet50Runner(trunk, torch.float32, f32_split="f16x2")
    assert r8.p8_blocks == (0, 1, 3, 4, 5)
    made = []
    orig = ops.P8.empty.__func__

    def spy(cls, shape, device):
        made.append(tuple(shape))
        return orig(cls, shape, device)

    ops.P8.empty = classmethod(spy)
    try:
        f8 = r8.forward(frames, [0, 1, 2, 3, 4, 5, 6]).cpu()
    finally:
        ops.P8.empty = classmethod(orig)
    assert made == [(6, 56, 56, 256)] * 2 + [(6, 28, 28, 512)] * 3
    r2 = ResNet50Runner(trunk, torch.float32, f32_split="f16x2")
    r2.p8_blocks = ()
    f2 = r2.forward(frames, [0, 1, 2, 3, 4, 5, 6]).cpu()
    # (53 batch-normalised layers amplify a 2^-19 perturbation ~20x: measured 3.9e-5; AVS_F16X2 itself sits 1-2.5e-4 from
    #  the fp32 oracle by the same conditioning, tests/test_gpu_f16x2.py)
    assert (f8 - f2).abs().max().item() <= 1e-4 * f2.abs().max().item()
    assert torch.equal(r8.forward(frames, [0, 1, 2, 3, 4, 5, 6]).cpu(), f8)     # deterministic
    # ragged groups fall back to 4-byte storage (the forms that read the format need equal groups)
    made.clear()
    r8.forward(frames, [0, 4, 6])
