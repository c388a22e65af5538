"""GPU parity tests of the individual C-ABI kernels against the oracle / plain torch fp32 CPU ops.
Every call goes through libavsum_hip.so (avsum_amd.ops -> ctypes)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    from avsum_amd import ops
    return ops


def _lib():
    from avsum_amd import _abi
    return _abi.lib()


def rel_err(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


# ----------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("m,n,k", [(128, 128, 16), (1, 1, 4), (37, 70, 52), (300, 512, 4096), (257, 64, 296),
                                   (1000, 402, 400), (129, 3072, 1024)])
@pytest.mark.parametrize("act", [0, 1])
def test_linear_f32(dev, m, n, k, act):
    ops = _ops()
    g = torch.Generator().manual_seed(m * 7 + n * 3 + k)
    x = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    b = torch.randn(n, generator=g)
    ref = F.linear(x, w, b)
    if act:
        ref = torch.relu(ref)
    out = ops.linear(x.to(dev), w.to(dev), b.to(dev), act).cpu()
    # fp32 MFMA is an exact fmaf chain: only the summation order differs from the CPU BLAS
    assert (out - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


def test_linear_f32_exact_integers(dev):
    """A = I against an ASYMMETRIC B on exact integer data: catches row/col swaps and k-permutation bugs."""
    ops = _ops()
    n = 160
    x = torch.eye(n)
    w = (torch.arange(n * n).reshape(n, n) % 251).float()  # asymmetric
    out = ops.linear(x.to(dev), w.to(dev)).cpu()
    assert torch.equal(out, w.t().contiguous())
    xr = torch.randint(-8, 9, (200, 96)).float()
    wr = torch.randint(-8, 9, (70, 96)).float()
    assert torch.equal(ops.linear(xr.to(dev), wr.to(dev)).cpu(), xr @ wr.t())


@pytest.mark.parametrize("m,n,k", [(128, 128, 32), (200, 70, 96), (513, 256, 576), (64, 2048, 512)])
def test_linear_bf16(dev, m, n, k):
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(m, k, generator=g).bfloat16()
    w = (torch.randn(n, k, generator=g) / k ** 0.5).bfloat16()
    b = torch.randn(n, generator=g)
    ref = x.float() @ w.float().t() + b
    out = ops.linear(x.to(dev), w.to(dev), b.to(dev)).cpu()
    assert out.dtype == torch.bfloat16
    assert (out.float() - ref).abs().max().item() <= 1.2e-2 * max(1.0, ref.abs().max().item())
    xi = torch.randint(-4, 5, (m, k)).bfloat16()
    wi = torch.randint(-4, 5, (n, k)).bfloat16()
    oi = ops.linear(xi.to(dev), wi.to(dev)).cpu().float()
    ri = (xi.float() @ wi.float().t()).bfloat16().float()
    assert torch.equal(oi, ri)


def test_gemm_f32_acc64_slices(dev):
    """fp64 accumulation of 32-element fp32 slices: closer to the exact product than a plain fp32 running sum."""
    ops = _ops()
    from avsum_amd._abi import AVS_F32_ACC64
    g = torch.Generator().manual_seed(4)
    m, n, k = 300, 402, 400
    a = torch.randn(m, k, generator=g) * 10 + 30  # large same-sign terms: long running sums lose bits
    b = torch.randn(n, k, generator=g)
    exact = a.double() @ b.double().t()
    c64 = torch.empty(m, n, device=dev)
    ops.gemm_nt_batched(AVS_F32_ACC64, m, n, k, a.to(dev), 0, k, 0, b.to(dev), 0, k, 0, c64, 0, n, 0)
    c32 = ops.linear(a.to(dev), b.to(dev)).cpu()
    e64 = (c64.cpu().double() - exact).abs().max().item()
    e32 = (c32.double() - exact).abs().max().item()
    assert e64 <= 8.0 * exact.abs().max().item() * 2 ** -24
    assert e64 < 0.6 * e32


def test_gemm_batched_strided(dev):
    ops = _ops()
    from avsum_amd._abi import BIAS_ROW
    g = torch.Generator().manual_seed(11)
    bsz, m, n, k = 3, 50, 45, 24
    a = torch.randn(bsz, m, k, generator=g)
    b = torch.randn(bsz, n, k, generator=g)
    bias = torch.randn(m, generator=g)
    ldc = 48
    c = torch.zeros(bsz, m, ldc)
    cd = c.to(dev)
    ops.gemm_nt_batched(0, m, n, k, a.to(dev), 0, k, m * k, b.to(dev), 0, k, n * k, cd, 0, ldc, m * ldc,
                        bias.to(dev), BIAS_ROW, 0, 0.5, 0, bsz)
    ref = 0.5 * torch.einsum("bmk,bnk->bmn", a, b) + bias[None, :, None]
    out = cd.cpu()
    assert (out[:, :, :n] - ref).abs().max().item() < 1e-5
    assert out[:, :, n:].abs().max().item() == 0.0


def test_abi_rejects_bad_arguments(dev):
    ops = _ops()
    from avsum_amd._abi import AvsError
    x = torch.randn(8, 6, device=dev)  # K=6 is not a multiple of 4
    w = torch.randn(4, 6, device=dev)
    with pytest.raises(AvsError):
        ops.linear(x, w)
    with pytest.raises(ValueError):
        ops.linear(torch.randn(8, 8), torch.randn(4, 8))  # host tensors: no CPU fallback


# ----------------------------------------------------------------------------- conv
def _conv_case(dev, dtype, n, h, w, cin, cout, kh, kw, stride, pad, tol, split=False, variant=0):
    ops = _ops()
    g = torch.Generator().manual_seed(h * 13 + cin)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, kh, kw, generator=g) / (cin * kh * kw) ** 0.5
    b = torch.randn(cout, generator=g)
    if dtype == torch.bfloat16:
        x, wt = x.bfloat16().float(), wt.bfloat16().float()
    ref = torch.relu(F.conv2d(x, wt, b, stride, pad))
    ho, wo = ref.shape[2], ref.shape[3]
    xn = x.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev)
    wk = wt.permute(0, 2, 3, 1).reshape(cout, -1).contiguous().to(dtype).to(dev)
    out = torch.empty((n, ho, wo, cout), dtype=dtype, device=dev)
    ops.conv2d(xn, wk, kh, kw, stride, pad, out, b.to(dev), 1, split=split, variant=variant)
    got = out.float().cpu().permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("cfg", [
    (2, 14, 14, 64, 64, 1, 1, 1, 0), (2, 14, 14, 64, 96, 3, 3, 1, 1), (3, 15, 15, 32, 48, 3, 3, 2, 1),
    (2, 16, 16, 128, 256, 1, 1, 2, 0), (1, 17, 17, 128, 192, 1, 7, 1, (0, 3)), (1, 17, 17, 128, 192, 7, 1, 1, (3, 0)),
    (2, 12, 12, 48, 64, 5, 5, 1, 2), (2, 9, 9, 80, 192, 3, 3, 1, 0), (1, 35, 35, 288, 384, 3, 3, 2, 0),
])
def test_conv2d_f32(dev, cfg):
    _conv_case(dev, torch.float32, *cfg, tol=2e-5)


@pytest.mark.parametrize("cfg", [
    (2, 14, 14, 64, 64, 1, 1, 1, 0), (2, 14, 14, 64, 96, 3, 3, 1, 1), (3, 15, 15, 32, 48, 3, 3, 2, 1),
    (2, 16, 16, 128, 256, 1, 1, 2, 0), (1, 17, 17, 128, 192, 1, 7, 1, (0, 3)), (2, 12, 12, 48, 64, 5, 5, 1, 2),
    (2, 9, 9, 80, 192, 3, 3, 1, 0), (1, 35, 35, 288, 384, 3, 3, 2, 0), (2, 28, 28, 16, 32, 1, 1, 1, 0),
])
def test_conv2d_f32_split(dev, cfg):
    """AVS_F32_SPLIT: fp32 operands, products on the bf16 matrix cores as hi*hi + hi*lo + lo*hi (both 64- and
    128-byte-row variants, the pipelined and the plain loop): ~2^-15 relative per product, i.e. ~1e-4 of the exact
    fp32 result's scale in the worst element, against 2e-5 for the exact fp32 mode and 1.2e-2 for bf16."""
    _conv_case(dev, torch.float32, *cfg, tol=1e-4, split=True)
    # the 256-row tiles (4 x 1 waves) forced on: same products in the same k order per output element => the same
    # values as the 128-row tiles, also with the fused BatchNorm statistics
    from avsum_amd import _abi
    ops = _ops()
    n, h, w, cin, cout, kh, kw, stride, pad = cfg
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(n, h, w, cin, generator=g).to(dev)
    wk = (torch.randn(cout, kh * kw * cin, generator=g) / (kh * kw * cin) ** 0.5).to(dev)
    ph, pw = pad if isinstance(pad, tuple) else (pad, pad)
    ho, wo = (h + 2 * ph - kh) // stride + 1, (w + 2 * pw - kw) // stride + 1
    gamma, beta = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    rpg = max(64, ho * wo)
    _conv_case(dev, torch.float32, *cfg, tol=1e-4, split=True, variant=_abi.TILE_256)
    tall = torch.empty((n, ho, wo, cout), device=dev)
    sc_t, sh_t = ops.conv2d(x, wk, kh, kw, stride, pad, tall, bnstats=(rpg, gamma, beta, 1e-5), split=True,
                            variant=_abi.TILE_256)
    base = torch.empty_like(tall)
    sc_b, sh_b = ops.conv2d(x, wk, kh, kw, stride, pad, base, bnstats=(rpg, gamma, beta, 1e-5), split=True,
                            variant=_abi.TILE_128)
    assert torch.equal(tall, base)
    assert (sc_t - sc_b).abs().max().item() < 1e-4 * sc_b.abs().max().item()
    assert (sh_t - sh_b).abs().max().item() < 1e-4 * max(1.0, sh_b.abs().max().item())


@pytest.mark.parametrize("cfg", [(2, 14, 14, 64, 64, 3, 3, 1, 1), (2, 16, 16, 128, 256, 1, 1, 2, 0),
                                 (1, 17, 17, 128, 192, 1, 7, 1, (0, 3))])
def test_conv2d_bf16(dev, cfg):
    _conv_case(dev, torch.bfloat16, *cfg, tol=1.2e-2)


@pytest.mark.parametrize("cfg", [(2, 14, 14, 64, 64, 3, 3, 1, 1), (3, 16, 16, 128, 256, 1, 1, 2, 0),
                                 (1, 17, 17, 128, 192, 1, 7, 1, (0, 3)), (5, 13, 13, 96, 64, 3, 3, 1, 1),
                                 (1, 35, 35, 288, 384, 3, 3, 2, 0), (2, 9, 9, 80, 200, 3, 3, 1, 0)])
def test_conv2d_bf16_tall_tiles(dev, cfg):
    """The 256-row tile variants (4 x 1 waves) forced on: same results as the 128-row tiles; also with the fused
    BatchNorm statistics and on a plain (no bias) convolution, ragged last tile included."""
    from avsum_amd import _abi
    ops = _ops()
    _conv_case(dev, torch.bfloat16, *cfg, tol=1.2e-2, variant=_abi.TILE_256)
    n, h, w, cin, cout, kh, kw, stride, pad = cfg
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(n, h, w, cin, generator=g).bfloat16().to(dev)
    wk = (torch.randn(cout, kh * kw * cin, generator=g) / (kh * kw * cin) ** 0.5).bfloat16().to(dev)
    ph, pw = pad if isinstance(pad, tuple) else (pad, pad)
    ho, wo = (h + 2 * ph - kh) // stride + 1, (w + 2 * pw - kw) // stride + 1
    gamma, beta = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    tall = torch.empty((n, ho, wo, cout), dtype=torch.bfloat16, device=dev)
    rpg = max(64, ho * wo)  # groups of fewer than 64 rows are declined by the fused-statistics form
    sc_t, sh_t = ops.conv2d(x, wk, kh, kw, stride, pad, tall, bnstats=(rpg, gamma, beta, 1e-5), variant=_abi.TILE_256)
    base = torch.empty_like(tall)
    sc_b, sh_b = ops.conv2d(x, wk, kh, kw, stride, pad, base, bnstats=(rpg, gamma, beta, 1e-5), variant=_abi.TILE_128)
    assert torch.equal(tall, base)          # same products, same k order per output element
    assert (sc_t - sc_b).abs().max().item() < 2e-3 * sc_b.abs().max().item()
    assert (sh_t - sh_b).abs().max().item() < 2e-3 * max(1.0, sh_b.abs().max().item())


@pytest.mark.parametrize("cfg", [(3, 14, 14, 64, 96, 3, 1, 1), (2, 9, 11, 32, 40, 1, 1, 0), (2, 12, 12, 128, 64, 3, 2, 1),
                                 (70, 14, 14, 256, 256, 3, 1, 1), (2, 8, 8, 1024, 136, 1, 1, 0), (1, 7, 7, 96, 24, 5, 1, 2)])
def test_conv2d_bf16_kstep_weight_layout(dev, cfg):
    """AVS_W_KSTEP32: the same weight matrix stored reduction-step major gives bit-identical outputs (every staging
    form: scalar tap walk / general gather, 64- and 128-byte steps, 128- and 256-row tiles); fp32 and reductions
    that are not multiples of 32 are refused."""
    ops = _ops()
    n, h, w_, cin, cout, k, s, p = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x = torch.randn(n, h, w_, cin, generator=g).bfloat16().to(dev)
    wt = (torch.randn(cout, k * k * cin, generator=g) / (k * k * cin) ** 0.5).bfloat16().to(dev)
    ho, wo = (h + 2 * p - k) // s + 1, (w_ + 2 * p - k) // s + 1
    bias = torch.randn(cout, generator=g).to(dev)
    ref = ops.conv2d(x, wt, k, k, s, p, torch.empty((n, ho, wo, cout), dtype=torch.bfloat16, device=dev), bias, ops.ACT_RELU)
    wk = ops.weights_kstep32(wt)
    assert wk.shape == wt.shape and (k * k * cin == 32 or not torch.equal(wk, wt))   # one step: the layouts coincide
    for tall in (1, 2):   # _abi.TILE_128, _abi.TILE_256: the per-call tile override of the descriptor
        got = ops.conv2d(x, wk, k, k, s, p, torch.empty_like(ref), bias, ops.ACT_RELU, w_layout=1, variant=tall)
        want = ops.conv2d(x, wt, k, k, s, p, torch.empty_like(ref), bias, ops.ACT_RELU, variant=tall)
        assert torch.equal(got, want)
    assert torch.equal(want, ref) or (want.float() - ref.float()).abs().max().item() < 0.05
    # fp32 (16-element steps), exact and split arithmetic
    x32, w32 = x.float(), wt.float()
    for split in (False, True):
        a = ops.conv2d(x32, w32, k, k, s, p, torch.empty_like(ref).float(), bias, ops.ACT_RELU, split=split)
        b = ops.conv2d(x32, ops.weights_kstep32(w32), k, k, s, p, torch.empty_like(ref).float(), bias, ops.ACT_RELU,
                       split=split, w_layout=1)
        assert torch.equal(a, b)
    with pytest.raises(ValueError):
        ops.weights_kstep32(wt[:, :24])
    with pytest.raises(RuntimeError):     # the layout flag on a reduction that is not whole steps
        ops.conv2d(x[..., :24].contiguous(), wt[:, :24 * k * k].contiguous(), k, k, s, p, torch.empty_like(ref), w_layout=1)


def test_conv2d_seeded_shape_sweep(dev):
    """Seeded sweep over convolution shapes against torch: channel counts on both sides of the fast-staging rule
    (cin a multiple of one reduction step or not), 1..49 taps (more than 32 taps use the general staging), strides,
    asymmetric kernels / padding, ragged tiles, both dtypes, with and without the 256-row tiles."""
    from avsum_amd import _abi
    rng = np.random.default_rng(20261004)
    kernels = [(1, 1), (3, 3), (5, 5), (7, 7), (1, 7), (7, 1), (3, 1), (2, 2)]
    cases = []
    for _ in range(36):
        kh, kw = kernels[rng.integers(len(kernels))]
        stride = int(rng.integers(1, 3))
        ph, pw = int(rng.integers(0, kh // 2 + 1)), int(rng.integers(0, kw // 2 + 1))
        h, w = int(rng.integers(max(kh, 5), 23)), int(rng.integers(max(kw, 5), 23))
        cin = int(rng.choice([8, 16, 24, 32, 48, 64, 96, 128, 160]))
        cout = int(rng.choice([8, 24, 64, 72, 128, 136, 200, 256]))
        n = int(rng.integers(1, 6))
        cases.append((n, h, w, cin, cout, kh, kw, stride, (ph, pw)))
    for i, cfg in enumerate(cases):
        dtype, tol = (torch.bfloat16, 1.2e-2) if i % 3 else (torch.float32, 2e-5)
        variant = (_abi.TILE_256 if i % 2 else _abi.TILE_128) | (_abi.STAGING_GENERIC if i % 5 == 4 else 0)
        _conv_case(dev, dtype, *cfg, tol=tol, variant=variant)


def test_conv2d_channel_slice_output(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 8, 8, 32, generator=g)
    w1 = torch.randn(16, 32, generator=g)
    w2 = torch.randn(24, 32, generator=g)
    buf = torch.zeros(2, 8, 8, 40, device=dev)
    ops.conv2d(x.to(dev), w1.to(dev), 1, 1, 1, 0, buf[..., :16])
    ops.conv2d(x.to(dev), w2.to(dev), 1, 1, 1, 0, buf[..., 16:])
    ref = torch.cat([x @ w1.t(), x @ w2.t()], -1)
    assert (buf.cpu() - ref).abs().max().item() < 1e-4


# ----------------------------------------------------------------------------- visual front end
def test_normalize_and_resize(dev):
    ops = _ops()
    from oracle import cnn as ocnn
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 256, (3, 224, 224, 3), dtype=np.uint8)
    d = torch.from_numpy(frames).to(dev)
    x = ops.frames_normalize(d, torch.float32, 1.0, ocnn.MEAN.flatten().tolist(), ocnn.STD.flatten().tolist(), 230, 232,
                             3, 3).cpu()
    ref = torch.cat([ocnn.preprocess_frame(f) for f in frames]).permute(0, 2, 3, 1)
    assert torch.equal(x[:, 3:227, 3:227, :3], ref)  # bit exact: same IEEE ops
    assert x[:, :3].abs().max() == 0 and x[:, 227:].abs().max() == 0 and x[..., 3].abs().max() == 0
    big = ops.resize_bilinear(d, 299, 299).cpu().numpy()
    for i in range(3):
        assert np.array_equal(big[i], ocnn.cv_resize_linear_u8(frames[i], 299, 299))
    small = rng.integers(0, 256, (2, 360, 480, 3), dtype=np.uint8)
    got = ops.resize_bilinear(torch.from_numpy(small).to(dev), 224, 224).cpu().numpy()
    for i in range(2):
        assert np.array_equal(got[i], ocnn.cv_resize_linear_u8(small[i], 224, 224))
    xi = ops.frames_normalize(torch.from_numpy(big).to(dev), torch.float32, 255.0, ocnn.MEAN.flatten().tolist(),
                              ocnn.STD.flatten().tolist(), 299, 300, 0, 0).cpu()
    refi = torch.cat([ocnn.preprocess_inception(f) for f in frames]).permute(0, 2, 3, 1)
    assert torch.equal(xi[:, :, :299, :3], refi)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bn_batch_stats_apply(dev, dtype):
    ops = _ops()
    from oracle import cnn as ocnn
    g = torch.Generator().manual_seed(2)
    hw, c = 49, 128
    sizes = [4, 4, 3, 1]
    n = sum(sizes)
    x = (torch.randn(n, c, 7, 7, generator=g) * 3 + 50).to(dtype).float()  # mean >> std: cancellation test
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    res = torch.randn(n, c, 7, 7, generator=g).to(dtype).float()
    refs, o = [], 0
    for s in sizes:
        refs.append(torch.relu(ocnn.bn_batch(x[o:o + s], gamma, beta) + res[o:o + s]))
        o += s
    ref = torch.cat(refs).permute(0, 2, 3, 1).reshape(-1, c)
    grow = torch.tensor(np.cumsum([0] + sizes) * hw, dtype=torch.int64, device=dev)
    x2 = x.permute(0, 2, 3, 1).reshape(-1, c).contiguous().to(dtype).to(dev)
    r2 = res.permute(0, 2, 3, 1).reshape(-1, c).contiguous().to(dtype).to(dev)
    scale, shift = ops.bn_batch_stats(x2, grow, gamma.to(dev), beta.to(dev), 1e-5)
    y = ops.bn_apply(x2, scale, shift, grow, 4 * hw, r2, 1).float().cpu()
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    assert (y - ref).abs().max().item() < tol


def test_pools_and_segment_mean(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 16, 15, 15, generator=g)
    xn = x.permute(0, 2, 3, 1).contiguous().to(dev)
    for mode, k, s, p in (("max", 3, 2, 1), ("max", 3, 2, 0), ("avg", 3, 1, 1)):
        ref = F.max_pool2d(x, k, s, p) if mode == "max" else F.avg_pool2d(x, k, s, p)
        out = torch.empty((2, ref.shape[2], ref.shape[3], 16), device=dev)
        ops.pool2d(xn, mode, k, s, p, out)
        assert (out.cpu().permute(0, 3, 1, 2) - ref).abs().max().item() < 1e-6
    # act(pool(x) + bias) into a channel slice of a wider buffer (Inception's branch_pool after its 1x1 convolution)
    bias = torch.randn(16, generator=g)
    wide = torch.zeros((2, 15, 15, 40), device=dev)
    ops.pool2d(xn, "avg", 3, 1, 1, wide[..., 8:24], bias.to(dev), ops.ACT_RELU)
    ref = torch.relu(F.avg_pool2d(x, 3, 1, 1) + bias.view(1, -1, 1, 1))
    assert (wide[..., 8:24].cpu().permute(0, 3, 1, 2) - ref).abs().max().item() < 1e-6
    assert wide[..., :8].abs().max().item() == 0 and wide[..., 24:].abs().max().item() == 0
    gap = ops.global_avgpool(xn).cpu()
    assert (gap - x.mean((2, 3))).abs().max().item() < 1e-6
    feats = torch.randn(9, 32, generator=g)
    seg = torch.tensor([0, 4, 4, 9], dtype=torch.int64)
    out = ops.segment_mean(feats.to(dev), seg.to(dev)).cpu().numpy()
    ref = np.stack([feats[0:4].numpy().mean(axis=0), np.zeros(32, np.float32), feats[4:9].numpy().mean(axis=0)])
    assert np.array_equal(out, ref)  # same sequential float32 sum as numpy's axis-0 mean


# ----------------------------------------------------------------------------- audio
def _wave(kind, t, seed=0, noise=0.05):
    g = torch.Generator().manual_seed(seed)
    ts = torch.arange(t) / 16000.0
    if kind == "sine":
        return 0.5 * torch.sin(2 * np.pi * 440.0 * ts)
    return (0.4 * torch.sin(2 * np.pi * 220 * ts) + 0.3 * torch.sin(2 * np.pi * 1333 * ts)
            + 0.2 * torch.sin(2 * np.pi * 5200 * ts) + noise * torch.randn(t, generator=g))


def test_stft_f64_exact_integers(dev):
    """fp64 MFMA lane layout check on exact integer data with an ASYMMETRIC basis (a wrong row/column map of
    the 16x16x4 f64 tile would silently permute the spectrum)."""
    ops = _ops()
    rng = np.random.default_rng(3)
    frames, hop, nfft, ncols = 75, 12, 40, 70
    x = rng.integers(-9, 10, (frames - 1) * hop + nfft).astype(np.float32)
    basis = rng.integers(-5, 6, (nfft, ncols)).astype(np.float64)
    bt = np.zeros((nfft, 128))
    bt[:, :ncols] = basis
    got = ops.stft_f64(torch.from_numpy(x).to(dev), frames, hop, nfft, torch.from_numpy(bt).to(dev), ncols).cpu().numpy()
    rows = np.stack([x[f * hop:f * hop + nfft] for f in range(frames)]).astype(np.float64)
    assert np.array_equal(got, (rows @ basis).astype(np.float32))


def _mel_bar(got, wave):
    """The parity bar for log2-mel.  `truth` = the reference's formula (torch.hann_window values, torchaudio
    filterbank) evaluated in float64.  The fp32 CPU reference (torch.stft) is itself up to ~1e-4..2e-3 away from
    it in rare quiet bins (|X|^2 ~ 1e-4 of its neighbours), so the bars are: within 1e-4 of the exact value,
    and consistent with the fp32 reference up to that reference's own error."""
    from oracle import audio as oa
    ref32 = oa.extract_mel(wave)
    truth = oa.extract_mel_f64(wave.numpy())
    err_ref = np.abs(ref32 - truth).max()
    err_got = np.abs(got - truth).max()
    assert got.shape == ref32.shape
    assert err_got < 1e-4, (err_got, err_ref)                       # north_star tolerance, against the exact value
    assert np.abs(got - ref32).max() <= err_ref + 1e-4
    assert (np.abs(got - ref32) > 1e-4).mean() < 2e-3               # and the two agree to 1e-4 almost everywhere


@pytest.mark.parametrize("t,noise", [(16000, 0.05), (160000, 0.05), (16123, 0.05), (401, 0.05), (16000, 0.01)])
def test_log2_mel_vs_oracle(dev, t, noise):
    from avsum_amd.audio import MelPlan
    wave = _wave("multi", t, noise=noise)
    got = MelPlan.get(16000, 128, 40, dev).log2_mel(wave.to(dev)).cpu().numpy()
    assert got.shape == (1 + t // 200, 128)
    _mel_bar(got, wave)


def test_mel_power_pure_sine(dev):
    """Pure tone: low-energy bins are conditioned by the +1e-6 floor, so compare the mel POWER relative to its peak."""
    from avsum_amd.audio import MelPlan
    from oracle import audio as oa
    wave = _wave("sine", 160000)
    ref = oa.mel_spectrogram(wave).t()
    got = MelPlan.get(16000, 128, 40, dev).mel_power(wave.to(dev)).cpu()
    assert (got - ref).abs().max().item() < 2e-6 * ref.max().item()


def test_mfcc_vs_oracle(dev):
    from avsum_amd.audio import MelPlan
    from oracle import audio as oa
    wave = _wave("multi", 48000, 3)
    ref = oa.mfcc(wave).t()
    got = MelPlan.get(16000, 128, 40, dev).mfcc(wave.to(dev)).cpu()
    assert got.shape == ref.shape
    # MFCC values are in dB * sqrt(128)-ish units (|x| up to ~1e3): 1e-4 relative to the largest coefficient
    assert (got - ref).abs().max().item() < 1e-4 * ref.abs().max().item()


@pytest.mark.parametrize("t", [201, 399, 400, 6399, 6400, 6601, 50000])
def test_fused_front_end_matches_the_unfused_kernels(dev, t):
    """avs_stft_mel_fused_f32 (span in LDS, folded DFT, |.|^2 + mel + log on chip) against the three-kernel sequence
    it replaces (reflect pad -> dense fp64 DFT -> power/mel): the same exact-product fp64 arithmetic summed in a
    different order, so the fp32 spectrum and everything after it agree to an fp32 rounding step; frame counts on
    both sides of the 32-frame block, the shortest legal clip, an unaligned slice; both outputs from one pass."""
    from avsum_amd import ops
    from avsum_amd.audio import HOP, N_BINS, N_FFT, MelPlan
    plan = MelPlan.get(16000, 128, 40, dev)
    wave = _wave("multi", t, seed=t).to(dev)
    old_power = ops.power_mel(plan.spectrum(wave), N_BINS, plan.fb, plan.fb_lo, plan.fb_hi, 2).cpu()
    new_power = plan.mel_power(wave).cpu()
    assert new_power.shape == (1 + t // HOP, 128)
    assert (new_power - old_power).abs().max().item() <= 2e-6 * old_power.max().item()
    old_log = ops.power_mel(plan.spectrum(wave), N_BINS, plan.fb, plan.fb_lo, plan.fb_hi, 0).cpu()
    assert (plan.log2_mel(wave).cpu() - old_log).abs().max().item() < 2e-5
    mel, mfcc = plan.log2_mel_and_mfcc(wave)
    assert torch.equal(mel, plan.log2_mel(wave)) and torch.equal(mfcc, plan.mfcc(wave))
    if t > 1000:
        shifted = torch.cat([torch.zeros(3, device=dev), wave])[3:]          # storage offset of 12 bytes
        assert shifted.data_ptr() % 16 != 0
        assert torch.equal(plan.log2_mel(shifted), plan.log2_mel(wave))
    with pytest.raises(RuntimeError, match="Padding size"):
        plan.log2_mel(wave[:N_FFT // 2])


# ----------------------------------------------------------------------------- scorer pieces
def test_segment_means_without_the_frame_matrices(dev):
    """avs_stft_mel_segmean_f32: per-shot time means of log2-mel and of the clamped dB mel (-> MFCC by linearity) against
    the means of the full per-frame matrices of the fused kernel; ragged segments, an empty one, deterministic."""
    from avsum_amd.audio import MelPlan
    ops = _ops()
    plan = MelPlan.get(16000, 128, 40, dev)
    g = torch.Generator().manual_seed(12)
    tt = 16000 * 7 + 123
    time = torch.arange(tt) / 16000.0
    wave = (0.5 * torch.sin(2 * np.pi * 440 * time) + 0.2 * torch.sin(2 * np.pi * 1234 * time)
            + 0.05 * torch.randn(tt, generator=g)).to(dev)
    frames = 1 + tt // 200
    bounds = [0, 1, 34, 34, 100, 163, 400, frames]      # 1-frame, 33-frame, EMPTY, ... segments; all frames covered
    table = plan.segment_table(bounds, dev)
    nseg = len(bounds) - 1
    outs = []
    for one_pass in (True, True, False):   # one DFT pass (dB rows through the workspace) twice, then the two-pass form
        plan.one_pass_means = one_pass
        m_log2 = torch.zeros((nseg, 130), device=dev)
        m_db = torch.zeros((nseg, 128), device=dev)
        plan.segment_means(wave, table, m_log2, m_db)
        outs.append((m_log2.clone(), m_db.clone()))
    plan.one_pass_means = True
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])     # deterministic
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])     # one pass == two passes, bitwise
    # a table that does NOT cover the track (frames 34.. only) falls back to the track-maximum pass: the same means
    sub = plan.segment_table(bounds[3:], dev)
    s_log2, s_db = torch.zeros((nseg - 3, 130), device=dev), torch.zeros((nseg - 3, 128), device=dev)
    plan.segment_means(wave, sub, s_log2, s_db)
    assert torch.equal(s_log2, outs[0][0][3:]) and torch.equal(s_db, outs[0][1][3:])
    mel, mfcc = plan.log2_mel_and_mfcc(wave)
    _, db, _, gmax = plan._fused(wave, db=True)
    ops.clamp_topdb(db, gmax, 80.0)
    for sgm, (a, b) in enumerate(zip(bounds[:-1], bounds[1:])):
        if b == a:
            assert outs[0][0][sgm].abs().max().item() == 0 and outs[0][1][sgm].abs().max().item() == 0
            continue
        assert (outs[0][0][sgm, :128] - mel[a:b].mean(0)).abs().max().item() < 2e-5
        assert (outs[0][1][sgm] - db[a:b].mean(0)).abs().max().item() < 2e-4     # dB values reach -100
        # the MFCC rows' mean = the DCT of the mean dB row
        got_mfcc = ops.linear(outs[0][1][sgm:sgm + 1].contiguous(), plan.dct)
        assert (got_mfcc[0] - mfcc[a:b].mean(0)).abs().max().item() < 1e-3
    assert outs[0][0][:, 128:].abs().max().item() == 0      # the columns past n_mels are not touched


def test_segment_means_batch_equals_per_track(dev):
    """avs_stft_mel_segmean_batch_f32 (every track of a batch in one set of launches, the tracks one after another in one
    buffer) gives bit for bit the per-track calls' means (features/extractors.py:232-246 per shot); tracks of different
    lengths, 1-frame and empty segments."""
    from avsum_amd.audio import MelPlan
    plan = MelPlan.get(16000, 128, 40, dev)
    g = torch.Generator().manual_seed(17)
    waves, bounds = [], []
    for i, tt in enumerate((16123, 50000, 801, 33333)):
        time = torch.arange(tt) / 16000.0
        waves.append((0.4 * torch.sin(2 * np.pi * (300 + 170 * i) * time) + 0.05 * torch.randn(tt, generator=g)).to(dev))
        frames = 1 + tt // 200
        cuts = sorted(set([0, 1, min(frames, 34), min(frames, 34), frames // 2, frames]))
        bounds.append([0] + [c for c in cuts if c > 0] if cuts[0] == 0 else cuts)
    bounds[1] = [0, 7, 7, 100, 1 + 50000 // 200]          # an empty segment
    tables = plan.batch_tables(waves, bounds, dev)
    nseg = sum(len(b) - 1 for b in bounds)
    got_log2, got_db = torch.zeros((nseg, 128), device=dev), torch.zeros((nseg, 128), device=dev)
    plan.segment_means_batch(tables, got_log2, got_db)
    again_log2, again_db = torch.zeros_like(got_log2), torch.zeros_like(got_db)
    plan.segment_means_batch(tables, again_log2, again_db)
    assert torch.equal(got_log2, again_log2) and torch.equal(got_db, again_db)
    row = 0
    for w, b in zip(waves, bounds):
        k = len(b) - 1
        ref_log2, ref_db = torch.zeros((k, 128), device=dev), torch.zeros((k, 128), device=dev)
        plan.segment_means(w, plan.segment_table(b, dev), ref_log2, ref_db)
        assert torch.equal(got_log2[row:row + k], ref_log2) and torch.equal(got_db[row:row + k], ref_db)
        row += k
    with pytest.raises(ValueError, match="cover"):
        plan.batch_tables(waves[:1], [[0, 5]], dev)


def test_lstm_vs_oracle(dev):
    ops = _ops()
    from oracle import scorer as osc
    g = torch.Generator().manual_seed(9)
    # (256, 512, [5000, 1800]): configs[3] / configs[1] lengths on the scorer's kernel (register / LDS-resident W_hh^T)
    for hid, inp, lens in ((256, 512, [37, 5, 120]), (16, 32, [9, 1, 30]), (20, 8, [7]), (256, 512, [5000, 1800])):
        rows = sum(lens)
        x = torch.randn(rows, inp, generator=g)
        ws = {k: (torch.rand(s, generator=g) - 0.5) * 2 / hid ** 0.5 for k, s in
              (("wi0", (4 * hid, inp)), ("wh0", (4 * hid, hid)), ("bi0", (4 * hid,)), ("bh0", (4 * hid,)),
               ("wi1", (4 * hid, inp)), ("wh1", (4 * hid, hid)), ("bi1", (4 * hid,)), ("bh1", (4 * hid,)))}
        ref = torch.zeros(rows, 2 * hid)
        o = 0
        for ln in lens:
            ref[o:o + ln, :hid] = osc.lstm_direction(x[o:o + ln], ws["wi0"], ws["wh0"], ws["bi0"], ws["bh0"], False)
            ref[o:o + ln, hid:] = osc.lstm_direction(x[o:o + ln], ws["wi1"], ws["wh1"], ws["bi1"], ws["bh1"], True)
            o += ln
        wih = torch.cat([ws["wi0"], ws["wi1"]]).to(dev)
        bih = torch.cat([ws["bi0"] + ws["bh0"], ws["bi1"] + ws["bh1"]]).to(dev)
        xproj = ops.linear(x.to(dev), wih, bih)
        whh_t = torch.stack([ws["wh0"].t().contiguous(), ws["wh1"].t().contiguous()]).to(dev)
        out = torch.zeros(rows, 2 * hid + 4, device=dev)
        seq = torch.tensor(np.cumsum([0] + lens), dtype=torch.int64, device=dev)
        ops.lstm(xproj, whh_t, hid, 2, 0b10, seq, out, 4)
        assert (out[:, 4:].cpu() - ref).abs().max().item() < 2e-5
        assert out[:, :4].abs().max().item() == 0
        if hid == 256:
            # the scorer's size keeps part of W_hh^T in registers / LDS (default); the generic streaming kernel and the
            # other resident split run the same fmaf chain in the same order: bit-identical
            from avsum_amd import _abi
            # (the default at this few recurrences: one recurrence split over four CUs, W_hh in registers - the same chain)
            for variant in (_abi.LSTM_STREAM, _abi.LSTM_RESIDENT_20_8, _abi.LSTM_RESIDENT_16_8, _abi.LSTM_SPLIT4):
                alt = torch.zeros_like(out)
                ops.lstm(xproj, whh_t, hid, 2, 0b10, seq, alt, 4, variant=variant)
                assert torch.equal(alt, out)
            assert ops.lstm_split_errors(dev) == 0


def test_lstm_training_kernels_resident_vs_streaming(dev):
    """hidden = 256 (the scorer's size): the training forward on the register / LDS-resident kernel is bit for bit the
    streaming kernel's (outputs, saved gates, cell states); the resident backward sweep (W_hh partly on chip, the saved
    inputs fetched a step ahead, sums over 16 row slices) equals the streaming sweep to rounding and is deterministic.
    Four recurrences in one launch, as the training step uses them (scripts/train_av_model.py:88-93)."""
    ops = _ops()
    from avsum_amd import _abi
    hid, ndir, lens = 256, 4, [700, 1, 333]
    rows = sum(lens)
    g = torch.Generator().manual_seed(21)
    xproj = torch.randn(rows, ndir * 4 * hid, generator=g).to(dev)
    whh = ((torch.rand(ndir, 4 * hid, hid, generator=g) - 0.5) * 2 / hid ** 0.5).to(dev)
    whh_t = whh.transpose(1, 2).contiguous()
    seq = torch.tensor(np.cumsum([0] + lens), dtype=torch.int64, device=dev)
    dout = torch.randn(rows, ndir * hid + 8, generator=g).to(dev)
    res = {}
    for variant in (_abi.LSTM_AUTO, _abi.LSTM_STREAM, _abi.LSTM_RESIDENT_20_8, _abi.LSTM_SPLIT4):
        out = torch.zeros(rows, ndir * hid + 8, device=dev)
        gates, cell = ops.lstm_train_fwd(xproj, whh_t, hid, ndir, 0b1010, seq, out, 8, variant=variant)
        dx = ops.lstm_bwd(dout, 8, gates, cell, whh, hid, ndir, 0b1010, seq, variant=variant)
        dx2 = ops.lstm_bwd(dout, 8, gates, cell, whh, hid, ndir, 0b1010, seq, variant=variant)
        assert torch.equal(dx, dx2)
        res[variant] = (out, gates, cell, dx)
    a, b = res[_abi.LSTM_AUTO], res[_abi.LSTM_STREAM]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    # inference kernel: the same h
    inf = torch.zeros(rows, ndir * hid + 8, device=dev)
    ops.lstm(xproj, whh_t, hid, ndir, 0b1010, seq, inf, 8)
    assert torch.equal(inf, a[0])
    scale = b[3].abs().max().item()
    assert (a[3] - b[3]).abs().max().item() <= 2e-5 * scale   # 700-step chains of fp32 sums in two orders
    assert torch.isfinite(a[3]).all()
    # the four-CU split (AUTO's choice at 12 recurrences) against the one-CU resident kernels: the same chains in the same
    # order, forward and backward - every saved tensor and the gradient bit for bit
    r, sp = res[_abi.LSTM_RESIDENT_20_8], res[_abi.LSTM_SPLIT4]
    for x, y, z in zip(r, sp, a):
        assert torch.equal(x, y) and torch.equal(y, z)
    assert ops.lstm_split_errors(dev) == 0


def test_lstm_split_many_recurrences_and_ragged_lengths(dev):
    """The split recurrence with more workgroups than the chip holds at once (96 recurrences x 4 = 384 workgroups of 512
    threads: partners are dispatched together, complete groups always finish) and ragged lengths down to 0 and 1 steps:
    bit for bit the one-CU kernel, no bounded wait ran out; twice in a row on the same workspace (tag ranges move on)."""
    ops = _ops()
    from avsum_amd import _abi
    hid, ndir = 256, 4
    g = torch.Generator().manual_seed(33)
    lens = [int(v) for v in torch.randint(0, 40, (24,), generator=g)]
    lens[3], lens[7], lens[11] = 0, 1, 2
    rows = sum(lens)
    xproj = torch.randn(rows, ndir * 4 * hid, generator=g).to(dev)
    whh = ((torch.rand(ndir, 4 * hid, hid, generator=g) - 0.5) * 2 / hid ** 0.5).to(dev)
    whh_t = whh.transpose(1, 2).contiguous()
    seq = torch.tensor(np.cumsum([0] + lens), dtype=torch.int64, device=dev)
    dout = torch.randn(rows, ndir * hid, generator=g).to(dev)
    ref_out = torch.zeros(rows, ndir * hid, device=dev)
    g0, c0 = ops.lstm_train_fwd(xproj, whh_t, hid, ndir, 0b0110, seq, ref_out, 0, variant=_abi.LSTM_RESIDENT_20_8)
    dx0 = ops.lstm_bwd(dout, 0, g0, c0, whh, hid, ndir, 0b0110, seq, variant=_abi.LSTM_RESIDENT_20_8)
    for _ in range(2):
        out = torch.zeros(rows, ndir * hid, device=dev)
        g1, c1 = ops.lstm_train_fwd(xproj, whh_t, hid, ndir, 0b0110, seq, out, 0, variant=_abi.LSTM_SPLIT4)
        dx1 = ops.lstm_bwd(dout, 0, g1, c1, whh, hid, ndir, 0b0110, seq, variant=_abi.LSTM_SPLIT4)
        inf = torch.zeros(rows, ndir * hid, device=dev)
        ops.lstm(xproj, whh_t, hid, ndir, 0b0110, seq, inf, 0, variant=_abi.LSTM_SPLIT4)
        assert torch.equal(out, ref_out) and torch.equal(inf, ref_out)
        assert torch.equal(g1, g0) and torch.equal(c1, c0) and torch.equal(dx1, dx0)
    assert ops.lstm_split_errors(dev) == 0


def test_softmax_score_head_mha(dev):
    ops = _ops()
    from oracle import scorer as osc
    g = torch.Generator().manual_seed(1)
    x = torch.randn(7, 1003, generator=g) * 4
    xd = torch.zeros(7, 1004)
    xd[:, :1003] = x
    xd = xd.to(dev)
    ops.softmax_rows(xd, 7, 1003, 1004)
    assert (xd.cpu()[:, :1003] - torch.softmax(x, -1)).abs().max().item() < 1e-6
    hid = torch.randn(50, 64, generator=g)
    w2, b2 = torch.randn(64, generator=g), torch.randn(1, generator=g)
    sc = ops.score_head(hid.to(dev), w2.to(dev), b2.to(dev)).cpu()
    assert (sc - torch.sigmoid(hid @ w2 + b2)).abs().max().item() < 1e-6
    b, t, e, h = 3, 11, 64, 4
    xs = torch.randn(b, t, e, generator=g)
    in_w, in_b = torch.randn(3 * e, e, generator=g) / 8, torch.randn(3 * e, generator=g)
    qkv = (xs.reshape(b * t, e) @ in_w.t() + in_b)
    ctx = ops.mha_batchaxis(qkv.to(dev).contiguous(), b, t, e, h).cpu().view(b, t, e)
    ident_w, zero_b = torch.eye(e), torch.zeros(e)
    ref = osc.mha_seq_first(xs, in_w, in_b, ident_w, zero_b, h)
    assert (ctx - ref).abs().max().item() < 1e-5


# ----------------------------------------------------------------------------- fusion
def test_cdist_dtw_gather(dev):
    ops = _ops()
    from oracle import fusion as ofu
    g = torch.Generator().manual_seed(8)
    v, a = torch.randn(70, 50, generator=g), torch.randn(45, 50, generator=g)
    a[3] = v[5]
    ref = ofu.compute_dtw(v, a)
    got = ops.cdist(v.to(dev), a.to(dev)).cpu().numpy()
    assert got.dtype == np.float64 and got[5, 3] == 0.0
    assert np.abs(got - ref).max() <= 1e-13 * ref.max()
    cost, path = ofu.dtw_path(ref)
    dpath, plen, dcost = ops.dtw_path(torch.from_numpy(ref).to(dev))
    n = int(plen.item())
    assert np.array_equal(dpath[:n].cpu().numpy(), path)  # index work: bit exact
    assert abs(dcost.item() - cost) <= 1e-12 * cost
    # ties: integer costs force the documented up/left/diag preference
    ti = torch.randint(0, 3, (23, 31), generator=g).double()
    assert np.array_equal(ops.dtw_path(ti.to(dev))[0][:int(ops.dtw_path(ti.to(dev))[1].item())].cpu().numpy(),
                          ofu.dtw_path(ti.numpy())[1])
    feats = torch.randn(70, 50, generator=g)
    refi = ofu.interpolate_features(feats, path, 40)
    uniq, counts = np.unique(path[:, 0], return_counts=True)
    w = counts / counts.sum()
    goti = ops.gather_scale(feats.to(dev), torch.from_numpy(uniq).to(dev), torch.from_numpy(w).to(dev)).cpu()[:40]
    assert torch.equal(goti, refi)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bn_maxpool_is_bn_apply_then_pool(dev, dtype):
    """The stem's fused BatchNorm apply + ReLU + max pooling must be bit-identical to the two-kernel sequence
    (ragged groups, negative scales, one-affine mode)."""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    n, h, w, c = 7, 18, 14, 64
    x = torch.randn(n, h, w, c, generator=g).to(dtype).to(dev)
    frames = torch.tensor([0, 3, 4, 7], dtype=torch.int64)
    rows = (frames * h * w).to(dev)
    scale = torch.randn(3, c, generator=g).to(dev)      # both signs
    shift = torch.randn(3, c, generator=g).to(dev)
    for k, s, p in ((3, 2, 1), (3, 2, 0), (2, 2, 0)):
        ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        for relu in (True, False):
            full = ops.bn_apply(x.view(-1, c), scale, shift, rows, 3 * h * w, None,
                                ops.ACT_RELU if relu else ops.ACT_NONE).view(n, h, w, c)
            want = ops.pool2d(full, "max", k, s, p, torch.empty((n, ho, wo, c), dtype=dtype, device=dev))
            got = ops.bn_maxpool(x, scale, shift, rows, relu, k, s, p, torch.empty_like(want))
            assert torch.equal(got, want)
    full = ops.bn_apply(x.view(-1, c), scale[:1], shift[:1], None, 0, None, ops.ACT_RELU).view(n, h, w, c)
    want = ops.pool2d(full, "max", 3, 2, 1, torch.empty((n, 9, 7, c), dtype=dtype, device=dev))
    assert torch.equal(ops.bn_maxpool(x, scale[:1], shift[:1], None, True, 3, 2, 1, torch.empty_like(want)), want)
    ref = F.max_pool2d(torch.relu(x.float().cpu().permute(0, 3, 1, 2) * scale[0].cpu().view(1, -1, 1, 1)
                                  + shift[0].cpu().view(1, -1, 1, 1)), 3, 2, 1).permute(0, 2, 3, 1)
    assert (want.float().cpu() - ref).abs().max().item() < (1e-5 if dtype == torch.float32 else 0.05)


def test_empty_and_degenerate_inputs(dev):
    """Zero-size calls of the newer entry points return cleanly (the reference's own edge: an empty shot gives
    zeros(4096), extractors.py:44-45; an empty clip zeros(296), :197-198)."""
    ops = _ops()
    from avsum_amd.audio import resample_to
    from avsum_amd.features.extractors import AudioFeatureExtractor, VisualFeatureExtractor
    from avsum_amd.features.shots import detect_shots
    from avsum_amd.vggish import VGGish
    bf = torch.bfloat16
    code = ops.dtype_code(bf)
    # no frames at all
    x = torch.empty((0, 14, 14, 64), dtype=bf, device=dev)
    y = torch.empty((0, 7, 7, 64), dtype=bf, device=dev)
    sc = torch.ones((1, 64), device=dev)
    assert ops.bn_maxpool(x, sc, sc, None, True, 3, 2, 1, y).shape == (0, 7, 7, 64)
    geom, xs = (0, 14, 14, 64, 1, 1, 1, 1, 0, 0, 14, 14, 64), (14 * 14 * 64, 14 * 64, 64)
    w = torch.zeros((64, 64), dtype=bf, device=dev)
    ops.conv2d_raw(code, *geom, x, *xs, w, 64, torch.empty((0, 14, 14, 64), dtype=bf, device=dev), 64,
                   bnlocal=(196, sc[0], sc[0], 1e-5, None))
    sc0, sh0 = ops.conv2d_raw(code, *geom, x, *xs, w, 64, torch.empty((0, 14, 14, 64), dtype=bf, device=dev), 64,
                              bnstats=(196, sc[0], sc[0], 1e-5))
    assert sc0.shape == (0, 64) and sh0.shape == (0, 64)
    assert ops.hsv_frame_diff(torch.empty((0, 8, 8, 3), dtype=torch.uint8, device=dev)).shape == (0, 3)
    assert detect_shots(torch.zeros((1, 8, 8, 3), dtype=torch.uint8, device=dev)) == []
    assert resample_to(torch.zeros(0, device=dev), 48000, 16000).shape == (0,)
    assert ops.quantize(torch.zeros(0, device=dev), -2, 2, 63.75).shape == (0,)
    assert VGGish()(np.zeros(100, np.float32), 16000).shape == (0, 128)
    assert np.array_equal(VisualFeatureExtractor(bf)([]), np.zeros(4096, np.float32))
    assert np.array_equal(AudioFeatureExtractor()(np.zeros(0, np.float32)), np.zeros(296, np.float32))


def _synthetic_video(n, h, w, cut_at, seed):
    """Slowly drifting noise scenes with abrupt content changes at `cut_at`."""
    rng = np.random.default_rng(seed)
    frames = np.zeros((n, h, w, 3), dtype=np.uint8)
    base = rng.integers(0, 256, (h, w, 3))
    for f in range(n):
        if f in cut_at:
            base = rng.integers(0, 256, (h, w, 3))
        frames[f] = np.clip(base + rng.integers(-3, 4, (h, w, 3)), 0, 255).astype(np.uint8)
    return frames


def test_shot_scan_vs_oracle(dev):
    """HSV frame-difference sums are integer work: bit exact against the oracle; cuts and scenes identical."""
    ops = _ops()
    from avsum_amd.features import shots as gshots
    from oracle import shots as oshots
    video = _synthetic_video(70, 36, 52, {20, 27, 48}, 7)   # 27 is < min_scene_len after 20: suppressed
    d = torch.from_numpy(video).to(dev)
    sums = ops.hsv_frame_diff(d, 1).cpu().numpy()
    ref = np.zeros((70, 3), dtype=np.int64)
    hsv = [oshots.bgr2hsv_u8(f).astype(np.int64) for f in video]
    for f in range(1, 70):
        ref[f] = np.abs(hsv[f] - hsv[f - 1]).reshape(-1, 3).sum(0)
    assert np.array_equal(sums, ref)
    assert np.array_equal(gshots.content_scores(d, 1), oshots.content_scores(video, 1))
    assert gshots.detect_shots(d) == oshots.detect_shots(video) == [(0, 20), (20, 48), (48, 70)]
    wide = _synthetic_video(20, 40, 600, {16}, 8)            # 600 px wide: PySceneDetect strides by 2
    assert gshots.downscale_factor(600) == 2
    assert np.array_equal(gshots.content_scores(torch.from_numpy(wide).to(dev)), oshots.content_scores(wide, 2))
    assert gshots.detect_shots(torch.from_numpy(_synthetic_video(30, 16, 16, set(), 9)).to(dev)) == []


def test_align_features_against_oracle(dev):
    """AudioFeatureExtractor._align_features (features/extractors.py:248-290) on seeded arrays: common feature
    dimension / length, DTW of vggish against mfcc and mel on the GPU cost matrix, rows gathered along the path.
    Parity unpinned (the reference's fastdtw call raises): the oracle restates the intent."""
    from avsum_amd.features.extractors import AudioFeatureExtractor
    from oracle import fusion as ofu
    rng = np.random.default_rng(77)
    ext = AudioFeatureExtractor()
    for (tm, dm), (tl, dl), (tv, dv) in (((30, 128), (30, 128), (9, 128)), ((17, 40), (25, 128), (12, 128)),
                                          ((6, 128), (6, 128), (6, 128)), ((1, 128), (4, 128), (3, 128))):
        mfcc, mel, vg = rng.normal(size=(tm, dm)), rng.normal(size=(tl, dl)), rng.normal(size=(tv, dv))
        a_mfcc, a_mel = ext._align_features(mfcc, mel, vg)
        r_mfcc, r_mel = ofu.align_features(mfcc, mel, vg)
        assert a_mfcc.shape == r_mfcc.shape and a_mel.shape == r_mel.shape
        assert np.array_equal(a_mfcc, r_mfcc) and np.array_equal(a_mel, r_mel)     # index work: bit-exact
    # 1-D inputs are promoted (np.atleast_2d); empty inputs give zeros(128) (extractors.py:252-259)
    a, b = ext._align_features(rng.normal(size=128), rng.normal(size=128), rng.normal(size=128))
    ra, rb = ofu.align_features(*[rng.normal(size=128)] * 3)
    assert a.shape == ra.shape == (1, 128) and b.shape == rb.shape
    z1, z2 = ext._align_features(np.zeros((0, 128)), rng.normal(size=(3, 128)), rng.normal(size=(3, 128)))
    assert np.array_equal(z1, np.zeros(128)) and np.array_equal(z2, np.zeros(128))


@pytest.mark.parametrize("rows,cols,ld,weighted", [(1800, 2048, 2048, False), (333, 64, 64, True), (1, 1, 1, False),
                                                   (257, 23, 40, True), (5000, 8, 8, False), (63, 1024, 1030, True)])
def test_colsum_bias_gradient_kernel(dev, rows, cols, ld, weighted):
    """avs_colsum_f32 (the bias gradients and the scoring head's weighted sum of loss.backward(),
    scripts/train_av_model.py:94): ragged row / column counts, a row stride wider than the columns, row weights; against a
    float64 sum; twice the same bits (fixed summation order)."""
    ops = _ops()
    g = torch.Generator().manual_seed(rows + cols)
    buf = torch.randn(max(rows, 1), ld, generator=g)
    x = buf[:rows, :cols]
    w = torch.randn(rows, generator=g) if weighted else None
    xd = buf.to(dev)[:rows, :cols]
    wd = w.to(dev) if weighted else None
    got = ops.colsum(xd, wd)
    ref = (x.double() * (w.double()[:, None] if weighted else 1.0)).sum(0)
    scale = max(1.0, (x.abs().double() * (w.abs().double()[:, None] if weighted else 1.0)).sum(0).max().item()) if rows else 1.0
    assert got.shape == (cols,)
    assert (got.cpu().double() - ref).abs().max().item() <= 2e-6 * scale
    assert torch.equal(ops.colsum(xd, wd), got)


@pytest.mark.parametrize("kind", ["f16x2", "bf16"])
def test_pool2d_3x3_windows_packed_formats(dev, kind):
    """avs_pool2d_nhwc on 3x3 windows in the storage formats of the two trunks (every pooling of ResNet-50 / Inception-v3,
    features/extractors.py:29,83): the nine loads of a window issued together, outside taps dropped - max pooling exact
    (a max of stored values is a stored value), average pooling to the format's rounding; odd sizes, both strides, padding,
    channel-slice destination with bias + ReLU."""
    ops = _ops()
    g = torch.Generator().manual_seed(77)
    n, h, w, c = 3, 17, 13, 24
    x32 = torch.randn(n, h, w, c, generator=g)
    if kind == "f16x2":
        xd = ops.f16x2_pack(x32.to(dev))
        xv = ops.f16x2_unpack(xd).cpu()
        mk = lambda shape: torch.zeros(shape, device=dev)
        rd = lambda t: ops.f16x2_unpack(t.contiguous()).cpu()
        code, tol = ops.dtype_code(torch.float32, "f16x2"), 5e-7
    else:
        xd = x32.to(dev).to(torch.bfloat16)
        xv = xd.float().cpu()
        mk = lambda shape: torch.zeros(shape, device=dev, dtype=torch.bfloat16)
        rd = lambda t: t.float().cpu()
        code, tol = None, 8e-3
    xc = xv.permute(0, 3, 1, 2)
    for mode, s, p in (("max", 2, 0), ("max", 2, 1), ("max", 1, 1), ("avg", 1, 1), ("avg", 2, 0)):
        ref = (F.max_pool2d(xc, 3, s, p) if mode == "max" else F.avg_pool2d(xc, 3, s, p)).permute(0, 2, 3, 1)
        out = mk((n, ref.shape[1], ref.shape[2], c))
        ops.pool2d(xd, mode, 3, s, p, out, code=code)
        got = rd(out)
        if mode == "max":
            assert torch.equal(got, ref)
        else:
            assert (got - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    bias = torch.randn(c, generator=g)
    wide = mk((n, h, w, c + 16))
    ops.pool2d(xd, "avg", 3, 1, 1, wide[..., 8:8 + c], bias.to(dev), ops.ACT_RELU, code=code)
    ref = torch.relu(F.avg_pool2d(xc, 3, 1, 1).permute(0, 2, 3, 1) + bias)
    assert (rd(wide[..., 8:8 + c]) - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    assert rd(wide[..., :8]).abs().max().item() == 0 and rd(wide[..., 8 + c:]).abs().max().item() == 0
