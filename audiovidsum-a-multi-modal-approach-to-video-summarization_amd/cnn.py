"""CNN trunks of the visual extractor on the MI355X.

The reference builds ``torchvision.models.resnet50(pretrained=True)`` minus its
``fc`` as an ``nn.Sequential`` (features/extractors.py:25,29) and
``inception_v3(pretrained=True, aux_logits=True)`` with ``fc = Identity``
(:26,32-36).  torchvision and its weight files are not available offline, so
this module supplies

  * parameter containers with torchvision's module structure and state-dict
    keys (weights come from ``load_state_dict`` or a seeded init), and
  * runners that execute the trunks through libavsum_hip.so on NHWC tensors:
    every convolution is the implicit-GEMM kernel (fp32 MFMA = parity mode,
    bf16 MFMA = throughput mode); ResNet BatchNorm runs in BATCH-STATISTICS
    mode per micro-batch group, because the reference never puts that trunk in
    eval mode; Inception BatchNorm (eval mode, eps 1e-3) is folded
    into the convolution weights.

The containers deliberately have no torch forward: the product path is HIP only.
"""
import torch
import torch.nn as nn

from . import ops

RESNET_MEAN = (0.485, 0.456, 0.406)
RESNET_STD = (0.229, 0.224, 0.225)


class _NoTorchForward(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - guard
        raise RuntimeError("this module only holds parameters; the trunk runs through the HIP runner")


# ============================================================================ ResNet-50 container
class Bottleneck(_NoTorchForward):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)  # v1.5: stride on the 3x3
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride


def _make_layer(inplanes, planes, blocks, stride):
    downsample = None
    if stride != 1 or inplanes != planes * 4:
        downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
    layers = [Bottleneck(inplanes, planes, stride, downsample)]
    for _ in range(1, blocks):
        layers.append(Bottleneck(planes * 4, planes))
    return nn.Sequential(*layers)


def resnet50_trunk():
    """``nn.Sequential(*list(resnet50().children())[:-1])`` with torchvision's init (kaiming fan_out, BN 1/0)."""
    trunk = nn.Sequential(
        nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1),
        _make_layer(64, 64, 3, 1), _make_layer(256, 128, 4, 2), _make_layer(512, 256, 6, 2),
        _make_layer(1024, 512, 3, 2), nn.AdaptiveAvgPool2d((1, 1)),
    )
    for m in trunk.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)
    return trunk


# ============================================================================ helpers
def _ohwi(weight, dtype):
    """OIHW conv weight -> [O, kh*kw*I] in the kernel's reduction order."""
    o, i, kh, kw = weight.shape
    return weight.detach().permute(0, 2, 3, 1).reshape(o, kh * kw * i).to(dtype).contiguous()


def _stem_weight(weight, px, dtype):
    """Stem conv [O,3,kh,kw] on the 4-channel pre-padded image: each kernel row becomes ``px`` pixels x 4
    channels (zero beyond kw and in channel 3) so that one reduction step is a contiguous run of pixels."""
    o, i, kh, kw = weight.shape
    w = torch.zeros((o, kh, px, 4), dtype=torch.float32, device=weight.device)
    w[:, :, :kw, :i] = weight.detach().permute(0, 2, 3, 1)
    return w.reshape(o, kh * px * 4).to(dtype).contiguous()


class _W:
    """A convolution weight in kernel layout: `rows` = [cout, kh*kw*cin] (what the 1x1 BatchNorm kernels and the stem
    read) and, when the reduction is a multiple of a 64-byte step, `kstep` = the same matrix reduction-step major
    (ops.weights_kstep32: whole cache lines per weight DMA instruction of the contraction kernel)."""
    __slots__ = ("rows", "kstep")

    def __init__(self, rows):
        self.rows = rows
        step = 32 if rows.dtype == torch.bfloat16 else 16   # elements of a 64-byte reduction step
        self.kstep = ops.weights_kstep32(rows) if rows.shape[1] % step == 0 else None

    def conv_operand(self):
        """(tensor, w_layout) for avs_conv2d_nhwc*."""
        return (self.kstep, 1) if self.kstep is not None else (self.rows, 0)


class ResNet50Runner:
    """Runs the container's parameters on uint8 frames [N,224,224,3] -> fp32 [N,2048].

    Batch-statistics BatchNorm (the reference's mode) has three forms here, chosen per layer; every one
    of them is deterministic (no float atomics: two runs give bit-identical features):
      local   bf16, equal-sized groups of <= 256 rows (14x14 / 7x7 maps): convolution + whole BatchNorm (+ residual
              + ReLU) in ONE launch, statistics inside a tile (avs_conv2d_nhwc_bnlocal) - nothing raw in HBM;
      gram    bf16 expanding 1x1 layers with 64 / 128 input channels (conv3 / downsample of layers 1-2): statistics
              from the input's Gram matrix (avs_bn_gram_affine_bf16), then ONE streaming pass with the affine in the
              epilogue (avs_conv1x1_affine_bf16);
      twopass other bf16 1x1 layers: one workgroup walks a group twice (avs_conv1x1_bn_bf16);
      split   convolution (+ per-tile partial statistics in its epilogue for bf16, folded in tile order) or
              avs_bn_batch_stats -> avs_bn_apply (fp32 parity mode, ragged groups, shapes the other forms decline)."""

    def __init__(self, trunk, dtype=torch.float32, bn_mode="batch", f32_split=False):
        """f32_split (fp32 only): True = activations and weights stay fp32 in HBM, the convolutions' products run on the
        bf16 matrix cores as hi*hi + hi*lo + lo*hi (AVS_F32_SPLIT: ~2^-15 relative per product instead of exact);
        "f16x2" = AVS_F16X2: activations and weights are STORED as fp16 hi | lo runs (22 significant bits, the byte
        size of fp32), products are three fp16 MFMAs with no arithmetic on the operands, every output is split once
        where it is produced, BatchNorm statistics are centred two-round sums: the fast parity-grade mode."""
        if bn_mode not in ("batch", "folded"):
            raise ValueError("bn_mode must be 'batch' (reference-faithful) or 'folded'")
        self.trunk, self.dtype, self.bn_mode = trunk, dtype, bn_mode
        self.h2 = f32_split == "f16x2" and dtype == torch.float32
        self.f32_split = bool(f32_split) and not self.h2 and dtype == torch.float32
        self.code = ops.dtype_code(dtype, "f16x2" if self.h2 else self.f32_split)   # the contraction entry points
        self.ecode = self.code if self.h2 else ops.dtype_code(dtype)                 # storage format (elementwise kernels)
        self.fuse_conv_bn = True
        self.fuse_min_rows, self.fuse_ratio_num, self.fuse_ratio_den = 128, 2, 1
        self.twopass_max_cin = 128   # wider inputs (256 -> 1024 at 4-frame groups): the second matrix pass costs more than
                                     # the split form's extra traffic (measured: 4.5 vs 3.4 ms per 8192 frames)
        self.bn_local = True         # the one-launch tile-local form where the library takes the shape
        self.gram_stats = True       # conv3 / downsample of layers 1-2: Gram-matrix statistics + one streaming pass
        self.bn_cluster = True       # AVS_F16X2: groups of several 14x14 maps in ONE launch (tiles exchange their statistics)
        self.fused_stem = True       # uint8 frames -> conv1 -> pooled raw map + partial sums in one kernel
        self.stem_raw = True         # bn1 + ReLU ride in the staging of layer 1's first conv1 / downsample (both take the
                                     # one-pass form on ONE Gram matrix of the stem output): no finishing pass
        self.defer_bn_apply = True   # bn2 + ReLU applied inside conv3's two-pass kernel (avs_conv1x1_bn_in_bf16)
        self.defer_res_apply = True  # the downsample's BatchNorm applied inside conv3's residual add (layer 2's first block)
        self.gram_finish_min_k = 128  # >= this many input channels: the Gram kernel stores the finished input in place, so
                                      # the convolution pass (N / 128 column slabs) does not transform it per slab
        self.affine_variant = 0      # tile override of the one-pass 1x1 form (study: _abi.TILE_128 / TILE_256)
        self.stats_1x1_variant = 0   # tile override of the 1x1 convolution + statistics form (study: _abi.TILE_128 / TILE_256)
        self.p8_blocks = (0, 1, 3, 4, 5) if self.h2 else ()   # AVS_F16X2: the outputs of these bottlenecks (the inner
                                     # blocks of layers 1-2, whose consumers are the next block's conv1 and residual add - both
                                     # HBM-bound) are stored as AVS_F16P8: fp16 hi + 8-bit remainder, 3 instead of 4 bytes
        self._key = None
        self._w = None
        self._plans = {}     # (n, group frames) -> per layer: does it take the tile-local form

    block_hook = None   # study hook: callable(block index, block output) -> block output

    # weights in kernel layout, rebuilt when the parameters change / move
    def _prepare(self):
        key = tuple((p.data_ptr(), p._version) for p in self.trunk.parameters()) + (self.dtype,)
        if self._w is not None and key == self._key:
            return self._w
        t, dt = self.trunk, self.dtype
        pack = ops.f16x2_pack if self.h2 else (lambda r: r)   # AVS_F16X2: every weight row as fp16 hi | lo runs, once

        def mkw(rows):
            return _W(pack(rows))

        w = {"stem": mkw(_stem_weight(t[0].weight, 8, dt)), "blocks": []}
        if self.h2 and tuple(t[0].weight.shape) == (64, 3, 7, 7):
            # the fused AVS_F16X2 stem: the input normalisation folded into the weights (the constant term in channel 3)
            w["stem_h2"] = ops.stem_h2_operands(t[0].weight, 1.0, RESNET_MEAN, RESNET_STD)

        def bn(m):
            return (m.weight.detach().float().contiguous(), m.bias.detach().float().contiguous(), float(m.eps),
                    m.running_mean.detach().float(), m.running_var.detach().float())

        w["bn1"] = bn(t[1])
        for li in range(4, 8):
            for blk in t[li]:
                d = {"c1": mkw(_ohwi(blk.conv1.weight, dt)), "b1": bn(blk.bn1), "c2": mkw(_ohwi(blk.conv2.weight, dt)),
                     "b2": bn(blk.bn2), "c3": mkw(_ohwi(blk.conv3.weight, dt)), "b3": bn(blk.bn3),
                     "stride": blk.stride, "planes": blk.conv1.out_channels}
                if blk.downsample is not None:
                    d["cd"] = mkw(_ohwi(blk.downsample[0].weight, dt))
                    d["bd"] = bn(blk.downsample[1])
                w["blocks"].append(d)
        # layer 1's first block: conv1 (64 -> 64) and the downsample (64 -> 256) read the same input, so ONE Gram matrix
        # gives the batch statistics of both: their weights / BatchNorm parameters stacked for avs_bn_gram_affine_bf16
        b0 = w["blocks"][0]
        if "cd" in b0 and b0["c1"].rows.shape[1] == b0["cd"].rows.shape[1] and b0["b1"][2] == b0["bd"][2]:
            w["cat0"] = (torch.cat([b0["c1"].rows, b0["cd"].rows]).contiguous(),
                         torch.cat([b0["b1"][0], b0["bd"][0]]).contiguous(),
                         torch.cat([b0["b1"][1], b0["bd"][1]]).contiguous(), b0["b1"][2], b0["c1"].rows.shape[0])
        self._w, self._key = w, key
        return w

    # ---- convolution geometry (shared by the workspace plan and the forward pass) ----
    @staticmethod
    def _stem_geom(n):
        # [N,230,232,4] pre-padded image; conv1 7x7/2 reads 8-pixel (32-element) runs: kh = 7 rows x 32 elements
        return (n, 230, 112, 32, 7, 1, 2, 1, 0, 0, 112, 112, 64), (230 * 232 * 4, 232 * 4, 8), 7 * 32

    @staticmethod
    def _nhwc_geom(n, h, cin, k, s, p, cout):
        ho = (h + 2 * p - k) // s + 1
        return (n, h, h, cin, k, k, s, s, p, p, ho, ho, cout), (h * h * cin, h * cin, cin), k * k * cin

    def _layer_geoms(self, n):
        """Every convolution of the trunk in forward order: (geometry, x strides, weight row stride)."""
        yield self._stem_geom(n)
        h, cin = 56, 64
        for li, (planes, blocks, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))):
            for b in range(blocks):
                s = stride if b == 0 else 1
                hout = h // s
                yield self._nhwc_geom(n, h, cin, 1, 1, 0, planes)
                yield self._nhwc_geom(n, h, planes, 3, s, 1, planes)
                if b == 0:
                    yield self._nhwc_geom(n, h, cin, 1, s, 0, planes * 4)
                yield self._nhwc_geom(n, hout, planes, 1, 1, 0, planes * 4)
                h, cin = hout, planes * 4

    def _local_plan(self, n, gsz):
        """Per layer (forward order): does the one-launch tile-local form take it for n frames in groups of gsz."""
        key = (n, gsz, self.bn_cluster)
        plan = self._plans.get(key)
        if plan is None:
            dcode = self.code   # (the tile-local form exists for bf16 and f16x2)
            plan = []
            for geom, xs, wrs in self._layer_geoms(n):
                ho, wo, cout = geom[10], geom[11], geom[12]
                rows = gsz * ho * wo
                if rows <= 256:
                    plan.append(ops.conv_bnlocal_tile_rows(dcode, *geom, *xs, wrs, cout, rows) is not None)
                else:
                    # a group larger than a tile: the clustered tile-local form where the map splits into k tiles of 193..224
                    # rows (14 x 14: k = 1, layer 3 with the reference's 4-frame micro-batches; 28 x 28: k = 4) and the group
                    # into gsz * k <= 16 of them - the tiles of a group exchange their statistics
                    cl = False
                    if self.h2 and self.bn_cluster:
                        cin_l, kh_l = geom[3], geom[4]
                        for k in (1, 2, 4, 8, 16):
                            # (maps of several tiles: only the wide 1x1 layer that would otherwise take convolution +
                            #  statistics + an apply pass - layer 3's first conv1, 512 -> 256 at 28 x 28; measured: layer 2's
                            #  narrow layers are no faster clustered than on their Gram / nine-tap / 3-byte forms)
                            if k > 1 and not (kh_l == 1 and cin_l >= 512 and cout >= 256):
                                continue
                            if (ho * wo) % k == 0 and 192 < (ho * wo) // k <= 224 and 2 <= gsz * k <= 16:
                                if ops.conv_bncluster_ok(dcode, *geom, *xs, wrs, cout, rows, gsz * k):
                                    cl = gsz * k
                                break
                    plan.append(cl)
            if len(self._plans) > 64:
                self._plans.clear()
            self._plans[key] = plan
        return plan

    def _twopass_ok(self, cin, cout, kh, sh, gmax):
        # measured on MI355X: the two-pass kernel wins over the split form where the layer is write-heavy
        # (cout >= 2*cin: the conv3 / downsample layers) and a group is several row tiles long
        return (kh == 1 and sh == 1 and self.fuse_conv_bn and gmax >= self.fuse_min_rows
                and cout * self.fuse_ratio_den >= cin * self.fuse_ratio_num and cin <= self.twopass_max_cin)

    def _gram_h2_ok(self, cin, cout, kh, sh, gmax):
        """AVS_F16X2: the expanding 1x1 layers whose groups are too large for a tile take Gram statistics + one
        streaming pass (64 / 128 input channels: conv3 of layers 1-2 and layer 1's downsample)."""
        return (self.h2 and self.gram_stats and kh == 1 and sh == 1 and cin in (64, 128) and cout % 32 == 0
                and cout >= 2 * cin and gmax >= self.fuse_min_rows)

    def _conv_bn(self, geom, xs, x, wt, bnp, groups, residual=None, relu=True, local=False, algo_k=None, pool=None,
                 defer=False, in_affine=None, res_affine=None, out_p8=False):
        """One convolution + BatchNorm (+ residual, + ReLU) -> NHWC activation; picks the form (class docstring).
        pool = (k, s, p): a max pooling follows (the stem) - on the split form it is fused with the BatchNorm apply
        (avs_bn_maxpool_nhwc: the normalised full-resolution map is never written).
        defer: return (raw convolution, (scale, shift)) WITHOUT applying the BatchNorm - the next layer's two-pass
        kernel applies it while staging its input (in_affine), so this layer needs no apply pass (bf16, equal groups).
        res_affine: the residual is a deferred (raw) downsample output; its BatchNorm rides in this layer's residual add
        (one-pass 1x1 form only).
        out_p8: the output as an AVS_F16P8 tensor (ops.P8; the one-pass 1x1 form only - the caller has checked that the
        layer takes it); x / residual may be P8 tensors where the forms that read them take that format."""
        n, ho, wo, cout = geom[0], geom[10], geom[11], geom[12]
        cin, kh, sh = geom[3], geom[4], geom[6]
        dev, dt = x.device, self.dtype
        dcode = self.code
        gamma, beta, eps, rmean, rvar = bnp
        act = ops.ACT_RELU if relu else ops.ACT_NONE
        y = torch.empty((n, ho, wo, cout), dtype=dt, device=dev)
        y2d = y.view(-1, cout)

        def conv(**kw):
            wsel, layout = wt.conv_operand()
            if kh == 1 and "bnstats" in kw and self.stats_1x1_variant:
                kw["variant"] = self.stats_1x1_variant
            return ops.conv2d_raw(dcode, *geom, x, *xs, wsel, wsel.stride(0), y, cout, algo_k=algo_k,
                                  algo_in_elems=x.numel() if algo_k is not None else None, w_layout=layout, **kw)

        def pooled(t):
            k, s, p = pool
            hp, wp = (ho + 2 * p - k) // s + 1, (wo + 2 * p - k) // s + 1
            return torch.empty((n, hp, wp, cout), dtype=dt, device=dev), k, s, p

        def finish(scale, shift, grows, gmax):
            if pool is not None and residual is None:
                out, k, s, p = pooled(y)
                return ops.bn_maxpool(y, scale, shift, grows, relu, k, s, p, out, code=self.ecode)
            ops.bn_apply(y2d, scale, shift, grows, gmax, residual, act, y2d, code=self.ecode)
            if pool is not None:
                out, k, s, p = pooled(y)
                return ops.pool2d(y, "max", k, s, p, out, code=self.ecode)
            return y

        if self.bn_mode != "batch":
            conv()
            scale = (gamma / torch.sqrt(rvar + eps)).contiguous()
            shift = (beta - rmean * scale).contiguous()
            return finish(scale.view(1, -1), shift.view(1, -1), None, 0)
        grows, gmax, uniform = groups[ho * wo]
        bf16 = dt == torch.bfloat16
        # statistics from the convolution's epilogue: the bf16 mode, and the fp32-split mode (whose products already carry
        # ~2^-15 of error: the E[y^2] - E[y]^2 form on fp32 sums costs nothing next to that); the exact fp32 parity mode
        # keeps the shifted statistics pass over the stored output
        fast = uniform and (bf16 or self.f32_split or self.h2)
        if fast and (bf16 or self.h2) and local:
            cluster = local if (local is not True and int(local) > 1) else 1   # (the plan: True = one group per tile)
            conv(act=act, bnlocal=(gmax, gamma, beta, eps, residual), cluster=cluster)
            if pool is not None:
                out, k, s, p = pooled(y)
                return ops.pool2d(y, "max", k, s, p, out, code=self.ecode)
            return y
        if self.h2 and (in_affine is not None or (fast and not local and self._gram_h2_ok(cin, cout, kh, sh, gmax))):
            # AVS_F16X2: statistics from the input's second moments (the pass that also applies the BatchNorm + ReLU of
            # the layer before, in place), then ONE streaming convolution pass with the affine in its epilogue
            x2d = x.view(-1, cin)
            sc, sf = ops.bn_gram_affine_h2(x2d, wt.rows, gmax, gamma, beta, eps, in_affine, store_input=in_affine is not None)
            wsel, layout = wt.conv_operand()
            if out_p8:
                y = ops.P8.empty((n, ho, wo, cout), dev)
            ops.conv2d_affine(dcode, n, geom[1], geom[2], cin, sh, geom[7], ho, wo, cout, x, *xs, wsel, wsel.stride(0), y, cout,
                              gmax, sc, sf, residual, relu, res_affine, w_layout=layout, variant=self.affine_variant)
            return y
        if out_p8 or isinstance(residual, ops.P8):
            raise RuntimeError("an AVS_F16P8 output / residual belongs to the one-pass 1x1 form")
        if in_affine is not None or (fast and bf16 and self._twopass_ok(cin, cout, kh, sh, gmax)):
            # statistics from the input's Gram matrix + ONE streaming pass where the shape allows it (the expanding
            # 1x1 layers of layers 1-2), else the two-pass kernel
            if self.gram_stats and ops.gram_supported(cin, cout):
                ops.conv1x1_gram_bn(x.view(-1, cin), wt.rows, gmax, gamma, beta, eps, y2d, residual, relu, in_affine, res_affine,
                                    finish_input=cin >= self.gram_finish_min_k)
            else:
                assert res_affine is None
                ops.conv1x1_bn(x.view(-1, cin), wt.rows, gmax, gamma, beta, eps, y2d, residual, relu, in_affine)
            return y
        assert res_affine is None
        affine = None
        if fast:
            # statistics from the convolution's epilogue: per-tile partial sums of the fp32 accumulators, folded in
            # tile order (E[x^2]-E[x]^2: fine for bf16 activations); None = groups too small for that form
            affine = conv(bnstats=(gmax, gamma, beta, eps))
        if affine is None and isinstance(x, ops.P8):
            raise RuntimeError("an AVS_F16P8 input belongs to the convolution + statistics form")
        if affine is None:
            # fp32 parity mode / ragged groups / tiny groups (the fused form declined before launching anything):
            # plain convolution, then the shifted statistics pass over the stored output
            conv()
            affine = ops.bn_batch_stats(y2d, grows, gamma, beta, eps, code=self.ecode)
        if defer:
            return y, affine
        return finish(affine[0], affine[1], grows, gmax)

    def forward(self, frames_u8, group_frames=None, out=None, mid_hook=None):
        """frames_u8: device uint8 [N,224,224,3] (already 224x224, extractors.py:132).
        group_frames: int64 CPU tensor / list [G+1] of frame offsets of the BatchNorm micro-batch groups
        (extractors.py:48-56); default = one group per frame.
        mid_hook: called (no arguments) once layers 1-2 - the HBM-bound half of the trunk - have been launched and
        before layers 3-4 - the matrix-core-bound half: the pipeline records a stream event there, so that the next
        pass (on another stream) runs its memory-bound half under this pass's compute-bound half."""
        n, h, w_, _ = frames_u8.shape
        if (h, w_) != (224, 224):
            raise ValueError("ResNet50Runner expects 224x224 frames (resize first)")
        if n == 0:
            return torch.zeros((0, 2048), dtype=torch.float32, device=frames_u8.device)
        w = self._prepare()
        dev, dt = frames_u8.device, self.dtype
        if group_frames is None:
            group_frames = torch.arange(n + 1, dtype=torch.int64)
        group_frames = torch.as_tensor(group_frames, dtype=torch.int64)
        if int(group_frames[0]) != 0 or int(group_frames[-1]) != n:
            raise ValueError("group_frames must start at 0 and end at N")
        sizes = group_frames[1:] - group_frames[:-1]
        gsz = int(sizes.max())
        uniform = bool((sizes == gsz).all())
        groups = {hw: ((group_frames * hw).to(dev), gsz * hw, uniform)
                  for hw in (112 * 112, 56 * 56, 28 * 28, 14 * 14, 7 * 7)}
        use_local = self.bn_local and self.bn_mode == "batch" and uniform and (dt == torch.bfloat16 or self.h2)
        plan = iter(self._local_plan(n, gsz)) if use_local else None

        def slot():
            return next(plan) if plan is not None else False

        # stem: (x - mean)/std without /255 (extractors.py:133-139), conv1 7x7/2 pad 3, bn1, ReLU, maxpool 3x3/2
        stem_local = slot()
        x_aff = None     # (scale, shift) of bn1 when x is the stem's RAW pooled map (applied by block 0's kernels)
        if self.fused_stem and self.bn_mode == "batch" and uniform and dt == torch.bfloat16:
            # one fused launch: the normalised image and the 112x112x64 map never reach HBM.  bn1 + ReLU: a finishing
            # pass in place, or (stem_raw) inside the staging of the first block's conv1 / downsample
            gamma, beta, eps = w["bn1"][:3]
            raw = (self.stem_raw and self.gram_stats and self.fuse_conv_bn and "cat0" in w
                   and ops.gram_supported(64, w["cat0"][0].shape[0]) and gsz * 56 * 56 >= self.fuse_min_rows)
            x, sc0, sh0 = ops.stem_conv_bn_pool(frames_u8, w["stem"].rows, 1.0, RESNET_MEAN, RESNET_STD, gsz, gamma, beta, eps,
                                                apply=not raw)
            if raw:
                x_aff = (sc0, sh0)
        elif (self.fused_stem and self.h2 and self.bn_mode == "batch" and uniform and "stem_h2" in w and "cat0" in w
              and self._gram_h2_ok(64, w["cat0"][0].shape[0], 1, 1, gsz * 56 * 56)):
            # AVS_F16X2: one fused launch (uint8 frames -> conv1 on the fp16 matrix cores, two MFMAs per product -> centred
            # statistics -> the pooled RAW map); bn1 + ReLU are applied by the first block's Gram pass, which reads the map
            # anyway and stores the finished activation in place
            gamma, beta, eps = w["bn1"][:3]
            x, sc0, sh0 = ops.stem_conv_pool_h2(frames_u8, w["stem_h2"], gsz, gamma, beta, eps)
            x_aff = (sc0, sh0)
        else:
            x0 = ops.frames_normalize(frames_u8, dt, 1.0, RESNET_MEAN, RESNET_STD, 230, 232, 3, 3, code=self.ecode)
            geom, xs, _ = self._stem_geom(n)
            x = self._conv_bn(geom, xs, x0, w["stem"], w["bn1"], groups, local=stem_local, algo_k=147, pool=(3, 2, 1))
            del x0
        hcur = 56
        for bi, blk in enumerate(w["blocks"]):
            if bi == 7 and mid_hook is not None:   # blocks 0-2 = layer 1, 3-6 = layer 2
                mid_hook()
            s, planes = blk["stride"], blk["planes"]
            cin = x.shape[3]
            hout = hcur // s
            s1, s2 = slot(), slot()
            sd = slot() if "cd" in blk else False
            s3 = slot()
            idn = None
            if x_aff is not None and self.h2:
                # AVS_F16X2, first block on the fused stem's RAW pooled map: the Gram pass applies bn1 + ReLU on the way in,
                # stores the finished activation in place (the identity input of nothing else: conv1 and the downsample
                # read it) and gives both BatchNorms' affines; then one streaming pass each
                wcat, gcat, bcat, eps, c1n = w["cat0"]
                gmax = gsz * hcur * hcur
                sc, sf = ops.bn_gram_affine_h2(x.view(-1, cin), wcat, gmax, gcat, bcat, eps, in_affine=x_aff, store_input=True)
                x_aff = None
                geom, xs, _ = self._nhwc_geom(n, hcur, cin, 1, 1, 0, planes)
                t1 = torch.empty((n, hcur, hcur, planes), dtype=dt, device=dev)
                wsel, layout = blk["c1"].conv_operand()
                ops.conv2d_affine(self.code, n, hcur, hcur, cin, 1, 1, hcur, hcur, planes, x, *xs, wsel, wsel.stride(0), t1,
                                  planes, gmax, sc[:, :c1n].contiguous(), sf[:, :c1n].contiguous(), None, True, None,
                                  w_layout=layout)
                idn = torch.empty((n * hcur * hcur, planes * 4), dtype=dt, device=dev)
                wsel, layout = blk["cd"].conv_operand()
                ops.conv2d_affine(self.code, n, hcur, hcur, cin, 1, 1, hcur, hcur, planes * 4, x, *xs, wsel, wsel.stride(0),
                                  idn, planes * 4, gmax, sc[:, c1n:].contiguous(), sf[:, c1n:].contiguous(), None, False,
                                  None, w_layout=layout)
            elif x_aff is not None:
                # first block on the stem's raw map: one Gram matrix -> the folded affines of conv1 and the downsample,
                # then one streaming pass each (bn1 + ReLU of the stem applied on the way in)
                wcat, gcat, bcat, eps, c1n = w["cat0"]
                x2d, gmax = x.view(-1, cin), gsz * hcur * hcur
                sc, sh = ops.bn_gram_affine(x2d, wcat, gmax, gcat, bcat, eps, x_aff)
                t1 = torch.empty((n, hcur, hcur, planes), dtype=dt, device=dev)
                ops.conv1x1_affine(x2d, blk["c1"].rows, gmax, sc[:, :c1n].contiguous(), sh[:, :c1n].contiguous(),
                                   t1.view(-1, planes), None, True, x_aff)
                idn = torch.empty((x2d.shape[0], planes * 4), dtype=dt, device=dev)
                ops.conv1x1_affine(x2d, blk["cd"].rows, gmax, sc[:, c1n:].contiguous(), sh[:, c1n:].contiguous(), idn, None,
                                   False, x_aff)
                x_aff = None
            elif (bi == 0 and self.h2 and "cat0" in w and self.bn_mode == "batch" and uniform and not s1 and not sd
                  and self._gram_h2_ok(cin, planes * 4, 1, 1, gsz * hcur * hcur)):
                # AVS_F16X2, first block: conv1 (64 -> 64) and the downsample (64 -> 256) read the same finished input, so
                # ONE Gram matrix gives both BatchNorms' affines and each layer is one streaming pass (conv1 would
                # otherwise take convolution + statistics + an apply pass over its output)
                wcat, gcat, bcat, eps, c1n = w["cat0"]
                gmax = gsz * hcur * hcur
                sc, sf = ops.bn_gram_affine_h2(x.view(-1, cin), wcat, gmax, gcat, bcat, eps)
                geom, xs, _ = self._nhwc_geom(n, hcur, cin, 1, 1, 0, planes)
                t1 = torch.empty((n, hcur, hcur, planes), dtype=dt, device=dev)
                wsel, layout = blk["c1"].conv_operand()
                ops.conv2d_affine(self.code, n, hcur, hcur, cin, 1, 1, hcur, hcur, planes, x, *xs, wsel, wsel.stride(0), t1,
                                  planes, gmax, sc[:, :c1n].contiguous(), sf[:, :c1n].contiguous(), None, True, None,
                                  w_layout=layout)
                idn = torch.empty((n * hcur * hcur, planes * 4), dtype=dt, device=dev)
                wsel, layout = blk["cd"].conv_operand()
                ops.conv2d_affine(self.code, n, hcur, hcur, cin, 1, 1, hcur, hcur, planes * 4, x, *xs, wsel, wsel.stride(0),
                                  idn, planes * 4, gmax, sc[:, c1n:].contiguous(), sf[:, c1n:].contiguous(), None, False,
                                  None, w_layout=layout)
            else:
                geom, xs, _ = self._nhwc_geom(n, hcur, cin, 1, 1, 0, planes)
                t1 = self._conv_bn(geom, xs, x, blk["c1"], blk["b1"], groups, local=s1)
            # bn2 + ReLU ride in conv3's input staging when conv3 takes the two-pass kernel: conv2 then only
            # writes its raw output and statistics (no apply pass over it)
            gmax3 = gsz * hout * hout
            defer2 = (self.defer_bn_apply and self.bn_mode == "batch" and uniform and not s2 and not s3 and
                      ((dt == torch.bfloat16 and planes <= 512 and self._twopass_ok(planes, planes * 4, 1, 1, gmax3))
                       or self._gram_h2_ok(planes, planes * 4, 1, 1, gmax3)))
            geom, xs, _ = self._nhwc_geom(n, hcur, planes, 3, s, 1, planes)
            t2 = self._conv_bn(geom, xs, t1, blk["c2"], blk["b2"], groups, local=s2, defer=defer2)
            aff2 = None
            if defer2:
                t2, aff2 = t2
            del t1
            affd = None
            if idn is not None:
                pass
            elif "cd" in blk:
                # a downsample branch that would take convolution + statistics + apply keeps its output RAW when conv3
                # is the one-pass form: its BatchNorm is folded into conv3's residual add
                deferd = (defer2 and self.defer_res_apply and not sd and self.gram_stats
                          and ops.gram_supported(planes, planes * 4)
                          and not self._twopass_ok(cin, planes * 4, 1, s, gsz * hout * hout))
                geom, xs, _ = self._nhwc_geom(n, hcur, cin, 1, s, 0, planes * 4)
                idn = self._conv_bn(geom, xs, x, blk["cd"], blk["bd"], groups, relu=False, local=sd, defer=deferd)
                if deferd:
                    idn, affd = idn
                idn = idn.view(-1, planes * 4)
            else:
                idn = x if isinstance(x, ops.P8) else x.view(-1, cin)
            geom, xs, _ = self._nhwc_geom(n, hout, planes, 1, 1, 0, planes * 4)
            # 3-byte storage of this block's output: conv3 is the one-pass form here and so is the next block's (same
            # layer), whose conv1 is the convolution + statistics form on dense rows
            p8 = (bi in self.p8_blocks and self.bn_mode == "batch" and uniform and not s3 and self.block_hook is None
                  and (aff2 is not None or self._gram_h2_ok(planes, planes * 4, 1, 1, gmax3)) and (planes * 4) % 32 == 0
                  # ... and the READER agrees: the next block's conv1 (dense 1x1 on this output, convolution + statistics)
                  and bi + 1 < len(w["blocks"]) and w["blocks"][bi + 1]["stride"] == 1 and "cd" not in w["blocks"][bi + 1]
                  and ops.conv_bnstats_p8_input_ok(self.code, n, hout, planes * 4, w["blocks"][bi + 1]["planes"], gmax3))
            x = self._conv_bn(geom, xs, t2, blk["c3"], blk["b3"], groups, residual=idn, relu=True, local=s3,
                              in_affine=aff2, res_affine=affd, out_p8=p8)
            del t2, idn
            if self.block_hook is not None:   # study tools only (tools/h3_storage_study.py): a block output's storage format
                x = self.block_hook(bi, x)
            hcur = hout
        return ops.global_avgpool(x, out, code=self.ecode)


# ============================================================================ Inception-v3 container
class BasicConv2d(_NoTorchForward):
    def __init__(self, cin, cout, **kw):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, bias=False, **kw)
        self.bn = nn.BatchNorm2d(cout, eps=0.001)


class InceptionA(_NoTorchForward):
    def __init__(self, cin, pool_features):
        super().__init__()
        self.branch1x1 = BasicConv2d(cin, 64, kernel_size=1)
        self.branch5x5_1 = BasicConv2d(cin, 48, kernel_size=1)
        self.branch5x5_2 = BasicConv2d(48, 64, kernel_size=5, padding=2)
        self.branch3x3dbl_1 = BasicConv2d(cin, 64, kernel_size=1)
        self.branch3x3dbl_2 = BasicConv2d(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = BasicConv2d(96, 96, kernel_size=3, padding=1)
        self.branch_pool = BasicConv2d(cin, pool_features, kernel_size=1)


class InceptionB(_NoTorchForward):
    def __init__(self, cin):
        super().__init__()
        self.branch3x3 = BasicConv2d(cin, 384, kernel_size=3, stride=2)
        self.branch3x3dbl_1 = BasicConv2d(cin, 64, kernel_size=1)
        self.branch3x3dbl_2 = BasicConv2d(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = BasicConv2d(96, 96, kernel_size=3, stride=2)


class InceptionC(_NoTorchForward):
    def __init__(self, cin, c7):
        super().__init__()
        self.branch1x1 = BasicConv2d(cin, 192, kernel_size=1)
        self.branch7x7_1 = BasicConv2d(cin, c7, kernel_size=1)
        self.branch7x7_2 = BasicConv2d(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7_3 = BasicConv2d(c7, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_1 = BasicConv2d(cin, c7, kernel_size=1)
        self.branch7x7dbl_2 = BasicConv2d(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_3 = BasicConv2d(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7dbl_4 = BasicConv2d(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_5 = BasicConv2d(c7, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch_pool = BasicConv2d(cin, 192, kernel_size=1)


class InceptionD(_NoTorchForward):
    def __init__(self, cin):
        super().__init__()
        self.branch3x3_1 = BasicConv2d(cin, 192, kernel_size=1)
        self.branch3x3_2 = BasicConv2d(192, 320, kernel_size=3, stride=2)
        self.branch7x7x3_1 = BasicConv2d(cin, 192, kernel_size=1)
        self.branch7x7x3_2 = BasicConv2d(192, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7x3_3 = BasicConv2d(192, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7x3_4 = BasicConv2d(192, 192, kernel_size=3, stride=2)


class InceptionE(_NoTorchForward):
    def __init__(self, cin):
        super().__init__()
        self.branch1x1 = BasicConv2d(cin, 320, kernel_size=1)
        self.branch3x3_1 = BasicConv2d(cin, 384, kernel_size=1)
        self.branch3x3_2a = BasicConv2d(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3_2b = BasicConv2d(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch3x3dbl_1 = BasicConv2d(cin, 448, kernel_size=1)
        self.branch3x3dbl_2 = BasicConv2d(448, 384, kernel_size=3, padding=1)
        self.branch3x3dbl_3a = BasicConv2d(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3dbl_3b = BasicConv2d(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch_pool = BasicConv2d(cin, 192, kernel_size=1)


class InceptionAux(_NoTorchForward):
    """Present in the checkpoint the reference loads (aux_logits=True at construction,
    extractors.py:26) but never executed (aux_logits set False at :36, eval mode)."""

    def __init__(self, cin, num_classes):
        super().__init__()
        self.conv0 = BasicConv2d(cin, 128, kernel_size=1)
        self.conv1 = BasicConv2d(128, 768, kernel_size=5)
        self.fc = nn.Linear(768, num_classes)


class Inception3(_NoTorchForward):
    """torchvision ``Inception3`` module tree with ``fc = Identity`` and ``transform_input = True``
    (what ``inception_v3(pretrained=True)`` sets)."""

    def __init__(self):
        super().__init__()
        self.aux_logits = True
        self.transform_input = True
        self.Conv2d_1a_3x3 = BasicConv2d(3, 32, kernel_size=3, stride=2)
        self.Conv2d_2a_3x3 = BasicConv2d(32, 32, kernel_size=3)
        self.Conv2d_2b_3x3 = BasicConv2d(32, 64, kernel_size=3, padding=1)
        self.maxpool1 = nn.MaxPool2d(kernel_size=3, stride=2)
        self.Conv2d_3b_1x1 = BasicConv2d(64, 80, kernel_size=1)
        self.Conv2d_4a_3x3 = BasicConv2d(80, 192, kernel_size=3)
        self.maxpool2 = nn.MaxPool2d(kernel_size=3, stride=2)
        self.Mixed_5b = InceptionA(192, 32)
        self.Mixed_5c = InceptionA(256, 64)
        self.Mixed_5d = InceptionA(288, 64)
        self.Mixed_6a = InceptionB(288)
        self.Mixed_6b = InceptionC(768, 128)
        self.Mixed_6c = InceptionC(768, 160)
        self.Mixed_6d = InceptionC(768, 160)
        self.Mixed_6e = InceptionC(768, 192)
        self.AuxLogits = InceptionAux(768, 1000)
        self.Mixed_7a = InceptionD(768)
        self.Mixed_7b = InceptionE(1280)
        self.Mixed_7c = InceptionE(2048)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.dropout = nn.Dropout(p=0.5)
        self.fc = nn.Identity()
        # Synthetic init (no pretrained weights offline): variance-preserving, so that eval-mode
        # BatchNorm with running stats (0, 1) keeps activations O(1) through 94 convolutions.
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_in", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self.eval()


INCEPTION_TRANSFORM = (0.229 / 0.5, 0.224 / 0.5, 0.225 / 0.5,
                       (0.485 - 0.5) / 0.5, (0.456 - 0.5) / 0.5, (0.406 - 0.5) / 0.5)


class InceptionV3Runner:
    """uint8 frames [N,299,299,3] -> fp32 [N,2048]; eval-mode BatchNorm folded into the convolutions."""

    def __init__(self, net, dtype=torch.float32, f32_split=False):
        """f32_split: as ResNet50Runner (True = AVS_F32_SPLIT, "f16x2" = AVS_F16X2 storage and arithmetic)."""
        self.net, self.dtype = net, dtype
        self.h2 = f32_split == "f16x2" and dtype == torch.float32
        self.f32_split = "f16x2" if self.h2 else (bool(f32_split) and dtype == torch.float32)
        self.ecode = ops.dtype_code(dtype, "f16x2") if self.h2 else ops.dtype_code(dtype)   # storage format
        self.pool_after_conv = True   # branch_pool: 1x1 convolution first, average pooling on its (narrow) output
        self.split_tail_columns = False  # cout = 128 k + r (r <= 64): two launches instead of a mostly empty last column tile
        #                                  (measured: no gain - these layers are not bound by the matrix work; off)
        self.stack_pool_head = True      # ... branch_pool's convolution too (its columns without bias / ReLU: they follow the pooling)
        self.stack_heads = True          # AVS_F16X2: the 1x1 heads of a block that read the block input run as ONE contraction
        #                                  over their stacked filters (avs_conv2d_nhwc_split): the input is fetched once
        self._key = None
        self._w = None

    def _fold(self, bc, stem_px=None):
        conv, bn = bc.conv, bc.bn
        with torch.no_grad():
            s = bn.weight.float() / torch.sqrt(bn.running_var.float() + bn.eps)
            wt = conv.weight.float() * s.view(-1, 1, 1, 1)
            bias = (bn.bias.float() - bn.running_mean.float() * s).contiguous()
            rows = _stem_weight(wt, stem_px, self.dtype) if stem_px else _ohwi(wt, self.dtype)
            w = _W(ops.f16x2_pack(rows) if self.h2 else rows)
            # column tiles are 128 wide (64 for cout <= 64): a layer with cout = 128 k + r, 0 < r <= 64 (192, 160, 320, 448:
            # most of Inception-v3) would spend a whole 128-column tile's matrix work on its last r columns - 25 - 37 % of
            # the layer.  Such a layer runs as TWO launches, the first 128 k columns on the wide tile and the last r on
            # the 64-column one, each writing its own channel slice: the same outputs, bit for bit
            parts = None
            cout = conv.out_channels
            r = cout % 128
            if self.split_tail_columns and not stem_px and cout > 128 and 0 < r <= 64:
                parts = []
                for a, b in ((0, cout - r), (cout - r, cout)):
                    pr = rows[a:b].contiguous()
                    parts.append((a, b, _W(ops.f16x2_pack(pr) if self.h2 else pr), bias[a:b].contiguous()))
        kh, kw = conv.kernel_size
        return {"w": w, "b": bias, "kh": kh, "kw": kw, "s": conv.stride[0], "ph": conv.padding[0],
                "pw": conv.padding[1], "cout": conv.out_channels, "parts": parts}

    def _prepare(self):
        key = tuple((p.data_ptr(), p._version) for p in self.net.parameters()) + \
            tuple((b.data_ptr(), b._version) for b in self.net.buffers()) + (self.dtype,)
        if self._w is not None and key == self._key:
            return self._w
        w = {}
        for name, m in self.net.named_modules():
            if isinstance(m, BasicConv2d) and not name.startswith("AuxLogits"):
                w[name] = self._fold(m, stem_px=4 if name == "Conv2d_1a_3x3" else None)
        self._w, self._key = w, key
        return w

    # conv + folded BN + ReLU into `out` (an NHWC view, possibly a channel slice)
    def _conv(self, w, name, x, out=None):
        c = w[name]
        n, h, ww, _ = x.shape
        ho = (h + 2 * c["ph"] - c["kh"]) // c["s"] + 1
        wo = (ww + 2 * c["pw"] - c["kw"]) // c["s"] + 1
        if out is None:
            out = torch.empty((n, ho, wo, c["cout"]), dtype=self.dtype, device=x.device)
        if c["parts"] is not None:
            for a, b, wp, bp in c["parts"]:
                wsel, layout = wp.conv_operand()
                ops.conv2d(x, wsel, c["kh"], c["kw"], c["s"], (c["ph"], c["pw"]), out[..., a:b], bp, ops.ACT_RELU,
                           split=self.f32_split, w_layout=layout)
            return out
        wsel, layout = c["w"].conv_operand()
        return ops.conv2d(x, wsel, c["kh"], c["kw"], c["s"], (c["ph"], c["pw"]), out, c["b"], ops.ACT_RELU,
                          split=self.f32_split, w_layout=layout)

    def _heads(self, w, p, x, first, others, first_out, pool=None):
        """The block's 1x1 convolutions that read x: `first` (or None) writes first_out (its slice of the block's
        concatenated output), `others` feed further convolutions, `pool` (or None) is branch_pool's convolution, whose
        average pooling, bias and ReLU follow (_pool_branch).  Returns the NHWC views of the others' outputs (+ the pool
        head's raw output, or None when it was not run here).  With stack_heads (AVS_F16X2): ONE contraction over the
        stacked filters, two destinations, the pool head's columns without bias / ReLU."""
        names = [p + "." + nm for nm in others]
        couts = [w[nm]["cout"] for nm in names]
        n, h, ww, _ = x.shape
        if not (self.h2 and self.stack_heads) or any(w[nm]["parts"] is not None for nm in names):
            if first is not None:
                self._conv(w, p + "." + first, x, first_out)
            return [self._conv(w, nm, x) for nm in names] + [None]
        with_pool = (pool is not None and first is not None and self.pool_after_conv and self.stack_pool_head
                     and w[p + "." + pool]["parts"] is None)
        key = ("stack", p, first, tuple(others), with_pool)
        st = w.get(key)
        if st is None:
            members = ([p + "." + first] if first is not None else []) + names
            # (an AVS_F16X2 row is packed by itself - runs of 8 inside the row: the stacked image is the rows one after another)
            rows = [w[nm]["w"].rows for nm in members]
            bias = [w[nm]["b"] for nm in members]
            if with_pool:
                rows.append(w[p + "." + pool]["w"].rows)
                bias.append(torch.zeros_like(w[p + "." + pool]["b"]))      # its bias comes after the pooling
            st = (_W(torch.cat(rows).contiguous()), torch.cat(bias).contiguous())
            w[key] = st
        wst, bst = st
        cpool = w[p + "." + pool]["cout"] if with_pool else 0
        tmp = torch.empty((n, h, ww, sum(couts) + cpool), dtype=self.dtype, device=x.device)
        wsel, layout = wst.conv_operand()
        if first is not None:
            ops.conv2d_split(x, wsel, first_out, w[p + "." + first]["cout"], tmp, bst, ops.ACT_RELU, w_layout=layout,
                             relu_cols=(wst.rows.shape[0] - cpool) if with_pool else 0)
        else:
            ops.conv2d(x, wsel, 1, 1, 1, (0, 0), tmp, bst, ops.ACT_RELU, split=self.f32_split, w_layout=layout)
        views, o = [], 0
        for c in couts:
            views.append(tmp[..., o:o + c])
            o += c
        views.append(tmp[..., o:o + cpool] if with_pool else None)
        return views

    def _pool_branch(self, w, name, x, out, z=None):
        """branch_pool = avg_pool2d(3, 1, 1) -> 1x1 conv -> folded BN -> ReLU, run as 1x1 conv (no bias) -> average ->
        + bias -> ReLU: the two linear maps commute (count_include_pad's zero padding included), and the pooling pass
        then moves cout (32-192) instead of cin (192-2048) channels."""
        if not self.pool_after_conv:
            return self._conv(w, name, self._pool(x, "avg", 3, 1, 1), out)
        c = w[name]
        n, h, ww, _ = x.shape
        if z is None:   # (else: the raw 1x1 output came out of the block's stacked-heads contraction)
            z = torch.empty((n, h, ww, c["cout"]), dtype=self.dtype, device=x.device)
            for a, b, wp, _ in (c["parts"] or [(0, c["cout"], c["w"], None)]):
                wsel, layout = wp.conv_operand()
                ops.conv2d(x, wsel, 1, 1, 1, (0, 0), z[..., a:b], None, ops.ACT_NONE, split=self.f32_split, w_layout=layout)
        return ops.pool2d(z, "avg", 3, 1, 1, out, c["b"], ops.ACT_RELU, code=self.ecode)

    def _pool(self, x, mode, k, s, p, out=None):
        n, h, ww, c = x.shape
        ho, wo = (h + 2 * p - k) // s + 1, (ww + 2 * p - k) // s + 1
        if out is None:
            out = torch.empty((n, ho, wo, c), dtype=self.dtype, device=x.device)
        return ops.pool2d(x, mode, k, s, p, out, code=self.ecode)

    def _cat_buffer(self, x, channels, stride=1):
        n, h, ww, _ = x.shape
        if stride == 2:
            h, ww = (h - 3) // 2 + 1, (ww - 3) // 2 + 1
        buf = torch.empty((n, h, ww, sum(channels)), dtype=self.dtype, device=x.device)
        offs = [0]
        for c in channels:
            offs.append(offs[-1] + c)
        return buf, [buf[..., offs[i]:offs[i + 1]] for i in range(len(channels))]

    def _block_a(self, w, p, x, pf):
        buf, (o1, o5, o3, op) = self._cat_buffer(x, [64, 64, 96, pf])
        t5, t3, zp = self._heads(w, p, x, "branch1x1", ["branch5x5_1", "branch3x3dbl_1"], o1, pool="branch_pool")
        self._conv(w, p + ".branch5x5_2", t5, o5)
        t = self._conv(w, p + ".branch3x3dbl_2", t3)
        self._conv(w, p + ".branch3x3dbl_3", t, o3)
        self._pool_branch(w, p + ".branch_pool", x, op, zp)
        return buf

    def _block_b(self, w, p, x):
        buf, (o3, od, op) = self._cat_buffer(x, [384, 96, x.shape[3]], stride=2)
        self._conv(w, p + ".branch3x3", x, o3)
        t = self._conv(w, p + ".branch3x3dbl_2", self._conv(w, p + ".branch3x3dbl_1", x))
        self._conv(w, p + ".branch3x3dbl_3", t, od)
        self._pool(x, "max", 3, 2, 0, op)
        return buf

    def _block_c(self, w, p, x):
        buf, (o1, o7, od, op) = self._cat_buffer(x, [192, 192, 192, 192])
        t7, td, zp = self._heads(w, p, x, "branch1x1", ["branch7x7_1", "branch7x7dbl_1"], o1, pool="branch_pool")
        t = self._conv(w, p + ".branch7x7_2", t7)
        self._conv(w, p + ".branch7x7_3", t, o7)
        t = td
        for i in (2, 3, 4):
            t = self._conv(w, f"{p}.branch7x7dbl_{i}", t)
        self._conv(w, p + ".branch7x7dbl_5", t, od)
        self._pool_branch(w, p + ".branch_pool", x, op, zp)
        return buf

    def _block_d(self, w, p, x):
        buf, (o3, o7, op) = self._cat_buffer(x, [320, 192, x.shape[3]], stride=2)
        t3, t7, _ = self._heads(w, p, x, None, ["branch3x3_1", "branch7x7x3_1"], None)
        self._conv(w, p + ".branch3x3_2", t3, o3)
        t = self._conv(w, p + ".branch7x7x3_2", t7)
        t = self._conv(w, p + ".branch7x7x3_3", t)
        self._conv(w, p + ".branch7x7x3_4", t, o7)
        self._pool(x, "max", 3, 2, 0, op)
        return buf

    def _block_e(self, w, p, x):
        buf, (o1, o3a, o3b, oda, odb, op) = self._cat_buffer(x, [320, 384, 384, 384, 384, 192])
        t, td, zp = self._heads(w, p, x, "branch1x1", ["branch3x3_1", "branch3x3dbl_1"], o1, pool="branch_pool")
        self._conv(w, p + ".branch3x3_2a", t, o3a)
        self._conv(w, p + ".branch3x3_2b", t, o3b)
        t = self._conv(w, p + ".branch3x3dbl_2", td)
        self._conv(w, p + ".branch3x3dbl_3a", t, oda)
        self._conv(w, p + ".branch3x3dbl_3b", t, odb)
        self._pool_branch(w, p + ".branch_pool", x, op, zp)
        return buf

    def forward(self, frames_u8, out=None):
        n, h, w_, _ = frames_u8.shape
        if (h, w_) != (299, 299):
            raise ValueError("InceptionV3Runner expects 299x299 frames (resize first)")
        if n == 0:
            return torch.zeros((0, 2048), dtype=torch.float32, device=frames_u8.device)
        w = self._prepare()
        dt, dev = self.dtype, frames_u8.device
        # (x/255 - mean)/std (extractors.py:151-153) then transform_input; one spare zero
        # pixel on the right so the stem's 4-pixel runs stay inside the row.
        affine = INCEPTION_TRANSFORM if self.net.transform_input else None
        x0 = ops.frames_normalize(frames_u8, dt, 255.0, RESNET_MEAN, RESNET_STD, 299, 300, 0, 0, affine, code=self.ecode)
        c = w["Conv2d_1a_3x3"]
        x = torch.empty((n, 149, 149, 32), dtype=dt, device=dev)
        wsel, layout = c["w"].conv_operand()
        ops.conv2d_raw(ops.dtype_code(dt, self.f32_split), n, 299, 149, 16, 3, 1, 2, 1, 0, 0, 149, 149, 32, x0, 299 * 300 * 4, 300 * 4, 8,
                       wsel, wsel.stride(0), x, 32, c["b"], ops.ACT_RELU, algo_k=27, algo_in_elems=x0.numel(),
                       w_layout=layout)
        del x0
        x = self._conv(w, "Conv2d_2a_3x3", x)
        x = self._conv(w, "Conv2d_2b_3x3", x)
        x = self._pool(x, "max", 3, 2, 0)
        x = self._conv(w, "Conv2d_3b_1x1", x)
        x = self._conv(w, "Conv2d_4a_3x3", x)
        x = self._pool(x, "max", 3, 2, 0)
        x = self._block_a(w, "Mixed_5b", x, 32)
        x = self._block_a(w, "Mixed_5c", x, 64)
        x = self._block_a(w, "Mixed_5d", x, 64)
        x = self._block_b(w, "Mixed_6a", x)
        for name in ("Mixed_6b", "Mixed_6c", "Mixed_6d", "Mixed_6e"):
            x = self._block_c(w, name, x)
        x = self._block_d(w, "Mixed_7a", x)
        x = self._block_e(w, "Mixed_7b", x)
        x = self._block_e(w, "Mixed_7c", x)
        return ops.global_avgpool(x, out, code=self.ecode)
