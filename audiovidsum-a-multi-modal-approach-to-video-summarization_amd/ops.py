"""Tensor-level wrappers over the C-ABI: argument checking + launch on torch's
current HIP stream.  torch is used for device memory and streams only; every
computation below is a call into libavsum_hip.so.
"""
import ctypes
from ctypes import c_float, c_void_p

import torch

from . import _abi
from ._abi import (ACT_NONE, ACT_RELU, AVS_BF16, AVS_F16X2, AVS_F32, AVS_F32_SPLIT, BIAS_COL, BIAS_NONE, BIAS_ROW, check,
                   lib)

__all__ = [
    "ACT_NONE", "ACT_RELU", "linear", "gemm_nt_batched", "conv2d", "conv2d_raw", "conv2d_split", "conv_bnlocal_tile_rows", "conv_bncluster_ok", "cluster_exchange_errors", "lstm_split_errors", "conv1x1_bn", "conv1x1_gram_bn", "bn_gram_affine", "gram_supported", "frames_normalize", "pull_copy", "stem_conv_bn_pool", "stem_h2_operands", "stem_conv_pool_h2", "resize_bilinear",
    "bn_batch_stats", "bn_apply", "bn_maxpool", "pool2d", "global_avgpool", "segment_mean", "hsv_frame_diff", "reflect_pad", "stft_f64", "stft_mel_fused", "power_mel",
    "clamp_topdb", "stft_mel_max", "stft_mel_segmean", "stft_mel_segmean_batch", "fill", "quantize", "resample", "lstm", "mha_batchaxis", "score_head", "mhsa_flash", "softmax_rows", "cdist", "dtw_path",
    "gather_scale", "dtype_code", "f16x2_pack", "f16x2_unpack", "bn_gram_affine_h2", "conv2d_affine",
]


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


class LaunchProfiler:
    """Optional per-launch HIP-event timing of the contraction kernel (bench.py's roofline leg):
    events are recorded on the stream the kernel is launched on, read back after a synchronize.
    ``count_only`` just counts the launches it would time (to size the event pool before a timed region:
    creating events is the expensive part, so they are made up front and only recorded inside it)."""

    def __init__(self, count_only=False, prealloc=0, sample_every=1):
        """sample_every = n: bracket every n-th launch of each kind only (an event pair is a barrier between kernels:
        timing EVERY launch costs the pipeline ~3 %); all launches are still counted."""
        self.records = []  # (kind, dtype_code, algorithmic FLOPs (bytes for the HBM-bound kinds), algorithmic bytes, start, end, form)
        self.count_only = count_only
        self.count = 0
        self.sample_every = max(1, int(sample_every))
        self.seen = {}     # (kind, dtype) -> launches seen (timed or not)
        self._pool = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                      for _ in range(prealloc)]

    def events(self):
        if self._pool:
            return self._pool.pop()
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for kind, dt, flops, nbytes, s, e, _ in self.records:
            d = out.setdefault((kind, dt), {"launches": 0, "flops": 0.0, "bytes": 0.0, "ms": 0.0})
            d["launches"] += 1
            d["flops"] += flops
            d["bytes"] += nbytes
            d["ms"] += s.elapsed_time(e)
        return out

    def forms(self, kind, dt):
        """The timed launches of one kind split by the FORM the wrapper tagged them with (epilogue form x filter size of
        the contraction kernel): {form: {"launches", "flops", "bytes", "ms"}}."""
        torch.cuda.synchronize()
        out = {}
        for k, d_, flops, nbytes, s, e, form in self.records:
            if k != kind or d_ != dt:
                continue
            d = out.setdefault(form or "other", {"launches": 0, "flops": 0.0, "bytes": 0.0, "ms": 0.0})
            d["launches"] += 1
            d["flops"] += flops
            d["bytes"] += nbytes
            d["ms"] += s.elapsed_time(e)
        return out


_profiler = None


def set_profiler(p):
    global _profiler
    _profiler = p


def _timed(kind, dtype, flops, fn, nbytes=0.0, form=None):
    if _profiler is None:
        return fn()
    if _profiler.count_only:
        _profiler.count += 1
        return fn()
    seen = _profiler.seen.get((kind, dtype), 0)
    _profiler.seen[(kind, dtype)] = seen + 1
    if seen % _profiler.sample_every:
        return fn()
    s, e = _profiler.events()
    s.record()
    r = fn()
    e.record()
    _profiler.records.append((kind, dtype, float(flops), float(nbytes), s, e, form))
    return r


def _p(t, offset_elems=0):
    if t is None:
        return None
    return c_void_p(t.data_ptr() + offset_elems * t.element_size())


def dtype_code(dtype, split=False):
    """C-ABI dtype code of a torch dtype.  split (fp32 only): True = AVS_F32_SPLIT, fp32 operands contracted on the bf16
    matrix cores as hi*hi + hi*lo + lo*hi (the contraction entry points only: avs_conv2d_nhwc*, avs_gemm_nt);
    "f16x2" = AVS_F16X2: the tensor's 4-byte slots hold fp16 hi | lo runs (carried in float32-typed torch tensors:
    same shapes, strides and byte size; f16x2_pack / f16x2_unpack convert)."""
    if dtype == torch.float32:
        if split == "f16x2":
            return AVS_F16X2
        return AVS_F32_SPLIT if split else AVS_F32
    if dtype == torch.bfloat16:
        return AVS_BF16
    raise TypeError(f"unsupported compute dtype {dtype}")


def f16x2_pack(x):
    """fp32 tensor (contiguous, numel a multiple of 8, rows of the innermost axis multiples of 8) -> the AVS_F16X2
    image of the same shape, carried in a float32-typed tensor (opaque slots)."""
    _dev(x)
    _f32(x, "x")
    if not x.is_contiguous() or x.numel() % 8 or (x.dim() and x.shape[-1] % 8):
        raise ValueError("f16x2_pack: contiguous fp32 tensor whose innermost extent is a multiple of 8")
    out = torch.empty_like(x)
    check(lib().avs_f16x2_pack_f32(_p(x), _p(out), x.numel(), _stream()), "avs_f16x2_pack_f32")
    return out


def f16x2_unpack(x):
    """The fp32 values (hi + lo) of an AVS_F16X2 tensor (contiguous)."""
    _dev(x)
    _f32(x, "x")
    if not x.is_contiguous() or x.numel() % 8:
        raise ValueError("f16x2_unpack: contiguous tensor, numel a multiple of 8")
    out = torch.empty_like(x)
    check(lib().avs_f16x2_unpack_f32(_p(x), _p(out), x.numel(), _stream()), "avs_f16x2_unpack_f32")
    return out


class P8:
    """An AVS_F16P8 activation (fp16 hi + 8-bit remainder, 3 bytes per value; include/avsum_hip.h): ``data`` is the
    uint8 image [rows, 3 * c], ``shape`` the logical NHWC shape.  Written by conv2d_affine(out_p8=...), read by it as a
    residual and by conv2d_raw(bnstats=...) as the input (the wide block outputs of ResNet layers 1-2)."""

    def __init__(self, data, shape):
        self.data, self.shape = data, tuple(shape)

    @property
    def device(self):
        return self.data.device

    @classmethod
    def empty(cls, shape, device):
        rows = 1
        for e in shape[:-1]:
            rows *= e
        if shape[-1] % 16:
            raise ValueError("AVS_F16P8 needs a channel count that is a multiple of 16")
        return cls(torch.empty((rows, 3 * shape[-1]), dtype=torch.uint8, device=device), shape)


def f16p8_pack(x):
    """fp32 [..., c] (contiguous, c a multiple of 16) -> P8 of the same logical shape."""
    _dev(x)
    _f32(x, "x")
    if not x.is_contiguous() or x.shape[-1] % 16:
        raise ValueError("f16p8_pack: contiguous fp32 tensor whose innermost extent is a multiple of 16")
    out = P8.empty(x.shape, x.device)
    check(lib().avs_f16p8_pack_f32(_p(x), _p(out.data), x.numel(), _stream()), "avs_f16p8_pack_f32")
    return out


def f16p8_unpack(t):
    """The fp32 values of a P8 tensor."""
    out = torch.empty(t.shape, dtype=torch.float32, device=t.device)
    check(lib().avs_f16p8_unpack_f32(_p(t.data), _p(out), out.numel(), _stream()), "avs_f16p8_unpack_f32")
    return out


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise ValueError("avsum HIP ops need device tensors (there is no CPU fallback)")


def _f32(t, name):
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")


def _rowmajor2d(t, name):
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError(f"{name} must be 2-D with unit column stride, got shape {tuple(t.shape)} strides {t.stride()}")


# --------------------------------------------------------------------------- GEMM / conv
def gemm_nt_batched(dtype, m, n, k, a, a_off, lda, sa, b, b_off, ldb, sb, c, c_off, ldc, sc, bias=None,
                    bias_mode=BIAS_NONE, sbias=0, alpha=1.0, act=ACT_NONE, batch=1):
    """Raw strided-batched C = act(alpha * A.B^T + bias); offsets/strides in elements."""
    _dev(a, b, c, bias)
    _timed("gemm", dtype, 2.0 * m * n * k * batch, lambda: check(
        lib().avs_gemm_nt(dtype, m, n, k, _p(a, a_off), lda, sa, _p(b, b_off), ldb, sb, _p(c, c_off), ldc, sc,
                          _p(bias), bias_mode, sbias, float(alpha), act, batch, _stream()), "avs_gemm_nt"))


def linear(x, w, bias=None, act=ACT_NONE, out=None, alpha=1.0):
    """out[M,N] = act(alpha * x[M,K] . w[N,K]^T + bias[N]) — nn.Linear semantics."""
    _rowmajor2d(x, "x")
    _rowmajor2d(w, "w")
    if x.dtype != w.dtype:
        raise TypeError(f"x {x.dtype} and w {w.dtype} differ")
    m, k = x.shape
    n = w.shape[0]
    if w.shape[1] != k:
        raise ValueError(f"x [{m},{k}] vs w {tuple(w.shape)}")
    if out is None:
        out = torch.empty((m, n), dtype=x.dtype, device=x.device)
    _rowmajor2d(out, "out")
    if out.shape != (m, n) or out.dtype != x.dtype:
        raise ValueError("out has the wrong shape or dtype")
    if bias is not None:
        _f32(bias, "bias")
        if bias.numel() != n or not bias.is_contiguous():
            raise ValueError("bias must be contiguous with N entries")
    lda = x.stride(0) if m > 1 else k
    ldb = w.stride(0) if n > 1 else k
    ldc = out.stride(0) if m > 1 else n
    gemm_nt_batched(dtype_code(x.dtype), m, n, k, x, 0, lda, 0, w, 0, ldb, 0, out, 0, ldc, 0, bias,
                    BIAS_COL if bias is not None else BIAS_NONE, 0, alpha, act, 1)
    return out


def conv_bnlocal_tile_rows(dtype, n, h, w, cin, kh, kw, sh, sw, ph, pw, ho, wo, cout, x_img_stride,
                           x_row_stride, x_px_stride, w_row_stride, y_px_stride, rows_per_group):
    """Rows of a 256-row tile the one-launch convolution + BatchNorm (avs_conv2d_nhwc_bnlocal) uses for this shape,
    or None when the library does not take it in that form (the caller then runs the unfused sequence)."""
    d = _abi.ConvDesc(dtype, n, h, w, cin, kh, kw, sh, sw, ph, pw, ho, wo, cout, x_img_stride, x_row_stride,
                      x_px_stride, w_row_stride, y_px_stride, ACT_NONE, 1.0)
    r = lib().avs_conv2d_bnlocal_tile_rows(ctypes.byref(d), int(rows_per_group))
    if r == _abi.E_UNSUPPORTED:
        return None
    if r < 0:
        check(int(r), "avs_conv2d_bnlocal_tile_rows")
    return int(r)


def conv_bncluster_ok(dtype, n, h, w, cin, kh, kw, sh, sw, ph, pw, ho, wo, cout, x_img_stride, x_row_stride, x_px_stride,
                      w_row_stride, y_px_stride, rows_per_group, cluster):
    """Does the library take this convolution + BatchNorm as ONE launch with groups of `cluster` tiles (the clustered
    tile-local form, avs_conv2d_nhwc_bncluster)?"""
    d = _abi.ConvDesc(dtype, n, h, w, cin, kh, kw, sh, sw, ph, pw, ho, wo, cout, x_img_stride, x_row_stride,
                      x_px_stride, w_row_stride, y_px_stride, ACT_NONE, 1.0)
    r = lib().avs_conv2d_bncluster_workspace_bytes(ctypes.byref(d), int(rows_per_group), int(cluster))
    if r == _abi.E_UNSUPPORTED:
        return False
    if r < 0:
        check(int(r), "avs_conv2d_bncluster_workspace_bytes")
    return True


def conv_bnstats_p8_input_ok(dtype, n, h, cin, cout, rows_per_group):
    """Does avs_conv2d_nhwc_bnstats take a dense 1x1 / stride-1 convolution [n,h,h,cin] -> cout with an AVS_F16P8 INPUT?
    (the planner's question before it stores a block output in that format: the next block's conv1 is the reader)"""
    d = _abi.ConvDesc(dtype, n, h, h, cin, 1, 1, 1, 1, 0, 0, h, h, cout, h * h * cin, h * cin, cin, cin, cout, ACT_NONE, 1.0,
                      0, 0, _abi.X_F16P8)
    r = lib().avs_conv2d_bnstats_workspace_bytes(ctypes.byref(d), int(rows_per_group))
    if r == _abi.E_UNSUPPORTED:
        return False
    if r < 0:
        check(int(r), "avs_conv2d_bnstats_workspace_bytes")
    return True


class _Exchange:
    """The granule buffer of avs_conv2d_nhwc_bncluster on one (device, stream): zeroed once, epochs count up per call."""

    def __init__(self):
        self.buf, self.epoch = None, 0

    def get(self, device, nbytes):
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = torch.zeros(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
            self.epoch = 0
        self.epoch += 1
        if self.epoch >= 0xFFFFFFF0:      # (32-bit tags: start over on a cleared buffer)
            self.buf.zero_()
            self.epoch = 1
        return self.buf, self.epoch


_exchanges = {}


def cluster_exchange_errors(device):
    """Waves whose bounded partner wait ran out in the clustered BatchNorm launches on this device's current stream so far
    (0 after healthy launches); reads the counter back (a host sync: for tests and debugging)."""
    ex = _exchanges.get(_ws_key(device))
    return 0 if ex is None or ex.buf is None else int(ex.buf[:4].view(torch.int32).item())


_stats_ws = {}


def _ws_key(device):
    # one scratch buffer per (device, stream): launches on ONE stream are ordered, so consecutive layers can share
    # it; two streams must not
    return (device, torch.cuda.current_stream(device).cuda_stream)


def _stats_workspace(device, nbytes):
    """Scratch for the per-tile partial sums of avs_conv2d_nhwc_bnstats: one buffer per device and stream, grown on
    demand."""
    key = _ws_key(device)
    ws = _stats_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _stats_ws[key] = ws
    return ws


def conv2d_raw(dtype, n, h, w, cin, kh, kw, sh, sw, ph, pw, ho, wo, cout, x, x_img_stride, x_row_stride, x_px_stride,
               wt, w_row_stride, y, y_px_stride, bias=None, act=ACT_NONE, alpha=1.0, x_off=0, y_off=0, algo_k=None,
               bnstats=None, bnlocal=None, algo_in_elems=None, w_layout=0, variant=0, cluster=1):
    """algo_k: the algorithmic reduction length when it differs from kh*kw*cin (zero-padded stem rows);
    algo_in_elems: input elements the launch reads when the geometry does not say (the re-viewed stem image).
    bnstats = (rows_per_group, gamma, beta, eps): the BatchNorm batch statistics of equal-sized row groups from the
    kernel's epilogue (no bias / activation; deterministic per-tile partial sums); returns the folded
    (scale, shift) [G, cout], or None when the library declines the shape (groups of < 64 rows).
    bnlocal = (rows_per_group, gamma, beta, eps, residual2d | None): the whole BatchNorm (+ residual, then `act`) in
    the convolution's launch (avs_conv2d_nhwc_bnlocal; shapes for which conv_bnlocal_tile_rows is not None).
    cluster > 1 (with bnlocal): a group = `cluster` tiles (avs_conv2d_nhwc_bncluster: AVS_F16X2, 14x14 maps).
    w_layout: _abi.AVS_W_ROWS (wt[cout, K]) or AVS_W_KSTEP32 (the image weights_kstep32() makes).
    variant: avs_conv_desc.variant (_abi.TILE_128 / TILE_256 | STAGING_GENERIC): a per-call override of the tile /
    staging choice, for tests and the study tools (the library has no global tuning state)."""
    formats = 0
    if isinstance(x, P8):   # AVS_F16P8 input: the 1x1 convolution + statistics form only (the library checks the shape)
        if bnstats is None:
            raise ValueError("an AVS_F16P8 input is taken by the convolution + statistics form (bnstats=...)")
        x, formats = x.data, _abi.X_F16P8
    _dev(x, wt, y, bias)
    d = _abi.ConvDesc(dtype, n, h, w, cin, kh, kw, sh, sw, ph, pw, ho, wo, cout, x_img_stride, x_row_stride,
                      x_px_stride, w_row_stride, y_px_stride, act, float(alpha), int(w_layout), int(variant), formats)
    flops = 2.0 * n * ho * wo * cout * (algo_k if algo_k is not None else kh * kw * cin)
    # algorithmic HBM bytes: the input map read once (the pixels a strided 1x1 skips are not needed), the output
    # written once (+ the residual read once); weights are negligible and L2-resident
    es = 2 if dtype == AVS_BF16 else 4
    touched = algo_in_elems if algo_in_elems is not None else n * (ho * wo if (kh == 1 and kw == 1) else h * w) * cin
    cbytes = (3.0 if formats else float(es)) * touched + float(es) * n * ho * wo * cout
    if bnlocal is not None:
        rpg, gamma, beta, eps, residual = bnlocal
        _dev(gamma, beta, residual)
        if bias is not None or bnstats is not None:
            raise ValueError("the one-launch convolution + BatchNorm takes no bias / separate statistics")
        if residual is not None:
            _rowmajor2d(residual, "residual")
            if residual.dtype != y.dtype or residual.shape != (n * ho * wo, cout):
                raise ValueError("residual must be [rows, cout] in the activation dtype")
        if cluster > 1:
            need = lib().avs_conv2d_bncluster_workspace_bytes(ctypes.byref(d), int(rpg), int(cluster))
            if need < 0:
                check(int(need), "avs_conv2d_bncluster_workspace_bytes")
            xbuf, epoch = _exchanges.setdefault(_ws_key(x.device), _Exchange()).get(x.device, need)
            _timed("conv", dtype, flops, lambda: check(
                lib().avs_conv2d_nhwc_bncluster(ctypes.byref(d), _p(x, x_off), _p(wt), _p(y, y_off), int(rpg), int(cluster),
                                                _p(gamma), _p(beta), float(eps), _p(residual),
                                                residual.stride(0) if residual is not None else 0, _p(xbuf), xbuf.numel(),
                                                int(epoch), _stream()),
                "avs_conv2d_nhwc_bncluster"),
                   cbytes + (float(es) * n * ho * wo * cout if residual is not None else 0.0),
                   form=f"clustered tile-local BatchNorm {kh}x{kw}")
            return None
        _timed("conv", dtype, flops, lambda: check(
            lib().avs_conv2d_nhwc_bnlocal(ctypes.byref(d), _p(x, x_off), _p(wt), _p(y, y_off), int(rpg), _p(gamma),
                                          _p(beta), float(eps), _p(residual),
                                          residual.stride(0) if residual is not None else 0, _stream()),
            "avs_conv2d_nhwc_bnlocal"),
               cbytes + (float(es) * n * ho * wo * cout if residual is not None else 0.0), form=f"tile-local BatchNorm {kh}x{kw}")
        return None
    if bnstats is None:
        _timed("conv", dtype, flops, lambda: check(
            lib().avs_conv2d_nhwc(ctypes.byref(d), _p(x, x_off), _p(wt), _p(bias), _p(y, y_off), _stream()),
            "avs_conv2d_nhwc"), cbytes, form=f"plain / bias+ReLU {kh}x{kw}")
        return None
    rpg, gamma, beta, eps = bnstats
    if bias is not None or act != ACT_NONE:
        raise ValueError("the fused-statistics convolution takes no bias / activation")
    rows = n * ho * wo
    groups = (rows + rpg - 1) // rpg
    need = lib().avs_conv2d_bnstats_workspace_bytes(ctypes.byref(d), int(rpg))
    if need == _abi.E_UNSUPPORTED:
        if formats:
            check(int(need), "avs_conv2d_bnstats_workspace_bytes")   # no other form reads an AVS_F16P8 input
        return None
    if need < 0:
        check(int(need), "avs_conv2d_bnstats_workspace_bytes")
    ws = _stats_workspace(x.device, need)
    scale = torch.empty((groups, cout), dtype=torch.float32, device=x.device)
    shift = torch.empty((groups, cout), dtype=torch.float32, device=x.device)
    _timed("conv", dtype, flops, lambda: check(
        lib().avs_conv2d_nhwc_bnstats(ctypes.byref(d), _p(x, x_off), _p(wt), _p(y, y_off), int(rpg), _p(gamma),
                                      _p(beta), float(eps), _p(scale), _p(shift), _p(ws), ws.numel(), _stream()),
        "avs_conv2d_nhwc_bnstats"), cbytes, form=f"convolution + statistics {kh}x{kw}")
    return scale, shift


def conv1x1_bn(x2d, wt, rows_per_group, gamma, beta, eps, out2d, residual=None, relu=True, in_affine=None):
    """Fused 1x1 convolution + batch-statistics BatchNorm (+residual, +ReLU), bf16.  x2d [rows, K] (rows =
    groups * rows_per_group), wt [N, K], out2d [rows, N]; every tensor row-major with unit column stride.
    in_affine = (scale, shift) fp32 [groups, K]: x2d is a RAW convolution output whose BatchNorm + ReLU is applied
    on the way in (avs_conv1x1_bn_in_bf16)."""
    _dev(x2d, wt, out2d, residual, gamma, beta)
    _rowmajor2d(x2d, "x")
    _rowmajor2d(wt, "w")
    _rowmajor2d(out2d, "out")
    rows, k = x2d.shape
    n = wt.shape[0]
    if x2d.dtype != torch.bfloat16 or wt.dtype != torch.bfloat16 or out2d.dtype != torch.bfloat16:
        raise TypeError("conv1x1_bn is the bf16 throughput path")
    if rows % rows_per_group or wt.shape[1] != k or out2d.shape != (rows, n):
        raise ValueError("conv1x1_bn: shapes do not match")
    if residual is not None:
        _rowmajor2d(residual, "residual")
    groups = rows // rows_per_group
    # algorithmic HBM bytes: read x once, write y once (+ read the residual); the second pass re-reads from L2
    nbytes = 2.0 * rows * (k + n * (2 if residual is not None else 1))
    if in_affine is not None:
        isc, ish = in_affine
        _dev(isc, ish)
        if isc.shape != (groups, k) or ish.shape != (groups, k) or not isc.is_contiguous() or not ish.is_contiguous():
            raise ValueError("in_affine must be contiguous fp32 [groups, K]")
        _timed("convbn", AVS_BF16, nbytes, lambda: check(
            lib().avs_conv1x1_bn_in_bf16(_p(x2d), x2d.stride(0), k, _p(isc), _p(ish), _p(wt), wt.stride(0), n,
                                         rows_per_group, groups, _p(gamma), _p(beta), float(eps), _p(residual),
                                         residual.stride(0) if residual is not None else 0, 1 if relu else 0,
                                         _p(out2d), out2d.stride(0), _stream()), "avs_conv1x1_bn_in_bf16"))
        return out2d
    _timed("convbn", AVS_BF16, nbytes, lambda: check(
        lib().avs_conv1x1_bn_bf16(_p(x2d), x2d.stride(0), k, _p(wt), wt.stride(0), n, rows_per_group, groups,
                                  _p(gamma), _p(beta), float(eps), _p(residual),
                                  residual.stride(0) if residual is not None else 0, 1 if relu else 0, _p(out2d),
                                  out2d.stride(0), _stream()), "avs_conv1x1_bn_bf16"))
    return out2d


def gram_supported(k, n):
    """Shapes avs_bn_gram_affine_bf16 takes (the expanding 1x1 layers of ResNet layers 1-2)."""
    return k in (64, 128) and n % 32 == 0


def bn_gram_affine(x2d, wt, rows_per_group, gamma, beta, eps, in_affine=None, store_input=False):
    """Folded BatchNorm affine (scale, shift) fp32 [groups, N] of y = a . wt^T per group of rows, computed from the
    Gram matrix of a WITHOUT forming y (avs_bn_gram_affine_bf16).  a = x2d, or bf16(relu(x2d * isc + ish)) with
    in_affine = (isc, ish) fp32 [groups, K].  K in {64, 128}, N a multiple of 32.
    store_input (with in_affine): x2d is overwritten IN PLACE with a."""
    _dev(x2d, wt, gamma, beta)
    _rowmajor2d(x2d, "x")
    _rowmajor2d(wt, "w")
    rows, k = x2d.shape
    n = wt.shape[0]
    if x2d.dtype != torch.bfloat16 or wt.dtype != torch.bfloat16:
        raise TypeError("bn_gram_affine is the bf16 throughput path")
    if rows % rows_per_group or wt.shape[1] != k:
        raise ValueError("bn_gram_affine: shapes do not match")
    groups = rows // rows_per_group
    isc = ish = None
    if in_affine is not None:
        isc, ish = in_affine
        _dev(isc, ish)
        if isc.shape != (groups, k) or ish.shape != (groups, k) or not isc.is_contiguous() or not ish.is_contiguous():
            raise ValueError("in_affine must be contiguous fp32 [groups, K]")
    scale = torch.empty((groups, n), dtype=torch.float32, device=x2d.device)
    shift = torch.empty((groups, n), dtype=torch.float32, device=x2d.device)
    # algorithmic HBM bytes: x read once
    if store_input and in_affine is None:
        raise ValueError("store_input needs in_affine")
    _timed("gram", AVS_BF16, 2.0 * rows * k * (2 if store_input else 1), lambda: check(
        lib().avs_bn_gram_affine_bf16(_p(x2d), x2d.stride(0), k, _p(isc), _p(ish), _p(wt), wt.stride(0), n,
                                      rows_per_group, groups, _p(gamma), _p(beta), float(eps), _p(scale), _p(shift),
                                      _p(x2d) if store_input else None, x2d.stride(0), _stream()),
        "avs_bn_gram_affine_bf16"))
    return scale, shift


def conv1x1_affine(x2d, wt, rows_per_group, scale, shift, out2d, residual=None, relu=True, in_affine=None,
                   res_affine=None):
    """out = act((a . wt^T) * scale[g] + shift[g] + residual), bf16, one streaming pass (avs_conv1x1_affine_bf16):
    scale / shift fp32 [groups, N] contiguous (a BatchNorm already folded, e.g. by bn_gram_affine); a = x2d or
    bf16(relu(x2d * isc + ish)) with in_affine = (isc, ish) fp32 [groups, K]; res_affine = (rsc, rsh) fp32
    [groups, N]: the residual is a raw convolution output, added as residual * rsc + rsh."""
    _dev(x2d, wt, out2d, residual, scale, shift)
    _rowmajor2d(x2d, "x")
    _rowmajor2d(wt, "w")
    _rowmajor2d(out2d, "out")
    rows, k = x2d.shape
    n = wt.shape[0]
    if x2d.dtype != torch.bfloat16 or wt.dtype != torch.bfloat16 or out2d.dtype != torch.bfloat16:
        raise TypeError("conv1x1_affine is the bf16 throughput path")
    if rows % rows_per_group or wt.shape[1] != k or out2d.shape != (rows, n):
        raise ValueError("conv1x1_affine: shapes do not match")
    groups = rows // rows_per_group
    for a in (scale, shift):
        if a.dtype != torch.float32 or a.shape != (groups, n) or not a.is_contiguous():
            raise ValueError("scale / shift must be contiguous fp32 [groups, N]")
    if residual is not None:
        _rowmajor2d(residual, "residual")
    isc, ish = in_affine if in_affine is not None else (None, None)
    if in_affine is not None:
        _dev(isc, ish)
        if isc.shape != (groups, k) or ish.shape != (groups, k) or not isc.is_contiguous() or not ish.is_contiguous():
            raise ValueError("in_affine must be contiguous fp32 [groups, K]")
    rsc, rsh = res_affine if res_affine is not None else (None, None)
    if res_affine is not None:
        _dev(rsc, rsh)
        if residual is None:
            raise ValueError("res_affine without a residual")
        for a in (rsc, rsh):
            if a.dtype != torch.float32 or a.shape != (groups, n) or not a.is_contiguous():
                raise ValueError("res_affine must be contiguous fp32 [groups, N]")
    # algorithmic HBM bytes: read x, write y (+ read the residual)
    nbytes = 2.0 * rows * (k + n * (2 if residual is not None else 1))
    _timed("convbn", AVS_BF16, nbytes, lambda: check(
        lib().avs_conv1x1_affine_bf16(_p(x2d), x2d.stride(0), k, _p(isc), _p(ish), _p(wt), wt.stride(0), n,
                                      rows_per_group, groups, _p(scale), _p(shift), _p(residual),
                                      residual.stride(0) if residual is not None else 0, _p(rsc), _p(rsh),
                                      1 if relu else 0, _p(out2d), out2d.stride(0), _stream()),
        "avs_conv1x1_affine_bf16"))
    return out2d


def conv1x1_gram_bn(x2d, wt, rows_per_group, gamma, beta, eps, out2d, residual=None, relu=True, in_affine=None,
                    res_affine=None, finish_input=False):
    """1x1 convolution + batch-statistics BatchNorm (+residual, +ReLU), bf16, same operands as conv1x1_bn, in ONE
    streaming pass over the output: the statistics come from the Gram matrix of the narrow input (bn_gram_affine),
    the convolution applies the folded affine in its epilogue (conv1x1_affine).
    finish_input (with in_affine): the statistics kernel overwrites x2d with the transformed input, and the convolution
    pass reads it finished instead of transforming it once per column slab."""
    if finish_input and in_affine is not None:
        scale, shift = bn_gram_affine(x2d, wt, rows_per_group, gamma, beta, eps, in_affine, store_input=True)
        in_affine = None
    else:
        scale, shift = bn_gram_affine(x2d, wt, rows_per_group, gamma, beta, eps, in_affine)
    return conv1x1_affine(x2d, wt, rows_per_group, scale, shift, out2d, residual, relu, in_affine, res_affine)


def bn_gram_affine_h2(x2d, wt, rows_per_group, gamma, beta, eps, in_affine=None, store_input=False):
    """AVS_F16X2 counterpart of bn_gram_affine (avs_bn_gram_affine_f16x2): x2d / wt are f16x2 tensors (float32-typed).
    Returns the folded affine (scale, shift) fp32 [groups, N] of the BatchNorm of y = a . wt^T.  in_affine = (isc, ish)
    fp32 [groups, K]: x2d is a raw convolution output, a = relu(x2d * isc + ish); store_input: a overwrites x2d."""
    _dev(x2d, wt, gamma, beta)
    _rowmajor2d(x2d, "x")
    _rowmajor2d(wt, "w")
    _f32(x2d, "x")
    _f32(wt, "w")
    rows, k = x2d.shape
    n = wt.shape[0]
    if rows % rows_per_group or wt.shape[1] != k:
        raise ValueError("bn_gram_affine_h2: shapes do not match")
    groups = rows // rows_per_group
    isc = ish = None
    if in_affine is not None:
        isc, ish = in_affine
        _dev(isc, ish)
        if isc.shape != (groups, k) or ish.shape != (groups, k) or not isc.is_contiguous() or not ish.is_contiguous():
            raise ValueError("in_affine must be contiguous fp32 [groups, K]")
    if store_input and in_affine is None:
        raise ValueError("store_input needs in_affine")
    scale = torch.empty((groups, n), dtype=torch.float32, device=x2d.device)
    shift = torch.empty((groups, n), dtype=torch.float32, device=x2d.device)
    _timed("gram", AVS_F16X2, 4.0 * rows * k * (2 if store_input else 1), lambda: check(
        lib().avs_bn_gram_affine_f16x2(_p(x2d), x2d.stride(0), k, _p(isc), _p(ish), _p(wt), wt.stride(0), n,
                                       rows_per_group, groups, _p(gamma), _p(beta), float(eps), _p(scale), _p(shift),
                                       _p(x2d) if store_input else None, x2d.stride(0), _stream()),
        "avs_bn_gram_affine_f16x2"))
    return scale, shift


def conv2d_affine(dtype, n, h, w, cin, sh, sw, ho, wo, cout, x, x_img_stride, x_row_stride, x_px_stride, wt,
                  w_row_stride, y, y_px_stride, rows_per_group, scale, shift, residual=None, relu=True,
                  res_affine=None, w_layout=0, variant=0):
    """1x1 convolution + a GIVEN per-group affine (+ residual, + its affine, + ReLU) in one streaming pass
    (avs_conv2d_nhwc_affine, AVS_F16X2).  scale / shift fp32 [groups, cout]; residual f16x2 [rows, cout]."""
    formats = 0
    yout = y
    if isinstance(y, P8):          # AVS_F16P8 output / residual: 3 bytes per value (strides stay in elements)
        if y.shape[-1] != cout or y_px_stride != cout:
            raise ValueError("an AVS_F16P8 output is dense [rows, cout]")
        y, formats = y.data, formats | _abi.Y_F16P8
    res_p8 = isinstance(residual, P8)
    if res_p8:
        if residual.shape[-1] != cout or residual.data.shape[0] != n * ho * wo:
            raise ValueError("an AVS_F16P8 residual is dense [rows, cout]")
        residual, formats = residual.data, formats | _abi.RES_F16P8
    _dev(x, wt, y, scale, shift, residual)
    d = _abi.ConvDesc(dtype, n, h, w, cin, 1, 1, sh, sw, 0, 0, ho, wo, cout, x_img_stride, x_row_stride, x_px_stride,
                      w_row_stride, y_px_stride, ACT_RELU if relu else ACT_NONE, 1.0, int(w_layout), int(variant), formats)
    rows = n * ho * wo
    groups = (rows + rows_per_group - 1) // rows_per_group
    for a in (scale, shift):
        if a.dtype != torch.float32 or a.shape != (groups, cout) or not a.is_contiguous():
            raise ValueError("scale / shift must be contiguous fp32 [groups, cout]")
    rsc, rsh = res_affine if res_affine is not None else (None, None)
    if residual is not None and not res_p8:
        _rowmajor2d(residual, "residual")
        if residual.shape != (rows, cout):
            raise ValueError("residual must be [rows, cout]")
    ldr = 0 if residual is None else (cout if res_p8 else residual.stride(0))
    flops = 2.0 * rows * cout * cin
    nbytes = (4.0 * rows * cin + (3.0 if formats & _abi.Y_F16P8 else 4.0) * rows * cout
              + (0.0 if residual is None else (3.0 if res_p8 else 4.0) * rows * cout))
    _timed("conv", dtype, flops, lambda: check(
        lib().avs_conv2d_nhwc_affine(ctypes.byref(d), _p(x), _p(wt), _p(y), int(rows_per_group), _p(scale), _p(shift),
                                     _p(residual), ldr, _p(rsc), _p(rsh), _stream()), "avs_conv2d_nhwc_affine"), nbytes,
           form="one-pass given-affine 1x1")
    return yout


def weights_kstep32(wt):
    """[cout, K] bf16 / fp32 (K a multiple of S = 32 / 16 elements = one 64-byte step) -> the same matrix stored
    reduction-step major, [K / S][cout][S] (AVS_W_KSTEP32), returned with the ROW shape [cout, K] so that shape checks
    read the same: the 64 bytes of a filter that one reduction step reads then sit next to the neighbouring filters'
    (whole cache lines per DMA instruction of the contraction kernel)."""
    cout, k = wt.shape
    if wt.dtype not in (torch.bfloat16, torch.float32):
        raise ValueError("weights_kstep32: bf16 or fp32 weights")
    step = 32 if wt.dtype == torch.bfloat16 else 16
    if k % step:
        raise ValueError(f"weights_kstep32: K = {k} is not a multiple of {step}")
    return wt.reshape(cout, k // step, step).permute(1, 0, 2).contiguous().view(cout, k)


def conv2d(x, wt, kh, kw, stride, pad, out, bias=None, act=ACT_NONE, bnstats=None, split=False, w_layout=0, variant=0):
    """x: NHWC view [n,h,w,cin] (unit channel stride); wt: [cout, kh*kw*cin]; out: NHWC view [n,ho,wo,cout]
    whose pixels are dense in (n,ho,wo) order (a channel slice of a dense buffer is fine).
    bnstats: see conv2d_raw (returns (scale, shift) then, else `out`)."""
    n, h, w, cin = x.shape
    sh, sw = stride if isinstance(stride, tuple) else (stride, stride)
    ph, pw = pad if isinstance(pad, tuple) else (pad, pad)
    n2, ho, wo, cout = out.shape
    if n2 != n or x.stride(3) != 1 or out.stride(3) != 1:
        raise ValueError("bad conv operands")
    if ho != (h + 2 * ph - kh) // sh + 1 or wo != (w + 2 * pw - kw) // sw + 1:
        raise ValueError(f"out extent {ho}x{wo} does not match conv arithmetic")
    yps = out.stride(2)
    if n * ho * wo > 1 and not (out.stride(1) == wo * yps and (n == 1 or out.stride(0) == ho * wo * yps)):
        raise ValueError("out pixels must be dense in (n,ho,wo) order")
    if wt.shape != (cout, kh * kw * cin) or wt.dtype != x.dtype or out.dtype != x.dtype:
        raise ValueError(f"weight {tuple(wt.shape)} / dtypes do not match")
    if bias is not None:
        _f32(bias, "bias")
    r = conv2d_raw(dtype_code(x.dtype, split), n, h, w, cin, kh, kw, sh, sw, ph, pw, ho, wo, cout, x, x.stride(0),
                   x.stride(1), x.stride(2), wt, wt.stride(0), out, yps, bias, act, bnstats=bnstats, w_layout=w_layout,
                   variant=variant)
    return out if bnstats is None else r


def conv2d_split(x, wt, out, n_split, out2, bias=None, act=ACT_NONE, w_layout=0, relu_cols=0):
    """A 1x1 / stride-1 convolution over STACKED filters with two destinations (avs_conv2d_nhwc_split, AVS_F16X2): x NHWC
    view [n,h,w,cin]; wt f16x2 [cout, cin]; output columns [0, n_split) -> out [n,h,w,n_split] (an NHWC view, e.g. a channel
    slice of a concatenation buffer), columns [n_split, cout) -> out2 [n,h,w,cout - n_split].  relu_cols > 0: with act =
    ReLU only the columns below it are rectified."""
    n, h, w, cin = x.shape
    cout = wt.shape[0]
    for o, c in ((out, n_split), (out2, cout - n_split)):
        if tuple(o.shape) != (n, h, w, c) or o.stride(3) != 1 or o.dtype != x.dtype:
            raise ValueError("conv2d_split: destinations must be NHWC views [n,h,w,n_split] and [n,h,w,cout - n_split]")
        if n * h * w > 1 and not (o.stride(1) == w * o.stride(2) and (n == 1 or o.stride(0) == h * w * o.stride(2))):
            raise ValueError("conv2d_split: destination pixels must be dense in (n,h,w) order")
    if wt.shape[1] != cin or wt.dtype != x.dtype or x.stride(3) != 1:
        raise ValueError("conv2d_split: bad operands")
    _dev(x, wt, out, out2, bias)
    d = _abi.ConvDesc(AVS_F16X2, n, h, w, cin, 1, 1, 1, 1, 0, 0, h, w, cout, x.stride(0), x.stride(1), x.stride(2),
                      wt.stride(0), out.stride(2), act, 1.0, int(w_layout), 0, 0)
    flops = 2.0 * n * h * w * cout * cin
    nbytes = 4.0 * n * h * w * (cin + cout)
    _timed("conv", AVS_F16X2, flops, lambda: check(
        lib().avs_conv2d_nhwc_split(ctypes.byref(d), _p(x), _p(wt), _p(bias), _p(out), int(n_split), _p(out2),
                                    out2.stride(2), int(relu_cols), _stream()), "avs_conv2d_nhwc_split"), nbytes,
           form="plain / bias+ReLU 1x1 (stacked heads)")
    return out, out2


# --------------------------------------------------------------------------- visual front end
def frames_normalize(frames_u8, dtype, denom, mean, std, out_h, out_w, pad_t, pad_l, affine=None, out=None, code=None):
    """frames_u8 [n,h,w,3] uint8 -> [n,out_h,out_w,4] normalised (4th channel 0), zero padded.
    code: C-ABI dtype code when it is not implied by `dtype` (AVS_F16X2 in a float32-typed tensor)."""
    _dev(frames_u8)
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or frames_u8.shape[3] != 3 or not frames_u8.is_contiguous():
        raise ValueError("frames must be contiguous uint8 [n,h,w,3]")
    n, h, w, _ = frames_u8.shape
    if out is None:
        out = torch.empty((n, out_h, out_w, 4), dtype=dtype, device=frames_u8.device)
    m3 = (c_float * 3)(*[float(v) for v in mean])
    s3 = (c_float * 3)(*[float(v) for v in std])
    a6 = (c_float * 6)(*[float(v) for v in affine]) if affine is not None else None
    check(lib().avs_frames_normalize_u8(dtype_code(dtype) if code is None else code, _p(frames_u8), n, h, w,
                                        float(denom), m3, s3, a6, _p(out),
                                        out_h, out_w, pad_t, pad_l, _stream()), "avs_frames_normalize_u8")
    return out


def pull_copy(host_pinned, dst, workgroups=16):
    """dst (device uint8, contiguous) <- host_pinned (pinned host uint8, contiguous, same numel) by avs_pull_copy_u8: a kernel
    of `workgroups` blocks reading the mapped host memory over PCIe, on the CURRENT stream."""
    if host_pinned.is_cuda or not host_pinned.is_pinned() or not host_pinned.is_contiguous():
        raise ValueError("pull_copy: the source must be a contiguous tensor in pinned host memory")
    _dev(dst)
    if not dst.is_contiguous() or dst.numel() * dst.element_size() != host_pinned.numel() * host_pinned.element_size():
        raise ValueError("pull_copy: destination must be contiguous and of the same byte size")
    check(lib().avs_pull_copy_u8(c_void_p(host_pinned.data_ptr()), _p(dst), dst.numel() * dst.element_size(), int(workgroups),
                                 _stream()), "avs_pull_copy_u8")
    return dst


_stem_ws = {}


def stem_conv_bn_pool(frames_u8, wt, denom, mean, std, frames_per_group, gamma, beta, eps, relu=True, apply=True):
    """The fused ResNet-50 stem of the bf16 path (avs_stem_conv_bn_pool_bf16): uint8 [n,224,224,3] -> bf16
    [n,56,56,64] = maxpool(relu(bn1(conv1((x / denom - mean) / std)))) with batch statistics per group of
    frames_per_group frames; wt = the stem weight in the 7 x 8 x 4 layout.  Returns (y, scale, shift).
    apply=False: y is the pooled RAW map and the consumers apply relu(scale * y + shift) while staging it (their
    in_affine operand) - the finishing pass over the map is saved."""
    _dev(frames_u8, wt, gamma, beta)
    if frames_u8.dtype != torch.uint8 or tuple(frames_u8.shape[1:]) != (224, 224, 3) or not frames_u8.is_contiguous():
        raise ValueError("frames must be contiguous uint8 [n,224,224,3]")
    if wt.dtype != torch.bfloat16 or wt.shape[0] != 64 or wt.shape[1] != 224:
        raise ValueError("stem weight must be bf16 [64, 224] (7 kernel rows x 8 pixels x 4 channels)")
    n = frames_u8.shape[0]
    dev = frames_u8.device
    y = torch.empty((n, 56, 56, 64), dtype=torch.bfloat16, device=dev)
    groups = n // frames_per_group if frames_per_group else 0
    scale = torch.empty((groups, 64), dtype=torch.float32, device=dev)
    shift = torch.empty((groups, 64), dtype=torch.float32, device=dev)
    need = int(lib().avs_stem_workspace_bytes(n))
    ws = _stem_ws.get(_ws_key(dev))
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
        _stem_ws[_ws_key(dev)] = ws
    m3 = (c_float * 3)(*[float(v) for v in mean])
    s3 = (c_float * 3)(*[float(v) for v in std])
    # algorithmic: 2 * 147 MACs per output, uint8 frames in, pooled bf16 map out
    _timed("stem", AVS_BF16, 2.0 * n * 112 * 112 * 64 * 147, lambda: check(
        lib().avs_stem_conv_bn_pool_bf16(_p(frames_u8), n, float(denom), m3, s3, _p(wt), wt.stride(0),
                                         int(frames_per_group), _p(gamma), _p(beta), float(eps), 1 if apply else 0,
                                         1 if relu else 0, _p(y), _p(scale), _p(shift), _p(ws), ws.numel(),
                                         _stream()),
        "avs_stem_conv_bn_pool_bf16"), float(frames_u8.numel()) + 2.0 * y.numel())
    return y, scale, shift


def stem_h2_operands(weight, denom, mean, std, pack=True):
    """The weight operand of avs_stem_conv_pool_f16x2 from the stem's OIHW weight [64,3,7,7], with the normalisation
    x = (v / denom - mean_c) / std_c folded in (prepared once per parameter version, in float64): an AVS_F16X2 image
    [64, 224] in the 7 x 8 x 4 layout - channels 0-2 = w / (denom std_c), channel 3 = -sum_c w mean_c / std_c (the kernel
    feeds it 1 for a pixel inside the image and 0 outside: the zero padding is of the NORMALISED input).
    pack=False: the fp32 rows before the AVS_F16X2 packing (host-side checks of the folding)."""
    w64 = weight.detach().double()                                   # [O, C, kh, kw]
    o, c, kh, kw = w64.shape
    if (o, c, kh, kw) != (64, 3, 7, 7):
        raise ValueError("the fused stem is ResNet-50's conv1: weight [64, 3, 7, 7]")
    std_t = torch.tensor([float(v) for v in std], dtype=torch.float64, device=w64.device).view(1, 3, 1, 1)
    mean_t = torch.tensor([float(v) for v in mean], dtype=torch.float64, device=w64.device).view(1, 3, 1, 1)
    wp = torch.zeros((o, kh, 8, 4), dtype=torch.float64, device=w64.device)
    wp[:, :, :kw, :c] = (w64 / (float(denom) * std_t)).permute(0, 2, 3, 1)
    wp[:, :, :kw, 3] = -(w64 * (mean_t / std_t)).sum(dim=1)
    rows = wp.reshape(o, kh * 8 * 4).float().contiguous()
    return f16x2_pack(rows) if pack else rows


def stem_conv_pool_h2(frames_u8, wimg, frames_per_group, gamma, beta, eps):
    """The fused ResNet-50 stem of the AVS_F16X2 path (avs_stem_conv_pool_f16x2): uint8 [n,224,224,3] -> the pooled RAW map
    (f16x2 slots in a float32-typed tensor [n,56,56,64]) + bn1's folded affine (scale, shift) [groups, 64]; the finished
    activation is relu(scale * y + shift), applied by the consumer (bn_gram_affine_h2's in_affine).
    wimg = stem_h2_operands(...)."""
    _dev(frames_u8, wimg, gamma, beta)
    if frames_u8.dtype != torch.uint8 or tuple(frames_u8.shape[1:]) != (224, 224, 3) or not frames_u8.is_contiguous():
        raise ValueError("frames must be contiguous uint8 [n,224,224,3]")
    if wimg.dtype != torch.float32 or tuple(wimg.shape) != (64, 224):
        raise ValueError("the weight operand must come from stem_h2_operands: f16x2 [64, 224]")
    n = frames_u8.shape[0]
    dev = frames_u8.device
    y = torch.empty((n, 56, 56, 64), dtype=torch.float32, device=dev)
    groups = n // frames_per_group if frames_per_group else 0
    scale = torch.empty((groups, 64), dtype=torch.float32, device=dev)
    shift = torch.empty((groups, 64), dtype=torch.float32, device=dev)
    need = int(lib().avs_stem_f16x2_workspace_bytes(n))
    ws = _stem_ws.get(_ws_key(dev))
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
        _stem_ws[_ws_key(dev)] = ws
    # algorithmic: 2 * 147 MACs per output, uint8 frames in, pooled f16x2 map out
    _timed("stem", AVS_F16X2, 2.0 * n * 112 * 112 * 64 * 147, lambda: check(
        lib().avs_stem_conv_pool_f16x2(_p(frames_u8), n, _p(wimg), wimg.stride(0), int(frames_per_group),
                                       _p(gamma), _p(beta), float(eps), _p(y), _p(scale), _p(shift), _p(ws), ws.numel(),
                                       _stream()),
        "avs_stem_conv_pool_f16x2"), float(frames_u8.numel()) + 4.0 * y.numel())
    return y, scale, shift


def resize_bilinear(frames_u8, dh, dw):
    _dev(frames_u8)
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or frames_u8.shape[3] != 3 or not frames_u8.is_contiguous():
        raise ValueError("frames must be contiguous uint8 [n,h,w,3]")
    n, h, w, _ = frames_u8.shape
    out = torch.empty((n, dh, dw, 3), dtype=torch.uint8, device=frames_u8.device)
    check(lib().avs_resize_bilinear_u8(_p(frames_u8), n, h, w, _p(out), dh, dw, _stream()), "avs_resize_bilinear_u8")
    return out


def bn_batch_stats(x2d, group_rows, gamma, beta, eps, code=None):
    """x2d [rows, C] (row stride >= C); group_rows int64 [G+1] device.  Returns scale, shift [G, C] fp32."""
    _dev(x2d, group_rows, gamma, beta)
    _rowmajor2d(x2d, "x")
    rows, c = x2d.shape
    g = group_rows.numel() - 1
    scale = torch.empty((g, c), dtype=torch.float32, device=x2d.device)
    shift = torch.empty((g, c), dtype=torch.float32, device=x2d.device)
    check(lib().avs_bn_batch_stats(dtype_code(x2d.dtype) if code is None else code, _p(x2d), rows, c, x2d.stride(0),
                                   _p(group_rows), g,
                                   _p(gamma), _p(beta), float(eps), _p(scale), _p(shift), _stream()),
          "avs_bn_batch_stats")
    return scale, shift


def bn_apply(x2d, scale, shift, group_rows=None, max_group_rows=0, residual=None, act=ACT_NONE, out=None, code=None):
    _dev(x2d, scale, shift, group_rows, residual)
    code = dtype_code(x2d.dtype) if code is None else code
    _rowmajor2d(x2d, "x")
    rows, c = x2d.shape
    if out is None:
        out = torch.empty((rows, c), dtype=x2d.dtype, device=x2d.device)
    g = group_rows.numel() - 1 if group_rows is not None else 0
    # algorithmic bytes: read x (+ residual), write y
    nbytes = float(rows) * c * x2d.element_size() * (3 if residual is not None else 2)
    _timed("bn_apply", code, nbytes, lambda: check(
        lib().avs_bn_apply(code, _p(x2d), rows, c, x2d.stride(0), _p(group_rows), g,
                           int(max_group_rows), _p(scale), _p(shift), _p(residual),
                           residual.stride(0) if residual is not None else 0, act, _p(out), out.stride(0),
                           _stream()), "avs_bn_apply"))
    return out


def pool2d(x, mode, k, s, p, out, bias=None, act=ACT_NONE, code=None):
    """x, out: NHWC views (unit channel stride, dense pixels).  mode 'max' | 'avg'.  out = act(pool(x) + bias) with an
    optional fp32 per-channel bias."""
    n, h, w, c = x.shape
    _, ho, wo, _ = out.shape
    if bias is not None:
        _f32(bias, "bias")
        if bias.numel() != c or not bias.is_contiguous():
            raise ValueError("pool2d: bias must be contiguous fp32 [c]")
    check(lib().avs_pool2d_nhwc(dtype_code(x.dtype) if code is None else code, 0 if mode == "max" else 1, _p(x), n, h,
                                w, c, x.stride(2), k, s,
                                p, _p(bias), int(act), _p(out), ho, wo, out.stride(2), _stream()), "avs_pool2d_nhwc")
    return out


def bn_maxpool(x, scale, shift, group_rows, relu, k, s, p, out, code=None):
    """out = maxpool(act(x*scale[g] + shift[g])) on NHWC views: avs_bn_apply + max pooling in one pass."""
    _dev(x, scale, shift, group_rows, out)
    n, h, w, c = x.shape
    _, ho, wo, _ = out.shape
    g = group_rows.numel() - 1 if group_rows is not None else 0
    # algorithmic bytes: read the raw map once, write the pooled map
    nbytes = float(x.numel() + out.numel()) * x.element_size()
    code = dtype_code(x.dtype) if code is None else code
    _timed("bn_apply", code, nbytes, lambda: check(
        lib().avs_bn_maxpool_nhwc(code, _p(x), n, h, w, c, x.stride(2), _p(group_rows), g, _p(scale),
                                  _p(shift), 1 if relu else 0, k, s, p, _p(out), ho, wo, out.stride(2), _stream()),
        "avs_bn_maxpool_nhwc"))
    return out


def global_avgpool(x, out=None, code=None):
    """x [n,h,w,c] dense NHWC -> fp32 [n,c]."""
    n, h, w, c = x.shape
    if not x.is_contiguous():
        raise ValueError("global_avgpool needs a dense NHWC tensor")
    if out is None:
        out = torch.empty((n, c), dtype=torch.float32, device=x.device)
    check(lib().avs_global_avgpool_nhwc(dtype_code(x.dtype) if code is None else code, _p(x), n, h * w, c, _p(out),
                                        out.stride(0), _stream()),
          "avs_global_avgpool_nhwc")
    return out


def segment_mean(x2d, seg, out=None):
    _f32(x2d, "x")
    _rowmajor2d(x2d, "x")
    nseg = seg.numel() - 1
    d = x2d.shape[1]
    if out is None:
        out = torch.empty((nseg, d), dtype=torch.float32, device=x2d.device)
    check(lib().avs_segment_mean_f32(_p(x2d), x2d.stride(0), d, _p(seg), nseg, _p(out), out.stride(0), _stream()),
          "avs_segment_mean_f32")
    return out


def hsv_frame_diff(frames_u8, step=1):
    """frames uint8 [n,h,w,3] on device -> int64 [n,3] sums of |dH|,|dS|,|dV| against the previous frame."""
    _dev(frames_u8)
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or frames_u8.shape[3] != 3 or not frames_u8.is_contiguous():
        raise ValueError("frames must be contiguous uint8 [n,h,w,3]")
    n, h, w, _ = frames_u8.shape
    sums = torch.empty((n, 3), dtype=torch.int32, device=frames_u8.device)
    check(lib().avs_hsv_frame_diff_u8(_p(frames_u8), n, h, w, step, _p(sums), _stream()), "avs_hsv_frame_diff_u8")
    return sums.to(torch.int64) & 0xFFFFFFFF


# --------------------------------------------------------------------------- audio front end
def reflect_pad(x, pad, out_len):
    _f32(x, "x")
    out = torch.empty(out_len, dtype=torch.float32, device=x.device)
    check(lib().avs_reflect_pad_f32(_p(x), x.numel(), pad, _p(out), out_len, _stream()), "avs_reflect_pad_f32")
    return out


def stft_f64(xpad, frames, hop, nfft, basis_t, ncols):
    """xpad fp32 [L]; basis_t float64 [nfft, ncols_pad]; returns fp32 [frames, ncols]."""
    _f32(xpad, "xpad")
    if basis_t.dtype != torch.float64 or not basis_t.is_contiguous() or basis_t.shape[0] != nfft:
        raise ValueError("basis_t must be contiguous float64 [nfft, ncols_pad]")
    spec = torch.empty((frames, ncols), dtype=torch.float32, device=xpad.device)
    check(lib().avs_stft_f64(_p(xpad), xpad.numel(), frames, hop, nfft, _p(basis_t), ncols, basis_t.shape[1], _p(spec),
                             _stream()), "avs_stft_f64")
    return spec


def power_mel(spec, nbins, fb, fb_lo, fb_hi, mode, gmax=None):
    frames = spec.shape[0]
    nmel = fb.shape[1]
    out = torch.empty((frames, nmel), dtype=torch.float32, device=spec.device)
    check(lib().avs_power_mel_f32(_p(spec), frames, nbins, _p(fb), _p(fb_lo), _p(fb_hi), nmel, mode, _p(out),
                                  _p(gmax), _stream()), "avs_power_mel_f32")
    return out


def stft_mel_fused(wave, window, cos_t, sin_t, fb, fb_lo, fb_hi, log2=False, db=False, power=False):
    """wave fp32 [T] on device (T > 200) -> (log2(mel + 1e-6) | None, 10 log10(max(mel, 1e-10)) | None, mel | None,
    max of the clamped mel power fp32 [1] | None), each [1 + T // 200, n_mels]: torchaudio's MelSpectrogram defaults
    (n_fft = win = 400, hop 200, center / reflect, power 2) in one kernel (avs_stft_mel_fused_f32)."""
    _dev(wave, window, cos_t, sin_t, fb)
    _f32(wave, "wave")
    if wave.dim() != 1 or not wave.is_contiguous():
        raise ValueError("wave must be a contiguous 1-D tensor")
    if window.dtype != torch.float64 or cos_t.dtype != torch.float64 or sin_t.dtype != torch.float64:
        raise TypeError("window and DFT tables must be float64")
    if tuple(cos_t.shape) != (204, 208) or tuple(sin_t.shape) != (200, 208) or window.numel() != 400:
        raise ValueError("tables must be cos [204, 208], -sin [200, 208], window [400]")
    t = wave.numel()
    frames = 1 + t // 200
    nmel = fb.shape[1]
    mk = lambda on: torch.empty((frames, nmel), dtype=torch.float32, device=wave.device) if on else None
    o_log2, o_db, o_pow = mk(log2), mk(db), mk(power)
    gmax = torch.zeros(1, dtype=torch.float32, device=wave.device) if db else None
    nbytes = 4.0 * t + 4.0 * frames * nmel * (int(log2) + int(db) + int(power))   # algorithmic: read x, write outputs
    _timed("audio", AVS_F32, nbytes, lambda: check(
        lib().avs_stft_mel_fused_f32(_p(wave), t, _p(window), _p(cos_t), _p(sin_t), _p(fb), _p(fb_lo), _p(fb_hi), nmel,
                                     _p(o_log2), _p(o_db), _p(o_pow), _p(gmax), _stream()), "avs_stft_mel_fused_f32"))
    return o_log2, o_db, o_pow, gmax


def stft_mel_max(wave, window, cos_t, sin_t, fb, fb_lo, fb_hi):
    """fp32 [1]: the largest clamped mel power of the track (what the MFCC's top_db clamp is relative to), from one
    pass of the fused front end that writes nothing else."""
    _dev(wave, window, cos_t, sin_t, fb)
    _f32(wave, "wave")
    gmax = torch.zeros(1, dtype=torch.float32, device=wave.device)
    t = wave.numel()
    _timed("audio", AVS_F32, 4.0 * t, lambda: check(
        lib().avs_stft_mel_fused_f32(_p(wave), t, _p(window), _p(cos_t), _p(sin_t), _p(fb), _p(fb_lo), _p(fb_hi),
                                     fb.shape[1], None, None, None, _p(gmax), _stream()), "avs_stft_mel_fused_f32"))
    return gmax


_segmean_ws = {}


def stft_mel_segmean(wave, window, cos_t, sin_t, fb, fb_lo, fb_hi, blocks, seg_block, seg_frames, gmax=None,
                     top_db=80.0, out_log2=None, out_db=None):
    """Time means per segment of log2(mel + 1e-6) (out_log2 [nseg, >= nmel]) and of the top_db-clamped dB mel (out_db)
    straight from the waveform (avs_stft_mel_segmean_f32).  gmax: the maximum the clamp is relative to (from
    stft_mel_max: a second pass of the DFT, nothing per frame in HBM); gmax=None with out_db: ONE pass of the DFT that also
    finds the maximum over the table's blocks (the whole track when the table covers it), the unclamped dB rows going
    through the workspace - bit-identical means.
    blocks int32 [nblocks, 3] = (first STFT frame, frames <= 32, segment); seg_block int32 [nseg + 1]; seg_frames
    int32 [nseg] - all on the device (audio.MelPlan.segment_table builds them)."""
    _dev(wave, blocks, seg_block, seg_frames, out_log2, out_db, gmax)
    _f32(wave, "wave")
    nmel = fb.shape[1]
    nblocks, nseg = blocks.shape[0], seg_frames.numel()
    find_max = int(out_db is not None and gmax is None)
    if find_max:
        gmax = torch.empty(1, dtype=torch.float32, device=wave.device)
    need = int(lib().avs_stft_mel_segmean_workspace_bytes(nblocks, nmel, int(out_log2 is not None),
                                                          int(out_db is not None), find_max))
    key = _ws_key(wave.device)
    ws = _segmean_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 16), dtype=torch.uint8, device=wave.device)
        _segmean_ws[key] = ws
    t = wave.numel()
    nbytes = 4.0 * t + 4.0 * nseg * nmel * (int(out_log2 is not None) + int(out_db is not None))
    _timed("audio", AVS_F32, nbytes, lambda: check(
        lib().avs_stft_mel_segmean_f32(_p(wave), t, _p(window), _p(cos_t), _p(sin_t), _p(fb), _p(fb_lo), _p(fb_hi), nmel,
                                       _p(blocks), nblocks, _p(seg_block), _p(seg_frames), nseg, _p(gmax), find_max,
                                       float(top_db), _p(out_log2), out_log2.stride(0) if out_log2 is not None else 0,
                                       _p(out_db), out_db.stride(0) if out_db is not None else 0, _p(ws), ws.numel(),
                                       _stream()),
        "avs_stft_mel_segmean_f32"))
    return out_log2, out_db


def stft_mel_segmean_batch(waves, track_off, track_len, window, cos_t, sin_t, fb, fb_lo, fb_hi, blocks, seg_block, seg_frames,
                           top_db=80.0, out_log2=None, out_db=None):
    """stft_mel_segmean for a batch of tracks in one set of launches (avs_stft_mel_segmean_batch_f32): waves = the tracks one
    after another (fp32, 16-byte aligned starts), track_off / track_len int64 [ntracks] on the device, blocks int32
    [nblocks, 4] = (first frame inside the track, frames <= 32, segment, track).  Returns (out_log2, out_db, gmax[ntracks])."""
    _dev(waves, track_off, track_len, blocks, seg_block, seg_frames, out_log2, out_db)
    _f32(waves, "waves")
    nmel = fb.shape[1]
    ntracks = track_len.numel()
    nblocks, nseg = blocks.shape[0], seg_frames.numel()
    gmax = torch.empty(max(ntracks, 1), dtype=torch.float32, device=waves.device)
    need = int(lib().avs_stft_mel_segmean_workspace_bytes(nblocks, nmel, int(out_log2 is not None), int(out_db is not None), 1))
    key = _ws_key(waves.device)
    ws = _segmean_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 16), dtype=torch.uint8, device=waves.device)
        _segmean_ws[key] = ws
    nbytes = 4.0 * waves.numel() + 4.0 * nseg * nmel * (int(out_log2 is not None) + int(out_db is not None))
    _timed("audio", AVS_F32, nbytes, lambda: check(
        lib().avs_stft_mel_segmean_batch_f32(_p(waves), _p(track_off), _p(track_len), ntracks, _p(window), _p(cos_t), _p(sin_t),
                                             _p(fb), _p(fb_lo), _p(fb_hi), nmel, _p(blocks), nblocks, _p(seg_block),
                                             _p(seg_frames), nseg, _p(gmax), float(top_db), _p(out_log2),
                                             out_log2.stride(0) if out_log2 is not None else 0, _p(out_db),
                                             out_db.stride(0) if out_db is not None else 0, _p(ws), ws.numel(), _stream()),
        "avs_stft_mel_segmean_batch_f32"))
    return out_log2, out_db, gmax


def clamp_topdb(x, gmax, top_db):
    check(lib().avs_clamp_topdb_f32(_p(x), x.numel(), _p(gmax), float(top_db), _stream()), "avs_clamp_topdb_f32")
    return x


def fill(x, value):
    _f32(x, "x")
    check(lib().avs_fill_f32(_p(x), x.numel(), float(value), _stream()), "avs_fill_f32")
    return x


def quantize(x, lo, hi, scale, out=None):
    _f32(x, "x")
    if out is None:
        out = torch.empty_like(x)
    check(lib().avs_quantize_f32(_p(x), x.numel(), float(lo), float(hi), float(scale), _p(out), _stream()),
          "avs_quantize_f32")
    return out


def resample(x, taps, up, down, width, out_len):
    """x fp32 [t] or interleaved [t, channels] on device; taps fp32 [up, ntaps] -> mono fp32 [out_len]."""
    _dev(x, taps)
    _f32(x, "x")
    _f32(taps, "taps")
    if not x.is_contiguous() or not taps.is_contiguous() or taps.shape[0] != up:
        raise ValueError("x must be contiguous [t] / [t, channels], taps contiguous [up, ntaps]")
    channels = 1 if x.dim() == 1 else x.shape[1]
    out = torch.empty(out_len, dtype=torch.float32, device=x.device)
    check(lib().avs_resample_f32(_p(x), x.shape[0], channels, _p(taps), up, down, taps.shape[1], width, _p(out),
                                 out_len, _stream()), "avs_resample_f32")
    return out


# --------------------------------------------------------------------------- scorer
# ---- the split recurrence (avs_lstm_split_f32 / avs_lstm_bwd_split_f32): workspace + tag ranges per (device, stream)
LSTM_SPLIT_MAX_RECURRENCES = 208    # AUTO takes the four-CU form up to this many recurrences per launch: 4 workgroups of 512
#                                     threads each, one per CU.  tools/lstm_study.py (profiles/r04_lstm_study.txt), sequences of
#                                     ~1800 steps: 48 recurrences 3.4 ms against 11.0 for one recurrence per CU, 100 (two rounds
#                                     on 256 CUs) 6.3 against 12.4, 200 (four rounds) 10.2 against 12.1; past ~250 one
#                                     recurrence per CU is the better use of the chip


class _LstmExchange:
    """Granule workspace of the split recurrence on one (device, stream): zeroed once; `epoch` = tags handed out so far."""

    def __init__(self):
        self.buf, self.epoch = None, 0

    def get(self, device, nbytes, steps):
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = torch.zeros(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
            self.epoch = 0
        if self.epoch + steps + 2 >= 0xFFFFFFF0:      # (32-bit tags: start over on a cleared buffer)
            self.buf.zero_()
            self.epoch = 0
        e = self.epoch
        self.epoch += int(steps) + 1
        return self.buf, e


_lstm_exchanges = {}


def _lstm_split_ws(device, ndir, nseq, rows):
    ex = _lstm_exchanges.setdefault(_ws_key(device), _LstmExchange())
    return ex.get(device, lib().avs_lstm_split_workspace_bytes(ndir, nseq), rows)


def lstm_split_errors(device):
    """Workgroups whose bounded partner wait ran out in the split-recurrence launches on this device's current stream so far
    (0 after healthy launches); reads the counter back (a host sync)."""
    ex = _lstm_exchanges.get(_ws_key(device))
    return 0 if ex is None or ex.buf is None else int(ex.buf[:4].view(torch.int32).item())


def _lstm_takes_split(variant, hidden, ndir, nseq):
    if variant == _abi.LSTM_SPLIT4:
        return True
    return variant == _abi.LSTM_AUTO and hidden == 256 and 0 < ndir * nseq <= LSTM_SPLIT_MAX_RECURRENCES


def lstm(xproj, whh_t, hidden, ndir, reverse_mask, seq_rows, out, out_col0, variant=0):
    _f32(xproj, "xproj")
    _f32(whh_t, "whh_t")
    _f32(out, "out")
    if xproj.shape[1] != ndir * 4 * hidden or not xproj.is_contiguous() or not whh_t.is_contiguous():
        raise ValueError("xproj must be contiguous [rows, ndir*4H], whh_t contiguous [ndir,H,4H]")
    if tuple(whh_t.shape) != (ndir, hidden, 4 * hidden):
        raise ValueError(f"whh_t shape {tuple(whh_t.shape)}")
    nseq = seq_rows.numel() - 1
    if _lstm_takes_split(variant, hidden, ndir, nseq):
        ws, epoch = _lstm_split_ws(xproj.device, ndir, nseq, xproj.shape[0])
        check(lib().avs_lstm_split_f32(_p(xproj), _p(whh_t), hidden, ndir, reverse_mask, _p(seq_rows), nseq, _p(out),
                                       out.stride(0), out_col0, None, None, _p(ws), ws.numel(), epoch, _stream()),
              "avs_lstm_split_f32")
        return out
    check(lib().avs_lstm_f32(_p(xproj), _p(whh_t), hidden, ndir, reverse_mask, _p(seq_rows), nseq, _p(out),
                             out.stride(0), out_col0, int(variant), _stream()), "avs_lstm_f32")
    return out


def mha_batchaxis(qkv, b, t, e, heads):
    _f32(qkv, "qkv")
    if tuple(qkv.shape) != (b * t, 3 * e) or not qkv.is_contiguous():
        raise ValueError("qkv must be contiguous [B*T, 3E]")
    ctx = torch.empty((b * t, e), dtype=torch.float32, device=qkv.device)
    check(lib().avs_mha_batchaxis_f32(_p(qkv), b, t, e, heads, _p(ctx), _stream()), "avs_mha_batchaxis_f32")
    return ctx


def score_head(hid, w2, b2):
    _f32(hid, "hid")
    _rowmajor2d(hid, "hid")
    rows, d = hid.shape
    out = torch.empty(rows, dtype=torch.float32, device=hid.device)
    check(lib().avs_score_head_f32(_p(hid), rows, d, hid.stride(0), _p(w2), _p(b2), _p(out), _stream()),
          "avs_score_head_f32")
    return out


def mhsa_flash(q, k, v, b, t, heads, split=True):
    """q, k, v: fp32 [b*t, E] (projected); returns ctx [b*t, E].  Head dim E/heads must be 64, 128 or 256.
    split (default): the operands are packed to fp16 hi | lo runs and the products run on the fp16 matrix cores as
    hi*hi + lo*hi + hi*lo (avs_mhsa_flash_f16x2); False: the exact fp32 MFMA kernel (avs_mhsa_flash_f32)."""
    for name, x in (("q", q), ("k", k), ("v", v)):
        _f32(x, name)
        _rowmajor2d(x, name)
    e = q.shape[1]
    ctx = torch.empty((b * t, e), dtype=torch.float32, device=q.device)
    if not (q.stride(0) == k.stride(0) == v.stride(0)):
        raise ValueError("q, k, v must share a row stride")
    if split and e % 8 == 0 and q.is_contiguous() and k.is_contiguous() and v.is_contiguous():
        qp, kp, vp = f16x2_pack(q), f16x2_pack(k), f16x2_pack(v)
        check(lib().avs_mhsa_flash_f16x2(_p(qp), _p(kp), _p(vp), qp.stride(0), b, t, heads, e // heads, _p(ctx), e,
                                         _stream()), "avs_mhsa_flash_f16x2")
        return ctx
    check(lib().avs_mhsa_flash_f32(_p(q), _p(k), _p(v), q.stride(0), b, t, heads, e // heads, _p(ctx), e, _stream()),
          "avs_mhsa_flash_f32")
    return ctx


def softmax_rows(x, rows, n, ldx):
    _f32(x, "x")
    check(lib().avs_softmax_rows_f32(_p(x), rows, n, ldx, _stream()), "avs_softmax_rows_f32")
    return x


def softmax_bwd_rows(p, dp, rows, n, ld, alpha):
    """dp <- alpha * p * (dp - rowsum(p * dp)), in place: the softmax backward w.r.t. the unscaled scores."""
    _f32(p, "p")
    _f32(dp, "dp")
    check(lib().avs_softmax_bwd_rows_f32(_p(p), _p(dp), rows, n, ld, float(alpha), _stream()),
          "avs_softmax_bwd_rows_f32")
    return dp


def transpose_into(src, src_off, rows, cols, ld_src, dst, dst_off, ld_dst):
    """dst[c, r] = src[r, c] on raw element offsets (pad columns of dst are left as they are)."""
    check(lib().avs_transpose_f32(_p(src, src_off), rows, cols, ld_src, _p(dst, dst_off), ld_dst, _stream()),
          "avs_transpose_f32")


# --------------------------------------------------------------------------- scorer backward
def transpose_padded(x2d, pad=4):
    """[rows, cols] fp32 -> [cols, rows_p] with rows_p = rows rounded up to `pad`, the tail columns zero."""
    _f32(x2d, "x")
    _rowmajor2d(x2d, "x")
    rows, cols = x2d.shape
    rp = (rows + pad - 1) // pad * pad
    out = torch.zeros((cols, rp), dtype=torch.float32, device=x2d.device)
    check(lib().avs_transpose_f32(_p(x2d), rows, cols, x2d.stride(0) if rows > 1 else cols, _p(out), rp, _stream()),
          "avs_transpose_f32")
    return out


def colsum(x2d, row_weight=None):
    _f32(x2d, "x")
    _rowmajor2d(x2d, "x")
    rows, cols = x2d.shape
    out = torch.empty(cols, dtype=torch.float32, device=x2d.device)
    check(lib().avs_colsum_f32(_p(x2d), rows, cols, x2d.stride(0) if rows > 1 else cols, _p(row_weight), _p(out),
                               _stream()), "avs_colsum_f32")
    return out


def relu_dropout_bwd(dy, relu_out, keep=None):
    out = torch.empty_like(dy)
    check(lib().avs_relu_dropout_bwd_f32(_p(dy), _p(relu_out), _p(keep), dy.numel(), _p(out), _stream()),
          "avs_relu_dropout_bwd_f32")
    return out


def mul(a, b):
    out = torch.empty_like(a)
    check(lib().avs_mul_f32(_p(a), _p(b), a.numel(), _p(out), _stream()), "avs_mul_f32")
    return out


def score_head_bwd(dscores, scores, hid, w2):
    rows, d = hid.shape
    dz = torch.empty(rows, dtype=torch.float32, device=hid.device)
    dpre = torch.empty((rows, d), dtype=torch.float32, device=hid.device)
    check(lib().avs_score_head_bwd_f32(_p(dscores), _p(scores), _p(hid), rows, d, hid.stride(0), _p(w2), _p(dz),
                                       _p(dpre), _stream()), "avs_score_head_bwd_f32")
    return dz, dpre


def lstm_train_fwd(xproj, whh_t, hidden, ndir, reverse_mask, seq_rows, out, out_col0, variant=0):
    rows = xproj.shape[0]
    gates = torch.empty((rows, ndir * 4 * hidden), dtype=torch.float32, device=xproj.device)
    cell = torch.empty((rows, ndir * hidden), dtype=torch.float32, device=xproj.device)
    nseq = seq_rows.numel() - 1
    if _lstm_takes_split(variant, hidden, ndir, nseq):
        ws, epoch = _lstm_split_ws(xproj.device, ndir, nseq, rows)
        check(lib().avs_lstm_split_f32(_p(xproj), _p(whh_t), hidden, ndir, reverse_mask, _p(seq_rows), nseq, _p(out),
                                       out.stride(0), out_col0, _p(gates), _p(cell), _p(ws), ws.numel(), epoch, _stream()),
              "avs_lstm_split_f32")
        return gates, cell
    if variant == _abi.LSTM_RESIDENT_20_8:      # (the C entry's AUTO: the one-CU resident kernel)
        variant = _abi.LSTM_AUTO
    check(lib().avs_lstm_train_fwd_f32(_p(xproj), _p(whh_t), hidden, ndir, reverse_mask, _p(seq_rows), nseq, _p(out),
                                       out.stride(0), out_col0, _p(gates), _p(cell), int(variant), _stream()),
          "avs_lstm_train_fwd_f32")
    return gates, cell


def lstm_bwd(dout, out_col0, gates, cell, whh, hidden, ndir, reverse_mask, seq_rows, variant=0):
    rows = gates.shape[0]
    dxproj = torch.empty((rows, ndir * 4 * hidden), dtype=torch.float32, device=gates.device)
    nseq = seq_rows.numel() - 1
    if _lstm_takes_split(variant, hidden, ndir, nseq):
        ws, epoch = _lstm_split_ws(gates.device, ndir, nseq, rows)
        check(lib().avs_lstm_bwd_split_f32(_p(dout), dout.stride(0), out_col0, _p(gates), _p(cell), _p(whh), hidden, ndir,
                                           reverse_mask, _p(seq_rows), nseq, _p(dxproj), _p(ws), ws.numel(), epoch, _stream()),
              "avs_lstm_bwd_split_f32")
        return dxproj
    if variant == _abi.LSTM_RESIDENT_20_8:
        variant = _abi.LSTM_AUTO
    check(lib().avs_lstm_bwd_f32(_p(dout), dout.stride(0), out_col0, _p(gates), _p(cell), _p(whh), hidden, ndir,
                                 reverse_mask, _p(seq_rows), nseq, _p(dxproj), int(variant), _stream()), "avs_lstm_bwd_f32")
    return dxproj


def grad_weight(dy, x):
    """dW [N, K] = dy[T, N]^T . x[T, K]  (NT GEMM on transposed, zero-padded copies)."""
    dyt = transpose_padded(dy)
    xt = transpose_padded(x)
    n, tp = dyt.shape
    k = xt.shape[0]
    out = torch.empty((n, k), dtype=torch.float32, device=dy.device)
    gemm_nt_batched(AVS_F32, n, k, tp, dyt, 0, tp, 0, xt, 0, tp, 0, out, 0, k, 0)
    return out


def grad_input(dy, w):
    """dx [T, K] = dy[T, N] . w[N, K]  (NT GEMM against w^T)."""
    wt = transpose_padded(w)  # [K, Np]
    t, n = dy.shape
    k, npad = wt.shape
    if n % 4:
        raise ValueError("grad_input needs the output width to be a multiple of 4")
    out = torch.empty((t, k), dtype=torch.float32, device=dy.device)
    gemm_nt_batched(AVS_F32, t, k, n, dy, 0, dy.stride(0) if t > 1 else n, 0, wt, 0, npad, 0, out, 0, k, 0)
    return out


# --------------------------------------------------------------------------- fusion
def cdist(v, a):
    _f32(v, "v")
    _f32(a, "a")
    if v.dim() != 2 or a.dim() != 2 or v.shape[1] != a.shape[1]:
        raise ValueError("XA and XB must have the same number of columns (i.e. feature dimension.)")
    v = v.contiguous()
    a = a.contiguous()
    out = torch.empty((v.shape[0], a.shape[0]), dtype=torch.float64, device=v.device)
    check(lib().avs_cdist_f64(_p(v), v.shape[0], _p(a), a.shape[0], v.shape[1], _p(out), _stream()), "avs_cdist_f64")
    return out


def dtw_path(cost):
    if cost.dtype != torch.float64 or cost.dim() != 2 or not cost.is_contiguous():
        raise ValueError("cost must be a contiguous float64 matrix")
    n, m = cost.shape
    ws_bytes = lib().avs_dtw_workspace_bytes(n, m)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=cost.device)
    path = torch.empty((n + m - 1, 2), dtype=torch.int64, device=cost.device)
    plen = torch.zeros(1, dtype=torch.int64, device=cost.device)
    total = torch.zeros(1, dtype=torch.float64, device=cost.device)
    check(lib().avs_dtw_path_f64(_p(cost), n, m, _p(ws), ws_bytes, _p(path), _p(plen), _p(total), _stream()),
          "avs_dtw_path_f64")
    return path, plen, total


def gather_scale(x, idx, w):
    _f32(x, "x")
    _rowmajor2d(x, "x")
    count = idx.numel()
    d = x.shape[1]
    out = torch.empty((count, d), dtype=torch.float32, device=x.device)
    check(lib().avs_gather_scale_f32(_p(x), x.stride(0), d, _p(idx), _p(w), count, _p(out), _stream()),
          "avs_gather_scale_f32")
    return out
