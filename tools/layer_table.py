#!/usr/bin/env python3
"""Per-layer accounting of the production ResNet-50 forward (bf16, batch-statistics BatchNorm, per-frame groups unless
--gf): every `_conv_bn` call (convolution + BatchNorm + residual + ReLU in whichever form the runner picks) and the
stem are bracketed with events IN PLACE, so the table is the real pass, not isolated kernels.

Per layer: time per pass, matrix rate (algorithmic FLOP / time), stream rate (bf16 input + output + residual bytes /
time) and the time an ideal kernel would take: max(FLOP / 1.2 PFLOP/s, bytes / 5.5 TB/s) - 1.2 PF is what the LDS-DMA
intake allows a 128x128-tile contraction here, 5.5 TB/s what a streaming kernel reaches on this chip.

Usage: python tools/layer_table.py [--n 8192] [--gf 1] [--dtype bf16|f32split]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avsum_amd import cnn, ops
from avsum_amd.features.extractors import VisualFeatureExtractor

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=8192)
ap.add_argument("--gf", type=int, default=1)
ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32split", "f32", "f16x2"])
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--set", action="append", default=[], help="runner attribute override, e.g. --set gram_finish_min_k=64")
ap.add_argument("--p8", default="on", choices=["on", "off"], help="f16x2: AVS_F16P8 storage of the inner block outputs of layers 1-2")
ap.add_argument("--short", type=int, default=0,
                help="study build (AVS_STUDY_LIB=1): reductions of at most this many bytes per row take 64-byte steps (rule: 2048)")
args = ap.parse_args()
if args.short:
    from avsum_amd import _abi
    _abi.lib().avs_tune_short_reduction_bytes(args.short)
dev = torch.device("cuda", 0)
dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
es = 2 if dt == torch.bfloat16 else 4
torch.manual_seed(0)
SPLIT = {"f32split": True, "f16x2": "f16x2"}.get(args.dtype, False)
MFMA_IDEAL = 1.2e15 / (3.0 if SPLIT else 1.0)   # three 16-bit MFMAs per product in the split modes
ext = VisualFeatureExtractor(dt, "batch", f32_split=SPLIT).to(dev)
runner = ext._resnet_runner if hasattr(ext, "_resnet_runner") else None
if runner is None:
    runner = next(v for v in vars(ext).values() if isinstance(v, cnn.ResNet50Runner))
for kv in args.set:
    k, v = kv.split("=")
    setattr(runner, k, type(getattr(runner, k))(int(v)))
if args.p8 == "off":
    runner.p8_blocks = ()
frames = torch.randint(0, 256, (args.n, 224, 224, 3), dtype=torch.uint8, device=dev)
gf = torch.arange(0, args.n + 1, args.gf, dtype=torch.int64)

records = []
orig_conv_bn = cnn.ResNet50Runner._conv_bn
orig_stem = ops.stem_conv_bn_pool


def ev():
    return torch.cuda.Event(enable_timing=True)


def timed_conv_bn(self, geom, xs, x, wt, bnp, groups, residual=None, **kw):
    e0, e1 = ev(), ev()
    e0.record()
    out = orig_conv_bn(self, geom, xs, x, wt, bnp, groups, residual=residual, **kw)
    e1.record()
    n, h, cin, kh, sh, ho, cout = geom[0], geom[1], geom[3], geom[4], geom[6], geom[10], geom[12]
    kk = kw.get("algo_k") or kh * geom[5] * cin
    rows = n * ho * geom[11]
    flops = 2.0 * rows * kk * cout
    in_elems = n * h * h * cin if sh == 1 or kh > 1 else rows * cin
    byts = (in_elems + rows * cout * (2 if residual is not None else 1)) * es
    loc = kw.get("local")
    form = ("cluster" if (loc and loc is not True and int(loc) > 1) else "local" if loc else "gram/2pass" if (kw.get("in_affine") is not None or (
        kh == 1 and cout >= 2 * cin and sh == 1)) else "split+defer" if kw.get("defer") else "split")
    records.append((f"{h}x{h} {kh}x{kh}/{sh} {cin}->{cout}" + (" +res" if residual is not None else ""), form,
                    flops, byts, e0, e1))
    return out


def timed_stem(frames_u8, *a, **kw):
    e0, e1 = ev(), ev()
    e0.record()
    out = orig_stem(frames_u8, *a, **kw)
    e1.record()
    n = frames_u8.shape[0]
    records.append(("stem 7x7/2 3->64 + pool", "fused", 2.0 * n * 112 * 112 * 147 * 64,
                    n * (224 * 224 * 3 + 56 * 56 * 64 * es), e0, e1))
    return out


inside = [0]
_inner_conv_bn = timed_conv_bn


def timed_conv_bn_outer(self, *a, **kw):
    inside[0] += 1
    try:
        return _inner_conv_bn(self, *a, **kw)
    finally:
        inside[0] -= 1


def timed_op(fn, label):
    # the first block's kernels on the stem's raw map are called from forward() directly, not through _conv_bn
    def wrapper(x2d, wt, rpg, *a, **kw):
        if inside[0]:
            return fn(x2d, wt, rpg, *a, **kw)
        e0, e1 = ev(), ev()
        e0.record()
        out = fn(x2d, wt, rpg, *a, **kw)
        e1.record()
        rows, k = x2d.shape
        nn = wt.shape[0]
        if label == "gram":
            records.append((f"56x56 gram {k} (for {nn} outputs)", "gram", 2.0 * rows * k * k, rows * k * es, e0, e1))
        else:
            records.append((f"56x56 1x1/1 {k}->{nn} (raw stem in)", "affine", 2.0 * rows * k * nn, rows * (k + nn) * es,
                            e0, e1))
        return out
    return wrapper


def timed_h2(fn, label):
    # AVS_F16X2: the first block's Gram matrix + its two streaming passes are called from forward() directly
    def wrapper(*a, **kw):
        if inside[0]:
            return fn(*a, **kw)
        e0, e1 = ev(), ev()
        e0.record()
        out = fn(*a, **kw)
        e1.record()
        if label == "gram":
            rows, k = a[0].shape
            records.append((f"56x56 gram {k} (for {a[1].shape[0]} outputs)", "gram", 2.0 * rows * k * k, rows * k * es, e0, e1))
        else:
            n, h, cin, cout = a[1], a[2], a[4], a[9]
            rows = n * h * h
            records.append((f"56x56 1x1/1 {cin}->{cout} (one Gram)", "affine", 2.0 * rows * cin * cout,
                            rows * (cin + cout) * es, e0, e1))
        return out
    return wrapper


cnn.ResNet50Runner._conv_bn = timed_conv_bn_outer
ops.bn_gram_affine_h2 = timed_h2(ops.bn_gram_affine_h2, "gram")
ops.conv2d_affine = timed_h2(ops.conv2d_affine, "affine")
ops.stem_conv_bn_pool = timed_stem
orig_stem_h2 = ops.stem_conv_pool_h2


def timed_stem_h2(frames_u8, *a, **kw):
    e0, e1 = ev(), ev()
    e0.record()
    out = orig_stem_h2(frames_u8, *a, **kw)
    e1.record()
    n = frames_u8.shape[0]
    records.append(("stem 7x7/2 3->64 + pool", "fused h2", 2.0 * n * 112 * 112 * 147 * 64,
                    n * (224 * 224 * 3 + 56 * 56 * 64 * es), e0, e1))
    return out


ops.stem_conv_pool_h2 = timed_stem_h2
ops.bn_gram_affine = timed_op(ops.bn_gram_affine, "gram")
ops.conv1x1_affine = timed_op(ops.conv1x1_affine, "affine")
for _ in range(2):
    runner.forward(frames, gf)
torch.cuda.synchronize()
acc = None
for _ in range(args.reps):
    records.clear()
    t0, t1 = ev(), ev()
    t0.record()
    runner.forward(frames, gf)
    t1.record()
    torch.cuda.synchronize()
    ms = [r[4].elapsed_time(r[5]) for r in records]
    acc = ms if acc is None else [min(a, b) for a, b in zip(acc, ms)]
    total = t0.elapsed_time(t1)
print(f"{args.n} frames, groups of {args.gf}, {args.dtype}: forward {total:.1f} ms = {args.n / total * 1e3:.0f} frames/s "
      f"(events between layers cost a few %)")
print(f"{'layer':34s} {'form':11s} {'ms':>8s} {'TFLOP/s':>8s} {'TB/s':>6s} {'ideal ms':>8s} {'x ideal':>7s} {'excess ms':>9s}")
tsum = isum = 0.0
agg = {}
for (name, form, flops, byts, _, _), ms in zip(records, acc):
    ideal = max(flops / MFMA_IDEAL, byts / 5.5e12) * 1e3
    tsum += ms
    isum += ideal
    k = (name, form)
    a = agg.setdefault(k, [0, 0.0, 0.0, flops, byts])
    a[0] += 1
    a[1] += ms
    a[2] += ideal
for (name, form), (cnt, ms, ideal, flops, byts) in agg.items():
    print(f"{(str(cnt) + ' x ' + name):34s} {form:11s} {ms:8.2f} {flops * cnt / ms / 1e9:8.0f} {byts * cnt / ms / 1e9:6.2f} "
          f"{ideal:8.2f} {ms / ideal:7.2f} {ms - ideal:9.2f}")
print(f"{'sum of layers':46s} {tsum:8.2f} {'':15s} {isum:8.2f} {tsum / isum:7.2f} {tsum - isum:9.2f}")
