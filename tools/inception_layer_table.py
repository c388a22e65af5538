#!/usr/bin/env python3
"""Per-launch accounting of the Inception-v3 runner (f16x2): every C-ABI call of one forward pass bracketed with events IN
PLACE (the real pass, not isolated kernels), grouped by the network's stages.
    python tools/inception_layer_table.py [--frames 1024]
Per launch: time per frame, algorithmic matrix rate, stream rate (4-byte input + output elements / time)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    from avsum_amd import ops
    from avsum_amd.cnn import Inception3, InceptionV3Runner
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = Inception3().eval().to(dev)
    g = torch.Generator(device=dev).manual_seed(1)
    frames = torch.randint(0, 256, (args.frames, 299, 299, 3), dtype=torch.uint8, device=dev, generator=g)
    r = InceptionV3Runner(net, torch.float32, f32_split="f16x2")
    r.forward(frames)
    torch.cuda.synchronize()
    rec = []
    live = [False]
    depth = [0]

    def wrap(name, describe):
        orig = getattr(ops, name)

        def f(*a, **kw):
            if not live[0] or depth[0]:          # (ops.conv2d calls ops.conv2d_raw: the outer call is the one recorded)
                return orig(*a, **kw)
            depth[0] += 1
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig(*a, **kw)
            e1.record()
            depth[0] -= 1
            rec.append((name, describe(*a, **kw), e0, e1))
            return out
        setattr(ops, name, f)

    def d_conv(x, wt, kh, kw_, s, pad, out, *a, **kw):
        n, h, w, cin = x.shape
        cout = wt.shape[0]
        ho, wo = out.shape[1], out.shape[2]
        return (f"{h}x{w}x{cin} -> {ho}x{wo}x{cout} k{kh}x{kw_} s{s}", 2.0 * n * ho * wo * cout * cin * kh * kw_,
                4.0 * (x.numel() + n * ho * wo * cout))

    def d_raw(code, n, h, ho, *a, **kw):
        return (f"stem 299x299x3 -> 149x149x32 k3x3 s2", 2.0 * n * 149 * 149 * 32 * 27, 4.0 * n * (299 * 300 * 4 + 149 * 149 * 32))

    def d_split(x, wt, out, n_split, out2, *a, **kw):
        n, h, w, cin = x.shape
        cout = wt.shape[0]
        return (f"{h}x{w}x{cin} -> {h}x{w}x({n_split}+{cout - n_split}) k1x1 stacked", 2.0 * n * h * w * cout * cin,
                4.0 * (x.numel() + n * h * w * cout))

    def d_pool(x, mode, k, s, p, out, *a, **kw):
        return (f"{mode}pool {x.shape[1]}x{x.shape[2]}x{x.shape[3]} k{k} s{s}", 0.0, 4.0 * (x.numel() + out.numel()))

    def d_norm(fr, *a, **kw):
        return ("normalise 299x299x3 u8 -> 299x300x4", 0.0, fr.numel() + 4.0 * fr.shape[0] * 299 * 300 * 4)

    def d_gap(x, *a, **kw):
        return (f"global avgpool {x.shape[1]}x{x.shape[2]}x{x.shape[3]}", 0.0, 4.0 * x.numel())

    wrap("conv2d", d_conv)
    wrap("conv2d_raw", d_raw)
    wrap("conv2d_split", d_split)
    wrap("pool2d", d_pool)
    wrap("frames_normalize", d_norm)
    wrap("global_avgpool", d_gap)
    live[0] = True
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(args.reps):
        r.forward(frames)
    t1.record()
    torch.cuda.synchronize()
    per = len(rec) // args.reps
    tot = 0.0
    n = args.frames
    print(f"{per} launches per pass, {n} frames; whole pass {t0.elapsed_time(t1) / args.reps * 1e3 / n:.2f} us/frame (with the event brackets)")
    print(f"{'#':>3} {'call':16s} {'shape':52s} {'us/frame':>9} {'TFLOP/s':>8} {'TB/s':>6}")
    rows = []
    for i in range(per):
        ms = sum(rec[i + k * per][2].elapsed_time(rec[i + k * per][3]) for k in range(args.reps)) / args.reps
        name, (desc, fl, by), _, _ = rec[i]
        rows.append((ms, name, desc, fl, by))
        tot += ms
        print(f"{i:3d} {name:16s} {desc:52s} {ms * 1e3 / n:9.3f} {fl / ms / 1e9:8.1f} {by / ms / 1e9:6.2f}")
    print(f"sum of the launches: {tot * 1e3 / n:.2f} us/frame")
    bykind = {}
    for ms, name, desc, fl, by in rows:
        k = name if name != "conv2d" else ("conv2d 1x1" if "k1x1" in desc else "conv2d kxk")
        d = bykind.setdefault(k, [0.0, 0.0, 0.0, 0])
        d[0] += ms; d[1] += fl; d[2] += by; d[3] += 1
    for k, (ms, fl, by, cnt) in sorted(bykind.items(), key=lambda kv: -kv[1][0]):
        print(f"{k:18s} {cnt:3d} launches {ms * 1e3 / n:7.2f} us/frame = {ms / tot * 100:5.1f} %  {fl / ms / 1e9:7.1f} TFLOP/s {by / ms / 1e9:5.2f} TB/s")


if __name__ == "__main__":
    main()
