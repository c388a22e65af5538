#!/usr/bin/env python3
"""The fused AVS_F16X2 stem alone (avs_stem_conv_pool_f16x2) against the unfused sequence it replaces
(avs_frames_normalize_u8 -> avs_conv2d_nhwc_bnstats -> avs_bn_maxpool_nhwc): ms per N frames, us per frame.
    python tools/stem_h2_study.py [--frames 4096] [--fpg 1]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--fpg", type=int, default=1)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    from avsum_amd import ops
    from avsum_amd.cnn import RESNET_MEAN, RESNET_STD, _stem_weight
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1)
    n = args.frames // args.fpg * args.fpg
    frames = torch.randint(0, 256, (n, 224, 224, 3), dtype=torch.uint8, device=dev, generator=g)
    torch.manual_seed(0)
    w4 = (torch.randn(64, 3, 7, 7) * (2.0 / (64 * 49)) ** 0.5).to(dev)
    gamma, beta = torch.randn(64).to(dev), torch.randn(64).to(dev)
    wimg = ops.stem_h2_operands(w4, 1.0, RESNET_MEAN, RESNET_STD)
    code = ops.dtype_code(torch.float32, "f16x2")
    wk = ops.f16x2_pack(_stem_weight(w4, 8, torch.float32).to(dev))

    def fused():
        return ops.stem_conv_pool_h2(frames, wimg, args.fpg, gamma, beta, 1e-5)

    def unfused():
        x0 = ops.frames_normalize(frames, torch.float32, 1.0, RESNET_MEAN, RESNET_STD, 230, 232, 3, 3, code=code)
        raw = torch.empty((n, 112, 112, 64), dtype=torch.float32, device=dev)
        geom, xs = (n, 230, 112, 32, 7, 1, 2, 1, 0, 0, 112, 112, 64), (230 * 232 * 4, 232 * 4, 8)
        sc, sh = ops.conv2d_raw(code, *geom, x0, *xs, wk, wk.stride(0), raw, 64,
                                bnstats=(args.fpg * 112 * 112, gamma, beta, 1e-5))
        rows = torch.arange(0, n + 1, args.fpg, dtype=torch.int64, device=dev) * 112 * 112
        return ops.bn_maxpool(raw, sc, sh, rows, True, 3, 2, 1, torch.empty((n, 56, 56, 64), device=dev), code=code)

    for name, fn in (("fused", fused), ("unfused", unfused)):
        fn()
        torch.cuda.synchronize()
        times = []
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        ms = sorted(times)[len(times) // 2]
        print(f"{name:8s} {n} frames: {ms:8.2f} ms = {ms * 1e3 / n:6.3f} us/frame  "
              f"({2.0 * n * 112 * 112 * 64 * 147 / ms / 1e9:7.1f} TFLOP/s algorithmic)")


if __name__ == "__main__":
    main()
