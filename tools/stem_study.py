#!/usr/bin/env python3
"""Kernel study: the fused stem (avs_stem_conv_bn_pool_bf16) against the unfused sequence on one pass of frames.
Usage: python tools/stem_study.py [frames]   (rocprofv3 --pmc ... -- python tools/stem_study.py for SQ counters)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from avsum_amd import ops
from avsum_amd.cnn import RESNET_MEAN, RESNET_STD, _stem_weight

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
frames = torch.randint(0, 256, (n, 224, 224, 3), dtype=torch.uint8, device=dev, generator=g)
wk = _stem_weight(torch.randn(64, 3, 7, 7) * 0.025, 8, torch.bfloat16).to(dev)
gamma, beta = torch.ones(64, device=dev), torch.zeros(64, device=dev)


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def fused():
    return ops.stem_conv_bn_pool(frames, wk, 1.0, RESNET_MEAN, RESNET_STD, 1, gamma, beta, 1e-5)


def unfused():
    x0 = ops.frames_normalize(frames, torch.bfloat16, 1.0, RESNET_MEAN, RESNET_STD, 230, 232, 3, 3)
    raw = torch.empty((n, 112, 112, 64), dtype=torch.bfloat16, device=dev)
    geom, xs = (n, 230, 112, 32, 7, 1, 2, 1, 0, 0, 112, 112, 64), (230 * 232 * 4, 232 * 4, 8)
    sc, sh = ops.conv2d_raw(ops.dtype_code(torch.bfloat16), *geom, x0, *xs, wk, wk.stride(0), raw, 64,
                            bnstats=(112 * 112, gamma, beta, 1e-5))
    rows = torch.arange(0, n + 1, dtype=torch.int64, device=dev) * 112 * 112
    return ops.bn_maxpool(raw, sc, sh, rows, True, 3, 2, 1, torch.empty((n, 56, 56, 64), dtype=torch.bfloat16, device=dev))


tf, tu = timeit(fused), timeit(unfused)
print(f"{n} frames: fused stem {tf:.2f} ms ({n / tf:.0f} frames/ms), unfused sequence {tu:.2f} ms")
