#!/usr/bin/env python3
"""Kernel study: the one-pass 1x1 convolution (Gram statistics + avs_conv1x1_affine_bf16) per shape: algorithmic TB/s of
the convolution pass alone and of the Gram pass, against the two-pass kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from avsum_amd import ops

dev = torch.device("cuda", 0)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4096


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, rpg, k, n, with_res, xf in (("l1.conv3 64->256 +res, XF", 3136, 64, 256, True, True), ("l1.ds 64->256", 3136, 64, 256, False, False),
                                      ("l2.conv3 128->512 +res, XF", 784, 128, 512, True, True), ("64->128 +res (one column tile)", 3136, 64, 128, True, True),
                                      ("64->128 no res", 3136, 64, 128, False, True), ("64->256 no res, XF", 3136, 64, 256, False, True)):
    rows = frames * rpg
    x = torch.randn(rows, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) / k ** 0.5).bfloat16()
    gamma, beta = torch.ones(n, device=dev), torch.zeros(n, device=dev)
    res = torch.randn(rows, n, device=dev).bfloat16() if with_res else None
    ia = (torch.rand(frames, k, device=dev) + 0.5, torch.randn(frames, k, device=dev)) if xf else None
    out = torch.empty(rows, n, dtype=torch.bfloat16, device=dev)
    t_gram = timeit(lambda: ops.bn_gram_affine(x, w, rpg, gamma, beta, 1e-5, ia))
    t_one = timeit(lambda: ops.conv1x1_gram_bn(x, w, rpg, gamma, beta, 1e-5, out, res, True, ia))
    t_two = timeit(lambda: ops.conv1x1_bn(x, w, rpg, gamma, beta, 1e-5, out, res, True, ia))
    byts = 2.0 * rows * (k + n * (2 if with_res else 1))
    print(f"{name:34s} gram {t_gram:6.2f} ms | conv pass {t_one - t_gram:6.2f} ms = {byts / (t_one - t_gram) / 1e9:5.2f} TB/s | "
          f"one-pass total {t_one:6.2f} | two-pass {t_two:6.2f} ms", flush=True)
    del x, res, out
