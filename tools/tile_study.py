#!/usr/bin/env python3
"""Kernel study: the convolution + per-tile statistics launch (the split form's first kernel) per ResNet-50 layer shape
with the 128-row and the 256-row tiles (avs_conv_desc.variant: AVS_TILE_128 / AVS_TILE_256, per call), the latter
also with the weights stored reduction-step major (AVS_W_KSTEP32).
Usage: python tools/tile_study.py [--n 4096]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avsum_amd import ops, _abi

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4096)
args = ap.parse_args()
dev = torch.device("cuda", 0)
L = _abi.lib()
dt = torch.bfloat16
code = ops.dtype_code(dt)
shapes = [  # name, hw, cin, cout, k, stride
    ("l1.conv1 1x1 256->64", 56, 256, 64, 1, 1),
    ("l1.conv2 3x3 64->64", 56, 64, 64, 3, 1),
    ("l2.conv1 1x1 512->128", 28, 512, 128, 1, 1),
    ("l2.conv2 3x3 128->128", 28, 128, 128, 3, 1),
    ("l2.b0.conv2 3x3/2 128->128", 56, 128, 128, 3, 2),
    ("l2.down 1x1/2 256->512", 56, 256, 512, 1, 2),
    ("l3.conv2 3x3 256->256", 14, 256, 256, 3, 1),
    ("l3.conv1 1x1 1024->256", 14, 1024, 256, 1, 1),
    ("l3.conv3 1x1 256->1024", 14, 256, 1024, 1, 1),
]
n = args.n
for name, hw, cin, cout, k, s in shapes:
    pad = k // 2
    ho = (hw + 2 * pad - k) // s + 1
    geom, xs, wrs = (n, hw, hw, cin, k, k, s, s, pad, pad, ho, ho, cout), (hw * hw * cin, hw * cin, cin), k * k * cin
    x = (torch.randn(n, hw, hw, cin, device=dev) + 0.3).to(dt)
    w = (torch.randn(cout, wrs, device=dev) / wrs ** 0.5).to(dt)
    y = torch.empty(n, ho, ho, cout, device=dev, dtype=dt)
    gamma, beta = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    flops = 2.0 * n * ho * ho * cout * wrs
    line = f"{name:28s}"
    wk = ops.weights_kstep32(w)   # reduction-step major image (AVS_W_KSTEP32)
    ref_y = None
    for tag, tall, kstep in (("128-row", 1, 0), ("256-row", 2, 0), ("256-row w-kstep", 2, 1)):
        wsel = wk if kstep else w

        def run():
            return ops.conv2d_raw(code, *geom, x, *xs, wsel, wrs, y, cout, bnstats=(ho * ho, gamma, beta, 1e-5),
                                  w_layout=kstep, variant=tall)
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 5
        if tag == "256-row":
            ref_y = y.clone()
        same = "" if not kstep else (" same" if torch.equal(y, ref_y) else " DIFFERENT")
        line += f" | {tag} {us:8.1f} us {flops / us / 1e6:6.0f} TFLOP/s{same}"
    print(line, flush=True)
