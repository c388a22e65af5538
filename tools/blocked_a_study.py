#!/usr/bin/env python3
"""Kernel study (study build): what a channel-blocked activation layout would give the contraction kernel's A operand.
1x1 layers of the trunk with the input stored reduction-step major, x[k / 32][M][32] (avs_debug_flags bit 6: a DMA
instruction then reads whole cache lines of A, as AVS_W_KSTEP32 does for the weights) against the NHWC rows; outputs
must agree bit for bit.  Usage: python tools/blocked_a_study.py [--n 4096]"""
import os
os.environ["AVS_STUDY_LIB"] = "1"
import argparse, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avsum_amd import ops, _abi

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4096)
args = ap.parse_args()
dev = torch.device("cuda", 0)
L = _abi.lib()
dt = torch.bfloat16
shapes = [("l1.conv1 1x1 256->64", 56, 256, 64), ("l2.conv1 1x1 512->128", 28, 512, 128),
          ("l3.conv1 1x1 1024->256", 14, 1024, 256), ("l3.conv3 1x1 256->1024", 14, 256, 1024),
          ("l4.conv1 1x1 2048->512", 7, 2048, 512), ("l4.conv3 1x1 512->2048", 7, 512, 2048)]
for name, hw, cin, cout in shapes:
    # the study switch walks the blocks with the kernel's 32-bit scalar offset: the whole input must stay below 2 GiB
    n = min(args.n, int(1.9e9 // (hw * hw * cin * 2)))
    m = n * hw * hw
    x = (torch.randn(n, hw, hw, cin, device=dev) + 0.3).to(dt)
    xb = x.view(m, cin // 32, 32).permute(1, 0, 2).contiguous().view(n, hw, hw, cin)   # [K/32][M][32] in the same bytes
    w = ops.weights_kstep32((torch.randn(cout, cin, device=dev) / cin ** 0.5).to(dt))
    y = torch.empty(n, hw, hw, cout, device=dev, dtype=dt)
    flops = 2.0 * m * cin * cout
    line, ref = f"{name:26s} {n:5d} frames", None
    for tag, flag, src in (("NHWC rows", 0, x), ("step-major A", 64, xb)):
        L.avs_debug_flags(flag)
        for _ in range(2):
            ops.conv2d(src, w, 1, 1, 1, 0, y, w_layout=1, variant=2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.conv2d(src, w, 1, 1, 1, 0, y, w_layout=1, variant=2)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 5
        if ref is None:
            ref = y.clone()
        line += f" | {tag} {us:8.1f} us {flops / us / 1e6:6.0f} TFLOP/s" + ("" if flag == 0 else (" same" if torch.equal(y, ref) else " DIFFERENT"))
    print(line, flush=True)
L.avs_debug_flags(0)
