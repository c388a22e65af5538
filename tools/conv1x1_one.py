#!/usr/bin/env python3
"""Run the one-pass 1x1 convolution (avs_conv1x1_affine_bf16, XF, + residual) on one shape a few times (for rocprofv3
--pmc studies: tools/ta_counters.sh).  usage: conv1x1_one.py frames rows_per_group k n"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avsum_amd import ops
frames, rpg, k, n = [int(v) for v in sys.argv[1:5]]
dev = torch.device("cuda", 0)
rows = frames * rpg
x = torch.randn(rows, k, device=dev).bfloat16()
w = (torch.randn(n, k, device=dev) / k ** 0.5).bfloat16()
res = torch.randn(rows, n, device=dev).bfloat16()
out = torch.empty(rows, n, dtype=torch.bfloat16, device=dev)
sc, sh = torch.rand(frames, n, device=dev) + 0.5, torch.randn(frames, n, device=dev)
ia = (torch.rand(frames, k, device=dev) + 0.5, torch.randn(frames, k, device=dev))
for _ in range(3):
    ops.conv1x1_affine(x, w, rpg, sc, sh, out, res, True, ia)
torch.cuda.synchronize()
