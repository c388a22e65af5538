#!/usr/bin/env python3
"""Run one convolution shape a few times (for rocprofv3 --pmc studies). usage: conv_one.py n hw cin cout k"""
import os
os.environ["AVS_STUDY_LIB"] = "1"  # the ablation switches live in the study build only (make -C <pkg>/csrc study)
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avsum_amd import ops, _abi
n, hw, cin, cout, k = [int(v) for v in sys.argv[1:6]]
if len(sys.argv) > 6:      # optional: debug flags (avs_debug_flags), tall mode
    _abi.lib().avs_debug_flags(int(sys.argv[6]))
variant = int(sys.argv[7]) if len(sys.argv) > 7 else 0   # avs_conv_desc.variant (1 = 128-row, 2 = 256-row tiles)
dev = torch.device("cuda", 0)
H2 = os.environ.get("AVS_ONE_F16X2") == "1"    # AVS_F16X2 operands instead of bf16
dt = torch.float32 if H2 else torch.bfloat16
x = torch.randn(n, hw, hw, cin, device=dev).to(dt)
w = (torch.randn(cout, k * k * cin, device=dev) / (k * k * cin) ** 0.5).to(dt)
if H2:
    x, w = ops.f16x2_pack(x), ops.weights_kstep32(ops.f16x2_pack(w))
y = torch.empty(n, hw, hw, cout, device=dev, dtype=dt)
for _ in range(3):
    ops.conv2d(x, w, k, k, 1, k // 2, y, variant=variant, **({"split": "f16x2", "w_layout": 1} if H2 else {}))
torch.cuda.synchronize()
