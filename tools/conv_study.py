#!/usr/bin/env python3
"""Kernel study: time single convolution shapes of the ResNet-50 trunk (1024-frame chunk) with ablations."""
import os
os.environ["AVS_STUDY_LIB"] = "1"  # the ablation switches live in the study build only (make -C <pkg>/csrc study)
import sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avsum_amd import ops, _abi

dev = torch.device("cuda", 0)
L = _abi.lib()
shapes = [  # name, n, hw, cin, cout, k, stride
    ("stem-like 3x3 32->64 s2", 1024, 112, 32, 64, 3, 2),
    ("l1.conv3 1x1 64->256", 1024, 56, 64, 256, 1, 1),
    ("l1.conv1 1x1 256->64", 1024, 56, 256, 64, 1, 1),
    ("l1.conv2 3x3 64->64", 1024, 56, 64, 64, 3, 1),
    ("l2.conv3 1x1 128->512", 1024, 28, 128, 512, 1, 1),
    ("l3.conv3 1x1 256->1024", 1024, 14, 256, 1024, 1, 1),
    ("l3.conv2 3x3 256->256", 1024, 14, 256, 256, 3, 1),
    ("l4.conv2 3x3 512->512", 1024, 7, 512, 512, 3, 1),
    ("l2.conv2 3x3 128->128", 1024, 28, 128, 128, 3, 1),
    ("l3.conv1 1x1 1024->256", 1024, 14, 1024, 256, 1, 1),
    ("l4.conv1 1x1 2048->512", 1024, 7, 2048, 512, 1, 1),
]
H2 = "--f16x2" in sys.argv   # AVS_F16X2 operands (fp16 hi | lo runs) instead of bf16
dt = torch.float32 if H2 else torch.bfloat16
for name, n, hw, cin, cout, k, s in shapes:
    x = torch.randn(n, hw, hw, cin, device=dev).to(dt)
    w = (torch.randn(cout, k * k * cin, device=dev) / (k * k * cin) ** 0.5).to(dt)
    if H2:
        x, w = ops.f16x2_pack(x), ops.f16x2_pack(w)
    ho = hw // s
    y = torch.empty(n, ho, ho, cout, device=dev, dtype=dt)
    flops = 2.0 * n * ho * ho * cout * k * k * cin
    byts = (x.numel() + y.numel() + w.numel()) * (4 if H2 else 2)
    line = f"{name:26s}"
    for tag, flags, rowb, pipe, tall in (("auto", 0, 2048, 1, 0), ("nostage", 8, 2048, 1, 0), ("noA", 16, 2048, 1, 0),
                                         ("noB", 32, 2048, 1, 0), ("noload", 2, 2048, 1, 0)):
        L.avs_debug_flags(flags)
        L.avs_tune_short_reduction_bytes(rowb)
        L.avs_tune_pipeline(pipe)
        for _ in range(2):
            ops.conv2d(x, w, k, k, s, k // 2, y, variant=tall, **({"split": "f16x2"} if H2 else {}))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.conv2d(x, w, k, k, s, k // 2, y, variant=tall, **({"split": "f16x2"} if H2 else {}))
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 5
        line += f" | {tag} {us:7.1f}us {flops / us / 1e6:6.0f}TF {byts / us / 1e6:5.2f}TB/s"
    print(line, flush=True)
L.avs_debug_flags(0)
L.avs_tune_pipeline(1)
L.avs_tune_short_reduction_bytes(2048)
