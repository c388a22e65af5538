#!/usr/bin/env python3
"""Kernel study: AVBiLSTMModel and MultiHeadSelfAttention forward latency at the sequence lengths of BASELINE.md
(the reference's own CPU numbers: scorer 24.9 / 160 / 434 ms and MHSA 4.9 / 89 / 400 ms at T = 300 / 1800 / 5000)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avsum_amd.models.av_model import AVBiLSTMModel
from avsum_amd.models.attention import MultiHeadSelfAttention

dev = torch.device("cuda", 0)
torch.manual_seed(0)
scorer = AVBiLSTMModel().eval().to(dev)
mhsa = MultiHeadSelfAttention(1024, 4).eval().to(dev)


def timeit(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


with torch.no_grad():
    for t in (300, 1800, 5000):
        v, a = torch.randn(1, t, 4096, device=dev), torch.zeros(1, t, 296, device=dev)
        x = torch.randn(1, t, 1024, device=dev)
        ms_s = timeit(lambda: scorer(v, a))
        mhsa.use_flash = False
        ms_m = timeit(lambda: mhsa(x))
        mhsa.use_flash = "f32"
        ms_f = timeit(lambda: mhsa(x))
        mhsa.use_flash = True
        ms_h = timeit(lambda: mhsa(x))
        print(f"T={t:5d}  AVBiLSTMModel {ms_s:8.3f} ms ({t / ms_s:8.1f} k steps/s)   MHSA(1024,4) GEMM path {ms_m:8.3f} ms "
              f"| fused fp32 MFMA {ms_f:8.3f} ms | fused fp16 split {ms_h:8.3f} ms "
              f"({4.0 * t * t * 1024 * 1e-9 / ms_h:6.1f} TFLOP/s on the two T^2 products, whole forward)", flush=True)
