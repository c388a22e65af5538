#!/usr/bin/env python3
"""Kernel study: throughput of the mel / MFCC front end (BASELINE configs[2] shape: minutes of 16 kHz audio)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avsum_amd.audio import MelPlan

dev = torch.device("cuda", 0)
plan = MelPlan.get(16000, 128, 40, dev)
for secs in (10, 240, 1800):
    t = 16000 * secs
    wave = torch.randn(t, device=dev) * 0.1
    from avsum_amd import ops
    from avsum_amd.audio import N_BINS
    unfused = lambda w: ops.power_mel(plan.spectrum(w), N_BINS, plan.fb, plan.fb_lo, plan.fb_hi, 0)
    for fn, name in ((plan.log2_mel, "log2_mel"), (plan.mfcc, "mfcc"), (plan.log2_mel_and_mfcc, "mel+mfcc"),
                     (unfused, "unfused")):
        for _ in range(2):
            fn(wave)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn(wave)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        frames = 1 + t // 200
        algo_bytes = 4 * t + 512 * frames          # 4 B/sample in + 512 B per STFT frame out
        flops = 2.0 * 400 * 402 * frames           # dense real DFT (the fused kernel's folded DFT does half of it)
        print(f"{name:9s} {secs:5d}s audio: {ms:8.3f} ms  {t / ms / 1e3:8.1f} Msamples/s  "
              f"{algo_bytes / ms / 1e6:7.1f} GB/s algorithmic  {flops / ms / 1e9:6.1f} TFLOP/s fp64 DFT", flush=True)
