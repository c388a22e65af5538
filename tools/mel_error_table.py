#!/usr/bin/env python3
"""Measured log2-mel / MFCC errors per test signal: |HIP - float64 value|, |fp32 CPU
reference (torch.stft) - float64 value| and |HIP - fp32 CPU reference| for the signals the -m gpu tests use."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from avsum_amd.audio import MelPlan
from oracle import audio as oa

dev = torch.device("cuda", 0)
plan = MelPlan.get(16000, 128, 40, dev)


def wave(kind, t, seed=0, noise=0.05):
    g = torch.Generator().manual_seed(seed)
    ts = torch.arange(t) / 16000.0
    if kind == "sine":
        return 0.5 * torch.sin(2 * np.pi * 440.0 * ts)
    return (0.4 * torch.sin(2 * np.pi * 220 * ts) + 0.3 * torch.sin(2 * np.pi * 1333 * ts)
            + 0.2 * torch.sin(2 * np.pi * 5200 * ts) + noise * torch.randn(t, generator=g))


print("signal | max|HIP - f64| | max|fp32 ref - f64| | max|HIP - fp32 ref| | share of bins with |HIP - fp32 ref| > 1e-4")
for name, w in (("3 tones + 0.05 noise, 1 s", wave("multi", 16000)), ("3 tones + 0.05 noise, 10 s", wave("multi", 160000)),
                ("3 tones + 0.01 noise, 1 s", wave("multi", 16000, noise=0.01)),
                ("3 tones + 0.001 noise, 1 s (60 dB)", wave("multi", 16000, noise=0.001)),
                ("440 Hz sine, 10 s", wave("sine", 160000)), ("white noise 0.1, 10 s", 0.1 * torch.randn(160000))):
    got = plan.log2_mel(w.to(dev)).cpu().numpy()
    ref32 = oa.extract_mel(w)
    truth = oa.extract_mel_f64(w.numpy())
    print(f"{name:36s} | {np.abs(got - truth).max():.2e} | {np.abs(ref32 - truth).max():.2e} | "
          f"{np.abs(got - ref32).max():.2e} | {(np.abs(got - ref32) > 1e-4).mean():.2e}")
w = wave("multi", 48000, 3)
ref = oa.mfcc(w).t()
got = plan.mfcc(w.to(dev)).cpu()
print(f"MFCC (3 s): max|HIP - fp32 ref| {(got - ref).abs().max().item():.2e} on coefficients up to {ref.abs().max().item():.1f}")
