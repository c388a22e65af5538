#!/usr/bin/env python3
"""Kernel study: how much of a matrix-core-bound layer's time depends on the DATA (not on the instructions).
l3.conv2 (14x14, 3x3, 256 -> 256, tile-local BatchNorm form, AVS_F16X2, 4096 frames) is timed with the same launch on
different operand contents: random activations, all zeros, all ones, random activations whose fp16 lo halves keep only
N mantissa bits (activations AND weights).  Same kernel, same loads, same instruction stream - on MI355X the all-zero
input runs 22 % faster, all-ones 10 %, lo halves cut to 6 / 3 / 0 mantissa bits 3 / 5 / 7 %: these layers run at a
POWER-limited clock, and an operand-staging ablation that feeds zeros ("no A loads") measures the clock, not the staging.

    python tools/mfma_power_study.py            # prints one line per variant
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avsum_amd import ops

dev = torch.device("cuda", 0)
code = ops.dtype_code(torch.float32, "f16x2")
n, hw, cin, cout, k = 4096, 14, 256, 256, 3
rpg = hw * hw
geom, xs, wrs = (n, hw, hw, cin, k, k, 1, 1, 1, 1, hw, hw, cout), (hw * hw * cin, hw * cin, cin), k * k * cin
gamma, beta = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
y = torch.empty(n, hw, hw, cout, device=dev)


def trim_lo(packed, bits):
    """Keep `bits` mantissa bits of every fp16 lo half (a run of 8 slots = 8 hi halves, then 8 lo halves)."""
    if bits < 10:
        packed.view(torch.int16).view(-1, 16)[:, 8:] &= ~((1 << (10 - bits)) - 1)
    return packed


def timed(x, w):
    def run():
        ops.conv2d_raw(code, *geom, x, *xs, w, wrs, y, cout, act=ops.ACT_RELU, bnlocal=(rpg, gamma, beta, 1e-5, None),
                       w_layout=1)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 100


torch.manual_seed(0)
xv = torch.randn(n, hw, hw, cin, device=dev) + 0.3
wv = torch.randn(cout, wrs, device=dev) / wrs ** 0.5
for tag, xin, lo in (("random", xv, 10), ("zeros", torch.zeros_like(xv), 10), ("ones", torch.ones_like(xv), 10),
                     ("random x 1e-3", xv * 1e-3, 10), ("random, lo halves 6 bits", xv, 6),
                     ("random, lo halves 3 bits", xv, 3), ("random, lo halves 0 bits", xv, 0), ("random", xv, 10)):
    x = trim_lo(ops.f16x2_pack(xin), lo)
    w = ops.weights_kstep32(trim_lo(ops.f16x2_pack(wv), lo))
    print(f"{tag:28s} {timed(x, w):8.1f} us", flush=True)
