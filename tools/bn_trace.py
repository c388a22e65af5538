#!/usr/bin/env python3
"""Kernel study: per-block timeline of the one-launch convolution + BatchNorm (dispatch order, XCD, wait length)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from avsum_amd import ops, _abi

dev = torch.device("cuda", 0)
L = _abi.lib()
dt = torch.bfloat16
code = ops.dtype_code(dt)
n, hw, cin, cout = 1024, 56, 64, 256
geom, xs, wrs = (n, hw, hw, cin, 1, 1, 1, 1, 0, 0, hw, hw, cout), (hw * hw * cin, hw * cin, cin), cin
x = (torch.randn(n, hw, hw, cin, device=dev) + 0.3).to(dt)
w = (torch.randn(cout, wrs, device=dev) / wrs ** 0.5).to(dt)
rpg = hw * hw
y = torch.empty(n, hw, hw, cout, device=dev, dtype=dt)
gamma, beta = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
nbytes = ops.conv_bnsync_workspace_bytes(code, *geom, *xs, wrs, cout, rpg)
ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
err = torch.zeros(1, dtype=torch.int32, device=dev)
tiles_m = (n * rpg + 127) // 128
blocks = (tiles_m + 7) // 8 * 8 * (cout // 128)
trace = torch.zeros(blocks * 4, dtype=torch.int64, device=dev)


def run():
    ws.zero_()
    ops.conv2d_raw(code, *geom, x, *xs, w, wrs, y, cout, act=ops.ACT_RELU, bnsync=(rpg, gamma, beta, 1e-5, None, ws, err))


run(); run()
torch.cuda.synchronize()
L.avs_debug_bnsync_trace(trace.data_ptr())
run()
torch.cuda.synchronize()
L.avs_debug_bnsync_trace(None)
t = trace.cpu().numpy().reshape(-1, 4)
ok = t[:, 1] > 0
t = t[ok]
idx = np.flatnonzero(ok)
t0 = t[:, 1].min()
start, arrive, done = (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0, (t[:, 3] - t0) / 100.0  # us
print("blocks", len(t), "kernel span us", done.max())
print("xcc id vs block%8 agreement:", [(int(b), np.bincount(t[idx % 8 == b, 0].astype(int), minlength=8).tolist()) for b in range(8)])
print("main loop us: median %.1f p90 %.1f" % (np.median(arrive - start), np.percentile(arrive - start, 90)))
print("wait us: median %.1f p90 %.1f max %.1f" % (np.median(done - arrive), np.percentile(done - arrive, 90), (done - arrive).max()))
order = np.argsort(start, kind="stable")
inv = np.abs(idx[order] - np.sort(idx))
print("dispatch displacement (|block at start-rank r - r-th block|): median %d p90 %d max %d" % (np.median(inv), np.percentile(inv, 90), inv.max()))
for lo in (0, 1000, 5000, 20000):
    sel = slice(lo, lo + 16)
    print("blocks", idx[sel][:16].tolist())
    print("  xcc  ", t[sel, 0].tolist())
    print("  start", np.round(start[sel], 1).tolist())
    print("  wait ", np.round((done - arrive)[sel], 1).tolist())
for q in (0.1, 0.5, 0.9):
    tq = done.max() * q
    res = np.sum((start <= tq) & (done + 2.0 > tq))
    waiting = np.sum((arrive <= tq) & (done > tq))
    print(f"t={tq:8.1f}us resident~{res} waiting {waiting}")
