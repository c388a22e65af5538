#!/usr/bin/env python3
"""Kernel study: convolution + batch-statistics BatchNorm forms on the ResNet-50 layer shapes (1024-frame chunk,
per-frame groups unless --gf): plain convolution (floor) | split (conv + per-tile statistics, fold, apply) | two-pass 1x1 |
one-launch tile-local form.  Usage: python tools/bn_study.py [--gf 1]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avsum_amd import ops, _abi

ap = argparse.ArgumentParser()
ap.add_argument("--gf", type=int, default=1)
ap.add_argument("--n", type=int, default=1024)
args = ap.parse_args()
dev = torch.device("cuda", 0)
L = _abi.lib()
dt = torch.bfloat16
code = ops.dtype_code(dt)
shapes = [  # name, hw, cin, cout, k, stride, residual
    ("stem 7x7/2 3->64", 0, 0, 64, 7, 2, False),
    ("l1.conv1 1x1 256->64", 56, 256, 64, 1, 1, False),
    ("l1.conv2 3x3 64->64", 56, 64, 64, 3, 1, False),
    ("l1.conv3 1x1 64->256 +res", 56, 64, 256, 1, 1, True),
    ("l2.conv1 1x1 512->128", 28, 512, 128, 1, 1, False),
    ("l2.conv2 3x3 128->128", 28, 128, 128, 3, 1, False),
    ("l2.conv3 1x1 128->512 +res", 28, 128, 512, 1, 1, True),
    ("l3.conv1 1x1 1024->256", 14, 1024, 256, 1, 1, False),
    ("l3.conv2 3x3 256->256", 14, 256, 256, 3, 1, False),
    ("l3.conv3 1x1 256->1024 +res", 14, 256, 1024, 1, 1, True),
    ("l4.conv1 1x1 2048->512", 7, 2048, 512, 1, 1, False),
    ("l4.conv2 3x3 512->512", 7, 512, 512, 3, 1, False),
    ("l4.conv3 1x1 512->2048 +res", 7, 512, 2048, 1, 1, True),
]


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
    return e0.elapsed_time(e1) * 1e3 / reps


n = args.n
tot = {"plain": 0.0, "split": 0.0, "best_old": 0.0, "local": 0.0}
for name, hw, cin, cout, k, s, with_res in shapes:
    if hw == 0:
        geom, xs, wrs = (n, 230, 112, 32, 7, 1, 2, 1, 0, 0, 112, 112, 64), (230 * 232 * 4, 232 * 4, 8), 224
        x = torch.randn(n, 230, 232, 4, device=dev).to(dt)
        ho = 112
    else:
        pad = k // 2
        ho = (hw + 2 * pad - k) // s + 1
        geom, xs, wrs = (n, hw, hw, cin, k, k, s, s, pad, pad, ho, ho, cout), (hw * hw * cin, hw * cin, cin), k * k * cin
        x = (torch.randn(n, hw, hw, cin, device=dev) + 0.3).to(dt)
    w = (torch.randn(cout, wrs, device=dev) / wrs ** 0.5).to(dt)
    rpg = args.gf * ho * ho
    rows = n * ho * ho
    y = torch.empty(n, ho, ho, cout, device=dev, dtype=dt)
    y2 = y.view(-1, cout)
    res = torch.randn(rows, cout, device=dev).to(dt) if with_res else None
    gamma, beta = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    grows = torch.arange(0, rows + 1, rpg, dtype=torch.int64, device=dev)
    byts = (x.numel() + y.numel() * (2 if with_res else 1)) * 2

    def plain():
        ops.conv2d_raw(code, *geom, x, *xs, w, wrs, y, cout)

    def split():
        aff = ops.conv2d_raw(code, *geom, x, *xs, w, wrs, y, cout, bnstats=(rpg, gamma, beta, 1e-5))
        if aff is None:   # groups of fewer than 64 rows: the fused statistics decline, the runner's sequence is this
            ops.conv2d_raw(code, *geom, x, *xs, w, wrs, y, cout)
            aff = ops.bn_batch_stats(y2, grows, gamma, beta, 1e-5)
        ops.bn_apply(y2, aff[0], aff[1], grows, rpg, res, ops.ACT_RELU, y2)

    def twopass():
        ops.conv1x1_bn(x.view(-1, cin), w, rpg, gamma, beta, 1e-5, y2, res, True)

    tile_rows = ops.conv_bnlocal_tile_rows(code, *geom, *xs, wrs, cout, rpg)

    def local():
        ops.conv2d_raw(code, *geom, x, *xs, w, wrs, y, cout, act=ops.ACT_RELU, bnlocal=(rpg, gamma, beta, 1e-5, res))

    t_plain, t_split = timeit(plain), timeit(split)
    t_two = timeit(twopass) if (k == 1 and s == 1 and hw) else float("nan")
    t_loc = timeit(local) if tile_rows is not None else float("nan")
    best_old = min(t_split, t_two) if t_two == t_two else t_split
    tot["plain"] += t_plain; tot["split"] += t_split; tot["best_old"] += best_old
    tot["local"] += t_loc if t_loc == t_loc else best_old
    print(f"{name:28s} plain {t_plain:7.1f}us {byts / t_plain / 1e6:5.2f}TB/s | split {t_split:7.1f} | twopass {t_two:7.1f} "
          f"| local {t_loc:7.1f}us {byts / t_loc / 1e6:5.2f}TB/s  x{best_old / t_loc:4.2f} vs best other", flush=True)
print("totals (one conv of each shape):", {k: round(v, 1) for k, v in tot.items()})
