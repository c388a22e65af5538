#!/usr/bin/env python3
"""Kernel study, AVS_F16X2: what each epilogue form costs on top of the plain contraction, per ResNet-50 layer shape.
plain = avs_conv2d_nhwc (raw output), stats = avs_conv2d_nhwc_bnstats, local = avs_conv2d_nhwc_bnlocal (without / with
residual), affine = avs_conv2d_nhwc_affine (with residual).  Usage: python tools/h2_forms_study.py [--n 4096]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avsum_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda", 0)
code = ops.dtype_code(torch.float32, "f16x2")
shapes = [  # name, hw, cin, cout, k, stride
    ("l1.conv1 1x1 256->64", 56, 256, 64, 1, 1),
    ("l1.conv2 3x3 64->64", 56, 64, 64, 3, 1),
    ("l1.conv3 1x1 64->256", 56, 64, 256, 1, 1),
    ("l2.conv1 1x1 512->128", 28, 512, 128, 1, 1),
    ("l2.conv2 3x3 128->128", 28, 128, 128, 3, 1),
    ("l2.conv3 1x1 128->512", 28, 128, 512, 1, 1),
    ("l3.conv1 1x1 1024->256", 14, 1024, 256, 1, 1),
    ("l3.conv2 3x3 256->256", 14, 256, 256, 3, 1),
    ("l3.conv3 1x1 256->1024", 14, 256, 1024, 1, 1),
    ("l3.0.conv2 3x3/2 256->256", 28, 256, 256, 3, 2),
    ("l3.0.ds 1x1/2 512->1024", 28, 512, 1024, 1, 2),
    ("l4.conv1 1x1 2048->512", 7, 2048, 512, 1, 1),
    ("l4.conv2 3x3 512->512", 7, 512, 512, 3, 1),
    ("l4.conv3 1x1 512->2048", 7, 512, 2048, 1, 1),
]
n = args.n


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / args.reps


print(f"{n} frames, per-frame groups; us per launch (algorithmic TFLOP/s)")
for name, hw, cin, cout, k, s in shapes:
    pad = k // 2
    ho = (hw + 2 * pad - k) // s + 1
    rpg = ho * ho
    geom, xs, wrs = (n, hw, hw, cin, k, k, s, s, pad, pad, ho, ho, cout), (hw * hw * cin, hw * cin, cin), k * k * cin
    x = ops.f16x2_pack(torch.randn(n, hw, hw, cin, device=dev) + 0.3)
    w = ops.weights_kstep32(ops.f16x2_pack(torch.randn(cout, wrs, device=dev) / wrs ** 0.5))
    y = torch.empty(n, ho, ho, cout, device=dev)
    res = ops.f16x2_pack(torch.randn(n * ho * ho, cout, device=dev))
    gamma, beta = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    flops = 2.0 * n * ho * ho * cout * wrs
    line = f"{name:26s}"

    def show(tag, us):
        return f" | {tag} {us:8.1f} ({flops / us / 1e6:4.0f})"
    line += show("plain", timed(lambda: ops.conv2d_raw(code, *geom, x, *xs, w, wrs, y, cout, w_layout=1)))
    if rpg >= 64:
        line += show("stats", timed(lambda: ops.conv2d_raw(code, *geom, x, *xs, w, wrs, y, cout,
                                                           bnstats=(rpg, gamma, beta, 1e-5), w_layout=1)))
        if True:
            line += show("stats/128", timed(lambda: ops.conv2d_raw(code, *geom, x, *xs, w, wrs, y, cout,
                                                                   bnstats=(rpg, gamma, beta, 1e-5), w_layout=1, variant=1)))
    if ops.conv_bnlocal_tile_rows(code, *geom, *xs, wrs, cout, rpg) is not None:
        line += show("local", timed(lambda: ops.conv2d_raw(code, *geom, x, *xs, w, wrs, y, cout, act=ops.ACT_RELU,
                                                           bnlocal=(rpg, gamma, beta, 1e-5, None), w_layout=1)))
        line += show("local+res", timed(lambda: ops.conv2d_raw(code, *geom, x, *xs, w, wrs, y, cout, act=ops.ACT_RELU,
                                                               bnlocal=(rpg, gamma, beta, 1e-5, res), w_layout=1)))
        if 192 < rpg <= 224 and cout % 128 == 0:   # the library's choice above is the 224-row tile: the 256-row one beside it
            line += show("local/256", timed(lambda: ops.conv2d_raw(code, *geom, x, *xs, w, wrs, y, cout, act=ops.ACT_RELU,
                                                                   bnlocal=(rpg, gamma, beta, 1e-5, None), w_layout=1,
                                                                   variant=2)))
            line += show("local+res/256", timed(lambda: ops.conv2d_raw(code, *geom, x, *xs, w, wrs, y, cout, act=ops.ACT_RELU,
                                                                       bnlocal=(rpg, gamma, beta, 1e-5, res), w_layout=1,
                                                                       variant=2)))
    if k == 1 and cin * 4 > 128:
        groups = n
        sc, sh = torch.rand(groups, cout, device=dev) + 0.5, torch.randn(groups, cout, device=dev)
        line += show("affine+res", timed(lambda: ops.conv2d_affine(code, n, hw, hw, cin, s, s, ho, ho, cout, x, *xs, w, wrs,
                                                                   y, cout, rpg, sc, sh, res, True, None, w_layout=1)))
        if cout > 64:
            line += show("affine+res/128", timed(lambda: ops.conv2d_affine(code, n, hw, hw, cin, s, s, ho, ho, cout, x, *xs, w,
                                                                           wrs, y, cout, rpg, sc, sh, res, True, None,
                                                                           w_layout=1, variant=1)))
    print(line, flush=True)

# the Gram statistics pass of the expanding 1x1 layers (it is also the apply pass of the layer before: reads the raw
# input, writes the finished one in place): GB/s over read + write
for name, hw, k, nout in (("gram l1 (56x56, 64 -> 256)", 56, 64, 256), ("gram l2 (28x28, 128 -> 512)", 28, 128, 512)):
    rows = n * hw * hw
    x = ops.f16x2_pack(torch.randn(rows, k, device=dev) + 0.3)
    w = ops.f16x2_pack(torch.randn(nout, k, device=dev) / k ** 0.5)
    gamma, beta = torch.rand(nout, device=dev) + 0.5, torch.randn(nout, device=dev)
    isc, ish = torch.rand(n, k, device=dev) + 0.5, torch.randn(n, k, device=dev) * 0.1
    us = timed(lambda: ops.bn_gram_affine_h2(x, w, hw * hw, gamma, beta, 1e-5, (isc, ish), store_input=True))
    us0 = timed(lambda: ops.bn_gram_affine_h2(x, w, hw * hw, gamma, beta, 1e-5))
    print(f"{name:28s} | affine in, stored {us:8.1f} us ({2 * rows * k * 4 / us / 1e3:6.0f} GB/s) | finished in {us0:8.1f} us "
          f"({rows * k * 4 / us0 / 1e3:6.0f} GB/s)", flush=True)
