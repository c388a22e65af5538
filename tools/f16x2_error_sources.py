#!/usr/bin/env python3
"""Where the f16x2 trunk's feature error against the CPU oracle comes from at 4-frame groups: the runner's round-4 switches
(fused stem, clustered tile-local BatchNorm) on / off.   python tools/f16x2_error_sources.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from avsum_amd.cnn import ResNet50Runner, resnet50_trunk
    from oracle import cnn as ocnn
    dev = torch.device("cuda", 0)
    torch.manual_seed(21)
    trunk = resnet50_trunk()
    sd = {k: v.clone() for k, v in trunk.state_dict().items()}
    trunk = trunk.to(dev)
    frames = np.random.default_rng(1).integers(0, 256, (8, 224, 224, 3), dtype=np.uint8)
    fd = torch.from_numpy(frames).to(dev)
    for gsize in (1, 4):
        with torch.no_grad():
            x = torch.cat([ocnn.preprocess_frame(f) for f in frames])
            ref = torch.cat([ocnn.resnet50_trunk_forward(sd, x[i:i + gsize]) for i in range(0, 8, gsize)])
        groups = list(range(0, 9, gsize))
        scale = max(1.0, ref.abs().max().item())
        g32 = ResNet50Runner(trunk, torch.float32, "batch").forward(fd, groups).cpu()
        print(f"[gsize {gsize}] exact fp32 (GPU) vs oracle {(g32 - ref).abs().max().item() / scale:.2e}")
        for stem, cluster, local, p8 in ((True, True, True, True), (False, True, True, True), (True, False, True, True),
                                         (False, False, True, True), (False, False, False, True), (False, False, False, False)):
            r = ResNet50Runner(trunk, torch.float32, "batch", f32_split="f16x2")
            r.fused_stem, r.bn_cluster, r.bn_local = stem, cluster, local
            if not p8:
                r.p8_blocks = ()
            got = r.forward(fd, groups).cpu()
            print(f"   fused stem {stem!s:5} clustered {cluster!s:5} tile-local {local!s:5} p8 {p8!s:5}: vs oracle "
                  f"{(got - ref).abs().max().item() / scale:.2e}, vs exact fp32 (GPU) {(got - g32).abs().max().item() / scale:.2e}")


if __name__ == "__main__":
    main()
