#!/usr/bin/env python3
"""Inception-v3 runner alone (f16x2): frames/s for the runner's switches.
    python tools/inception_study.py [--frames 2048]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2048)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    from avsum_amd import ops
    from avsum_amd.cnn import Inception3, InceptionV3Runner
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = Inception3().eval().to(dev)
    g = torch.Generator(device=dev).manual_seed(1)
    frames = torch.randint(0, 256, (args.frames, 299, 299, 3), dtype=torch.uint8, device=dev, generator=g)
    ref = None
    for name, kw in (("baseline", dict(split_tail_columns=False, stack_heads=False)),
                     ("tail columns split", dict(split_tail_columns=True, stack_heads=False)),
                     ("stacked heads", dict(split_tail_columns=False, stack_heads=True, stack_pool_head=False)),
                     ("stacked heads + pool", dict(split_tail_columns=False, stack_heads=True, stack_pool_head=True))):
        r = InceptionV3Runner(net, torch.float32, f32_split="f16x2")
        for k, v in kw.items():
            if not hasattr(r, k):
                break
            setattr(r, k, v)
        else:
            out = r.forward(frames)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                out = r.forward(frames)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.reps
            if ref is None:
                ref = out
            print(f"{name:20s}: {dt * 1e3:8.1f} ms = {args.frames / dt:9.1f} frames/s = {dt * 1e6 / args.frames:6.2f} us/frame; "
                  f"max |diff| vs baseline {(out - ref).abs().max().item():.2e} (features up to {ref.abs().max().item():.2f})")
            continue
        print(f"{name:20s}: switch not available")


if __name__ == "__main__":
    main()
