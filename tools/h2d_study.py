#!/usr/bin/env python3
"""PCIe-inclusive headline step: frames in pinned host memory vs resident in HBM, for several lead-pass sizes
(FrameScoringPipeline.host_lead_frames: the first pass's upload is the only one with nothing to hide behind).
    python tools/h2d_study.py [--steps 2]
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
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--leads", default="1024,0")
    ap.add_argument("--pull", default="0,4,8,16,32", help="0 = torch copy_ (copy engine); n = avs_pull_copy_u8 with n workgroups")
    args = ap.parse_args()
    from avsum_amd import synthetic
    from avsum_amd.features.extractors import VisualFeatureExtractor
    from avsum_amd.models.av_model import AVBiLSTMModel
    from avsum_amd.pipeline import FrameScoringPipeline
    dev = torch.device("cuda", 0)
    torch.manual_seed(7)
    ext = VisualFeatureExtractor(torch.float32, "batch", f32_split="f16x2").to(dev)
    scorer = AVBiLSTMModel().eval().to(dev)
    cfg = synthetic.config(1, 0, 1)
    offsets = synthetic.offsets_of(cfg["lengths"])
    total = offsets[-1]
    frames = synthetic.make_frames_uniform(total, dev, cfg["seed"])
    host = torch.empty(frames.shape, dtype=torch.uint8, pin_memory=True)
    host.copy_(frames)
    torch.cuda.synchronize()
    pipe = FrameScoringPipeline(ext, scorer, use_inception=False, chunk_frames=12288, frames_per_group=1)

    def timed(src, stream=False):
        nb = (src, offsets) if stream else None
        pipe.score(src, offsets, next_batch=nb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            s = pipe.score(src, offsets, next_batch=nb)
        torch.cuda.synchronize()
        return total * args.steps / (time.perf_counter() - t0), s

    base, s_dev = timed(frames)
    print(f"resident in HBM: {base:9.1f} frames/s")
    # the raw upload rate alone
    buf = torch.empty_like(frames[:12288])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    buf.copy_(host[:12288], non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"one 12288-frame upload alone: {dt * 1e3:.1f} ms = {12288 * 150528 / dt / 1e9:.1f} GB/s")
    for pull in [int(v) for v in args.pull.split(",")]:
        pipe.host_pull_workgroups = pull
        if pull:
            from avsum_amd import ops
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ops.pull_copy(host[:12288], buf, pull)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"pull kernel, {pull} workgroups: one 12288-frame upload alone {dt * 1e3:.1f} ms = "
                  f"{12288 * 150528 / dt / 1e9:.1f} GB/s; equal: {torch.equal(buf, frames[:12288])}")
        rate, s_host = timed(host, stream=True)
        print(f"pinned host, {'pull kernel x' + str(pull) if pull else 'copy engine'}, stream of batches (next first pass prefetched): "
              f"{rate:9.1f} frames/s = {rate / base:.4f} of resident; scores identical: {torch.equal(s_host, s_dev)}")
        for lead in [int(v) for v in args.leads.split(",")]:
            pipe.host_lead_frames = lead
            rate, s_host = timed(host)
            print(f"pinned host, {'pull kernel x' + str(pull) if pull else 'copy engine'}, lead pass {lead:5d}: {rate:9.1f} frames/s = "
                  f"{rate / base:.4f} of resident; scores identical: {torch.equal(s_host, s_dev)}")


if __name__ == "__main__":
    main()
