#!/usr/bin/env python3
"""Where the PCIe-inclusive step loses its ~2 %: the stream-of-batches host path (a) as it is, (b) with the pull kernel
reading a DEVICE mirror of the host frames (the same launches, events and staging buffers, no PCIe), (c) with the uploads
skipped altogether (staging buffers pre-filled: only the events and the pass structure remain), against the resident path.
    python tools/h2d_where_study.py [--steps 3] [--fpg 4]
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
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--fpg", type=int, default=4)
    ap.add_argument("--pull", default="16", help="workgroups of the pull kernel, comma separated (the PCIe rows are repeated per value)")
    args = ap.parse_args()
    from avsum_amd import ops, synthetic
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
    pipe = FrameScoringPipeline(ext, scorer, use_inception=False, chunk_frames=12288, frames_per_group=args.fpg)
    fb = frames[0].numel()

    def timed(src, stream=False):
        nb = (src, offsets) if stream else None
        for _ in range(2):
            pipe.score(src, offsets, next_batch=nb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            s = pipe.score(src, offsets, next_batch=nb)
        torch.cuda.synchronize()
        return total * args.steps / (time.perf_counter() - t0), s

    real_pull = ops.pull_copy

    from ctypes import c_void_p
    from avsum_amd import _abi

    def mirror_pull(src, dst, wgs):   # the same kernel, its source a device pointer (the wrapper insists on pinned memory)
        off = (src.data_ptr() - host.data_ptr()) // fb
        m = frames[off:off + src.shape[0]]
        _abi.check(_abi.lib().avs_pull_copy_u8(c_void_p(m.data_ptr()), c_void_p(dst.data_ptr()), dst.numel(), int(wgs),
                                               c_void_p(torch.cuda.current_stream().cuda_stream)), "avs_pull_copy_u8")
        return dst

    def no_pull(src, dst, wgs):
        return dst

    def sleep_pull(src, dst, wgs):    # a kernel that only SPINS for about as long as the PCIe pull of the same frames would take
        torch.cuda._sleep(int(src.numel() / 57e9 * 2.4e9))
        return dst

    import avsum_amd.pipeline as pl
    for rep in range(2):
        base, s0 = timed(frames)
        print(f"resident in HBM                         : {base:9.1f} frames/s", flush=True)
        for wg in [int(v) for v in args.pull.split(",")][1:]:
            pipe.host_pull_workgroups = wg
            r, s = timed(host, stream=True)
            print(f"host, pull kernel x{wg:<3d} over PCIe        : {r:9.1f} frames/s = {r / base:.4f} of resident", flush=True)
        pipe.host_pull_workgroups = int(args.pull.split(",")[0])
        for name, fn in (("host, pull kernel over PCIe", real_pull), ("pull kernel from a device mirror", mirror_pull),
                         ("uploads skipped (events only)", no_pull),
                         ("a spinning kernel of the pull's duration", sleep_pull)):
            ops.pull_copy = fn
            pl.ops.pull_copy = fn
            r, s = timed(host, stream=True)
            print(f"{name:40s}: {r:9.1f} frames/s = {r / base:.4f} of resident; scores identical: "
                  f"{bool(torch.equal(s, s0)) if fn in (real_pull, mirror_pull) else 'n/a'}", flush=True)
        ops.pull_copy = real_pull
        pl.ops.pull_copy = real_pull


if __name__ == "__main__":
    main()
