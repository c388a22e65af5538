#!/usr/bin/env python3
"""Accuracy study of the bf16 throughput path against the fp32 parity path (both HIP): how far are the features,
the importance scores and the selection of the benchmarked configuration from the fp32 result?

The fp32 HIP path is itself held to the oracle by the -m gpu tests (scores <= 1e-4), so it is the yardstick here;
tests/test_gpu_accuracy.py repeats the end-to-end numbers against oracle/ directly.

    python tools/accuracy_study.py [--videos 2] [--frames 300] [--spread 6]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from avsum_amd.evaluation.metrics import compute_temporal_f1, segments_from_indices, select_frames
from avsum_amd.features.extractors import VisualFeatureExtractor
from avsum_amd.models.av_model import AVBiLSTMModel
from avsum_amd.pipeline import FrameScoringPipeline


def gt_segments(n, seed):
    """Seeded synthetic ground truth: ~15 % of the frames in 4 segments."""
    rng = np.random.default_rng(seed)
    starts = np.sort(rng.choice(n - n // 25, 4, replace=False))
    segs, last = [], 0
    for s in starts:
        s = max(int(s), last)
        e = min(n, s + n // 25)
        if e > s:
            segs.append((s, e))
        last = e
    return segs


def report(tag, scores, ref, offsets):
    err = np.abs(scores - ref).max()
    agree, drift = [], []
    for v, (a, b) in enumerate(zip(offsets[:-1], offsets[1:])):
        s, r = scores[a:b], ref[a:b]
        agree.append(np.mean((s > s.mean()) == (r > r.mean())))
        gt = gt_segments(b - a, 900 + v)
        f = compute_temporal_f1(segments_from_indices(select_frames(s)), gt, b - a)
        fr = compute_temporal_f1(segments_from_indices(select_frames(r)), gt, b - a)
        drift.append(abs(f - fr))
    print(f"{tag:34s} score max|err| {err:.3e}  (range {ref.max() - ref.min():.3e})  selection agreement "
          f"{np.mean(agree):.4f}  F1 drift max {max(drift):.4f} mean {np.mean(drift):.4f}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--videos", type=int, default=2)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--spread", type=float, default=6.0)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(7)
    ext32 = VisualFeatureExtractor(torch.float32, "batch")
    scorer = AVBiLSTMModel().eval()
    with torch.no_grad():
        scorer.scorer[0].weight.mul_(args.spread)
        scorer.scorer[2].weight.mul_(args.spread)
    ext16 = VisualFeatureExtractor(torch.bfloat16, "batch")
    ext16.load_state_dict(ext32.state_dict())
    ext32, ext16, scorer = ext32.to(dev), ext16.to(dev), scorer.to(dev)
    n = args.videos * args.frames
    offsets = [i * args.frames for i in range(args.videos + 1)]
    g = torch.Generator(device=dev).manual_seed(1000)
    frames = torch.randint(0, 256, (n, 224, 224, 3), dtype=torch.uint8, device=dev, generator=g)
    for fpg in (1, 4):
        p32 = FrameScoringPipeline(ext32, scorer, use_inception=False, chunk_frames=256, frames_per_group=fpg)
        p16 = FrameScoringPipeline(ext16, scorer, use_inception=False, chunk_frames=12288, frames_per_group=fpg)
        with torch.no_grad():
            f32 = p32.embed(frames, offsets)[:, :2048].cpu()
            s32 = p32.score(frames, offsets).cpu().numpy()
        for tag, knobs in (("bf16 default", {}), ("bf16 no local", {"bn_local": False}),
                           ("bf16 no fused conv+bn", {"bn_local": False, "fuse_conv_bn": False, "defer_bn_apply": False})):
            r = ext16._resnet_runner
            saved = {k: getattr(r, k) for k in knobs}
            for k, v in knobs.items():
                setattr(r, k, v)
            with torch.no_grad():
                f16 = p16.embed(frames, offsets)[:, :2048].cpu()
                s16 = p16.score(frames, offsets).cpu().numpy()
                s16b = p16.score(frames, offsets).cpu().numpy()
            for k, v in saved.items():
                setattr(r, k, v)
            rel = ((f16 - f32).norm() / f32.norm()).item()
            cos = torch.nn.functional.cosine_similarity(f16, f32, dim=1).min().item()
            print(f"fpg={fpg} {tag}: feature rel L2 {rel:.4f}, min cosine {cos:.5f}, run-to-run identical: "
                  f"{np.array_equal(s16, s16b)}")
            report(f"fpg={fpg} {tag}", s16, s32, offsets)
        # what a feature perturbation of a given relative size does to the selection (the scorer's sensitivity)
        for eps in (1e-3, 1e-2):
            gg = torch.Generator().manual_seed(5)
            pert = f32 + eps * f32.norm() / f32.numel() ** 0.5 * torch.randn(f32.shape, generator=gg)
            vis = torch.cat([pert, torch.zeros(n, 2048)], 1).to(dev)
            seq = torch.tensor(offsets, dtype=torch.int64, device=dev)
            with torch.no_grad():
                sp = scorer.score_rows(vis, torch.zeros(n, 296, device=dev), seq).cpu().numpy()
            report(f"fpg={fpg} fp32 + {eps:g} gaussian noise", sp, s32, offsets)


if __name__ == "__main__":
    main()
