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

from avsum_amd import synthetic
from avsum_amd.evaluation.accuracy import accuracy_report
from avsum_amd.features.extractors import VisualFeatureExtractor
from avsum_amd.models.av_model import AVBiLSTMModel
from avsum_amd.pipeline import FrameScoringPipeline


def report(tag, scores, ref, offsets):
    r = accuracy_report(scores, ref, offsets)
    print(f"{tag:40s} score max|err| {r['score_max_abs_err']:.3e}  (range {r['score_range']:.3e})  selection agreement "
          f"{r['selection_agreement']:.4f}  F1 drift max {r['f1_drift_max']:.4f} mean {r['f1_drift_mean']:.4f}", flush=True)


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
    for kind, fpg in (("uniform", 1), ("uniform", 4), ("scenes", 1), ("scenes", 4)):
        frames = (synthetic.make_frames_uniform(n, dev, 1000) if kind == "uniform"
                  else synthetic.make_frames_scenes([args.frames] * args.videos, dev, 1000))
        print(f"---- {kind} frames, {fpg} frame(s) per BatchNorm group")
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
            print(f"{kind} fpg={fpg} {tag}: feature rel L2 {rel:.4f}, min cosine {cos:.5f}, run-to-run identical: "
                  f"{np.array_equal(s16, s16b)}")
            report(f"{kind} fpg={fpg} {tag}", s16, s32, offsets)
        # what a feature perturbation of a given relative size does to the selection (the scorer's sensitivity)
        for eps in (1e-3, 1e-2):
            gg = torch.Generator().manual_seed(5)
            pert = f32 + eps * f32.norm() / f32.numel() ** 0.5 * torch.randn(f32.shape, generator=gg)
            vis = torch.cat([pert, torch.zeros(n, 2048)], 1).to(dev)
            seq = torch.tensor(offsets, dtype=torch.int64, device=dev)
            with torch.no_grad():
                sp = scorer.score_rows(vis, torch.zeros(n, 296, device=dev), seq).cpu().numpy()
            report(f"{kind} fpg={fpg} fp32 + {eps:g} gaussian noise", sp, s32, offsets)


if __name__ == "__main__":
    main()
