"""Synthetic inputs of the five BASELINE.json configs, seeded and shape-faithful.

There is no network for SumMe / TVSum, so every config is generated: uint8 frames [N,224,224,3] on the device
(6.8 GB for configs[1], 37.6 GB for one rank's share of configs[3]: they live in HBM, never on the host), mono
16 kHz waveforms, video lengths from the distributions the configs name.  ``config(i, rank, world)`` returns the
description of config i for one rank; the ``make_*`` functions materialise it.

  configs[0]  one 300-frame video + 10 s 440 Hz sine, 10 uniform shots (the CPU-runnable case)
  configs[1]  SumMe shape: 25 videos, lengths ~N(1800, 300) clipped to [900, 2700]           (the bench headline)
  configs[2]  TVSum shape: 50 videos, lengths ~U[2000, 10000] frames at 30 fps, audio = 3 sines + noise
  configs[3]  400 videos x 5000 frames sharded over the ranks (longest-first assignment): 50 per rank at 8 ranks
  configs[4]  training on synthetic labels ~U[1,5]: scripts.train_av_model.SyntheticShotDataset
"""
import math

import numpy as np
import torch

from . import dist as avd

FPS = 30.0
SAMPLE_RATE = 16000


def _normal_lengths(num, mean, std, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    ln = (torch.randn(num, generator=g) * std + mean).round().clamp(lo, hi).long()
    return [int(v) for v in ln]


def _uniform_lengths(num, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return [int(v) for v in torch.randint(lo, hi + 1, (num,), generator=g)]


def config(index, rank=0, world=1, videos=None, mean_frames=None):
    """Per-rank description of BASELINE config `index`: {"name", "video_ids" (global), "lengths", "num_videos"
    (global), "seed"}.  configs[1] / [2] are weak-scaling shapes: every rank gets its own batch of the same
    distribution (global id = rank * videos + i); configs[3] is ONE global list sharded over the ranks."""
    if index == 0:
        return {"name": "configs[0]: one 300-frame video", "video_ids": [rank], "lengths": [300],
                "num_videos": world, "seed": 1001 + rank}
    if index == 1:
        v = videos or 25
        m = mean_frames or 1800
        return {"name": f"configs[1]: SumMe-shape batch, {v} videos x ~{m} frames",
                "video_ids": [rank * v + i for i in range(v)],
                "lengths": _normal_lengths(v, m, 300, m // 2, m * 3 // 2, 2002 + rank),
                "num_videos": world * v, "seed": 1000 + rank}
    if index == 2:
        v = videos or 50
        return {"name": f"configs[2]: TVSum-shape batch, {v} videos x U[2000,10000] frames + 16 kHz audio",
                "video_ids": [rank * v + i for i in range(v)], "lengths": _uniform_lengths(v, 2000, 10000, 3003 + rank),
                "num_videos": world * v, "seed": 3000 + rank}
    if index == 3:
        # the config is 400 videos at 8 ranks; with fewer ranks the same 50-videos-per-rank share is kept (the
        # whole list is 301 GB of frames: it only fits the node, not one GPU)
        per_rank = videos or 50
        total = per_rank * world
        lengths = [mean_frames or 5000] * total
        mine = avd.shard_videos(lengths, world)[rank]
        return {"name": f"configs[3]: {total} videos x {lengths[0]} frames sharded over {world} rank(s)",
                "video_ids": mine, "lengths": [lengths[i] for i in mine], "num_videos": total, "seed": 4004 + rank}
    raise ValueError("configs[4] is the training loop: scripts.train_av_model.train_synthetic")


def offsets_of(lengths):
    offs = [0]
    for ln in lengths:
        offs.append(offs[-1] + int(ln))
    return offs


def make_frames_uniform(total, device, seed, step=2048):
    """uint8 [total,224,224,3] ~ U{0..255}, generated on the device in slabs (random 224x224x3)."""
    g = torch.Generator(device=device).manual_seed(seed)
    frames = torch.empty((total, 224, 224, 3), dtype=torch.uint8, device=device)
    for a in range(0, total, step):
        b = min(total, a + step)
        frames[a:b] = torch.randint(0, 256, (b - a, 224, 224, 3), dtype=torch.uint8, device=device, generator=g)
    return frames


def make_frames_scenes(lengths, device, seed, scene_frames=60, noise=6.0):
    """Scene-structured videos: every `scene_frames` frames a new smooth random image (a 7x7x3 field upsampled to
    224x224), drifting by one pixel per frame, plus sensor-like noise.  Unlike uniform noise, different scenes
    give clearly different embeddings - the regime the accuracy figures of a real video live in."""
    g = torch.Generator(device=device).manual_seed(seed)
    total = int(sum(lengths))
    frames = torch.empty((total, 224, 224, 3), dtype=torch.uint8, device=device)
    o = 0
    for ln in lengths:
        for s0 in range(0, ln, scene_frames):
            n = min(scene_frames, ln - s0)
            field = torch.rand((1, 3, 7, 7), device=device, generator=g) * 255.0
            base = torch.nn.functional.interpolate(field, size=(224 + scene_frames, 224), mode="bilinear",
                                                   align_corners=False)[0].permute(1, 2, 0)      # [H+drift, W, 3]
            idx = torch.arange(n, device=device)[:, None] + torch.arange(224, device=device)[None, :]
            clip = base[idx]                                                                     # [n,224,224,3]
            clip = clip + noise * torch.randn(clip.shape, device=device, generator=g)
            frames[o + s0:o + s0 + n] = clip.clamp_(0, 255).to(torch.uint8)
        o += ln
    return frames


def make_waveform(num_samples, seed, kind="tones"):
    """float32 [num_samples] at 16 kHz: 'sine' = 0.5 sin(2 pi 440 t) (configs[0]); 'tones' = three sines + 0.01 N(0,1)
    (configs[2])."""
    t = torch.arange(num_samples, dtype=torch.float64) / SAMPLE_RATE
    if kind == "sine":
        return (0.5 * torch.sin(2 * math.pi * 440.0 * t)).float()
    g = torch.Generator().manual_seed(seed)
    freqs = 200.0 + 3000.0 * torch.rand(3, generator=g, dtype=torch.float64)
    x = sum(0.25 * torch.sin(2 * math.pi * f * t) for f in freqs)
    return (x + 0.01 * torch.randn(num_samples, generator=g, dtype=torch.float64)).float()


def uniform_shots(length, shot_frames=30):
    """[(start, end)] shots of `shot_frames` frames covering a video (configs[0]: 10 shots of 30)."""
    return [(a, min(a + shot_frames, length)) for a in range(0, length, shot_frames)]


def host_frames_uniform(count, seed):
    """numpy uint8 [count,224,224,3] for the CPU-baseline / accuracy sample (same distribution as the device frames)."""
    return np.random.default_rng(seed).integers(0, 256, (count, 224, 224, 3), dtype=np.uint8)
