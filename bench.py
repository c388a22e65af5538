#!/usr/bin/env python3
"""bench.py — frames/sec end-to-end (extract + fuse + score) on synthetic 224x224x3 frames.

    python bench.py --gpus N --steps K --warmup W

Workload at N=1 = BASELINE.json configs[1]: SumMe-shape batch, 25 videos x ~1.8k frames
(lengths ~N(1800,300) clipped to [900,2700], seed 2002), visual-only (audio = the literal
zeros(296)), ResNet-50 extractor in the reference's batch-statistics BatchNorm mode
(every frame its own one-frame shot / micro-batch), bf16 MFMA, then the AVBiLSTMModel scorer
(fp32) and the mean-threshold selection.  One step = one pass of that path over the whole batch,
uint8 frames already resident in HBM.  For N>1 every rank runs its own batch of the same shape
(weak scaling; videos are independent, so the data path has no collective) and the per-video
scores are all-gathered (C2) inside the timed region.

Prints ONE JSON line on rank 0 (contract in the task description), with
  roofline     — the dominant kernel (implicit-GEMM conv, MFMA-bound): algorithmic FLOPs of all its
                 launches in the timed region / their summed HIP-event durations;
  cpu_baseline — the oracle (CPU restatement of the same path) timed on the host cores on a bounded
                 sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r01_e_pmc_traffic.json")
RESNET50_FLOP_PER_FRAME = 2 * 4.0878e9


T_START = time.perf_counter()


def log(msg):
    """Progress on stderr (stdout carries exactly one JSON line)."""
    print(f"[bench {time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


def video_lengths(num_videos, mean, std, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    ln = (torch.randn(num_videos, generator=g) * std + mean).round().clamp(lo, hi).long()
    return [int(v) for v in ln]


def make_frames(total, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    frames = torch.empty((total, 224, 224, 3), dtype=torch.uint8, device=device)
    step = 2048
    for a in range(0, total, step):
        b = min(total, a + step)
        frames[a:b] = torch.randint(0, 256, (b - a, 224, 224, 3), dtype=torch.uint8, device=device, generator=g)
    return frames


def cpu_baseline(trunk_sd, scorer_sd, sample_frames, use_inception, inception_sd):
    """Oracle leg: the CPU restatement (oracle/) of the same per-frame path on a bounded sample."""
    from oracle import cnn as ocnn, scorer as osc
    # the GPU box grants a CPU share of 16 cores per GPU; asking torch for every visible core (256 on the host)
    # oversubscribes that share and is ~1000x slower
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(16, visible)
    torch.set_num_threads(cores)
    rng = np.random.default_rng(7)
    frames = rng.integers(0, 256, (sample_frames, 224, 224, 3), dtype=np.uint8)
    t0 = time.perf_counter()
    with torch.no_grad():
        feats = []
        for f in frames:  # per-frame mode: every frame is its own micro-batch (train-mode BN over one frame)
            r = ocnn.resnet50_trunk_forward(trunk_sd, ocnn.preprocess_frame(f))
            if use_inception:
                i = ocnn.inception_v3_forward(inception_sd, ocnn.preprocess_inception(f))
            else:
                i = torch.zeros(1, 2048)
            feats.append(torch.cat([r, i], 1))
        visual = torch.cat(feats).unsqueeze(0)
        scores = osc.av_bilstm_forward(scorer_sd, visual, torch.zeros(1, sample_frames, 296))
        _ = np.flatnonzero(scores.numpy() > scores.numpy().mean())
    dt = time.perf_counter() - t0
    return {"value": sample_frames / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{sample_frames} frames of the same workload (one video), oracle/ torch-CPU fp32, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--videos", type=int, default=25)
    ap.add_argument("--mean-frames", type=int, default=1800)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--extractor", default="resnet50", choices=["resnet50", "resnet50+inception3"])
    ap.add_argument("--chunk", type=int, default=12288,
                    help="frames per pass of the trunk (activations of one pass: ~60 GB of the 288 GB at 12288)")
    ap.add_argument("--cpu-sample", type=int, default=512, help="frames for the CPU baseline (0 = skip)")
    ap.add_argument("--frames-per-group", type=int, default=1,
                    help="BatchNorm micro-batch inside a video: 1 = every frame its own shot (the per-frame scoring "
                         "reading of north_star, default); 4 = the reference's micro-batches of 4 (extractors.py:48)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--profile-every", type=int, default=1,
                    help="bracket every n-th launch of each kernel kind with HIP events (1 = all: ~3%% slower)")
    ap.add_argument("--bn-local", default="on", choices=["on", "off"],
                    help="tuning: one-launch tile-local convolution + BatchNorm on the layers that take it")
    ap.add_argument("--short-k-bytes", type=int, default=None, help="tuning: avs_tune_short_reduction_bytes")
    ap.add_argument("--tall", default=None, help="tuning: mode[,min_tiles[,min_k_bytes]] of avs_tune_tall_tiles")
    ap.add_argument("--fuse", default=None, help="tuning: min_rows,ratio_num,ratio_den of the one-kernel conv+BN")
    args = ap.parse_args()

    from avsum_amd import dist as avd, ops
    from avsum_amd.features.extractors import VisualFeatureExtractor
    from avsum_amd.models.av_model import AVBiLSTMModel
    from avsum_amd.pipeline import FrameScoringPipeline

    if args.short_k_bytes is not None:
        from avsum_amd import _abi
        _abi.lib().avs_tune_short_reduction_bytes(args.short_k_bytes)
    if args.tall is not None:
        from avsum_amd import _abi
        tv = [int(v) for v in args.tall.split(",")] + [0, -1]
        _abi.lib().avs_tune_tall_tiles(tv[0], tv[1], tv[2])
    rank, world, local = avd.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    use_inception = args.extractor != "resnet50"

    # seeded random-init weights of the reference architectures (no pretrained files offline)
    torch.manual_seed(7)
    extractor = VisualFeatureExtractor(dtype, "batch")
    scorer = AVBiLSTMModel().eval()
    sd_cpu = None
    if rank == 0 and args.cpu_sample > 0 and world == 1:
        sd_cpu = ({k: v.clone() for k, v in extractor.resnet.state_dict().items()},
                  {k: v.clone() for k, v in scorer.state_dict().items()},
                  {k: v.clone() for k, v in extractor.inception.state_dict().items()} if use_inception else None)
    extractor._resnet_runner.bn_local = args.bn_local == "on"
    if args.fuse is not None:
        r = extractor._resnet_runner
        r.fuse_min_rows, r.fuse_ratio_num, r.fuse_ratio_den = [int(v) for v in args.fuse.split(",")]
    extractor = extractor.to(dev)
    scorer = scorer.to(dev)
    avd.broadcast_module(extractor, 0)  # C1
    avd.broadcast_module(scorer, 0)

    lengths = video_lengths(args.videos, args.mean_frames, 300, args.mean_frames // 2,
                            args.mean_frames * 3 // 2, 2002 + rank)
    offsets = [0]
    for ln in lengths:
        offsets.append(offsets[-1] + ln)
    total = offsets[-1]
    log(f"rank {rank}: {args.videos} videos, {total} frames; generating frames in HBM")
    frames = make_frames(total, dev, 1000 + rank)
    torch.cuda.synchronize()
    log("frames ready")
    pipe = FrameScoringPipeline(extractor, scorer, use_inception=use_inception, chunk_frames=args.chunk,
                                frames_per_group=args.frames_per_group)

    def step():
        scores = pipe.score(frames, offsets)
        if world > 1:
            avd.gather_video_scores(scores, list(range(args.videos)), lengths, args.videos)
        return pipe.select(scores, offsets)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    counter = ops.LaunchProfiler(count_only=True)
    per_step = 0
    for i in range(args.warmup):
        counter.count = 0
        ops.set_profiler(counter)
        step()
        ops.set_profiler(None)
        per_step = counter.count
        torch.cuda.synchronize()
        log(f"warmup step {i} done")
    prof = None
    if not args.no_profile:
        # the HIP events of the timed launches are created here, outside the timed region (creation is the costly
        # part); inside it they are only recorded, on the stream the kernels are launched on
        prof = ops.LaunchProfiler(prealloc=per_step * args.steps // args.profile_every + 64,
                                   sample_every=args.profile_every)
        ops.set_profiler(prof)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        selected = step()
        log(f"timed step {i} issued")
    barrier()
    elapsed = time.perf_counter() - t0
    ops.set_profiler(None)

    t_all = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    n_all = torch.tensor([float(total)], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t_all, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(n_all, op=torch.distributed.ReduceOp.SUM)
    t_max = float(t_all.item())
    frames_all = float(n_all.item())

    if rank == 0:
        roofline = None
        if prof is not None:
            summ = prof.summary()
            code = ops.dtype_code(dtype)
            conv = summ.get(("conv", code))
            if conv and conv["ms"] > 0:
                achieved = conv["flops"] / (conv["ms"] * 1e-3) / 1e12
                peak = MFMA_BF16_PEAK_TFLOPS if dtype == torch.bfloat16 else MFMA_F32_PEAK_TFLOPS
                # HBM bytes per launch from the separate rocprofv3 --pmc passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE,
                # KiB -> bytes), summarised by profiles/summarize_pmc.py; null when no such summary is committed
                traffic = None
                if os.path.exists(PMC_TRAFFIC_FILE):
                    try:
                        traffic = json.load(open(PMC_TRAFFIC_FILE)).get("igemm_kernel", {}).get("hbm_bytes_per_launch")
                    except (OSError, ValueError):
                        traffic = None
                roofline = {"bound": "mfma", "kernel": "igemm_kernel (avs_conv2d_nhwc[_bnstats|_bnlocal])",
                            "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                            "frac": round(achieved / peak, 4), "traffic": traffic,
                            "launches": prof.seen.get(("conv", code), conv["launches"]),
                            "timed_launches": conv["launches"],
                            "avg_launch_us": round(conv["ms"] * 1e3 / conv["launches"], 2),
                            "algorithmic_flop_per_launch": round(conv["flops"] / conv["launches"], 1),
                            "algorithmic_bytes_per_launch": round(conv["bytes"] / conv["launches"], 1),
                            "share_of_step": round(conv["ms"] * 1e-3 / elapsed *
                                                   prof.seen.get(("conv", code), conv["launches"]) / conv["launches"], 3)}
            others = []
            for kind, label in (("convbn", "conv1x1_bn_kernel (avs_conv1x1_bn[_in]_bf16)"),
                                ("bn_apply", "bn_apply_kernel / bn_maxpool_kernel (avs_bn_apply, avs_bn_maxpool_nhwc)")):
                rec = summ.get((kind, code))
                if rec and rec["ms"] > 0:
                    gbs = rec["flops"] / (rec["ms"] * 1e-3) / 1e9
                    others.append({"bound": "hbm", "kernel": label, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                   "launches": prof.seen.get((kind, code), rec["launches"]),
                                   "timed_launches": rec["launches"],
                                   "avg_launch_us": round(rec["ms"] * 1e3 / rec["launches"], 2),
                                   "algorithmic_bytes_per_launch": round(rec["flops"] / rec["launches"], 1),
                                   "share_of_step": round(rec["ms"] * 1e-3 / elapsed *
                                                          prof.seen.get((kind, code), rec["launches"]) / rec["launches"], 3)})
            if roofline is not None and others:
                roofline["other_kernels"] = others
        cpu = None
        if sd_cpu is not None:
            log(f"timed region {elapsed:.2f}s; CPU baseline on {args.cpu_sample} frames")
            cpu = cpu_baseline(sd_cpu[0], sd_cpu[1], args.cpu_sample, use_inception, sd_cpu[2])
        out = {
            "metric": "frames/sec end-to-end (extract+fuse+score), 224x224 + 16kHz",
            "value": round(frames_all * args.steps / t_max, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(t_max * 1e3 / args.steps, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic (uniform uint8 frames, seeded random-init weights; audio = literal zeros(296))",
            "config": {"workload": f"configs[1]: SumMe-shape batch, {args.videos} videos x ~{args.mean_frames} frames "
                                   f"({total} frames/GPU), visual-only {args.extractor} extractor (batch-stat BN, "
                                   f"{'per-frame shots' if args.frames_per_group == 1 else str(args.frames_per_group) + '-frame micro-batches'}) "
                                   f"+ AVBiLSTM attention scorer + mean-threshold selection",
                       "frames_per_gpu": total, "videos_per_gpu": args.videos, "extractor": args.extractor,
                       "chunk_frames": args.chunk, "parallelism": f"videos sharded x{world}, no data-path collective",
                       "selected_frames_rank0": int(sum(len(s) for s in selected))},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
