"""Multi-GPU layer: one process per GPU, torch.distributed over RCCL ("nccl" backend on ROCm) / xGMI.

The hot path shards over VIDEOS: every stage is per-video, so the data path has no
collective.  The only exchanges are
  C1  one-time broadcast of the weights from rank 0,
  C2  gather of the per-video scores: all-gather of the lengths (int64), then one all-gather of a
      buffer padded to the largest shard (8 MB in BASELINE config 4; a single fused collective —
      on the point-to-point xGMI mesh that is 7 concurrent 1-hop writes per GPU, latency-bound).
Works unchanged with the gloo backend on CPU tensors (tests/test_dist_cpu.py, world_size 2).
"""
import os

import torch
import torch.distributed as dist

# True: the three exchanges below run their real collective path even in a process group of ONE rank (normally a
# single rank returns early).  A one-GPU box can then exercise the product's flat-bucket broadcast, padded score
# all-gather and gradient bucket on RCCL device tensors (tests/test_gpu_configs.py); set by tests only.
FORCE_COLLECTIVES = os.environ.get("AVS_DIST_FORCE_COLLECTIVES") == "1"


def _collective_path():
    return dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


def init_from_env(backend=None):
    """Initialise from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun); returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # REHEARSAL on a box with fewer GPUs than ranks (tests only): AVS_DIST_REHEARSAL=1 maps every rank onto the devices that
    # exist (local % device_count) and runs the collectives on gloo (RCCL refuses two ranks on one GPU) - the multi-rank
    # program flow on real kernels, not a measurement
    if os.environ.get("AVS_DIST_REHEARSAL") == "1" and torch.cuda.is_available():
        local = local % max(1, torch.cuda.device_count())
        backend = backend or "gloo"
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_videos(lengths, world):
    """Longest-processing-time-first assignment of videos to ranks by frame count.
    Returns a list (per rank) of video indices, each in ascending order.  Deterministic."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    loads = [0] * world
    shards = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += int(lengths[i])
    return [sorted(s) for s in shards]


def broadcast_module(module, src=0):
    """C1: weights (parameters and buffers) from rank `src` to every rank.  The copies go through the parameters
    themselves (not ``.data``), so their version counters move and the kernel-layout weight caches of the runners
    (keyed on data_ptr + _version) are rebuilt even if a forward ran before the broadcast."""
    if not _collective_path():
        return
    with torch.no_grad():
        tensors = list(module.parameters()) + list(module.buffers())
        # one flat bucket per dtype: few large collectives instead of hundreds of small ones
        by_dtype = {}
        for t in tensors:
            by_dtype.setdefault(t.dtype, []).append(t)
        for _, group in sorted(by_dtype.items(), key=lambda kv: str(kv[0])):
            flat = torch.cat([t.detach().reshape(-1) for t in group])
            dist.broadcast(flat, src)
            o = 0
            for t in group:
                t.copy_(flat[o:o + t.numel()].view_as(t))
                o += t.numel()


def allreduce_gradients(module, average=True):
    """C3: data-parallel gradient exchange for the training loop (scripts/train_av_model.py:94-96 run on one
    video per rank): one flat fp32 bucket (38.7 MB for the scorer) all-reduced over RCCL, then averaged.
    On the 8-GPU xGMI mesh a bucket this size is per-link bound (2*(7/8)*38.7 MB / 153 GB/s ~ 0.44 ms)."""
    if not _collective_path():
        return
    grads = [p.grad for p in module.parameters() if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()
    o = 0
    for g in grads:
        g.copy_(flat[o:o + g.numel()].view_as(g))
        o += g.numel()


def gather_video_scores(local_scores, local_video_ids, local_lengths, num_videos):
    """C2: every rank contributes the concatenated scores of its videos; every rank receives the list
    of per-video score tensors in global video order."""
    if not _collective_path():
        out, o = [None] * num_videos, 0
        for vid, ln in zip(local_video_ids, local_lengths):
            out[vid] = local_scores[o:o + ln]
            o += ln
        return out
    world = dist.get_world_size()
    dev = local_scores.device
    # (video id, length) per slot, built on the host and moved once; one host read-back of all ranks' tables
    rows = [[int(v), int(n)] for v, n in zip(local_video_ids, local_lengths)]
    rows += [[-1, -1]] * (num_videos - len(rows))
    meta = torch.tensor(rows, dtype=torch.int64).reshape(num_videos, 2).to(dev)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    metas = list(torch.stack(metas).cpu())
    totals = [int(m[:, 1].clamp(min=0).sum()) for m in metas]
    pad = max(max(totals), 1)
    buf = torch.zeros(pad, dtype=local_scores.dtype, device=dev)
    buf[:local_scores.numel()] = local_scores
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    out = [None] * num_videos
    for r in range(world):
        o = 0
        for vid, ln in metas[r].tolist():
            if vid < 0:
                continue
            out[vid] = bufs[r][o:o + ln]
            o += ln
    return out
