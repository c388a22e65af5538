"""Multi-process logic on CPU with the gloo backend (world_size 2): sharding plan, weight broadcast,
ragged score gather.  The data path itself has no collective (videos are independent)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from avsum_amd import dist as avd
    r, w, _ = avd.init_from_env("gloo")
    assert (r, w) == (rank, world)
    lengths = [5, 9, 2, 7, 4, 4, 1]
    shards = avd.shard_videos(lengths, world)
    mine = shards[rank]
    torch.manual_seed(rank)
    lin = torch.nn.Linear(8, 3)
    bn = torch.nn.BatchNorm1d(3)
    mod = torch.nn.Sequential(lin, bn)
    avd.broadcast_module(mod, 0)
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Linear(8, 3), torch.nn.BatchNorm1d(3))
    for a, b in zip(mod.state_dict().values(), ref.state_dict().values()):
        assert torch.equal(a, b)
    local = torch.cat([torch.full((lengths[v],), float(v)) + torch.arange(lengths[v]) / 100 for v in mine]) \
        if mine else torch.zeros(0)
    out = avd.gather_video_scores(local, mine, [lengths[v] for v in mine], len(lengths))
    for v, ln in enumerate(lengths):
        assert torch.equal(out[v], torch.full((ln,), float(v)) + torch.arange(ln) / 100)
    # C3: gradient all-reduce (averaged) of a module whose grads differ per rank
    for i, p_ in enumerate(mod.parameters()):
        p_.grad = torch.full_like(p_, float(rank + 1 + i))
    avd.allreduce_gradients(mod)
    for i, p_ in enumerate(mod.parameters()):
        assert torch.allclose(p_.grad, torch.full_like(p_, 1.5 + i))
    ret[rank] = True
    dist.destroy_process_group()


def test_gloo_world2():
    port = 29500 + os.getpid() % 2000
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
        assert dict(ret) == {0: True, 1: True}


def test_shard_videos_balanced_and_deterministic():
    sys.path.insert(0, ROOT)
    from avsum_amd.dist import shard_videos
    lengths = [5000] * 400
    shards = shard_videos(lengths, 8)
    assert all(len(s) == 50 for s in shards)
    assert sorted(sum(shards, [])) == list(range(400))
    ragged = [1800 + (i * 37) % 900 for i in range(25)]
    s2 = shard_videos(ragged, 4)
    loads = [sum(ragged[i] for i in s) for s in s2]
    assert max(loads) - min(loads) <= max(ragged)
    assert s2 == shard_videos(ragged, 4)
    assert shard_videos([3, 1], 4) == [[0], [1], [], []]
