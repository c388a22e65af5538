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
    assert all(p_._version > 0 for p_ in mod.parameters())   # weight caches keyed on _version see the broadcast
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


class _TinyScorer(torch.nn.Module):
    """CPU stand-in with AVBiLSTMModel's call signature (the HIP model itself has no CPU path): exercises the
    distributed logic of the training loop only."""

    def __init__(self):
        super().__init__()
        self.v = torch.nn.Linear(12, 6)
        self.a = torch.nn.Linear(5, 6)
        self.head = torch.nn.Linear(6, 1)

    def forward(self, visual, audio):
        return torch.sigmoid(self.head(torch.relu(self.v(visual) + self.a(audio)))).squeeze()


def _train_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from avsum_amd import dist as avd
    from avsum_amd.scripts import train_av_model as tr
    avd.init_from_env("gloo")
    torch.manual_seed(100 + rank)                 # replicas start DIFFERENT; the loop must make them agree
    model = _TinyScorer()
    ds = tr.SyntheticShotDataset(num_videos=16, shots=(4, 9), seed=5, visual_dim=12, audio_dim=5)
    seen = []
    real_step = tr.train_step

    def spy(model, optimizer, features, frame_scores, device="cuda"):
        seen.append(float(features["visual"].sum()))
        return real_step(model, optimizer, features, frame_scores, device)

    tr.train_step = spy
    try:
        tr.train_on_dataset(ds, epochs=3, lr=1e-2, model=model, device="cpu")
    finally:
        tr.train_step = real_step
    ret[rank] = ([p.detach().clone() for p in model.parameters()], seen)
    dist.destroy_process_group()


def test_gloo_world2_training_loop_keeps_replicas_identical():
    """weights broadcast before the first step, every rank a different video of the same shuffled batch,
    gradients averaged -> parameters bit-identical across ranks after the run."""
    port = 31500 + os.getpid() % 2000
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_train_worker, args=(2, port, ret), nprocs=2, join=True)
        (p0, seen0), (p1, seen1) = ret[0], ret[1]
    assert len(seen0) == len(seen1) == 6                       # 3 epochs x 2 batches of 8
    assert all(a != b for a, b in zip(seen0, seen1))           # different videos on the two ranks at every step
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)


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
