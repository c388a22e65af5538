"""Counterpart of the reference's scripts/train_av_model.py:11-96 on the MI355X.

``train_on_dataset`` is the reference's loop step for step (DataLoader(batch_size=8, shuffle=True,
collate_fn=lambda x: x[0]) — i.e. one of every eight videos per step, SURVEY Q12 —, one shot boundary
(0, num_shots), fps 30, AdamW lr 1e-4, MSE against the single broadcast target, Dropout active).  The model's
forward/backward run through libavsum_hip.so; the loss, the optimiser and the data loader are the caller's
torch, exactly as in the reference.  ``train()`` reads the TVSum HDF5 annotations like the reference and needs
h5py + the dataset on disk; ``train_synthetic`` is the BASELINE config-5 harness (synthetic labels ~U[1,5]).
With torch.distributed initialised (one process per GPU) the gradients are all-reduced before each step.
"""
import torch
import torch.nn.functional as F
from torch.utils.data import DataLoader

from .. import dist as avd, ops
from ..models.av_model import AVBiLSTMModel
from ..utils.alignments import align_shots_to_annotations


def train_step(model, optimizer, features, frame_scores, device="cuda"):
    """Lines 72-96 of the reference for one (features, frame_scores) item.  Returns the loss value."""
    num_shots = features["visual"].shape[0]
    shot_scores = align_shots_to_annotations(shot_boundaries=[(0, num_shots)], annotations=frame_scores.numpy(),
                                             fps=30)
    visual = features["visual"].unsqueeze(0).to(device)
    audio = features["audio"].unsqueeze(0).to(device)
    preds = model(visual, audio)
    loss = F.mse_loss(preds, shot_scores.to(device).float())
    optimizer.zero_grad()
    loss.backward()
    avd.allreduce_gradients(model)
    optimizer.step()
    return float(loss.item())


def train_on_dataset(dataset, epochs=100, lr=1e-4, model=None, on_step=None, device="cuda"):
    """One process: the reference's loop exactly (its DataLoader, its global-RNG shuffle, item 0 of every batch of 8).
    With torch.distributed initialised (one process per GPU) it is data-parallel over that loop: rank 0's weights are
    broadcast first (C1), every rank draws the SAME shuffled batches of 8 (the shuffle seed comes from rank 0) and
    takes item `rank` of each (rank 0 the item the reference would take), gradients are averaged (C3) before every
    AdamW step - so the replicas and their optimiser states stay identical."""
    import torch.distributed as tdist
    model = (model or AVBiLSTMModel()).to(device)
    rank, world, gen = 0, 1, None
    if tdist.is_initialized() and tdist.get_world_size() > 1:
        rank, world = tdist.get_rank(), tdist.get_world_size()
        avd.broadcast_module(model, 0)
        seed = torch.randint(0, 2 ** 31 - 1, (1,), dtype=torch.int64).to(device)
        tdist.broadcast(seed, 0)
        gen = torch.Generator().manual_seed(int(seed.item()))
    loader = DataLoader(dataset, batch_size=8, shuffle=True, generator=gen,
                        collate_fn=lambda items: items[rank % len(items)])
    optimizer = torch.optim.AdamW(model.parameters(), lr=lr)
    for _ in range(epochs):
        model.train()
        for features, frame_scores in loader:
            loss = train_step(model, optimizer, features, frame_scores, device)
            if on_step is not None:
                on_step(loss)
        # the recurrences run split over four CUs (ops.lstm*): no bounded wait may have run out during the epoch
        dev = next(model.parameters()).device
        if dev.type == "cuda":
            bad = ops.lstm_split_errors(dev)
            if bad:
                raise RuntimeError(f"split LSTM recurrence: {bad} workgroup(s) gave up waiting for a partner's step vector")
    return model


class SyntheticShotDataset(torch.utils.data.Dataset):
    """TVSumDataset-shaped items: ({"visual": [S,4096], "audio": [S,296]}, frame scores [n_frames] ~ U[1,5])."""

    def __init__(self, num_videos=8, shots=(20, 60), seed=5005, visual_dim=4096, audio_dim=296):
        g = torch.Generator().manual_seed(seed)
        self.items = []
        for _ in range(num_videos):
            s = int(torch.randint(shots[0], shots[1] + 1, (1,), generator=g))
            feats = {"visual": torch.randn(s, visual_dim, generator=g), "audio": torch.zeros(s, audio_dim)}
            self.items.append((feats, torch.rand(s * 30, generator=g) * 4 + 1))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def train_synthetic(steps=20, seed=7, **dataset_kw):
    """BASELINE config 5 on synthetic labels: returns the list of per-step losses."""
    torch.manual_seed(seed)
    losses = []
    ds = SyntheticShotDataset(**dataset_kw)
    epochs = max(1, -(-steps * 8 // len(ds)) // 1)
    train_on_dataset(ds, epochs=epochs, on_step=losses.append)
    return losses[:steps]


def train():
    """The reference's entry point: TVSum .mat (HDF5) -> DataFrame -> TVSumDataset -> the loop above."""
    try:
        import h5py
    except ImportError as e:
        raise RuntimeError("train() reads ydata-tvsum50.mat with h5py, which is not installed here; "
                           "use train_on_dataset(dataset) or train_synthetic()") from e
    import pandas as pd
    from ..data.dataset import TVSumDataset
    with h5py.File("Evaluation/TVSum/ydata-tvsum50-matlab/matlab/ydata-tvsum50.mat", "r") as f:
        titles_ref = f["tvsum50/title"][:]
        videos_ref = f["tvsum50/video"][:]
        titles = ["".join(chr(c) for c in f[ref][:].flatten()) for ref in titles_ref.squeeze()]
        videos = ["".join(chr(c) for c in f[ref][:].flatten()) for ref in videos_ref.squeeze()]
        user_anno = f["tvsum50/user_anno"][:]
        rows = []
        for vid_idx in range(50):
            user_annotations = f[user_anno[vid_idx, 0]][:]
            for user_idx in range(20):
                rows.append({"Video Title": titles[vid_idx], "Video File Name": videos[vid_idx],
                             "User ID": user_idx + 1, "Annotations": user_annotations[user_idx].flatten()})
    return train_on_dataset(TVSumDataset(pd.DataFrame(rows), "data/processed"))
