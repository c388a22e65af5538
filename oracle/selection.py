"""ORACLE — TEST INFRASTRUCTURE ONLY.  Frame selection and metrics.

  * select_frames        scripts/evaluate.py:26        (pred > mean(pred))
  * f1_binary            scripts/evaluate.py:26-33
  * evaluate_metrics     scripts/evaluate.py:21-42     (the per-video loop after inference)
  * compute_temporal_f1  evaluation/metrics.py:1-9
  * align_shots          utils/alignments.py:4-22
Pinned against the reference functions (importable here) in tests/test_oracle_pins.py.
"""
import numpy as np
import torch
from scipy.stats import kendalltau, spearmanr


def select_frames(pred):
    pred = np.asarray(pred)
    return np.flatnonzero(pred > np.mean(pred))


def f1_binary(pred, target):
    bp = (pred > np.mean(pred)).astype(int)
    bt = (target > np.mean(target)).astype(int)
    tp = np.logical_and(bp, bt).sum()
    precision = tp / bp.sum()
    recall = tp / bt.sum()
    return 2 * (precision * recall) / (precision + recall + 1e-8)


def evaluate_metrics(all_preds, all_targets):
    f1s, sps, kts = [], [], []
    for pred, target in zip(all_preds, all_targets):
        f1s.append(f1_binary(pred, target))
        sps.append(spearmanr(pred, target).correlation)
        kts.append(kendalltau(pred, target).correlation)
    return {"f1": np.mean(f1s), "spearman": np.mean(sps), "kendall": np.mean(kts)}


def compute_temporal_f1(pred_shots, gt_shots, total_frames=None):
    overlap = sum(max(0, min(pe, ge) - max(ps, gs)) for ps, pe in pred_shots for gs, ge in gt_shots)
    precision = overlap / sum(pe - ps for ps, pe in pred_shots)
    recall = overlap / sum(ge - gs for gs, ge in gt_shots)
    return 2 * (precision * recall) / (precision + recall + 1e-8)


def align_shots(shot_boundaries, annotations, fps):
    out = []
    for start, end in shot_boundaries:
        s = int((start / fps) // 2)
        e = int((end / fps) // 2) + 1
        out.append(annotations[s:e].mean())
    return torch.tensor(out)


def segments_from_indices(idx):
    """Runs of consecutive selected indices as half-open (start, end) — conversion defined by this
    build (the reference has none)."""
    idx = np.asarray(idx)
    if idx.size == 0:
        return []
    breaks = np.flatnonzero(np.diff(idx) != 1)
    starts = np.concatenate([[idx[0]], idx[breaks + 1]])
    ends = np.concatenate([idx[breaks] + 1, [idx[-1] + 1]])
    return [(int(s), int(e)) for s, e in zip(starts, ends)]
