"""Host-side selection rule and metrics — integer/boolean work kept on the
host so the selected indices are bit-exact with numpy: scripts/evaluate.py:21-42 and
evaluation/metrics.py:1-9 of the reference."""
import numpy as np


def compute_temporal_f1(pred_shots, gt_shots, total_frames):
    overlap = sum(max(0, min(p_end, g_end) - max(p_start, g_start))
                  for p_start, p_end in pred_shots for g_start, g_end in gt_shots)
    precision = overlap / sum(p_end - p_start for p_start, p_end in pred_shots)
    recall = overlap / sum(g_end - g_start for g_start, g_end in gt_shots)
    return 2 * (precision * recall) / (precision + recall + 1e-8)


def select_frames(pred):
    """np.flatnonzero(pred > pred.mean()) — the reference's selection rule (scripts/evaluate.py:26)."""
    pred = np.asarray(pred)
    return np.flatnonzero(pred > np.mean(pred))


def segments_from_indices(idx):
    idx = np.asarray(idx)
    if idx.size == 0:
        return []
    cut = np.flatnonzero(np.diff(idx) != 1)
    starts = np.concatenate([[idx[0]], idx[cut + 1]])
    ends = np.concatenate([idx[cut] + 1, [idx[-1] + 1]])
    return [(int(s), int(e)) for s, e in zip(starts, ends)]


def binary_f1(pred, target):
    bp = (pred > np.mean(pred)).astype(int)
    bt = (target > np.mean(target)).astype(int)
    tp = np.logical_and(bp, bt).sum()
    precision = tp / bp.sum()
    recall = tp / bt.sum()
    return 2 * (precision * recall) / (precision + recall + 1e-8)


def summarize_scores(pairs):
    """Mean over videos of the mean-threshold F1, Spearman and Kendall correlations (scripts/evaluate.py:21-42).
    Like the reference it does not guard a constant prediction (NaN precision / correlation)."""
    from scipy.stats import kendalltau, spearmanr
    f1 = [binary_f1(p, t) for p, t in pairs]
    rho = [spearmanr(p, t).correlation for p, t in pairs]
    tau = [kendalltau(p, t).correlation for p, t in pairs]
    return {"f1": np.mean(f1), "spearman": np.mean(rho), "kendall": np.mean(tau)}
