"""Accuracy figures of a scored batch against reference scores: the three numbers north_star's bars are stated in
(importance scores within 1e-4, selected frame indices, frame-selection F1 within 0.001).

Host numpy, like the selection rule itself (scripts/evaluate.py:26-32, evaluation/metrics.py:1-9 of the
reference).  The reference holds no ground-truth summaries for synthetic videos, so the F1 of a selection is taken
against a seeded synthetic segment list (``synthetic_gt_segments``) and what is reported is the DRIFT of that F1
between the two score vectors.
"""
import numpy as np

from .metrics import compute_temporal_f1, segments_from_indices, select_frames


def synthetic_gt_segments(n, seed):
    """Seeded synthetic ground truth for an n-frame video: four segments of n // 25 frames (about 15 % of it)."""
    rng = np.random.default_rng(seed)
    width = max(1, n // 25)
    starts = np.sort(rng.choice(max(1, n - width), min(4, max(1, n - width)), replace=False))
    segs, last = [], 0
    for s in starts:
        s = max(int(s), last)
        e = min(n, s + width)
        if e > s:
            segs.append((s, e))
        last = e
    return segs


def _f1(scores, gt, n):
    sel = select_frames(scores)
    if sel.size == 0:
        return 0.0
    return float(compute_temporal_f1(segments_from_indices(sel), gt, n))


def accuracy_report(scores, ref_scores, video_offsets, gt_seed=900):
    """scores, ref_scores: float arrays [N] of concatenated videos (offsets [V+1]).  Returns a dict:
    score_max_abs_err, score_range (of the reference), selection_agreement (share of frames on the same side of
    their video's mean-threshold), selected / selected_ref counts, f1_drift_max / f1_drift_mean over the videos."""
    scores = np.asarray(scores, dtype=np.float32)
    ref = np.asarray(ref_scores, dtype=np.float32)
    same, drift, nsel, nsel_ref = 0, [], 0, 0
    for v, (a, b) in enumerate(zip(video_offsets[:-1], video_offsets[1:])):
        s, r = scores[a:b], ref[a:b]
        ms, mr = s > np.mean(s), r > np.mean(r)
        same += int((ms == mr).sum())
        nsel += int(ms.sum())
        nsel_ref += int(mr.sum())
        gt = synthetic_gt_segments(b - a, gt_seed + v)
        drift.append(abs(_f1(s, gt, b - a) - _f1(r, gt, b - a)))
    n = int(video_offsets[-1] - video_offsets[0])
    return {"score_max_abs_err": float(np.abs(scores - ref).max()) if n else 0.0,
            "score_range": float(ref.max() - ref.min()) if n else 0.0,
            "selection_agreement": same / max(n, 1),
            "selected": nsel, "selected_ref": nsel_ref,
            "f1_drift_max": float(max(drift)) if drift else 0.0,
            "f1_drift_mean": float(np.mean(drift)) if drift else 0.0,
            "videos": len(drift), "frames": n}
