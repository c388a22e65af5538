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


def _f1_of_mask(mask, gt, n):
    sel = np.flatnonzero(mask)
    if sel.size == 0:
        return 0.0
    return float(compute_temporal_f1(segments_from_indices(sel), gt, n))


def _f1(scores, gt, n):
    return _f1_of_mask(np.asarray(scores) > np.mean(scores), gt, n)


SCORE_TOL = 1e-4      # north_star: importance scores within 1e-4 (fp32) of the reference CPU path
F1_TOL = 1e-3         # north_star: frame-selection F1 within 0.001 of the reference


def accuracy_report(scores, ref_scores, video_offsets, gt_seed=900, guard=2.0 * SCORE_TOL):
    """scores, ref_scores: float arrays [N] of concatenated videos (offsets [V+1]).  Returns a dict:
    score_max_abs_err, score_range (of the reference), selection_agreement (share of frames on the same side of
    their video's mean-threshold), selected / selected_ref counts, f1_drift_max / f1_drift_mean over the videos.

    Guard band: two score vectors that agree to `tol` may put a frame whose reference score lies within
    2 * tol of its video's mean on either side of the threshold, so such frames say nothing about the arithmetic.
    With guard > 0 the frames with |ref - mean(ref)| < guard are counted (`guarded_frames`) and taken out of the
    decision figures: `agreement_outside_guard`, `indices_identical_outside_guard`, and `f1_drift_guarded_*` (the F1
    of the selection in which the guarded frames take the reference's decision).  The unguarded figures stay beside
    them.  `bars_met`: score error <= SCORE_TOL, indices identical outside the guard band, guarded F1 drift <= F1_TOL
    - ONE definition, shared by bench.py and tests/test_gpu_accuracy.py."""
    scores = np.asarray(scores, dtype=np.float32)
    ref = np.asarray(ref_scores, dtype=np.float32)
    same, drift, nsel, nsel_ref = 0, [], 0, 0
    guarded, same_safe, drift_g = 0, 0, []
    for v, (a, b) in enumerate(zip(video_offsets[:-1], video_offsets[1:])):
        s, r = scores[a:b], ref[a:b]
        ms, mr = s > np.mean(s), r > np.mean(r)
        same += int((ms == mr).sum())
        nsel += int(ms.sum())
        nsel_ref += int(mr.sum())
        gt = synthetic_gt_segments(b - a, gt_seed + v)
        f1_ref = _f1_of_mask(mr, gt, b - a)
        drift.append(abs(_f1_of_mask(ms, gt, b - a) - f1_ref))
        safe = np.abs(r - np.mean(r)) >= guard
        guarded += int((~safe).sum())
        same_safe += int((ms == mr)[safe].sum())
        drift_g.append(abs(_f1_of_mask(np.where(safe, ms, mr), gt, b - a) - f1_ref))
    n = int(video_offsets[-1] - video_offsets[0])
    err = float(np.abs(scores - ref).max()) if n else 0.0
    rep = {"score_max_abs_err": err,
           "score_range": float(ref.max() - ref.min()) if n else 0.0,
           "selection_agreement": same / max(n, 1),
           "selected": nsel, "selected_ref": nsel_ref,
           "f1_drift_max": float(max(drift)) if drift else 0.0,
           "f1_drift_mean": float(np.mean(drift)) if drift else 0.0,
           "guard_band": float(guard), "guarded_frames": guarded,
           "agreement_outside_guard": same_safe / max(n - guarded, 1),
           "indices_identical_outside_guard": bool(same_safe == n - guarded),
           "f1_drift_guarded_max": float(max(drift_g)) if drift_g else 0.0,
           "f1_drift_guarded_mean": float(np.mean(drift_g)) if drift_g else 0.0,
           "videos": len(drift), "frames": n}
    rep["bars"] = {"score_abs": SCORE_TOL, "f1_drift": F1_TOL, "guard_band": float(guard)}
    rep["bars_met"] = bool(err <= SCORE_TOL and rep["indices_identical_outside_guard"]
                           and rep["f1_drift_guarded_max"] <= F1_TOL)
    return rep
