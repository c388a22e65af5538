"""Mirror of the reference's utils/shot_metrics.py:4-16 (same formula as evaluation/metrics.py)."""
from ..evaluation.metrics import compute_temporal_f1


def calculate_overlap(pred_segments, gt_segments):
    return sum(max(0, min(pe, ge) - max(ps, gs)) for ps, pe in pred_segments for gs, ge in gt_segments)


def compute_f1(pred_segments, gt_segments, video_length):
    return compute_temporal_f1(pred_segments, gt_segments, video_length)
