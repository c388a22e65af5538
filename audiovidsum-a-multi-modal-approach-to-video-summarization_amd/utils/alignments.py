"""Host helper mirroring the reference's utils/alignments.py:4-22 (shot -> mean of 2-second annotation bins)."""
import torch


def align_shots_to_annotations(shot_boundaries, annotations, fps):
    shot_scores = []
    for start, end in shot_boundaries:
        start_idx = int((start / fps) // 2)
        end_idx = int((end / fps) // 2) + 1
        shot_scores.append(annotations[start_idx:end_idx].mean())
    return torch.tensor(shot_scores)
