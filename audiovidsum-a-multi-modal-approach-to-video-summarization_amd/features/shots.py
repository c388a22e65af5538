"""Shot segmentation on the MI355X: the front end that features/extractors.py:388-393 delegates
to ``scenedetect.detect(video_path, ContentDetector())``.

PySceneDetect is a third-party dependency absent here and unpinned; restated from its published algorithm
[3P-memory, v0.6.x defaults: threshold 27.0, min_scene_len 15, weights hue/sat/lum = 1/1/1, edges 0,
auto-downscale to ~256 px wide by striding]: a frame's content score is the mean of the three mean absolute
differences of OpenCV's 8-bit HSV against the previous frame; a cut is placed where the score reaches the
threshold and at least min_scene_len frames passed since the last cut; scenes are the intervals between cuts
(empty when there is no cut).  The per-frame sums come from the GPU scan; the thresholding is a host loop over
N numbers.
"""
import numpy as np
import torch

from .. import ops

DEFAULT_MIN_WIDTH = 256  # scenedetect.scene_manager.DEFAULT_MIN_WIDTH


def downscale_factor(frame_width, effective_width=DEFAULT_MIN_WIDTH):
    if frame_width < effective_width:
        return 1
    return int(frame_width / float(effective_width))


def content_scores(frames_u8, step=None):
    """frames uint8 [n,h,w,3] (device) -> float64 [n] content scores (score[0] = 0)."""
    n, h, w, _ = frames_u8.shape
    step = downscale_factor(w) if step is None else step
    sums = ops.hsv_frame_diff(frames_u8, step).cpu().numpy().astype(np.float64)
    pixels = float(len(range(0, h, step)) * len(range(0, w, step)))
    return (sums / pixels).sum(axis=1) / 3.0


def cuts_from_scores(scores, threshold=27.0, min_scene_len=15):
    cuts, last = [], 0
    for f in range(1, len(scores)):
        if scores[f] >= threshold and f - last >= min_scene_len:
            cuts.append(f)
            last = f
    return cuts


def detect_shots(frames_u8, threshold=27.0, min_scene_len=15):
    """[(start_frame, end_frame)] like scenedetect.detect(...): empty when no cut was found."""
    if not torch.is_tensor(frames_u8):
        frames_u8 = torch.from_numpy(np.ascontiguousarray(frames_u8))
    if not frames_u8.is_cuda:
        frames_u8 = frames_u8.cuda()
    n = frames_u8.shape[0]
    cuts = cuts_from_scores(content_scores(frames_u8.contiguous()), threshold, min_scene_len)
    if not cuts:
        return []
    bounds = [0] + cuts + [n]
    return [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1)]
