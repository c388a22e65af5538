"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the shot segmentation the reference delegates to
PySceneDetect (features/extractors.py:388-393: ``detect(video_path, ContentDetector())``).

Third-party, absent from /root/reference and unpinned (scenedetect; OpenCV for the colour conversion):
restated from the published algorithms [3P-memory] => PARITY UNPINNED; pinned by analytic known answers in
tests/test_oracle_analytic.py (primary colours -> H 0/60/120, grey -> S = H = 0, cuts of a synthetic video).
"""
import numpy as np


def bgr2hsv_u8(img):
    """cv2.cvtColor(img, cv2.COLOR_BGR2HSV) for uint8: H in [0,180), 12-bit fixed-point division tables."""
    b, g, r = [img[..., i].astype(np.int64) for i in range(3)]
    v = np.maximum(b, np.maximum(g, r))
    vmin = np.minimum(b, np.minimum(g, r))
    diff = v - vmin
    sdiv = np.zeros(256, dtype=np.int64)
    hdiv = np.zeros(256, dtype=np.int64)
    idx = np.arange(1, 256)
    sdiv[1:] = np.rint((255 << 12) / (1.0 * idx)).astype(np.int64)
    hdiv[1:] = np.rint((180 << 12) / (6.0 * idx)).astype(np.int64)
    s = (diff * sdiv[v] + (1 << 11)) >> 12
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * hdiv[diff] + (1 << 11)) >> 12
    h = np.where(h < 0, h + 180, h)
    return np.stack([h, s, v], -1).astype(np.uint8)


def content_scores(frames, step=1):
    scores = np.zeros(len(frames))
    prev = None
    for i, f in enumerate(frames):
        hsv = bgr2hsv_u8(f[::step, ::step]).astype(np.int32)
        if prev is not None:
            n = float(hsv.shape[0] * hsv.shape[1])
            scores[i] = sum(np.sum(np.abs(hsv[..., c] - prev[..., c])) / n for c in range(3)) / 3.0
        prev = hsv
    return scores


def detect_shots(frames, threshold=27.0, min_scene_len=15, step=1):
    scores = content_scores(frames, step)
    cuts, last = [], 0
    for f in range(1, len(frames)):
        if scores[f] >= threshold and f - last >= min_scene_len:
            cuts.append(f)
            last = f
    if not cuts:
        return []
    b = [0] + cuts + [len(frames)]
    return [(b[i], b[i + 1]) for i in range(len(b) - 1)]
