"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of features/fusion.py.

  * compute_dtw            features/fusion.py:7-12  (scipy cdist 'euclidean', float64 out;
                           SciPy IS installed here, so this one is pinned by direct call)
  * compute_optimal_path   features/fusion.py:15-18 — the literal call raises TypeError
                           (fastdtw needs two series).  Restated as its intent:
                           exact DTW over the cost matrix with fastdtw's documented recursion and
                           tie order (i-1,j), (i,j-1), (i-1,j-1) (third-party fastdtw, absent and
                           unpinned => PARITY UNPINNED; pinned by brute-force path enumeration on
                           small matrices in tests/test_oracle_pins.py::test_dtw_path_is_optimal_bruteforce)
  * interpolate_features   features/fusion.py:21-32 (pinned against the reference function itself,
                           imported with a stub `fastdtw` module, in tests/test_oracle_pins.py)
  * align_features         features/extractors.py:248-290 (AudioFeatureExtractor._align_features) with the
                           fastdtw call (which raises and falls back to zeros in the reference)
                           restated as its intent: exact DTW on the Euclidean cost matrix; PARITY UNPINNED
"""
import numpy as np
import torch
from scipy.spatial.distance import cdist


def compute_dtw(visual, audio):
    return cdist(visual.numpy(), audio.numpy(), metric="euclidean")


def dtw_path(cost):
    """Exact DTW on a cost matrix.  Returns (total_cost, path int64 [L,2]) from (0,0) to (n-1,m-1)."""
    cost = np.asarray(cost, dtype=np.float64)
    n, m = cost.shape
    acc = np.full((n, m), np.inf)
    step = np.zeros((n, m), dtype=np.int8)
    for i in range(n):
        for j in range(m):
            if i == 0 and j == 0:
                acc[0, 0] = cost[0, 0]
                continue
            up = acc[i - 1, j] if i > 0 else np.inf
            left = acc[i, j - 1] if j > 0 else np.inf
            diag = acc[i - 1, j - 1] if (i > 0 and j > 0) else np.inf
            best, code = up, 0
            if left < best:
                best, code = left, 1
            if diag < best:
                best, code = diag, 2
            acc[i, j] = cost[i, j] + best
            step[i, j] = code
    i, j = n - 1, m - 1
    path = [(i, j)]
    while i > 0 or j > 0:
        c = step[i, j]
        if c == 0:
            i -= 1
        elif c == 1:
            j -= 1
        else:
            i -= 1
            j -= 1
        path.append((i, j))
    path.reverse()
    return float(acc[n - 1, m - 1]), np.array(path, dtype=np.int64)


def compute_optimal_path(dtw_matrix):
    return dtw_path(dtw_matrix)[1]


def interpolate_features(features, path, target_length):
    aligned_indices = path[:, 0]
    unique_indices, counts = np.unique(aligned_indices, return_counts=True)
    weights = counts / counts.sum()
    aligned = [features[idx] * weight for idx, weight in zip(unique_indices, weights)]
    return torch.stack(aligned)[:target_length]


def align_features(mfcc, mel, vggish):
    """features/extractors.py:248-290: atleast_2d, empty -> zeros(128); truncate the three streams to the common
    feature dimension and the common length; warp mfcc and mel onto vggish: rows feat[p[1]] along the path."""
    vggish, mfcc, mel = np.atleast_2d(vggish), np.atleast_2d(mfcc), np.atleast_2d(mel)
    if vggish.size == 0 or mfcc.size == 0 or mel.size == 0:
        return np.zeros(128), np.zeros(128)
    dim = min(vggish.shape[1], mfcc.shape[1], mel.shape[1])
    vggish, mfcc, mel = vggish[:, :dim], mfcc[:, :dim], mel[:, :dim]
    length = min(vggish.shape[0], mfcc.shape[0], mel.shape[0])
    if length == 0:
        return np.zeros(dim), np.zeros(dim)
    vggish, mfcc, mel = vggish[:length], mfcc[:length], mel[:length]
    out = []
    for feat in (mfcc, mel):
        cost = cdist(np.asarray(vggish, dtype=np.float32), np.asarray(feat, dtype=np.float32), metric="euclidean")
        path = dtw_path(cost)[1]
        out.append(np.array([feat[j] for _, j in path]))
    return out[0], out[1]
