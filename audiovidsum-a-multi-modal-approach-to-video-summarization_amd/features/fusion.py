"""Drop-in for the reference's features/fusion.py on the MI355X.

  compute_dtw(visual, audio)            fusion.py:7-12   Tensor[Tv,D], Tensor[Ta,D] -> np.float64 [Tv,Ta]
  compute_optimal_path(dtw_matrix)      fusion.py:15-18  -> np.int64 [L,2]
  interpolate_features(feats, path, n)  fusion.py:21-32  -> Tensor [min(U,n), D] float32

``compute_optimal_path`` cannot run as written in the reference (fastdtw is called without its
second series); this build implements the evident intent — the exact DTW path over
the cost matrix with fastdtw's tie order — and says so.  Inputs are host tensors/arrays as in
the reference; they are moved to the HIP device, computed there, and returned on the host.
"""
import numpy as np
import torch

from .. import ops


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError("avsum_amd needs an MI355X (HIP device); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def compute_dtw_device(visual, audio):
    """Device tensors in, device float64 [Tv,Ta] out."""
    return ops.cdist(visual.float(), audio.float())


def compute_dtw(visual, audio):
    """Compute DTW cost matrix (Euclidean, float64)."""
    dev = _dev()
    v = torch.as_tensor(visual)
    a = torch.as_tensor(audio)
    if v.dim() != 2 or a.dim() != 2:
        raise ValueError("XA must be a 2-dimensional array.")
    return compute_dtw_device(v.to(dev), a.to(dev)).cpu().numpy()


def compute_optimal_path(dtw_matrix):
    """Optimal warping path through the cost matrix, start -> end, as int64 [L,2]."""
    dev = _dev()
    cost = torch.as_tensor(np.ascontiguousarray(dtw_matrix, dtype=np.float64)).to(dev)
    path, plen, _ = ops.dtw_path(cost)
    n = int(plen.item())
    return path[:n].cpu().numpy()


def interpolate_features(features, path, target_length):
    """features[idx] * (count/sum(count)) for the unique first-column indices of path, first target_length rows."""
    dev = _dev()
    aligned_indices = np.asarray(path)[:, 0]
    unique_indices, counts = np.unique(aligned_indices, return_counts=True)  # host: a few thousand ints
    weights = counts / counts.sum()
    feats = torch.as_tensor(features).float()
    out = ops.gather_scale(feats.to(dev).contiguous(), torch.from_numpy(unique_indices.astype(np.int64)).to(dev),
                           torch.from_numpy(weights.astype(np.float64)).to(dev))
    return out[:target_length].cpu()
