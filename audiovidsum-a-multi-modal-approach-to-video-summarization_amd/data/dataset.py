"""Host-side readers of the on-disk feature format the extractor stage writes
(data/processed/<vid>/{visual,audio}.npy float32 [S,4096] / [S,296]; reference data/dataset.py:8-62,
scripts/preprocess.py:74-81).  Pure I/O — SURVEY §8 row F1; no GPU work here."""
import os

import numpy as np
import torch


class BaseDataset(torch.utils.data.Dataset):
    def __init__(self, feature_dir, annotation_path=None):
        self.feature_dir = feature_dir
        self.video_ids = sorted(os.listdir(feature_dir))
        self.annotations = None

    def __len__(self):
        return len(self.video_ids)

    def _features(self, vid):
        return {
            "visual": torch.from_numpy(np.load(os.path.join(self.feature_dir, vid, "visual.npy"))),
            "audio": torch.from_numpy(np.load(os.path.join(self.feature_dir, vid, "audio.npy"))),
        }

    def __getitem__(self, idx):
        vid = self.video_ids[idx]
        scores = torch.from_numpy(np.load(os.path.join(self.feature_dir, vid, "scores.npy")))
        return self._features(vid), scores


class TVSumDataset(BaseDataset):
    """mat_annotations_df: DataFrame with columns "Video File Name" and "Annotations" (one row per user)."""

    def __init__(self, mat_annotations_df, feature_dir):
        self.annotations_df = mat_annotations_df
        self.video_ids = self.annotations_df["Video File Name"].unique()
        self.feature_dir = feature_dir

    def __getitem__(self, idx):
        vid = self.video_ids[idx]
        annos = self.annotations_df[self.annotations_df["Video File Name"] == vid]["Annotations"]
        avg = np.mean([a for a in annos], axis=0)
        return self._features(vid), torch.tensor(avg).float()


class SumMeDataset(BaseDataset):
    def _process_mat(self, mat_path):
        from scipy.io import loadmat
        return loadmat(mat_path)["gt_score"].squeeze()


def save_features(output_dir, video_name, visual, audio):
    """The writer side of the format (scripts/preprocess.py:66-81): validates 4096/296 and saves float32."""
    if visual.shape[1] != 4096:
        raise ValueError(f"Invalid visual shape: {visual.shape}")
    if audio.shape[1] != 296:
        raise ValueError(f"Invalid audio shape: {audio.shape}")
    d = os.path.join(output_dir, video_name)
    os.makedirs(d, exist_ok=True)
    np.save(os.path.join(d, "visual.npy"), visual.astype(np.float32))
    np.save(os.path.join(d, "audio.npy"), audio.astype(np.float32))
