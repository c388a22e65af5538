"""Counterpart of the reference's scripts/preprocess.py:32-85: walk a directory of .mp4
files, run AVProcessor.process_video, validate the 4096 / 296 widths, save data/processed/<vid>/{visual,audio}.npy
as float32, skip videos whose outputs already exist, delete a partial output directory on failure.
Unlike the reference (which runs preprocess_dataset() at import time, :88-89) nothing runs on import.
Decoding needs cv2 / pydub(ffmpeg) / scenedetect on the host; the feature arithmetic runs on the MI355X."""
import os
import shutil

import numpy as np

from ..data.dataset import save_features
from ..features.extractors import AVProcessor


def preprocess_dataset(input_dir="data/raw", output_dir="data/processed", processor=None):
    processor = processor or AVProcessor()
    os.makedirs(output_dir, exist_ok=True)
    done, failed = [], []
    for video_file in sorted(os.listdir(input_dir)):
        if not video_file.endswith(".mp4"):
            continue
        name = os.path.splitext(video_file)[0]
        vdir = os.path.join(output_dir, name)
        if all(os.path.exists(os.path.join(vdir, f)) for f in ("visual.npy", "audio.npy")):
            continue  # preprocess.py:46-55: resume by skipping finished videos
        try:
            visual, audio = processor.process_video(os.path.join(input_dir, video_file))
            save_features(output_dir, name, np.asarray(visual), np.asarray(audio))
            done.append(name)
        except Exception as e:  # preprocess.py:83-85: catch-all, remove the partial directory
            print(f"Failed processing {video_file}: {e}")
            shutil.rmtree(vdir, ignore_errors=True)
            failed.append(name)
    return done, failed


if __name__ == "__main__":
    preprocess_dataset()
