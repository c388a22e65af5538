#!/usr/bin/env python3
"""What a plain fp16-STORAGE mode (fp16 activations and weights, fp32 accumulation: BASELINE configs[4]'s "fp16 MFMA conv
path" read as a storage format) would do to the scores - measured, without building its kernels: the AVS_F16X2 trunk
with every lo half forced to zero computes exactly that arithmetic (make -C <pkg>/csrc fp16emu; the library is loaded
because this script sets AVS_STUDY_LIB=fp16emu before importing the package).  Same pipeline, inputs and bars as
tests/test_gpu_accuracy.py / bench.py's accuracy leg.   Usage: python tools/fp16_storage_study.py"""
import os
import sys
os.environ["AVS_STUDY_LIB"] = "fp16emu"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from avsum_amd import synthetic
from avsum_amd.evaluation.accuracy import accuracy_report
from avsum_amd.features.extractors import VisualFeatureExtractor
from avsum_amd.models.av_model import AVBiLSTMModel
from avsum_amd.pipeline import FrameScoringPipeline
from oracle import cnn as ocnn, scorer as osc

dev = torch.device("cuda", 0)
torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
for kind, fpg in (("uniform", 1), ("uniform", 4), ("scenes", 1)):
    torch.manual_seed(7)
    ext = VisualFeatureExtractor(torch.float32, "batch", f32_split="f16x2")
    scorer = AVBiLSTMModel().eval()
    with torch.no_grad():
        scorer.scorer[0].weight.mul_(6.0)
        scorer.scorer[2].weight.mul_(6.0)
    rsd = {k: v.clone() for k, v in ext.resnet.state_dict().items()}
    ssd = {k: v.clone() for k, v in scorer.state_dict().items()}
    lengths = [300, 300]
    offsets = synthetic.offsets_of(lengths)
    frames = (synthetic.make_frames_uniform(sum(lengths), dev, 1000) if kind == "uniform"
              else synthetic.make_frames_scenes(lengths, dev, 1000))
    host = frames.cpu().numpy()
    ref = []
    with torch.no_grad():
        for a, b in zip(offsets[:-1], offsets[1:]):
            feats = [ocnn.resnet50_trunk_forward(rsd, torch.cat([ocnn.preprocess_frame(f) for f in host[g:min(g + fpg, b)]]))
                     for g in range(a, b, fpg)]
            visual = torch.cat([torch.cat(feats), torch.zeros(b - a, 2048)], 1).unsqueeze(0)
            ref.append(osc.av_bilstm_forward(ssd, visual, torch.zeros(1, b - a, 296)).reshape(-1))
    ref = torch.cat(ref).numpy()
    pipe = FrameScoringPipeline(ext.to(dev), scorer.to(dev), use_inception=False, chunk_frames=256, frames_per_group=fpg)
    got = pipe.score(frames, offsets).cpu().numpy()
    rep = accuracy_report(got, ref, offsets)
    print(f"[{kind} fpg={fpg}] fp16 storage (emulated): score_max_abs_err {rep['score_max_abs_err']:.2e} (range "
          f"{rep['score_range']:.3f}), agreement outside the guard band {rep['agreement_outside_guard']:.4f}, guarded F1 "
          f"drift {rep['f1_drift_guarded_max']:.4f}, bars_met {rep['bars_met']}", flush=True)
