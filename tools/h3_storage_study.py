#!/usr/bin/env python3
"""What a 3-BYTE storage format for the wide block outputs of the ResNet-50 trunk (fp16 hi + an 8-bit lo instead of
fp16 hi | fp16 lo) would do to the scores - measured by emulation, before any kernel for it exists: the AVS_F16X2 trunk
with the chosen block outputs re-quantised in place (unpack -> hi = fp16(x), lo narrowed -> pack).  Two candidate lo
encodings: "e5m2" = the top byte of the fp16 lo half (round to nearest: sign, 5 exponent, 2 mantissa bits; decoding is
a byte permute), "i8" = the remainder as a signed 8-bit fraction of ulp(hi) / 256 (19-20 significant bits; decoding is
a convert + a scale from hi's exponent).  Same pipeline, inputs and bars as tests/test_gpu_accuracy.py / bench.py.
Layers 1-2 of the f16x2 trunk run at the HBM roofline of their dataflow: bytes per stored value are
the remaining lever there.   Usage: python tools/h3_storage_study.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from avsum_amd import ops, synthetic
from avsum_amd.evaluation.accuracy import accuracy_report
from avsum_amd.features.extractors import VisualFeatureExtractor
from avsum_amd.models.av_model import AVBiLSTMModel
from avsum_amd.pipeline import FrameScoringPipeline
from oracle import cnn as ocnn, scorer as osc

dev = torch.device("cuda", 0)
torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))


def requant(x, enc):
    v = ops.f16x2_unpack(x.contiguous())
    hi = v.half().float()
    r = v - hi
    if enc == "e5m2":
        lo = r.to(torch.float8_e5m2).float()
    elif enc == "i8":
        e = torch.floor(torch.log2(hi.abs().clamp_min(2.0 ** -14)))            # exponent of hi (denormals: 2^-14)
        step = torch.exp2(e - 10 - 8)                                            # ulp(hi) / 256
        lo = torch.round(r / step).clamp_(-128, 127) * step
    elif enc == "none":                                                          # plain fp16 storage of these tensors
        lo = torch.zeros_like(r)
    else:
        raise ValueError(enc)
    return ops.f16x2_pack((hi + lo).contiguous()).view_as(x)


SETS = {"l1-2 inner (blocks 0,1,3,4,5)": {0, 1, 3, 4, 5}, "layers 1-2 (blocks 0-6)": set(range(7)),
        "layers 1-3 (blocks 0-12)": set(range(13)), "all 16 blocks": set(range(16))}
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
    ext = ext.to(dev)
    pipe = FrameScoringPipeline(ext, scorer.to(dev), use_inception=False, chunk_frames=128, frames_per_group=fpg)
    runner = ext._resnet_runner
    cases = [("f16x2 as it is", None, None)] + [(f"{enc} lo, {name}", enc, blocks)
                                                 for enc in ("i8", "e5m2", "none") for name, blocks in SETS.items()]
    for label, enc, blocks in cases:
        runner.block_hook = None if enc is None else (lambda bi, x, enc=enc, blocks=blocks: requant(x, enc) if bi in blocks else x)
        got = pipe.score(frames, offsets).cpu().numpy()
        rep = accuracy_report(got, ref, offsets)
        print(f"[{kind} fpg={fpg}] {label}: score_max_abs_err {rep['score_max_abs_err']:.2e} (range "
              f"{rep['score_range']:.3f}), agreement outside the guard band {rep['agreement_outside_guard']:.4f}, guarded F1 "
              f"drift {rep['f1_drift_guarded_max']:.4f}, bars_met {rep['bars_met']}", flush=True)
    runner.block_hook = None
