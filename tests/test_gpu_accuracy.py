"""Accuracy gate of the benchmarked pipeline: the exact bench path - uint8 frames -> ResNet-50
(batch-statistics BatchNorm per micro-batch group) -> AVBiLSTM scorer -> mean-threshold selection - against the
fp32 ORACLE (oracle/, CPU), in both arithmetic modes:

  fp32 (parity mode)      north_star's bars hold: scores within 1e-4, selected indices identical outside a 2e-6
                          guard band, frame-selection F1 within 0.001;
  bf16 (throughput mode)  what is TRUE is asserted and the numbers are printed: with 8 significant bits per stored
                          activation through 53 batch-normalised layers the features are ~5 % (per-frame groups) /
                          ~11 % (4-frame groups) from fp32 in relative L2, and on these inputs - random-weight
                          scorer, score range ~0.05, every frame statistically alike - that moves the scores by up
                          to ~0.015 and flips about a quarter of the mean-threshold decisions.  The 0.001 F1 bar is
                          NOT met in bf16 (measured drift 0.01-0.09); it needs fp32-class arithmetic here: a 0.1 %
                          relative perturbation of the fp32 features already costs 0.001-0.008 of F1
                          (tools/accuracy_study.py).  DESIGN.md section 4 records the measured table.
The bf16 path is deterministic (no float atomics), so these figures are reproducible run to run.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

VIDEOS, FRAMES = 2, 300


def _setup(seed):
    from avsum_amd.features.extractors import VisualFeatureExtractor
    from avsum_amd.models.av_model import AVBiLSTMModel
    torch.manual_seed(seed)
    ext32 = VisualFeatureExtractor(torch.float32, "batch")
    scorer = AVBiLSTMModel().eval()
    with torch.no_grad():   # spread scorer weights as in tests/test_gpu_models.py::_seeded_scorer
        scorer.scorer[0].weight.mul_(6.0)
        scorer.scorer[2].weight.mul_(6.0)
    ext16 = VisualFeatureExtractor(torch.bfloat16, "batch")
    ext16.load_state_dict(ext32.state_dict())
    rsd = {k: v.clone() for k, v in ext32.resnet.state_dict().items()}
    ssd = {k: v.clone() for k, v in scorer.state_dict().items()}
    return ext32, ext16, scorer, rsd, ssd


def _oracle_scores(rsd, ssd, frames, offsets, fpg):
    """The reference's computation on the CPU: micro-batches of fpg frames inside each video through the train-mode
    trunk (features/extractors.py:48-65), zeros for the Inception half (visual-only config) and the audio vector
    (SURVEY Q5), one AVBiLSTM call per video (scripts/evaluate.py:13-15)."""
    from oracle import cnn as ocnn, scorer as osc
    out = []
    with torch.no_grad():
        for a, b in zip(offsets[:-1], offsets[1:]):
            feats = []
            for g in range(a, b, fpg):
                x = torch.cat([ocnn.preprocess_frame(f) for f in frames[g:min(g + fpg, b)]])
                feats.append(ocnn.resnet50_trunk_forward(rsd, x))
            visual = torch.cat([torch.cat(feats), torch.zeros(b - a, 2048)], 1).unsqueeze(0)
            out.append(osc.av_bilstm_forward(ssd, visual, torch.zeros(1, b - a, 296)).reshape(-1))
    return torch.cat(out).numpy()


@pytest.mark.parametrize("kind,fpg", [("uniform", 1), ("uniform", 4), ("scenes", 1)])
def test_bench_pipeline_against_oracle(dev, kind, fpg):
    from avsum_amd import synthetic
    from avsum_amd.evaluation.accuracy import accuracy_report
    from avsum_amd.pipeline import FrameScoringPipeline
    ext32, ext16, scorer, rsd, ssd = _setup(7)
    lengths = [FRAMES] * VIDEOS
    offsets = synthetic.offsets_of(lengths)
    frames = (synthetic.make_frames_uniform(sum(lengths), dev, 1000) if kind == "uniform"
              else synthetic.make_frames_scenes(lengths, dev, 1000))
    ref = _oracle_scores(rsd, ssd, frames.cpu().numpy(), offsets, fpg)
    ext32, ext16, scorer = ext32.to(dev), ext16.to(dev), scorer.to(dev)

    # ---- fp32 parity mode: north_star's bars
    p32 = FrameScoringPipeline(ext32, scorer, use_inception=False, chunk_frames=256, frames_per_group=fpg)
    s32 = p32.score(frames, offsets).cpu().numpy()
    r32 = accuracy_report(s32, ref, offsets)
    print(f"\n[{kind} fpg={fpg}] fp32 vs oracle: {r32}")
    assert r32["score_max_abs_err"] < 1e-4
    for a, b in zip(offsets[:-1], offsets[1:]):
        safe = np.abs(ref[a:b] - ref[a:b].mean()) > 2e-6
        assert np.array_equal((s32[a:b] > s32[a:b].mean())[safe], (ref[a:b] > ref[a:b].mean())[safe])
        if safe.all():
            assert np.array_equal(p32.select(torch.from_numpy(s32[a:b]), [0, b - a])[0],
                                  np.flatnonzero(ref[a:b] > ref[a:b].mean()))
    assert r32["selection_agreement"] >= 1.0 - 2.0 / FRAMES and r32["f1_drift_max"] <= 1e-3
    assert r32["bars_met"], r32

    # ---- fp32-split mode: fp32 storage, convolution products on the bf16 matrix cores as hi*hi + hi*lo + lo*hi
    from avsum_amd.features.extractors import VisualFeatureExtractor
    exts = VisualFeatureExtractor(torch.float32, "batch", f32_split=True)
    exts.load_state_dict(ext32.state_dict())
    ps = FrameScoringPipeline(exts.to(dev), scorer, use_inception=False, chunk_frames=256, frames_per_group=fpg)
    ss = ps.score(frames, offsets).cpu().numpy()
    rs = accuracy_report(ss, ref, offsets)
    print(f"[{kind} fpg={fpg}] fp32-split vs oracle: {rs}")
    assert rs["score_max_abs_err"] < 1e-4                              # north_star's score bar holds
    assert rs["selection_agreement"] >= 0.99
    assert rs["bars_met"], rs                                          # ... and so do the guard-banded decision bars

    # ---- f16x2 mode (the bench headline): activations / weights stored as fp16 hi | lo runs, three fp16 MFMAs per
    # product, centred BatchNorm statistics; must meet north_star's bars as bench.py evaluates them (same helper)
    exth = VisualFeatureExtractor(torch.float32, "batch", f32_split="f16x2")
    exth.load_state_dict(ext32.state_dict())
    ph = FrameScoringPipeline(exth.to(dev), scorer, use_inception=False, chunk_frames=256, frames_per_group=fpg)
    sh = ph.score(frames, offsets).cpu().numpy()
    assert np.array_equal(sh, ph.score(frames, offsets).cpu().numpy())   # deterministic
    rh = accuracy_report(sh, ref, offsets)
    print(f"[{kind} fpg={fpg}] f16x2 vs oracle: {rh}")
    assert rh["score_max_abs_err"] < 2e-5                              # measured ~5e-6: fp32-class
    assert rh["bars_met"], rh
    assert rh["selection_agreement"] >= 1.0 - 2.0 / FRAMES

    # ---- bf16 throughput mode (the bench's): deterministic; measured deviations, asserted as they are
    p16 = FrameScoringPipeline(ext16, scorer, use_inception=False, chunk_frames=12288, frames_per_group=fpg)
    s16 = p16.score(frames, offsets).cpu().numpy()
    assert np.array_equal(s16, p16.score(frames, offsets).cpu().numpy())
    r16 = accuracy_report(s16, ref, offsets)
    print(f"[{kind} fpg={fpg}] bf16 vs oracle: {r16}")
    assert np.isfinite(s16).all()
    assert not r16["bars_met"]                                        # bf16 does NOT carry parity: said, not hidden
    # measured deviations + a small margin, so that a regression of a fused bf16 form shows here
    assert r16["score_max_abs_err"] < 0.25 * r16["score_range"]      # measured 0.13-0.20 of the range
    assert r16["selection_agreement"] > 0.68                          # measured 0.72-0.78
    assert r16["f1_drift_max"] < 0.12                                 # measured 0.02-0.09: NOT within 0.001
    assert np.corrcoef(s16, ref)[0, 1] > 0.75                         # the ranking signal survives
