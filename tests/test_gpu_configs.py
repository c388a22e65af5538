"""BASELINE.json configs as parity cases (the bench line is configs[1]; these are the others, at sizes the oracle
finishes in about a minute)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_config0_single_video_extract_score_f1(dev):
    """configs[0]: single 300-frame synthetic video (random 224x224x3 + 16 kHz sine), extractors -> av_model score,
    F1 via evaluation/metrics.py.  10 uniform shots of 30 frames; the reference's sampling rule keeps every third
    frame (10 per shot); audio features are the literal zeros."""
    from avsum_amd.evaluation.metrics import compute_temporal_f1, segments_from_indices, select_frames
    from avsum_amd.features.extractors import AVProcessor, sample_shot_indices
    from avsum_amd.models.av_model import AVBiLSTMModel
    from oracle import audio as oa, cnn as ocnn, scorer as osc, selection as osel
    rng = np.random.default_rng(1001)
    frames = rng.integers(0, 256, (300, 224, 224, 3), dtype=np.uint8)
    wave = (0.5 * np.sin(2 * np.pi * 440 * np.arange(160000) / 16000.0)).astype(np.float32)
    shots = [(30 * i, 30 * (i + 1)) for i in range(10)]
    torch.manual_seed(7)
    proc = AVProcessor(torch.float32, "batch")
    model = AVBiLSTMModel().eval()
    with torch.no_grad():
        model.scorer[0].weight.mul_(6.0)
        model.scorer[2].weight.mul_(6.0)
    rsd = {k: v.clone() for k, v in proc.visual_extractor.resnet.state_dict().items()}
    isd = {k: v.clone() for k, v in proc.visual_extractor.inception.state_dict().items()}
    ssd = {k: v.clone() for k, v in model.state_dict().items()}

    # oracle: the reference's per-shot loop
    vis_ref, aud_ref = [], []
    for s, e in shots:
        picked = [frames[i] for i in sample_shot_indices(s, e)]
        assert len(picked) == 10
        vis_ref.append(ocnn.visual_forward(rsd, isd, picked))
        aud_ref.append(oa.audio_forward_literal(wave[int(s / 30 * 16000):int(e / 30 * 16000)]))
    vis_ref, aud_ref = np.array(vis_ref), np.array(aud_ref)
    ref = osc.av_bilstm_forward(ssd, torch.from_numpy(vis_ref).float().unsqueeze(0),
                                torch.from_numpy(aud_ref).float().unsqueeze(0)).numpy()

    proc.visual_extractor.to(dev)
    vis, aud = proc.process_decoded(frames, wave, 30.0, shots)
    assert vis.shape == (10, 4096) and aud.shape == (10, 296) and aud.dtype == np.float64 and not aud.any()
    assert np.abs(vis - vis_ref).max() < 5e-4 * max(1.0, np.abs(vis_ref).max())
    with torch.no_grad():
        got = model.to(dev)(torch.from_numpy(vis).float().unsqueeze(0).to(dev),
                            torch.from_numpy(aud).float().unsqueeze(0).to(dev)).cpu().numpy()
    assert np.abs(got - ref).max() < 1e-4                      # importance scores within 1e-4 fp32
    safe = np.abs(ref - ref.mean()) > 2e-6
    assert np.array_equal((got > got.mean())[safe], (ref > ref.mean())[safe])  # selected shots identical
    gt = [(60, 120), (210, 270)]                               # seeded synthetic ground-truth segments (frames)
    def shot_segments(sel):
        return [(shots[i][0], shots[i][1]) for i in sel]
    f1 = compute_temporal_f1(shot_segments(select_frames(got)), gt, 300)
    f1_ref = osel.compute_temporal_f1(shot_segments(osel.select_frames(ref)), gt, 300)
    assert abs(f1 - f1_ref) <= 1e-3                            # north_star: F1 within 0.001 of the reference
    assert segments_from_indices(select_frames(got)) == osel.segments_from_indices(osel.select_frames(ref))


def test_decoded_video_with_detected_shots(dev):
    """The whole decoded-input front end of AVProcessor.process_video (extractors.py:339-362): shot detection on the
    GPU (row F2), the every-third-frame sampling rule, resize to 224 / 299, 4-frame micro-batches with ragged tails,
    per-shot means - against the oracle run shot by shot on the oracle's own cuts."""
    from avsum_amd.features.extractors import AVProcessor, sample_shot_indices
    from oracle import cnn as ocnn, shots as oshots
    rng = np.random.default_rng(12)
    n, h, w = 64, 40, 56
    frames = np.zeros((n, h, w, 3), dtype=np.uint8)
    base = rng.integers(0, 256, (h, w, 3))
    for f in range(n):
        if f in (22, 45):
            base = rng.integers(0, 256, (h, w, 3))
        frames[f] = np.clip(base + rng.integers(-3, 4, (h, w, 3)), 0, 255).astype(np.uint8)
    wave = np.zeros(int(n / 30 * 16000), dtype=np.float32)
    torch.manual_seed(9)
    proc = AVProcessor(torch.float32, "batch")
    rsd = {k: v.clone() for k, v in proc.visual_extractor.resnet.state_dict().items()}
    isd = {k: v.clone() for k, v in proc.visual_extractor.inception.state_dict().items()}
    shots = oshots.detect_shots(frames)
    assert shots == [(0, 22), (22, 45), (45, 64)]
    want = np.array([ocnn.visual_forward(rsd, isd, [frames[i] for i in sample_shot_indices(s, e)]) for s, e in shots])
    proc.visual_extractor.to(dev)
    vis, aud = proc.process_decoded(frames, wave, 30.0)          # shots=None: detected on the GPU
    assert vis.shape == (3, 4096) and aud.shape == (3, 296) and not aud.any()
    assert np.abs(vis - want).max() < 5e-4 * max(1.0, np.abs(want).max())


def test_config2_audio_visual_fusion(dev):
    """configs[2] at reduced size: mel / MFCC of a multi-sine + noise waveform, CNN embeddings, then
    features/fusion.py on the two 512-d embedded streams (cost matrix float64, DTW path, gather)."""
    from avsum_amd.features import fusion
    from avsum_amd.features.extractors import AudioFeatureExtractor
    from avsum_amd.models.av_model import AVBiLSTMModel
    from avsum_amd import ops
    from oracle import audio as oa, fusion as ofu
    g = torch.Generator().manual_seed(3003)
    t = 16000 * 8
    ts = torch.arange(t) / 16000.0
    wave = (0.4 * torch.sin(2 * np.pi * 300 * ts) + 0.3 * torch.sin(2 * np.pi * 1700 * ts)
            + 0.2 * torch.sin(2 * np.pi * 4100 * ts) + 0.05 * torch.randn(t, generator=g))
    ext = AudioFeatureExtractor(strict_reference=True)
    mel = ext._extract_mel(wave)
    truth = oa.extract_mel_f64(wave.numpy())
    assert mel.shape == (641, 128) and np.abs(mel - truth).max() < 1e-4
    torch.manual_seed(5)
    model = AVBiLSTMModel().eval()
    vis = torch.randn(40, 4096, generator=g)
    aud = torch.randn(55, 296, generator=g)
    with torch.no_grad():
        v512 = torch.relu(vis @ model.visual_fc[0].weight.t() + model.visual_fc[0].bias)
        a512 = torch.relu(aud @ model.audio_fc[0].weight.t() + model.audio_fc[0].bias)
        v_dev = ops.linear(vis.to(dev), model.visual_fc[0].weight.detach().to(dev), model.visual_fc[0].bias.detach().to(dev), 1)
        a_dev = ops.linear(aud.to(dev), model.audio_fc[0].weight.detach().to(dev), model.audio_fc[0].bias.detach().to(dev), 1)
    assert (v_dev.cpu() - v512).abs().max().item() < 1e-4
    cost = fusion.compute_dtw(v_dev.cpu(), a_dev.cpu())
    cost_ref = ofu.compute_dtw(v_dev.cpu(), a_dev.cpu())
    assert cost.dtype == np.float64 and np.abs(cost - cost_ref).max() <= 1e-12 * cost_ref.max()
    path = fusion.compute_optimal_path(cost_ref)
    assert np.array_equal(path, ofu.compute_optimal_path(cost_ref))
    assert torch.equal(fusion.interpolate_features(v_dev.cpu(), path, 40), ofu.interpolate_features(v_dev.cpu(), path, 40))


def test_batch_and_chunk_invariance_fp32(dev):
    """Size-independent properties of the sharded path (configs[3]): a video's scores do not depend on which
    other videos share the batch, nor on how the frames are cut into CNN chunks — bitwise in the fp32 parity
    mode (every stage is per-frame / per-video and deterministic there)."""
    from avsum_amd.features.extractors import VisualFeatureExtractor
    from avsum_amd.models.av_model import AVBiLSTMModel
    from avsum_amd.pipeline import FrameScoringPipeline
    torch.manual_seed(61)
    ext = VisualFeatureExtractor(torch.float32, "batch").to(dev)
    model = AVBiLSTMModel().eval().to(dev)
    rng = np.random.default_rng(4004)
    frames = torch.from_numpy(rng.integers(0, 256, (11, 224, 224, 3), dtype=np.uint8)).to(dev)
    offsets = [0, 3, 8, 11]
    ref = FrameScoringPipeline(ext, model, use_inception=False, chunk_frames=1024).score(frames, offsets).cpu()
    small = FrameScoringPipeline(ext, model, use_inception=False, chunk_frames=2).score(frames, offsets).cpu()
    assert torch.equal(ref, small)
    alone = FrameScoringPipeline(ext, model, use_inception=False).score(frames[3:8].contiguous(), [0, 5]).cpu()
    assert torch.equal(alone, ref[3:8])
    sel = FrameScoringPipeline.select(ref, offsets)
    assert [len(s) for s in sel] and all((np.diff(s) > 0).all() for s in sel if len(s) > 1)  # sorted indices


def test_ragged_micro_batches_fp32(dev):
    """frames_per_group = 4 on videos whose lengths are not multiples of 4: the pipeline's equal-size sets (full
    groups gathered across videos, tails per size) give every frame the features of ITS group - identical, in the
    deterministic fp32 mode, to one pass over explicitly ragged groups."""
    from avsum_amd.features.extractors import VisualFeatureExtractor
    from avsum_amd.pipeline import FrameScoringPipeline
    torch.manual_seed(63)
    ext = VisualFeatureExtractor(torch.float32, "batch").to(dev)
    rng = np.random.default_rng(6)
    frames = torch.from_numpy(rng.integers(0, 256, (14, 224, 224, 3), dtype=np.uint8)).to(dev)
    offsets = [0, 6, 11, 14]                       # groups 4+2 | 4+1 | 3
    pipe = FrameScoringPipeline(ext, None, use_inception=False, chunk_frames=8, frames_per_group=4)
    got = pipe.embed(frames, offsets)[:, :2048].cpu()
    want = ext._resnet_runner.forward(frames, pipe._group_offsets(offsets)).cpu()
    assert torch.equal(got, want)


def test_pinned_host_frames_are_uploaded_pass_by_pass(dev):
    """Frames in pinned host memory (the PCIe-inclusive path): a short lead pass, then equal passes uploaded by the copy
    stream while the previous one computes; fp32 features are bit-identical to the HBM-resident path for every pass
    structure (every BatchNorm group is one frame here), host tensors that are not pinned are refused."""
    from avsum_amd.features.extractors import VisualFeatureExtractor
    from avsum_amd.pipeline import FrameScoringPipeline
    torch.manual_seed(63)
    ext = VisualFeatureExtractor(torch.float32, "batch").to(dev)
    rng = np.random.default_rng(7)
    host = torch.from_numpy(rng.integers(0, 256, (23, 224, 224, 3), dtype=np.uint8))
    offsets = [0, 9, 23]
    want = FrameScoringPipeline(ext, None, use_inception=False, chunk_frames=64).embed(host.to(dev), offsets)
    pipe = FrameScoringPipeline(ext, None, use_inception=False, chunk_frames=8)
    pipe.host_lead_frames = 3                       # passes of 3 | 7 | 7 | 6 frames
    got = pipe.embed(host.pin_memory(), offsets)
    assert got.is_cuda and torch.equal(got, want)
    pipe.host_lead_frames = 0                       # no lead pass: 8 | 8 | 7
    assert torch.equal(pipe.embed(host.pin_memory(), offsets), want)
    # two uploads back to back with NO host synchronisation in between (the second call's staging buffers may be
    # the blocks the first call's kernels still read; the stager orders its copy stream behind the compute stream)
    host2 = torch.from_numpy(rng.integers(0, 256, (23, 224, 224, 3), dtype=np.uint8))
    want2 = FrameScoringPipeline(ext, None, use_inception=False, chunk_frames=64).embed(host2.to(dev), offsets)
    torch.cuda.synchronize()
    p1, p2 = host.pin_memory(), host2.pin_memory()
    g1 = pipe.embed(p1, offsets)
    g2 = pipe.embed(p2, offsets)
    g3 = pipe.embed(p1, offsets)
    assert torch.equal(g1, want) and torch.equal(g2, want2) and torch.equal(g3, want)
    # a stream of batches: the call is told the NEXT batch and uploads its first pass under its own last pass; the next call
    # finds it staged (no lead pass), a call with ANOTHER batch than announced uploads afresh - same features every time
    g4 = pipe.embed(p1, offsets, next_batch=(p2, offsets))
    assert "prefetched" in pipe._stager_cache
    g5 = pipe.embed(p2, offsets, next_batch=(p1, offsets))      # consumes the staged pass, announces p1
    g6 = pipe.embed(p2, offsets)                                  # not the announced batch: staged pass discarded
    assert "prefetched" not in pipe._stager_cache
    assert torch.equal(g4, want) and torch.equal(g5, want2) and torch.equal(g6, want2)
    with pytest.raises(ValueError, match="pinned"):
        pipe.embed(host, offsets)


def test_bf16_batch_invariance(dev):
    """Same property in the bf16 throughput mode.  Every BatchNorm form is deterministic and per-group, but the
    chunking decides which FORM a layer takes (a 3-frame pass and a 9-frame pass tile differently), so two chunkings
    agree to bf16 rounding noise; the same chunking twice agrees bit for bit."""
    from avsum_amd.features.extractors import VisualFeatureExtractor
    from avsum_amd.models.av_model import AVBiLSTMModel
    from avsum_amd.pipeline import FrameScoringPipeline
    torch.manual_seed(62)
    ext = VisualFeatureExtractor(torch.bfloat16, "batch").to(dev)
    model = AVBiLSTMModel().eval().to(dev)
    rng = np.random.default_rng(5)
    frames = torch.from_numpy(rng.integers(0, 256, (9, 224, 224, 3), dtype=np.uint8)).to(dev)
    a = FrameScoringPipeline(ext, model, use_inception=False, chunk_frames=1024).score(frames, [0, 4, 9]).cpu()
    b = FrameScoringPipeline(ext, model, use_inception=False, chunk_frames=3).score(frames, [0, 4, 9]).cpu()
    assert (a - b).abs().max().item() < 2e-3
    again = FrameScoringPipeline(ext, model, use_inception=False, chunk_frames=3).score(frames, [0, 4, 9]).cpu()
    assert torch.equal(b, again)


def test_bench_contract_line(dev):
    """bench.py on a tiny instance of configs[1]: exactly one JSON line on stdout with the fields the driver reads,
    the roofline measured live (HIP events) and the CPU-baseline leg."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--videos", "2", "--mean-frames", "24",
                          "--chunk", "16", "--steps", "1", "--warmup", "1", "--cpu-sample", "4"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-800:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["steps"] == 1 and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f16x2" and d["value"] > 0 and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["launches"] > 0 and r["avg_launch_us"] > 0 and r["algorithmic_bytes_per_launch"] > 0
    assert abs(r["peak"] - 2500.0 / 3.0) < 1.0 and abs(r["mfma_rate_tflops"] - 3 * r["achieved"]) < 0.5
    assert r["traffic"] is None or "profiles/" in r["traffic_source"]   # replayed from a committed PMC summary, labelled
    assert abs(r["frac_of_nominal_peak"] - r["achieved"] / 2500.0) < 1e-3   # the other reading, side by side
    assert r["forms"] and abs(sum(f["share_of_step"] for f in r["forms"]) - r["share_of_step"]) < 0.02
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "frames/s" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    assert c["runs"] == 5 and c["min"] <= c["value"] <= c["max"]       # median of 5 runs
    a = d["accuracy"]                                                   # the timed arithmetic mode against the oracle
    assert a["mode"] == "f16x2" and a["deterministic"] is True and a["frames"] == 20 and a["videos"] == 5
    assert 0.0 <= a["selection_agreement"] <= 1.0 and a["score_max_abs_err"] >= 0
    assert a["bars_met"] is True and a["guard_band"] == 2e-4 and a["guarded_frames"] >= 0   # the headline carries parity
    assert d["sub_results"] is None                                     # only with the default headline


@pytest.mark.parametrize("dtype,config", [("f32split", "1"), ("f32", "1"), ("bf16", "1"), ("f16x2", "3")])
def test_bench_other_headlines(dev, dtype, config):
    """The non-default headlines of bench.py stay runnable: the fp32 parity / fp32-split arithmetic modes (whose
    roofline is priced against the fp32 peak / a third of the bf16 peak) and one rank's share of configs[3]."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", config, "--dtype", dtype, "--videos", "2",
                          "--mean-frames", "24", "--chunk", "16", "--steps", "1", "--warmup", "1", "--cpu-sample", "0",
                          "--sub", "none"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-800:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] > 0 and d["dtype"] == dtype
    r = d["roofline"]
    assert r is not None and r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    if dtype == "f32split":
        assert abs(r["peak"] - 2500.0 / 3.0) < 1.0


def test_bench_config4_training_leg(dev):
    """BASELINE configs[4] at world size 1 (scripts/train_av_model.py:70-96 on synthetic labels): the leg bench.py reports
    as sub_results.config4_training - steps/s at T = 300 and 1800, the LSTM sweeps per time step, and the first six losses
    against the oracle's (<= 1e-4 relative)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "4", "--steps", "3"],
                         capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-800:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    leg = json.loads(lines[0])["detail"]
    assert leg["loss_trajectory"]["within_bar"] is True and leg["loss_trajectory"]["steps"] == 6
    for t_len in ("300", "1800"):
        r = leg["lengths"][t_len]
        assert r["steps_per_s"] > 0 and r["lstm_forward_us_per_time_step"] > 0
        # the backward sweep keeps W_hh on chip like the forward: within 2x of its per-step time
        assert r["lstm_backward_over_forward"] <= 2.0, r


def test_bench_under_the_launcher_one_rank(dev):
    """The driver's launch line at N = 1: python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment, rendezvous on 127.0.0.1)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29600 + os.getpid() % 300
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                          "--gpus", "1", "--videos", "2", "--mean-frames", "24", "--chunk", "16", "--steps", "1",
                          "--warmup", "1", "--cpu-sample", "0"],
                         capture_output=True, text=True, timeout=600, cwd=root,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stderr[-800:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["cpu_baseline"] is None and d["accuracy"] is None


def test_rccl_world1_collectives_on_device(dev):
    """SURVEY 4.5 / VERDICT r1 #7a: bring RCCL up on the MI355X (backend "nccl", world_size 1, rendezvous on
    127.0.0.1) and run the three exchanges of the hot path on DEVICE tensors: C1 weight broadcast, C2 ragged score
    gather, C3 gradient all-reduce.  (N > 1 needs more GPUs than a box has; the rank logic is covered on gloo.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys, torch
sys.path.insert(0, os.environ["AVS_ROOT"])
import torch.distributed as dist
from avsum_amd import dist as avd
avd.FORCE_COLLECTIVES = True      # the product's exchanges take their real collective path in this one-rank group
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
t = torch.arange(8, dtype=torch.float32, device=dev)
dist.all_reduce(t)                                   # a real RCCL collective on the device
dist.broadcast(t, 0)
assert torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32))
parts = [torch.empty_like(t)]
dist.all_gather(parts, t)
assert torch.equal(parts[0], t)
mod = torch.nn.Sequential(torch.nn.Linear(8, 3), torch.nn.BatchNorm1d(3)).to(dev)
before = [p.detach().clone() for p in mod.parameters()]
versions = [p._version for p in mod.parameters()]
avd.broadcast_module(mod, 0)                         # C1: flat bucket per dtype through RCCL, copied back
assert all(torch.equal(a, b) for a, b in zip(before, mod.parameters()))
assert all(p._version > v for p, v in zip(mod.parameters(), versions))   # the bucket path really ran
lengths = [5, 9, 2]
local = torch.cat([torch.full((n,), float(v), device=dev) for v, n in enumerate(lengths)])
out = avd.gather_video_scores(local, [2, 0, 1], lengths, 4)      # C2: padded all-gather, global ids, one id absent
assert out[3] is None and [out[v].shape[0] for v in (2, 0, 1)] == lengths
assert [float(out[v][0]) for v in (2, 0, 1)] == [0.0, 1.0, 2.0]
assert all(o.data_ptr() != local.data_ptr() for o in out if o is not None)   # slices of the GATHERED buffers
for i, p in enumerate(mod.parameters()):
    p.grad = torch.full_like(p, float(i + 1))
avd.allreduce_gradients(mod)                         # C3
assert all(torch.allclose(p.grad, torch.full_like(p, float(i + 1))) for i, p in enumerate(mod.parameters()))
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_OK", torch.cuda.get_device_name(0))
'''
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29000 + os.getpid() % 2000), AVS_ROOT=root,
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "RCCL_OK" in out.stdout, out.stderr[-1500:]


def test_two_rank_rehearsal_of_the_bench_on_one_gpu(tmp_path):
    """The multi-rank program flow on real kernels: two ranks of `bench.py --gpus 2` share this box's GPU (AVS_DIST_REHEARSAL=1:
    ranks map onto the devices that exist, collectives on gloo - RCCL refuses two ranks on one GPU), started the way the
    driver starts N > 1 (torch.distributed.run).  Weights broadcast from rank 0, each rank its own batch, scores gathered under
    global ids, the slowest rank's time, ONE JSON line from rank 0 with n_gpus = 2 - for the headline path and the configs[4]
    training leg (gradients averaged over the ranks before AdamW).  Not a measurement."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AVS_DIST_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for port, extra, unit in ((29521, ["--steps", "1", "--warmup", "1", "--videos", "4", "--mean-frames", "400", "--cpu-sample", "0"],
                               "frames/s"),
                              (29522, ["--config", "4", "--steps", "3"], "videos/s")):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2"] + extra
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["unit"] == unit and d["value"] > 0 and d["scaling"] == "weak"
