"""CPU-side checks: the C-ABI library loads and exports every declared symbol (no compute calls without a
GPU), argument validation that happens before any launch, host logic of the mirrors."""
import ctypes
import os

import numpy as np
import pytest
import torch


def test_library_exports_every_declared_symbol():
    from avsum_amd import _abi
    names = _abi.declared_symbols()
    assert len(names) >= 24 and "avs_conv2d_nhwc" in names and "avs_lstm_f32" in names
    lib = _abi.lib()
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_abi._SIGNATURES)
    assert lib.avs_abi_version() == 5
    assert os.path.dirname(_abi.LIB_PATH).startswith(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_shipped_library_has_no_global_tuning_state():
    """the C-ABI holds no process-global mutable state.  The tile / staging variant of a convolution and
    the LSTM variant are per-call arguments; the rule thresholds are compile-time constants; the avs_tune_* setters and
    avs_debug_flags exist in the kernel-study build only - the shipped library does not even export the symbols."""
    import re
    import subprocess
    from avsum_amd import _abi
    out = subprocess.run(["nm", "-D", "--defined-only", _abi.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\b(avs_[a-z0-9_]+)\b", out))
    assert exported, "nm found no avs_ symbols"
    assert not [n for n in exported if n.startswith(("avs_tune", "avs_debug"))]
    assert exported == set(_abi.declared_symbols())          # exactly the header's functions, nothing else
    header = open(_abi.HEADER_PATH).read()
    assert "avs_tune_" not in re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    # no writable file-scope variables in the kernel sources outside the study build and the thread-local error text
    src_dir = os.path.join(os.path.dirname(_abi.LIB_PATH), "..", "csrc")
    for name in sorted(os.listdir(src_dir)):
        if not name.endswith(".hip"):
            continue
        text = open(os.path.join(src_dir, name)).read()
        text = re.sub(r"#ifdef AVS_STUDY.*?#(?:else|endif)", "", text, flags=re.S)
        for m in re.finditer(r"^static (?!constexpr|inline|int [a-z_0-9]+\(|void|bool [a-z_0-9]+\()[^;(]*;", text, flags=re.M):
            assert "thread_local" in m.group(0) or "const" in m.group(0), (name, m.group(0))


def test_validation_before_launch_needs_no_gpu():
    """Bad arguments are rejected with a status code + message before any HIP call."""
    from avsum_amd import _abi
    lib = _abi.lib()
    st = lib.avs_gemm_nt(0, 8, 8, 6, None, 6, 0, None, 6, 0, None, 8, 0, None, 0, 0, 1.0, 0, 1, None)
    assert st == -1 or st == -2
    assert lib.avs_last_error()
    st = lib.avs_gemm_nt(7, 8, 8, 8, None, 8, 0, None, 8, 0, None, 8, 0, None, 0, 0, 1.0, 0, 1, None)
    assert st == -1 and b"dtype" in lib.avs_last_error()
    assert lib.avs_dtw_workspace_bytes(10, 20) >= 200
    assert lib.avs_lstm_f32(None, None, 0, 2, 0, None, 1, None, 0, 0, 0, None) == -2
    assert lib.avs_cdist_f64(None, 4, None, 4, 0, None, None) == -2
    # the split recurrence: 64 bytes of error word + two slots of 1024 granules per recurrence; built for hidden = 256;
    # a workspace that is too small is refused
    assert lib.avs_lstm_split_workspace_bytes(4, 25) == 64 + 4 * 25 * 2 * 1024 * 8 and lib.avs_lstm_split_workspace_bytes(0, 3) == 0
    assert lib.avs_lstm_split_f32(None, None, 128, 2, 0, None, 1, None, 256, 0, None, None, None, 0, 0, None) == -6
    assert lib.avs_lstm_split_f32(None, None, 256, 2, 0, None, 1, None, 512, 0, None, None, None, 0, 0, None) == -5
    assert lib.avs_lstm_bwd_split_f32(None, 512, 0, None, None, None, 256, 2, 0, None, 1, None, None, 0, 0, None) == -5
    assert b"workspace" in lib.avs_last_error()


def test_missing_library_fails_loudly(monkeypatch):
    from avsum_amd import _abi
    monkeypatch.setattr(_abi, "_lib", None)
    monkeypatch.setattr(_abi, "LIB_PATH", "/nonexistent/libavsum_hip.so")
    with pytest.raises(_abi.AvsError, match="no CPU fallback"):
        _abi.lib()


def test_product_path_never_imports_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "audiovidsum-a-multi-modal-approach-to-video-summarization_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dp, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, os.path.join(dp, f)


def test_host_tensors_are_rejected_not_computed_on_cpu():
    from avsum_amd import ops
    from avsum_amd.models.av_model import AVBiLSTMModel
    with pytest.raises(ValueError, match="no CPU fallback"):
        ops.linear(torch.zeros(4, 8), torch.zeros(4, 8))
    m = AVBiLSTMModel(16, 8, 8).eval()
    with pytest.raises(RuntimeError, match="HIP path only"):
        m(torch.zeros(1, 3, 16), torch.zeros(1, 3, 8))


def test_sampling_rule_and_selection():
    from avsum_amd.features.extractors import sample_shot_indices
    from avsum_amd.evaluation.metrics import binary_f1, segments_from_indices, select_frames
    assert sample_shot_indices(0, 10) == [0, 3, 6, 9]
    assert sample_shot_indices(4, 11) == [6, 9]          # ABSOLUTE index % 3 (extractors.py:406)
    assert len(sample_shot_indices(0, 1000)) == 100      # max 100 frames (extractors.py:400,404)
    assert sample_shot_indices(5, 5) == []
    pred = np.array([0.1, 0.9, 0.8, 0.2, 0.7, 0.1], dtype=np.float32)
    assert select_frames(pred).tolist() == [1, 2, 4]
    assert segments_from_indices(select_frames(pred)) == [(1, 3), (4, 5)]
    assert segments_from_indices(np.array([], dtype=np.int64)) == []
    tgt = np.array([0, 1, 0, 0, 1, 1], dtype=np.float32)
    p, r = 2 / 3, 2 / 3
    assert binary_f1(pred, tgt) == 2 * (p * r) / (p + r + 1e-8)


def test_extractor_containers_have_torchvision_keys():
    from avsum_amd.cnn import Inception3, resnet50_trunk
    t = resnet50_trunk()
    keys = list(t.state_dict().keys())
    assert keys[0] == "0.weight" and "4.0.downsample.0.weight" in keys and "7.2.bn3.running_var" in keys
    assert sum(p.numel() for p in t.parameters()) == 23508032       # resnet50 minus fc
    i = Inception3()
    ik = i.state_dict().keys()
    assert "Conv2d_1a_3x3.conv.weight" in ik and "Mixed_7c.branch_pool.bn.running_mean" in ik
    assert "AuxLogits.conv0.conv.weight" in ik and not any(k.startswith("fc.") for k in ik)
    assert sum(p.numel() for n, p in i.named_parameters() if not n.startswith("AuxLogits")) == 21785568
    assert t.training and not i.training  # only the Inception net is put in eval mode


def test_binding_brings_up_torchs_hip_runtime_first():
    """One HIP runtime per process: the binding must import torch before it dlopens libavsum_hip.so (which links
    /opt/rocm's libamdhip64 of the same SONAME); in the other order the second runtime finds no device - seen as
    build() followed by smoke() in one process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import avsum_amd._abi as a; assert 'torch' in sys.modules; "
            "a.lib(); print('ok')" % root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-500:]


def test_preprocess_orchestration_and_disk_format(tmp_path):
    """scripts/preprocess.py counterpart (reference scripts/preprocess.py:32-85): walks *.mp4, writes
    data/processed/<vid>/{visual,audio}.npy as float32 [S,4096] / [S,296], skips finished videos, removes a partial
    directory on failure, and the dataset reader (data/dataset.py:35-62) returns what was written.  Host logic only:
    the processor is a stand-in."""
    import pandas as pd
    from avsum_amd.data.dataset import TVSumDataset
    from avsum_amd.scripts.preprocess import preprocess_dataset
    raw, out = tmp_path / "raw", tmp_path / "processed"
    raw.mkdir()
    for name in ("a.mp4", "b.mp4", "bad.mp4", "notes.txt"):
        (raw / name).write_bytes(b"x")
    calls = []

    class FakeProcessor:
        def process_video(self, path):
            calls.append(os.path.basename(path))
            if "bad" in path:
                os.makedirs(out / "bad", exist_ok=True)          # a partial output directory
                raise RuntimeError("decode failed")
            s = 3 if path.endswith("a.mp4") else 5
            return np.full((s, 4096), 0.5, np.float64), np.zeros((s, 296))   # audio: the literal float64 zeros

    done, failed = preprocess_dataset(str(raw), str(out), FakeProcessor())
    assert done == ["a", "b"] and failed == ["bad"] and calls == ["a.mp4", "b.mp4", "bad.mp4"]
    assert not (out / "bad").exists()                             # preprocess.py:83-85
    v = np.load(out / "a" / "visual.npy")
    a = np.load(out / "b" / "audio.npy")
    assert v.dtype == np.float32 and v.shape == (3, 4096) and a.dtype == np.float32 and a.shape == (5, 296)
    calls.clear()
    done, failed = preprocess_dataset(str(raw), str(out), FakeProcessor())
    assert done == [] and calls == ["bad.mp4"]                    # preprocess.py:46-55: finished videos are skipped
    df = pd.DataFrame({"Video File Name": ["a", "a", "b"],
                       "Annotations": [np.ones(4), 3 * np.ones(4), np.arange(4.0)]})
    ds = TVSumDataset(df, str(out))
    feats, scores = ds[0]
    assert feats["visual"].shape == (3, 4096) and feats["audio"].shape == (3, 296)
    assert torch.equal(scores, torch.full((4,), 2.0))             # mean over the users' annotations


def test_pipeline_group_offsets():
    from avsum_amd.pipeline import FrameScoringPipeline
    p = FrameScoringPipeline(None, None, frames_per_group=4)
    assert p._group_offsets([0, 6, 6, 15]) == [0, 4, 6, 10, 14, 15]   # groups never straddle a video
    p1 = FrameScoringPipeline(None, None, frames_per_group=1)
    assert p1._group_offsets([0, 3]) == [0, 1, 2, 3]


def test_pipeline_uniform_sets_partition_the_frames():
    """Ragged micro-batches (a video whose length is not a multiple of the group size) are split into sets of
    equal-sized groups that together cover every frame exactly once and never mix videos inside a group."""
    from avsum_amd.pipeline import FrameScoringPipeline
    p = FrameScoringPipeline(None, None, frames_per_group=4)
    offs = [0, 6, 6, 15, 23, 24]                  # lengths 6, 0, 9, 8, 1
    sets = p._uniform_sets(offs)
    assert [g for g, _ in sets] == [4, 1, 2]
    cover = np.concatenate([np.arange(*w) if isinstance(w, tuple) else w for _, w in sets])
    assert sorted(cover.tolist()) == list(range(24))
    ref_groups = p._group_offsets(offs)
    ref = {tuple(range(a, b)) for a, b in zip(ref_groups[:-1], ref_groups[1:])}
    got = set()
    for g, w in sets:
        idx = np.arange(*w) if isinstance(w, tuple) else w
        got |= {tuple(idx[i:i + g].tolist()) for i in range(0, len(idx), g)}
    assert got == ref                              # the same groups the per-video rule makes
    assert p._uniform_sets([0, 8, 16]) == [(4, (0, 16))]
    assert FrameScoringPipeline(None, None, frames_per_group=1)._uniform_sets([0, 5, 9]) == [(1, (0, 9))]
    assert p._uniform_sets([0, 0]) == []


def test_audio_constants_match_oracle_formulas():
    from avsum_amd.audio import dct_matrix, mel_filterbank, windowed_dft_basis
    from oracle import audio as oa
    fb = mel_filterbank(16000, 128)
    assert np.array_equal(fb, oa.melscale_fbanks().numpy())
    assert ((fb > 0).sum(), (fb.sum(0) == 0).sum(), (fb > 0).sum(1).max()) == (394, 4, 2)
    d = dct_matrix(40, 128)
    assert np.array_equal(d.T, oa.create_dct().numpy())
    assert np.abs(d.astype(np.float64) @ d.T.astype(np.float64) - np.eye(40)).max() < 1e-5  # orthonormal rows
    b = windowed_dft_basis()
    assert b.shape == (402, 400)
    w = torch.hann_window(400).double().numpy()
    assert np.abs(b[0] - w).max() < 1e-12 and np.abs(b[201]).max() == 0  # k=0: cos=1, sin=0


class _FakeCapture:
    """Stands in for cv2.VideoCapture: frame i is filled with the value i; reads fail past `total`."""

    def __init__(self, total, shape=(4, 4, 3)):
        self.total, self.shape, self.pos, self.sets = total, shape, 0, []

    def set(self, prop, value):
        self.sets.append((prop, value))
        self.pos = int(value)
        return True

    def read(self):
        if self.pos >= self.total:
            return False, None
        frame = np.full(self.shape, self.pos % 256, dtype=np.uint8)
        self.pos += 1
        return True, frame


def test_avprocessor_extract_frames_follows_the_reference_rule():
    """AVProcessor._extract_frames (features/extractors.py:395-413): seek, absolute index % 3, <= 100 frames, stop at
    a failed read, 1-/4-channel frames forced to 3 channels."""
    from avsum_amd.features.extractors import AVProcessor, sample_shot_indices
    extract = AVProcessor._extract_frames
    cap = _FakeCapture(1000)
    got = extract(None, cap, 4, 20)
    assert cap.sets == [(1, 4)]                                   # cv2.CAP_PROP_POS_FRAMES == 1
    assert [int(f[0, 0, 0]) for f in got] == [6, 9, 12, 15, 18] == sample_shot_indices(4, 20)
    assert all(f.shape == (4, 4, 3) and f.dtype == np.uint8 for f in got)
    assert len(extract(None, _FakeCapture(1000), 0, 1000)) == 100  # max_frames
    assert [int(f[0, 0, 0]) for f in extract(None, _FakeCapture(8), 3, 50)] == [3, 6]   # the video ends at frame 8
    assert extract(None, _FakeCapture(10), 5, 5) == []
    grey = extract(None, _FakeCapture(4, (4, 4, 1)), 0, 4)
    assert grey[0].shape == (4, 4, 3) and (grey[1] == 3).all()
    assert extract(None, _FakeCapture(4, (4, 4, 4)), 0, 1)[0].shape == (4, 4, 3)


def test_avprocessor_audio_boundary_errors_and_wav_reader(tmp_path):
    """_extract_audio wraps every failure as the reference does (:385-386: RuntimeError 'Audio extraction failed');
    the WAV reader reproduces torchaudio.load(...).mean(0) for PCM16 (:326-328)."""
    import wave
    from avsum_amd.features.extractors import AVProcessor
    with pytest.raises(RuntimeError, match="Audio extraction failed"):
        AVProcessor._extract_audio(None, str(tmp_path / "missing.mp4"), str(tmp_path / "a.wav"))
    pcm = (np.arange(-8, 8, dtype=np.int16) * 1000).reshape(-1, 2)       # 8 stereo frames
    path = str(tmp_path / "s.wav")
    with wave.open(path, "wb") as wf:
        wf.setnchannels(2)
        wf.setsampwidth(2)
        wf.setframerate(16000)
        wf.writeframes(pcm.tobytes())
    got = AVProcessor._load_wav_mono(path)
    assert got.dtype == np.float32 and np.array_equal(got, (pcm.astype(np.float32) / 32768.0).mean(axis=1))


def test_avprocessor_has_the_reference_surface():
    """the drop-in surface of features/extractors.py:298-413."""
    import inspect
    from avsum_amd.features.extractors import AVProcessor
    for name, params in (("process_video", ["self", "video_path"]), ("_extract_audio", ["self", "video_path", "audio_path"]),
                         ("_detect_shots", ["self", "video_path"]), ("_extract_frames", ["self", "cap", "start", "end"])):
        assert list(inspect.signature(getattr(AVProcessor, name)).parameters) == params


def test_synthetic_config_generators():
    """the five BASELINE configs as seeded generators (lengths, global video ids, sharding); frames and
    waveforms have the stated shapes / statistics (generated on the CPU here, in HBM in the bench)."""
    from avsum_amd import synthetic as sy
    c1 = sy.config(1)
    assert len(c1["lengths"]) == 25 and sum(c1["lengths"]) == 45143 and min(c1["lengths"]) >= 900 and max(c1["lengths"]) <= 2700
    assert c1["video_ids"] == list(range(25)) and c1["num_videos"] == 25
    assert sy.config(1) == c1                                             # seeded
    r3 = sy.config(1, rank=3, world=8)
    assert r3["video_ids"] == list(range(75, 100)) and r3["num_videos"] == 200 and r3["lengths"] != c1["lengths"]
    c2 = sy.config(2)
    assert len(c2["lengths"]) == 50 and all(2000 <= n <= 10000 for n in c2["lengths"])
    seen = []
    for r in range(8):                                                    # configs[3]: one global list, sharded
        c3 = sy.config(3, r, 8)
        assert c3["num_videos"] == 400 and c3["lengths"] == [5000] * 50
        seen += c3["video_ids"]
    assert sorted(seen) == list(range(400))
    assert sy.config(3, 0, 1)["num_videos"] == 50                         # the one-GPU share
    assert sy.config(0)["lengths"] == [300] and sy.uniform_shots(300) == [(30 * i, 30 * i + 30) for i in range(10)]
    with pytest.raises(ValueError):
        sy.config(4)
    assert sy.offsets_of([3, 5]) == [0, 3, 8]
    f = sy.make_frames_uniform(5, torch.device("cpu"), 1)
    assert f.shape == (5, 224, 224, 3) and f.dtype == torch.uint8 and 120 < f.float().mean() < 135
    assert torch.equal(f, sy.make_frames_uniform(5, torch.device("cpu"), 1))
    s = sy.make_frames_scenes([70, 50], torch.device("cpu"), 2, scene_frames=60)
    assert s.shape == (120, 224, 224, 3)
    within = (s[1].float() - s[2].float()).abs().mean()                   # neighbours of one scene: drift + noise
    across = (s[1].float() - s[65].float()).abs().mean()                  # different scenes
    assert within < 15 and across > 2 * within
    w = sy.make_waveform(16000, 5)
    assert w.shape == (16000,) and w.dtype == torch.float32 and 0.3 < w.abs().max() < 1.0
    sine = sy.make_waveform(16000, 0, "sine")
    assert abs(float(sine.abs().max()) - 0.5) < 1e-3


def test_accuracy_report_counts():
    from avsum_amd.evaluation.accuracy import accuracy_report, synthetic_gt_segments
    ref = np.linspace(0.0, 1.0, 100, dtype=np.float32)
    rep = accuracy_report(ref, ref, [0, 60, 100])
    assert rep["selection_agreement"] == 1.0 and rep["f1_drift_max"] == 0.0 and rep["videos"] == 2 and rep["frames"] == 100
    flipped = ref.copy()
    flipped[:60] = flipped[:60][::-1]                                     # first video: selection mirrored
    rep = accuracy_report(flipped, ref, [0, 60, 100])
    assert rep["selection_agreement"] == 0.4 and rep["score_max_abs_err"] > 0.5
    segs = synthetic_gt_segments(300, 900)
    assert segs == synthetic_gt_segments(300, 900) and all(0 <= a < b <= 300 for a, b in segs)


def test_pipeline_pass_sizes_cover_every_frame_in_equal_passes():
    from avsum_amd.pipeline import FrameScoringPipeline
    f = FrameScoringPipeline._pass_frames
    assert f(45143, 24576, 1) == 22572 and f(45143, 24576, 4) == 22572 and f(250000, 24576, 1) == 22728
    assert f(11, 4, 4) == 4 and f(11, 1024, 1) == 11 and f(5, 8, 4) == 8 and f(7, 2, 1) == 2 and f(1, 256, 1) == 1
    for count, chunk, gsz in ((45143, 24576, 1), (99991, 12288, 4), (13, 5, 3), (24577, 24576, 1), (8, 8, 8), (9, 8, 8)):
        per = f(count, chunk, gsz)
        assert per % gsz == 0 and gsz <= per <= max(gsz, chunk // gsz * gsz)
        passes = -(-count // per)
        assert passes == max(1, -(-count // max(gsz, chunk // gsz * gsz)))     # never more passes than the cap needs
        assert (passes - 1) * per < count <= passes * per


def test_audio_batch_tables_host_logic():
    """MelPlan.batch_tables: tracks laid out one after another at 16-byte aligned offsets, block rows (first frame inside
    the track, frames <= 32, segment over the batch, track)."""
    import torch
    from avsum_amd.audio import MelPlan
    waves = [torch.arange(1001, dtype=torch.float32), torch.ones(7000)]
    bounds = [[0, 2, 1 + 1001 // 200], [0, 33, 33, 1 + 7000 // 200]]
    cat, toff, tlen, blocks, seg_block, seg_frames = MelPlan.batch_tables(waves, bounds, torch.device("cpu"))
    assert toff.tolist() == [0, 1004] and tlen.tolist() == [1001, 7000] and cat.numel() == 1004 + 7000
    assert torch.equal(cat[:1001], waves[0]) and cat[1001:1004].abs().sum() == 0 and torch.equal(cat[1004:], waves[1])
    assert seg_frames.tolist() == [2, 4, 33, 0, 3] and seg_block.tolist() == [0, 1, 2, 4, 4, 5]
    assert blocks.tolist() == [[0, 2, 0, 0], [2, 4, 1, 0], [0, 32, 2, 1], [32, 1, 2, 1], [33, 3, 4, 1]]


def test_stem_normalisation_folded_into_the_weights():
    """ops.stem_h2_operands: conv1 of the NORMALISED image ((v - mean_c) / std_c, zero padded AFTER the normalisation,
    features/extractors.py:126-140 + the trunk's conv1) equals the convolution of the RAW bytes + an "inside the image"
    channel with the folded weights - border rows / columns included (float64 check of the algebra the fused f16x2 stem
    relies on; its kernel is tested on the GPU)."""
    import torch
    import torch.nn.functional as F
    from avsum_amd import ops
    from avsum_amd.cnn import RESNET_MEAN, RESNET_STD
    g = torch.Generator().manual_seed(4)
    w = torch.randn(64, 3, 7, 7, generator=g)
    v = torch.randint(0, 256, (2, 3, 37, 41), generator=g).double()
    mean = torch.tensor(RESNET_MEAN, dtype=torch.float64).view(1, 3, 1, 1)
    std = torch.tensor(RESNET_STD, dtype=torch.float64).view(1, 3, 1, 1)
    ref = F.conv2d((v - mean) / std, w.double(), None, 2, 3)
    rows = ops.stem_h2_operands(w, 1.0, RESNET_MEAN, RESNET_STD, pack=False)          # [64, 7 * 8 * 4]
    w4 = rows.double().view(64, 7, 8, 4)[:, :, :7, :].permute(0, 3, 1, 2)             # [64, 4, 7, 7]
    v4 = torch.cat([v, torch.ones(2, 1, 37, 41, dtype=torch.float64)], 1)             # the 4th channel: inside the image
    got = F.conv2d(v4, w4, None, 2, 3)
    assert rows.view(64, 7, 8, 4)[:, :, 7].abs().max() == 0                            # the 8th pixel of a kernel row is padding
    assert (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()            # (the folded rows are fp32)
