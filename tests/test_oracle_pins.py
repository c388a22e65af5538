"""Pins the oracle (CPU restatement) against (a) the golden fixtures generated from the reference's own
classes by tests/golden/make_golden.py and (b), where /root/reference exists, those classes directly."""
import hashlib
import os
import sys
import types

import numpy as np
import pytest
import torch

from oracle import fusion as ofu, scorer as osc, selection as osel

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF = "/root/reference"


def _sd(z, prefix="sd_"):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


def test_scorer_small_golden():
    z = np.load(os.path.join(GOLD, "scorer_small.npz"))
    sd = _sd(z)
    assert len(sd) == 28
    for name in ("b1", "b3", "t1"):
        out = osc.av_bilstm_forward(sd, torch.from_numpy(z[name + "_visual"]), torch.from_numpy(z[name + "_audio"]))
        ref = torch.from_numpy(z[name + "_out"])
        assert out.shape == ref.shape  # squeeze rules: [37], [3,11], []
        assert (out - ref).abs().max().item() < 1e-6


def _hash(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().numpy().tobytes())
    return h.hexdigest()


def test_scorer_full_golden_and_construction_order():
    """The mirror class built under the same seed must reproduce the reference's parameters bit for bit
    (same sub-modules, same construction order) and the same 28 state-dict keys."""
    from avsum_amd.models.av_model import AVBiLSTMModel
    z = np.load(os.path.join(GOLD, "scorer_full.npz"))
    torch.manual_seed(int(z["seed"]))
    m = AVBiLSTMModel().eval()
    sd = m.state_dict()
    assert sorted(sd.keys()) == list(z["keys"])
    assert sum(p.numel() for p in m.parameters()) == int(z["n_params"]) == 9667713
    assert _hash(sd) == str(z["sd_sha256"])
    v = torch.randn(1, 300, 4096, generator=torch.Generator().manual_seed(int(z["input_seed"])))
    out = osc.av_bilstm_forward(sd, v, torch.zeros(1, 300, 296))
    assert (out - torch.from_numpy(z["out"])).abs().max().item() < 1e-6


@pytest.mark.parametrize("t_len", [1800, 5000])
def test_scorer_long_golden(t_len):
    """configs[1] / configs[3] sequence lengths: the oracle's recurrences over 1800 and 5000 steps against the scores the
    reference class (models/av_model.py:6-46) produced for the same seeded parameters and input."""
    from avsum_amd.models.av_model import AVBiLSTMModel
    z = np.load(os.path.join(GOLD, "scorer_long.npz"))
    torch.manual_seed(int(z["seed"]))
    sd = AVBiLSTMModel().eval().state_dict()
    assert _hash(sd) == str(z["sd_sha256"])
    v = torch.randn(1, t_len, 4096, generator=torch.Generator().manual_seed(int(z[f"t{t_len}_input_seed"])))
    out = osc.av_bilstm_forward(sd, v, torch.zeros(1, t_len, 296))
    ref = torch.from_numpy(z[f"t{t_len}_out"])
    assert out.shape == ref.shape == (t_len,)
    assert (out - ref).abs().max().item() < 1e-6


def test_mhsa_long_golden():
    """MultiHeadSelfAttention(1024, 4) at T = 5000 (models/attention.py:15-25): 64 stored output rows of the reference
    class + the absolute sum of the whole output."""
    from avsum_amd.models.attention import MultiHeadSelfAttention
    z = np.load(os.path.join(GOLD, "mhsa_long.npz"))
    torch.manual_seed(int(z["seed"]))
    sd = MultiHeadSelfAttention(1024, 4).eval().state_dict()
    assert _hash(sd) == str(z["sd_sha256"])
    x = torch.randn(1, int(z["t"]), 1024, generator=torch.Generator().manual_seed(int(z["input_seed"])))
    out = osc.mhsa_forward(sd, x, 4)
    assert (out[0, z["rows"]] - torch.from_numpy(z["out_rows"])).abs().max().item() < 1e-6
    assert abs(float(out.double().abs().sum()) / float(z["out_abs_sum"]) - 1) < 1e-6


def test_mhsa_golden_and_keys():
    from avsum_amd.models.attention import MultiHeadSelfAttention
    z = np.load(os.path.join(GOLD, "mhsa_small.npz"))
    sd = _sd(z)
    torch.manual_seed(99)
    assert sorted(MultiHeadSelfAttention(64, 4).state_dict().keys()) == sorted(sd.keys())
    out = osc.mhsa_forward(sd, torch.from_numpy(z["x"]), 4)
    assert (out - torch.from_numpy(z["out"])).abs().max().item() < 1e-6


def test_fusion_metrics_golden():
    z = np.load(os.path.join(GOLD, "fusion_metrics.npz"))
    v, a = torch.from_numpy(z["v"]), torch.from_numpy(z["a"])
    cost = ofu.compute_dtw(v, a)
    assert cost.dtype == np.float64 and np.array_equal(cost, z["cost"])
    interp = ofu.interpolate_features(v, z["path"], 3)
    assert str(interp.dtype) == str(z["interp_dtype"]) and np.array_equal(interp.numpy(), z["interp"])
    al = osel.align_shots([(0, 50), (60, 200), (10, 11)], z["ann"], 30)
    assert np.array_equal(al.numpy(), z["align"])
    assert osel.compute_temporal_f1([(0, 10), (20, 30)], [(5, 15), (20, 25)], 30) == float(z["tf1"])
    assert abs(float(z["tf1"]) - 0.5714285665) < 1e-9  # SURVEY A.10 probe


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "models")), reason="reference tree not present")
def test_against_reference_classes_directly():
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k.split(".")[0] in
             ("models", "features", "utils", "evaluation", "scripts")}
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    try:
        sys.modules.setdefault("fastdtw", types.SimpleNamespace(fastdtw=None))
        from models.av_model import AVBiLSTMModel
        from models.attention import MultiHeadSelfAttention
        from features import fusion as rfu
        from utils.shot_metrics import compute_f1
        from evaluation.metrics import compute_temporal_f1
        g = torch.Generator().manual_seed(77)
        torch.manual_seed(3)
        m = AVBiLSTMModel(32, 12, 16).eval()
        for b, t in ((1, 50), (4, 9), (2, 1)):
            v, a = torch.randn(b, t, 32, generator=g), torch.randn(b, t, 12, generator=g)
            with torch.no_grad():
                ref = m(v, a)
            out = osc.av_bilstm_forward(m.state_dict(), v, a)
            assert out.shape == ref.shape and (out - ref).abs().max().item() < 1e-6
        # backward: torch autograd through the restatement == autograd through the reference module
        m.zero_grad()
        vg, ag = torch.randn(1, 13, 32, generator=g), torch.randn(1, 13, 12, generator=g)
        m(vg, ag).sum().backward()
        sd = {k: p.detach().clone().requires_grad_(True) for k, p in m.named_parameters()}
        ones = torch.ones(13, 16)
        grads = torch.autograd.grad(osc.av_bilstm_forward_train(sd, vg, ag, ones, ones).sum(), list(sd.values()),
                                    allow_unused=True)
        for (k, p), gr in zip(m.named_parameters(), grads):
            want = p.grad if p.grad is not None else torch.zeros_like(p)
            got = gr if gr is not None else torch.zeros_like(p)
            assert (got - want).abs().max().item() < 1e-6, k
        at = MultiHeadSelfAttention(48, 6).eval()
        x = torch.randn(3, 21, 48, generator=g)
        with torch.no_grad():
            assert (osc.mhsa_forward(at.state_dict(), x, 6) - at(x)).abs().max().item() < 1e-6
        vv, aa = torch.randn(15, 8, generator=g), torch.randn(11, 8, generator=g)
        assert np.array_equal(ofu.compute_dtw(vv, aa), rfu.compute_dtw(vv, aa))
        _, path = ofu.dtw_path(ofu.compute_dtw(vv, aa))
        assert torch.equal(ofu.interpolate_features(vv, path, 9), rfu.interpolate_features(vv, path, 9))
        segs = ([(0, 10), (20, 30)], [(5, 15), (20, 25)])
        assert osel.compute_temporal_f1(*segs, 30) == compute_temporal_f1(*segs, 30) == compute_f1(*segs, 30)
        # the metric block of scripts/evaluate.py:21-42 (its model loop needs .cuda(); restated lines only)
        rng = np.random.default_rng(0)
        preds = [rng.random(40).astype(np.float32) for _ in range(3)]
        tgts = [rng.random(40).astype(np.float32) for _ in range(3)]
        from scipy.stats import kendalltau, spearmanr
        f1 = []
        for p, t in zip(preds, tgts):
            bp, bt = (p > np.mean(p)).astype(int), (t > np.mean(t)).astype(int)
            tp = np.logical_and(bp, bt).sum()
            pr, rc = tp / bp.sum(), tp / bt.sum()
            f1.append(2 * (pr * rc) / (pr + rc + 1e-8))
        got = osel.evaluate_metrics(preds, tgts)
        assert got["f1"] == np.mean(f1)
        assert got["spearman"] == np.mean([spearmanr(p, t).correlation for p, t in zip(preds, tgts)])
        assert got["kendall"] == np.mean([kendalltau(p, t).correlation for p, t in zip(preds, tgts)])
    finally:
        sys.path.remove(REF)
        for k in list(sys.modules):
            if k.split(".")[0] in ("models", "features", "utils", "evaluation", "scripts"):
                del sys.modules[k]
        sys.modules.update(saved)


def test_dtw_path_is_optimal_bruteforce():
    """Pins the exact-DTW restatement (fastdtw itself is absent): on small matrices the returned path must be
    a valid warping path whose cost equals the minimum over ALL warping paths, and ties follow up/left/diag."""
    rng = np.random.default_rng(5)

    def all_paths(n, m):
        def rec(i, j):
            if i == n - 1 and j == m - 1:
                yield [(i, j)]
                return
            for di, dj in ((1, 0), (0, 1), (1, 1)):
                if i + di < n and j + dj < m:
                    for rest in rec(i + di, j + dj):
                        yield [(i, j)] + rest
        return rec(0, 0)

    for n, m in ((1, 1), (1, 5), (4, 1), (4, 5), (5, 4)):
        c = rng.random((n, m))
        total, path = ofu.dtw_path(c)
        best = min(sum(c[i, j] for i, j in p) for p in all_paths(n, m))
        assert abs(total - best) < 1e-12
        assert abs(sum(c[i, j] for i, j in path) - total) < 1e-12
        assert tuple(path[0]) == (0, 0) and tuple(path[-1]) == (n - 1, m - 1)
        d = np.diff(path, axis=0)
        assert ((d >= 0) & (d <= 1)).all() and (d.sum(1) >= 1).all()
    # all-zero costs: every predecessor ties; "up" (i-1, j) is preferred at every cell, so walking back from
    # (2,2) climbs column 2 first: the path runs along row 0, then down the last column
    _, p = ofu.dtw_path(np.zeros((3, 3)))
    assert p.tolist() == [[0, 0], [0, 1], [0, 2], [1, 2], [2, 2]]
