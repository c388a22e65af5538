#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ by IMPORTING the reference's own Python
(/root/reference/models, features/fusion.py with a stub `fastdtw` module, utils, evaluation,
scripts/evaluate.py) in this container and recording inputs + outputs.  Only data is written:
no reference source text or bytecode.  Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.dont_write_bytecode = True


def ref_modules():
    sys.path.insert(0, REF)
    sys.modules.setdefault("fastdtw", types.SimpleNamespace(fastdtw=None))  # fusion.py imports it at module level
    from models.av_model import AVBiLSTMModel
    from models.attention import MultiHeadSelfAttention
    from features import fusion
    from utils.alignments import align_shots_to_annotations
    from evaluation.metrics import compute_temporal_f1
    sys.path.pop(0)
    return AVBiLSTMModel, MultiHeadSelfAttention, fusion, align_shots_to_annotations, compute_temporal_f1


def sd_hash(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().numpy().tobytes())
    return h.hexdigest()


def main():
    AVBiLSTMModel, MHSA, fusion, align, tf1 = ref_modules()
    torch.manual_seed(1234)
    m = AVBiLSTMModel(64, 24, 32).eval()
    g = torch.Generator().manual_seed(5)
    out = {"sd_" + k: v.numpy() for k, v in m.state_dict().items()}
    for name, (b, t) in {"b1": (1, 37), "b3": (3, 11), "t1": (1, 1)}.items():
        v, a = torch.randn(b, t, 64, generator=g), torch.randn(b, t, 24, generator=g)
        with torch.no_grad():
            y = m(v, a)
        out[f"{name}_visual"], out[f"{name}_audio"], out[f"{name}_out"] = v.numpy(), a.numpy(), y.numpy()
    np.savez_compressed(os.path.join(HERE, "scorer_small.npz"), **out)

    # full dims: parameters are NOT stored; they are reproduced from the seed by constructing the model in the
    # reference's order (that the mirror class does so is part of the drop-in contract) and checked by hash.
    torch.manual_seed(7)
    mf = AVBiLSTMModel().eval()
    gv = torch.Generator().manual_seed(1007)
    v = torch.randn(1, 300, 4096, generator=gv)
    a = torch.zeros(1, 300, 296)
    with torch.no_grad():
        y = mf(v, a)
    np.savez_compressed(os.path.join(HERE, "scorer_full.npz"), seed=7, input_seed=1007,
                        sd_sha256=sd_hash(mf.state_dict()), keys=np.array(sorted(mf.state_dict().keys())),
                        out=y.numpy(), n_params=sum(p.numel() for p in mf.parameters()))

    torch.manual_seed(99)
    at = MHSA(64, 4).eval()
    x = torch.randn(2, 19, 64, generator=g)
    with torch.no_grad():
        ya = at(x)
    oa = {"sd_" + k: v.numpy() for k, v in at.state_dict().items()}
    oa.update(x=x.numpy(), out=ya.numpy())
    np.savez_compressed(os.path.join(HERE, "mhsa_small.npz"), **oa)

    vv, aa = torch.randn(23, 16, generator=g), torch.randn(31, 16, generator=g)
    cost = fusion.compute_dtw(vv, aa)
    path = np.array([[0, 0], [0, 1], [1, 2], [2, 2], [2, 3], [2, 4], [5, 5], [5, 6]])
    interp = fusion.interpolate_features(vv, path, 3)
    ann = np.arange(100, dtype=np.float64)
    al = align([(0, 50), (60, 200), (10, 11)], ann, 30)
    np.savez_compressed(os.path.join(HERE, "fusion_metrics.npz"), v=vv.numpy(), a=aa.numpy(), cost=cost, path=path,
                        interp=interp.numpy(), interp_dtype=str(interp.dtype), ann=ann, align=al.numpy(),
                        tf1=tf1([(0, 10), (20, 30)], [(5, 15), (20, 25)], 30))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
