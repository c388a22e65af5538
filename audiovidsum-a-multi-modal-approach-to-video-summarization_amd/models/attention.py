"""MultiHeadSelfAttention — drop-in for the reference's models/attention.py:5-25.

Same constructor, same sub-module names (query, key, value, out), batch-first
[B, T, E] -> [B, T, E], no mask, no dropout.  ``forward`` runs on the MI355X:

    Q, K projections          -> avs_gemm_nt                         attention.py:17-18
    V projection, TRANSPOSED  -> avs_gemm_nt with A = W_v, B = x     attention.py:19
                                 (V^T[e, t], so the second einsum is an NT GEMM too)
    head dim 64 / 128 / 256:  -> avs_f16x2_pack_f32 + avs_mhsa_flash_f16x2: QK^T, online softmax (one wave shuffle
                                 per tile) and PV fused on the fp16 matrix cores (hi*hi + lo*hi + hi*lo), the [T,T]
                                 scores never reach memory (avs_mhsa_flash_f32 = the exact-fp32 form) attention.py:21-24
    other head dims:
    scores = Q.K^T / sqrt(d)  -> avs_gemm_nt, batch = heads          attention.py:21
    softmax over keys         -> avs_softmax_rows_f32 (wave shuffle)  attention.py:22
    context = P.V             -> avs_gemm_nt, batch = heads          attention.py:23-24
    out projection            -> avs_gemm_nt                         attention.py:25

The key axis is padded to a multiple of 4 columns (16-byte rows) with zeros;
the softmax only touches the real T columns.
"""
import math

import torch
import torch.nn as nn

from .. import ops
from .._abi import BIAS_NONE, BIAS_ROW


class MultiHeadSelfAttention(nn.Module):
    def __init__(self, embed_dim, num_heads):
        super().__init__()
        self.query = nn.Linear(embed_dim, embed_dim)
        self.key = nn.Linear(embed_dim, embed_dim)
        self.value = nn.Linear(embed_dim, embed_dim)
        self.out = nn.Linear(embed_dim, embed_dim)
        self.num_heads = num_heads
        self.dim_head = embed_dim // num_heads
        # "auto" / True / "f32" / False.  The fused kernels (head dim 64, 128 or 256) never materialise the [T,T] scores:
        # "auto" / True = the split-precision form on the fp16 matrix cores (avs_mhsa_flash_f16x2, 16-query tiles), "f32" =
        # the exact fp32 MFMA form (avs_mhsa_flash_f32), False = the batched-GEMM path, which DOES materialise [H,T,T]
        # (400 MB at T = 5000).  Measured on MI355X, E = 1024, H = 4, whole forward, T = 300 / 1800 / 5000: split fused
        # 0.375 / 0.63 / 1.40 ms, fp32 fused 0.49 / 1.11 / 2.69 ms, batched GEMMs 0.376 / 0.60 / 2.30 ms.
        self.use_flash = "auto"

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("MultiHeadSelfAttention runs on the MI355X HIP path only: move inputs with .cuda()")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()) and x.requires_grad:
            raise NotImplementedError("MultiHeadSelfAttention HIP path: autograd is not implemented yet")
        b, t, e = x.shape
        h, d = self.num_heads, self.dim_head
        if h * d != e:
            # the reference's .view(B, T, H, d) raises for this too (attention.py:17)
            raise RuntimeError(f"shape '[{b}, {t}, {h}, {d}]' is invalid for input of size {b * t * e}")
        if d % 4 != 0:
            raise ValueError("HIP path needs head dim to be a multiple of 4 (16-byte rows)")
        x2 = x.reshape(b * t, e).float().contiguous()
        dev = x2.device
        q = ops.linear(x2, self.query.weight, self.query.bias)
        k = ops.linear(x2, self.key.weight, self.key.bias)
        flash = self.use_flash is not False
        if d in (64, 128, 256) and flash:
            # fused core: scores never materialised
            v = ops.linear(x2, self.value.weight, self.value.bias)
            ctx = ops.mhsa_flash(q, k, v, b, t, h, split=self.use_flash != "f32")
            return ops.linear(ctx, self.out.weight, self.out.bias).view(b, t, e)
        tp = (t + 3) // 4 * 4
        vt = torch.zeros((b, e, tp), dtype=torch.float32, device=dev)
        # V^T[b] [E, T] = W_v [E,E] . x[b] [T,E]^T + bias per row
        ops.gemm_nt_batched(ops.dtype_code(torch.float32), e, t, e, self.value.weight, 0, e, 0, x2, 0, e, t * e, vt, 0,
                            tp, e * tp, self.value.bias, BIAS_ROW, 0, 1.0, ops.ACT_NONE, b)
        ctx = torch.empty((b * t, e), dtype=torch.float32, device=dev)
        alpha = 1.0 / math.sqrt(d)
        f32 = ops.dtype_code(torch.float32)
        for bi in range(b):
            scores = torch.zeros((h, t, tp), dtype=torch.float32, device=dev)
            ops.gemm_nt_batched(f32, t, t, d, q, bi * t * e, e, d, k, bi * t * e, e, d, scores, 0, tp, t * tp, None,
                                BIAS_NONE, 0, alpha, ops.ACT_NONE, h)
            ops.softmax_rows(scores, h * t, t, tp)
            ops.gemm_nt_batched(f32, t, d, tp, scores, 0, tp, t * tp, vt, bi * e * tp, tp, d * tp, ctx, bi * t * e, e, d,
                                None, BIAS_NONE, 0, 1.0, ops.ACT_NONE, h)
        out = ops.linear(ctx, self.out.weight, self.out.bias)
        return out.view(b, t, e)
