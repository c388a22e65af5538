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

Autograd (the reference class is an ordinary autograd module, attention.py:5-25; nothing in the reference trains it,
SURVEY Q11): with gradients enabled ``forward`` goes through ``_MHSAFunction`` - the same forward kernels, and a
backward made of C-ABI calls: the probabilities are RECOMPUTED per batch element (avs_gemm_nt + avs_softmax_rows_f32),
dP = dctx.V^T, dS = alpha P (dP - rowsum(P dP)) (avs_softmax_bwd_rows_f32), dQ = dS.K, dK = dS^T.Q, dV = P^T.dctx as
NT GEMMs on transposed copies (avs_transpose_f32), the projection gradients as in the scorer's backward
(ops.grad_weight / grad_input / colsum).  The backward materialises [H, T, T] (it is a training path: T is a clip).
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
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return _MHSAFunction.apply(self, x, self.query.weight, self.query.bias, self.key.weight, self.key.bias,
                                       self.value.weight, self.value.bias, self.out.weight, self.out.bias)
        return self._forward(x)[0]

    def _forward(self, x, keep=False):
        """-> (out [B,T,E], (x2, q, k, ctx) when ``keep``)."""
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
            return ops.linear(ctx, self.out.weight, self.out.bias).view(b, t, e), ((x2, q, k, ctx) if keep else None)
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
        return out.view(b, t, e), ((x2, q, k, ctx) if keep else None)


class _MHSAFunction(torch.autograd.Function):
    """forward = MultiHeadSelfAttention._forward; backward = C-ABI calls only (module docstring)."""

    @staticmethod
    def forward(ctx, mod, x, wq, bq, wk, bk, wv, bv, wo, bo):
        out, (x2, q, k, att) = mod._forward(x.detach(), keep=True)
        ctx.mod_dims = (x.shape, mod.num_heads, mod.dim_head, x.dtype)
        ctx.save_for_backward(x2, q, k, att, wq, wk, wv, bv, wo)
        return out.to(x.dtype)

    @staticmethod
    def backward(ctx, dout):
        x2, q, k, att, wq, wk, wv, bv, wo = ctx.saved_tensors
        (b, t, e), h, d, xdtype = ctx.mod_dims
        dev = x2.device
        f32 = ops.dtype_code(torch.float32)
        alpha = 1.0 / math.sqrt(d)
        tp = (t + 3) // 4 * 4
        dout2 = dout.reshape(b * t, e).float().contiguous()
        v = ops.linear(x2, wv, bv)
        g_wo = ops.grad_weight(dout2, att)
        g_bo = ops.colsum(dout2)
        datt = ops.grad_input(dout2, wo)                              # [B*T, E], heads side by side
        dq = torch.empty((b * t, e), dtype=torch.float32, device=dev)
        dk = torch.empty_like(dq)
        dv = torch.empty_like(dq)
        for bi in range(b):
            o = bi * t * e
            # P = softmax(alpha Q K^T), recomputed                                    attention.py:21-22
            prob = torch.zeros((h, t, tp), dtype=torch.float32, device=dev)
            ops.gemm_nt_batched(f32, t, t, d, q, o, e, d, k, o, e, d, prob, 0, tp, t * tp, None, BIAS_NONE, 0, alpha,
                                ops.ACT_NONE, h)
            ops.softmax_rows(prob, h * t, t, tp)
            # dP = dctx . V^T                                                          attention.py:23
            dp = torch.zeros((h, t, tp), dtype=torch.float32, device=dev)
            ops.gemm_nt_batched(f32, t, t, d, datt, o, e, d, v, o, e, d, dp, 0, tp, t * tp, None, BIAS_NONE, 0, 1.0,
                                ops.ACT_NONE, h)
            # transposed copies [E, T] of this batch element's dctx, K, Q (pad columns zero) and P^T per head
            dattT = torch.zeros((e, tp), dtype=torch.float32, device=dev)
            kT = torch.zeros((e, tp), dtype=torch.float32, device=dev)
            qT = torch.zeros((e, tp), dtype=torch.float32, device=dev)
            ops.transpose_into(datt, o, t, e, e, dattT, 0, tp)
            ops.transpose_into(k, o, t, e, e, kT, 0, tp)
            ops.transpose_into(q, o, t, e, e, qT, 0, tp)
            probT = torch.zeros((h, t, tp), dtype=torch.float32, device=dev)
            for hh in range(h):
                ops.transpose_into(prob, hh * t * tp, t, t, tp, probT, hh * t * tp, tp)
            # dV[h] = P[h]^T . dctx[h]
            ops.gemm_nt_batched(f32, t, d, tp, probT, 0, tp, t * tp, dattT, 0, tp, d * tp, dv, o, e, d, None, BIAS_NONE,
                                0, 1.0, ops.ACT_NONE, h)
            # dS = alpha P (dP - rowsum(P dP)), in place on dP
            ops.softmax_bwd_rows(prob, dp, h * t, t, tp, alpha)
            # dQ[h] = dS[h] . K[h];  dK[h] = dS[h]^T . Q[h]                            attention.py:21
            ops.gemm_nt_batched(f32, t, d, tp, dp, 0, tp, t * tp, kT, 0, tp, d * tp, dq, o, e, d, None, BIAS_NONE, 0,
                                1.0, ops.ACT_NONE, h)
            for hh in range(h):   # probT's storage is free again: dS^T
                ops.transpose_into(dp, hh * t * tp, t, t, tp, probT, hh * t * tp, tp)
            ops.gemm_nt_batched(f32, t, d, tp, probT, 0, tp, t * tp, qT, 0, tp, d * tp, dk, o, e, d, None, BIAS_NONE, 0,
                                1.0, ops.ACT_NONE, h)
        need = ctx.needs_input_grad
        dx = None
        if need[1]:
            dx = (ops.grad_input(dq, wq) + ops.grad_input(dk, wk) + ops.grad_input(dv, wv)).view(b, t, e).to(xdtype)
        return (None, dx, ops.grad_weight(dq, x2), ops.colsum(dq), ops.grad_weight(dk, x2), ops.colsum(dk),
                ops.grad_weight(dv, x2), ops.colsum(dv), g_wo, g_bo)
