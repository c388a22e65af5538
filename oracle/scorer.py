"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the importance scorer.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product path (avsum_amd) never does.

Restates, op by op on torch CPU tensors (fp32), what the reference executes:

  * av_bilstm_forward      models/av_model.py:33-46   (AVBiLSTMModel.forward)
  * lstm_direction         nn.LSTM as used at models/av_model.py:18-23,39-40
                           (gate order i,f,g,o; h0=c0=0)
  * mha_seq_first          nn.MultiheadAttention(1024,4) fed [B,T,E] WITHOUT
                           batch_first, models/av_model.py:26,44:
                           softmax over the B axis, per time-step and head
  * mhsa_forward           models/attention.py:15-25 (MultiHeadSelfAttention)

Pinned (tests/test_oracle_pins.py, run where /root/reference exists) against
the reference's own classes imported from /root/reference/models, and against
the fixtures under tests/golden/ generated from those classes by
tests/golden/make_golden.py.
"""
import math

import torch


def linear(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


def lstm_direction(x, w_ih, w_hh, b_ih, b_hh, reverse=False):
    """x [T, I] -> h [T, H]; one direction of one sequence."""
    t_len = x.shape[0]
    hid = w_hh.shape[1]
    h = torch.zeros(hid, dtype=x.dtype)
    c = torch.zeros(hid, dtype=x.dtype)
    outs = [None] * t_len
    xp = x @ w_ih.t() + b_ih
    order = range(t_len - 1, -1, -1) if reverse else range(t_len)
    for t in order:
        g = xp[t] + (w_hh @ h + b_hh)
        i = torch.sigmoid(g[0:hid])
        f = torch.sigmoid(g[hid:2 * hid])
        gg = torch.tanh(g[2 * hid:3 * hid])
        o = torch.sigmoid(g[3 * hid:4 * hid])
        c = f * c + i * gg
        h = o * torch.tanh(c)
        outs[t] = h
    return torch.stack(outs) if t_len else torch.zeros((0, hid), dtype=x.dtype)


def bilstm(x, sd, prefix):
    """x [B, T, I] -> [B, T, 2H] (batch_first, bidirectional, one layer)."""
    outs = []
    for b in range(x.shape[0]):
        fwd = lstm_direction(x[b], sd[prefix + "weight_ih_l0"], sd[prefix + "weight_hh_l0"],
                             sd[prefix + "bias_ih_l0"], sd[prefix + "bias_hh_l0"], False)
        rev = lstm_direction(x[b], sd[prefix + "weight_ih_l0_reverse"], sd[prefix + "weight_hh_l0_reverse"],
                             sd[prefix + "bias_ih_l0_reverse"], sd[prefix + "bias_hh_l0_reverse"], True)
        outs.append(torch.cat([fwd, rev], -1))
    return torch.stack(outs)


def mha_seq_first(x, in_w, in_b, out_w, out_b, heads):
    """x [L, N, E] in nn.MultiheadAttention's default (seq-first) convention: attention over L."""
    l, n, e = x.shape
    d = e // heads
    q = linear(x, in_w[0:e], in_b[0:e]).reshape(l, n, heads, d)
    k = linear(x, in_w[e:2 * e], in_b[e:2 * e]).reshape(l, n, heads, d)
    v = linear(x, in_w[2 * e:3 * e], in_b[2 * e:3 * e]).reshape(l, n, heads, d)
    q = q * (1.0 / math.sqrt(d))
    s = torch.einsum("lnhd,mnhd->nhlm", q, k)
    a = torch.softmax(s, dim=-1)
    ctx = torch.einsum("nhlm,mnhd->lnhd", a, v).reshape(l, n, e)
    return linear(ctx, out_w, out_b)


def av_bilstm_forward(sd, visual, audio, heads=4):
    """sd: state_dict of AVBiLSTMModel (eval mode: Dropout = identity).  visual [B,T,Dv], audio [B,T,Da]."""
    v_emb = torch.relu(linear(visual, sd["visual_fc.0.weight"], sd["visual_fc.0.bias"]))
    a_emb = torch.relu(linear(audio, sd["audio_fc.0.weight"], sd["audio_fc.0.bias"]))
    v_out = bilstm(v_emb, sd, "visual_bilstm.")
    a_out = bilstm(a_emb, sd, "audio_bilstm.")
    fused = torch.cat([v_out, a_out], -1)
    # fed as (L=B, N=T, E): the module attends across the batch axis
    attn = mha_seq_first(fused, sd["attention.in_proj_weight"], sd["attention.in_proj_bias"],
                         sd["attention.out_proj.weight"], sd["attention.out_proj.bias"], heads)
    hid = torch.relu(linear(attn, sd["scorer.0.weight"], sd["scorer.0.bias"]))
    return torch.sigmoid(linear(hid, sd["scorer.2.weight"], sd["scorer.2.bias"])).squeeze()


def av_bilstm_forward_train(sd, visual, audio, keep_v, keep_a, heads=4):
    """Training-mode forward (models/av_model.py:33-46 with Dropout(0.3) active, as scripts/train_av_model.py:71
    runs it): identical to av_bilstm_forward except that the two embeddings are multiplied by the inverted-
    dropout masks keep_* (0 or 1/(1-p)); the masks are inputs so that both sides of a test use the same draw.
    Differentiable with torch autograd (the oracle for the backward kernels)."""
    v_emb = torch.relu(linear(visual, sd["visual_fc.0.weight"], sd["visual_fc.0.bias"])) * keep_v
    a_emb = torch.relu(linear(audio, sd["audio_fc.0.weight"], sd["audio_fc.0.bias"])) * keep_a
    v_out = bilstm(v_emb, sd, "visual_bilstm.")
    a_out = bilstm(a_emb, sd, "audio_bilstm.")
    fused = torch.cat([v_out, a_out], -1)
    attn = mha_seq_first(fused, sd["attention.in_proj_weight"], sd["attention.in_proj_bias"],
                         sd["attention.out_proj.weight"], sd["attention.out_proj.bias"], heads)
    hid = torch.relu(linear(attn, sd["scorer.0.weight"], sd["scorer.0.bias"]))
    return torch.sigmoid(linear(hid, sd["scorer.2.weight"], sd["scorer.2.bias"])).squeeze()


def mhsa_forward(sd, x, heads):
    """models/attention.py:15-25."""
    b, t, e = x.shape
    d = e // heads
    q = linear(x, sd["query.weight"], sd["query.bias"]).view(b, t, heads, d)
    k = linear(x, sd["key.weight"], sd["key.bias"]).view(b, t, heads, d)
    v = linear(x, sd["value.weight"], sd["value.bias"]).view(b, t, heads, d)
    scores = torch.einsum("bqhd,bkhd->bhqk", q, k) / (d ** 0.5)
    attn = torch.softmax(scores, dim=-1)
    ctx = torch.einsum("bhqk,bkhd->bqhd", attn, v).reshape(b, t, -1)
    return linear(ctx, sd["out.weight"], sd["out.bias"])
