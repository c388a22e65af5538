"""Training-mode forward + backward of AVBiLSTMModel on the MI355X (scripts/train_av_model.py:86-96).

A torch.autograd.Function whose forward and backward are sequences of libavsum_hip.so calls: torch only owns
the tensors, the RNG that draws the Dropout masks, and (in the caller's script) the loss and AdamW.  One
sequence per call (the reference trains with B = 1, train_av_model.py:64,86-88); with B = 1 the attention is
out_proj(v_proj(x)) and the query/key projections receive exactly zero gradient.
"""
import torch

from .. import ops

DROPOUT_P = 0.3  # models/av_model.py:11,14


class ScorerTrainFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, visual, audio, keep_v, keep_a, *params):
        names = [n for n, _ in model.named_parameters()]
        pm = dict(zip(names, params))
        dev = visual.device
        t = visual.shape[0]
        seq = torch.tensor([0, t], dtype=torch.int64, device=dev)
        hidden = pm["visual_fc.0.weight"].shape[0]
        e = 2 * hidden

        # both BiLSTMs = four independent recurrences: ONE launch (direction d writes fused[:, d*H:]), as in score_rows
        saved = {}
        fused = torch.empty((t, e), dtype=torch.float32, device=dev)
        hid = pm["visual_bilstm.weight_hh_l0"].shape[1]
        xproj = torch.empty((t, 16 * hid), dtype=torch.float32, device=dev)   # [v fwd | v rev | a fwd | a rev] x 4H
        whh_all = []
        col = 0
        for tag, x, keep, fc, lstm in (("v", visual, keep_v, "visual_fc.0.", "visual_bilstm."),
                                       ("a", audio, keep_a, "audio_fc.0.", "audio_bilstm.")):
            r = ops.linear(x, pm[fc + "weight"], pm[fc + "bias"], ops.ACT_RELU)     # Linear + ReLU
            emb = ops.mul(r, keep)                                                   # Dropout (inverted scaling)
            wih = torch.cat([pm[lstm + "weight_ih_l0"], pm[lstm + "weight_ih_l0_reverse"]], 0)
            bih = torch.cat([pm[lstm + "bias_ih_l0"] + pm[lstm + "bias_hh_l0"],
                             pm[lstm + "bias_ih_l0_reverse"] + pm[lstm + "bias_hh_l0_reverse"]])
            whh_all += [pm[lstm + "weight_hh_l0"], pm[lstm + "weight_hh_l0_reverse"]]
            ops.linear(emb, wih, bih, out=xproj[:, col * 4:(col + 2 * hid) * 4])
            saved[tag] = (x, r, keep, emb, wih, col)
            col += 2 * hid
        whh = torch.stack(whh_all).contiguous()                  # [4, 4H, H]: the backward's layout
        whh_t = whh.transpose(1, 2).contiguous()                 # [4, H, 4H]: the forward's
        gates, cell = ops.lstm_train_fwd(xproj, whh_t, hid, 4, 0b1010, seq, fused, 0)
        ctx.lstm = (whh, gates, cell, hid)
        w_in, b_in = pm["attention.in_proj_weight"], pm["attention.in_proj_bias"]
        w_v, b_v = w_in[2 * e:3 * e].contiguous(), b_in[2 * e:3 * e].contiguous()
        w_o, b_o = pm["attention.out_proj.weight"], pm["attention.out_proj.bias"]
        val = ops.linear(fused, w_v, b_v)
        attn = ops.linear(val, w_o, b_o)
        hid64 = ops.linear(attn, pm["scorer.0.weight"], pm["scorer.0.bias"], ops.ACT_RELU)
        scores = ops.score_head(hid64, pm["scorer.2.weight"].reshape(-1).contiguous(), pm["scorer.2.bias"])

        ctx.names, ctx.saved, ctx.seq, ctx.e = names, saved, seq, e
        ctx.tail = (fused, val, attn, hid64, scores, w_v, w_o)
        ctx.pm = pm
        ctx.need_inputs = (ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return scores

    @staticmethod
    def backward(ctx, dscores):
        pm, e = ctx.pm, ctx.e
        fused, val, attn, hid64, scores, w_v, w_o = ctx.tail
        g = {}
        dscores = dscores.contiguous().float()
        # scoring head: Sigmoid, Linear(64,1), ReLU of scorer.0
        w2 = pm["scorer.2.weight"].reshape(-1).contiguous()
        dz, dpre = ops.score_head_bwd(dscores, scores, hid64, w2)
        g["scorer.2.weight"] = ops.colsum(hid64, dz).view(1, -1)
        g["scorer.2.bias"] = ops.colsum(dz.view(-1, 1))
        g["scorer.0.weight"] = ops.grad_weight(dpre, attn)
        g["scorer.0.bias"] = ops.colsum(dpre)
        dattn = ops.grad_input(dpre, pm["scorer.0.weight"])
        # attention with a single key: out_proj(v_proj(fused))
        g["attention.out_proj.weight"] = ops.grad_weight(dattn, val)
        g["attention.out_proj.bias"] = ops.colsum(dattn)
        dval = ops.grad_input(dattn, w_o)
        gw_in = torch.zeros_like(pm["attention.in_proj_weight"])
        gb_in = torch.zeros_like(pm["attention.in_proj_bias"])
        gw_in[2 * e:3 * e] = ops.grad_weight(dval, fused)
        gb_in[2 * e:3 * e] = ops.colsum(dval)
        g["attention.in_proj_weight"], g["attention.in_proj_bias"] = gw_in, gb_in
        dfused = ops.grad_input(dval, w_v)
        dinputs = {}
        whh, gates, cell, hid = ctx.lstm
        t = fused.shape[0]
        da_all = ops.lstm_bwd(dfused, 0, gates, cell, whh, hid, 4, 0b1010, ctx.seq)   # [T, 4 * 4H], one launch
        for tag, fc, lstm in (("v", "visual_fc.0.", "visual_bilstm."), ("a", "audio_fc.0.", "audio_bilstm.")):
            x, r, keep, emb, wih, col = ctx.saved[tag]
            da = da_all[:, col * 4:(col + 2 * hid) * 4]                                  # this BiLSTM's [T, 2*4H]
            gwih = ops.grad_weight(da, emb)                                             # [8H, hidden]
            gb = ops.colsum(da)
            # h_{t-1} of each direction: forward = the previous row, reverse = the next row (zero at the start)
            hprev = torch.zeros((t, 2 * hid), dtype=torch.float32, device=x.device)
            if t > 1:
                hprev[1:, :hid] = fused[:-1, col:col + hid]
                hprev[:-1, hid:] = fused[1:, col + hid:col + 2 * hid]
            g[lstm + "weight_ih_l0"], g[lstm + "weight_ih_l0_reverse"] = gwih[:4 * hid], gwih[4 * hid:]
            g[lstm + "bias_ih_l0"] = g[lstm + "bias_hh_l0"] = gb[:4 * hid]
            g[lstm + "bias_ih_l0_reverse"] = g[lstm + "bias_hh_l0_reverse"] = gb[4 * hid:]
            g[lstm + "weight_hh_l0"] = ops.grad_weight(da[:, :4 * hid], hprev[:, :hid])
            g[lstm + "weight_hh_l0_reverse"] = ops.grad_weight(da[:, 4 * hid:], hprev[:, hid:])
            demb = ops.grad_input(da, wih)
            dr = ops.relu_dropout_bwd(demb, r, keep)
            g[fc + "weight"] = ops.grad_weight(dr, x)
            g[fc + "bias"] = ops.colsum(dr)
            dinputs[tag] = dr
        dvis = ops.grad_input(dinputs["v"], pm["visual_fc.0.weight"]) if ctx.need_inputs[0] else None
        daud = ops.grad_input(dinputs["a"], pm["audio_fc.0.weight"]) if ctx.need_inputs[1] else None
        return (None, dvis, daud, None, None) + tuple(g[n].reshape(pm[n].shape) for n in ctx.names)


def dropout_keep(shape, device, p=DROPOUT_P):
    """The multiplier torch's Dropout applies in training mode: 0 with probability p, else 1/(1-p)."""
    return torch.nn.functional.dropout(torch.ones(shape, dtype=torch.float32, device=device), p, True)
