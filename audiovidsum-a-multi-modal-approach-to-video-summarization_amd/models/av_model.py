"""AVBiLSTMModel — drop-in for the reference's models/av_model.py:6-46.

Same constructor signature, same sub-module names and construction order (so
``torch.manual_seed(s); AVBiLSTMModel()`` yields the reference's parameters and
the 28 state-dict keys are interchangeable), same output shape rules
(``.squeeze()``, av_model.py:46).  ``forward`` does not call the sub-modules:
they only hold parameters.  The computation runs on the MI355X through
libavsum_hip.so:

    visual_fc / audio_fc      -> avs_gemm_nt (+bias, ReLU)          av_model.py:35-36
    LSTM input projections    -> avs_gemm_nt, fwd|rev stacked       av_model.py:39-40
    recurrences               -> avs_lstm_f32                       av_model.py:39-40
    nn.MultiheadAttention     -> attends over dim 0 (no batch_first):
                                 B == 1: out_proj(v_proj(x)) exactly (softmax over one key);
                                 B  > 1: in_proj GEMM + avs_mha_batchaxis_f32 + out_proj
    scorer                    -> avs_gemm_nt (+ReLU) + avs_score_head_f32 (sigmoid)
"""
import torch
import torch.nn as nn

from .. import ops
from .attention import MultiHeadSelfAttention  # noqa: F401  (imported, unused — as in av_model.py:3)


def _param_key(module):
    return tuple((p.data_ptr(), p._version) for p in module.parameters())


class AVBiLSTMModel(nn.Module):
    def __init__(self, visual_dim=4096, audio_dim=296, hidden_dim=512):
        super().__init__()
        self.visual_fc = nn.Sequential(nn.Linear(visual_dim, hidden_dim), nn.ReLU(), nn.Dropout(0.3))
        self.audio_fc = nn.Sequential(nn.Linear(audio_dim, hidden_dim), nn.ReLU(), nn.Dropout(0.3))
        self.visual_bilstm = nn.LSTM(hidden_dim, hidden_dim // 2, bidirectional=True, batch_first=True)
        self.audio_bilstm = nn.LSTM(hidden_dim, hidden_dim // 2, bidirectional=True, batch_first=True)
        self.attention = nn.MultiheadAttention(embed_dim=hidden_dim * 2, num_heads=4)
        self.scorer = nn.Sequential(nn.Linear(hidden_dim * 2, 64), nn.ReLU(), nn.Linear(64, 1), nn.Sigmoid())
        self._prepared = None
        self._prepared_key = None

    # ------------------------------------------------------------------ weights in kernel layout
    def _prepare(self):
        key = _param_key(self)
        if self._prepared is not None and key == self._prepared_key:
            return self._prepared
        with torch.no_grad():
            prep = {}
            # all four recurrences (visual fwd/rev, audio fwd/rev) as ONE launch: direction d writes fused[:, d*H:]
            prep["whh_t4"] = torch.cat([torch.stack([l.weight_hh_l0.t().contiguous(),
                                                     l.weight_hh_l0_reverse.t().contiguous()])
                                        for l in (self.visual_bilstm, self.audio_bilstm)]).contiguous()
            for name, lstm in (("v", self.visual_bilstm), ("a", self.audio_bilstm)):
                hid = lstm.hidden_size
                # input projection of both directions as one GEMM: rows [fwd 4H | rev 4H]
                prep[name + "_wih"] = torch.cat([lstm.weight_ih_l0, lstm.weight_ih_l0_reverse], 0).contiguous()
                prep[name + "_bih"] = torch.cat([lstm.bias_ih_l0 + lstm.bias_hh_l0,
                                                 lstm.bias_ih_l0_reverse + lstm.bias_hh_l0_reverse]).contiguous()
                # recurrent weights transposed: [dir][k][gate row]
                prep[name + "_whh_t"] = torch.stack([lstm.weight_hh_l0.t().contiguous(),
                                                     lstm.weight_hh_l0_reverse.t().contiguous()]).contiguous()
                prep[name + "_hid"] = hid
            e = self.attention.embed_dim
            prep["w_in"] = self.attention.in_proj_weight.contiguous()
            prep["b_in"] = self.attention.in_proj_bias.contiguous()
            prep["w_v"] = self.attention.in_proj_weight[2 * e:3 * e].contiguous()
            prep["b_v"] = self.attention.in_proj_bias[2 * e:3 * e].contiguous()
            prep["w_o"] = self.attention.out_proj.weight.contiguous()
            prep["b_o"] = self.attention.out_proj.bias.contiguous()
        self._prepared, self._prepared_key = prep, key
        return prep

    # ------------------------------------------------------------------ scoring of concatenated sequences
    def score_rows(self, visual_rows, audio_rows, seq_rows, attn_batch=1):
        """Scores for rows of concatenated sequences.

        visual_rows [R, visual_dim], audio_rows [R, audio_dim] (fp32, device);
        seq_rows int64 [S+1] device row offsets of the S sequences (each is an
        independent LSTM recurrence).  ``attn_batch`` is the B of the reference's
        [B, T, .] call: rows are then ordered b-major with S == B sequences of
        equal length T, and the attention couples them per time-step.
        Returns fp32 [R].
        """
        if self.training:
            raise NotImplementedError("score_rows is the inference entry; training goes through forward() (B = 1)")
        p = self._prepare()
        vfc, afc = self.visual_fc[0], self.audio_fc[0]
        rows = visual_rows.shape[0]
        dev = visual_rows.device
        hidden = vfc.out_features
        v_emb = ops.linear(visual_rows, vfc.weight, vfc.bias, ops.ACT_RELU)
        a_emb = ops.linear(audio_rows, afc.weight, afc.bias, ops.ACT_RELU)
        fused = torch.empty((rows, 2 * hidden), dtype=torch.float32, device=dev)
        hid = p["v_hid"]
        xproj = torch.empty((rows, 16 * hid), dtype=torch.float32, device=dev)  # [v fwd | v rev | a fwd | a rev] x 4H
        ops.linear(v_emb, p["v_wih"], p["v_bih"], out=xproj[:, :8 * hid])
        ops.linear(a_emb, p["a_wih"], p["a_bih"], out=xproj[:, 8 * hid:])
        ops.lstm(xproj, p["whh_t4"], hid, 4, 0b1010, seq_rows, fused, 0)
        e = self.attention.embed_dim
        if attn_batch == 1:
            # softmax over a single key is exactly 1: attention == out_proj(v_proj(x))
            val = ops.linear(fused, p["w_v"], p["b_v"])
        else:
            t = rows // attn_batch
            qkv = ops.linear(fused, p["w_in"], p["b_in"])
            val = ops.mha_batchaxis(qkv, attn_batch, t, e, self.attention.num_heads)
        attn_out = ops.linear(val, p["w_o"], p["b_o"])
        s0, s2 = self.scorer[0], self.scorer[2]
        hid64 = ops.linear(attn_out, s0.weight, s0.bias, ops.ACT_RELU)
        return ops.score_head(hid64, s2.weight.reshape(-1), s2.bias)

    def forward(self, visual, audio):
        if not visual.is_cuda or not audio.is_cuda:
            raise RuntimeError("AVBiLSTMModel runs on the MI355X HIP path only: move inputs with .cuda()")
        if visual.dim() != 3 or audio.dim() != 3 or visual.shape[:2] != audio.shape[:2]:
            raise ValueError(f"expected visual [B,T,Dv] and audio [B,T,Da], got {tuple(visual.shape)} / "
                             f"{tuple(audio.shape)}")
        b, t = visual.shape[:2]
        v = visual.reshape(b * t, -1).float().contiguous()
        a = audio.reshape(b * t, -1).float().contiguous()
        needs_grad = torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters())
                                                  or visual.requires_grad or audio.requires_grad)
        if self.training or (needs_grad and b == 1):
            # scripts/train_av_model.py:86-96: one video per step, Dropout active, autograd through the HIP path
            if b != 1:
                raise NotImplementedError("the HIP training path takes one sequence per call (B = 1), as the "
                                          "reference's training loop does")
            if t == 0:
                raise ValueError("empty sequence")
            from ._scorer_train import ScorerTrainFunction, dropout_keep
            hidden = self.visual_fc[0].out_features
            if self.training:
                masks = getattr(self, "_dropout_keep", None)  # tests inject fixed masks here
                keep_v, keep_a = masks if masks is not None else (dropout_keep((t, hidden), v.device),
                                                                  dropout_keep((t, hidden), v.device))
            else:
                keep_v = keep_a = torch.ones((t, hidden), dtype=torch.float32, device=v.device)
            scores = ScorerTrainFunction.apply(self, v, a, keep_v, keep_a, *[p for _, p in self.named_parameters()])
            return scores.view(b, t, 1).squeeze()
        if needs_grad:
            # eval mode, B > 1, autograd on: the inference path below builds no graph - refuse instead of returning
            # a tensor whose .backward() would silently do nothing
            raise NotImplementedError("gradients through AVBiLSTMModel need one sequence per call (B = 1), as the "
                                      "reference's training loop uses it; wrap inference in torch.no_grad()")
        seq_rows = torch.arange(0, (b + 1) * t, max(t, 1), dtype=torch.int64, device=v.device)[: b + 1] if t > 0 \
            else torch.zeros(b + 1, dtype=torch.int64, device=v.device)
        scores = self.score_rows(v, a, seq_rows, attn_batch=b)
        return scores.view(b, t, 1).squeeze()
