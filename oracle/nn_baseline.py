"""TEST / BENCH INFRASTRUCTURE ONLY -- the CPU baseline of bench.py, composed of the SAME torch.nn modules the reference composes.

/root/reference does not exist on the GPU box, so the reference's own ``Model`` cannot be timed there.  ``oracle/univtg_oracle.py`` is a
functional restatement that is convenient for parity (explicit dropout masks) but cheaper on a CPU than the reference: it never builds
``nn.MultiheadAttention``'s head-averaged attention weights and never permutes to the (S, B, d) layout.  This module restores the
reference's cost structure with the library modules it is built from:

  * ``nn.LayerNorm`` -> ``nn.Dropout`` -> ``nn.Linear`` -> ReLU blocks for the two input projections (model/univtg.py:91-100,385-406)
  * ``nn.Embedding`` token types, sine position table (model/position_encoding.py:44-83)
  * post-norm encoder layers around ``nn.MultiheadAttention(d, H, dropout)`` called with ``need_weights`` left at its default (True:
    the (B, S, S) head-averaged weights are materialised and thrown away, model/transformer_encoder_droppath.py:118), on the
    (S, B, d) layout behind ``permute(1, 0, 2)`` (:49-54), erf-GELU FFN, per-sample DropPath through two more permutes (:154-183)
  * ``nn.Conv1d(k=3, padding=1)`` x3 heads on (B, d, L) (model/univtg.py:365-382), weighted text pooling + cosine saliency (:36-49,143-147)

Module attribute names follow the reference's ``state_dict`` keys, so ``load_state_dict(oracle.init_params(cfg))`` is strict.  It is pinned
to the oracle (hence to the golden vectors of the real reference) by ``tests/test_oracle_golden.py::test_nn_baseline_matches_oracle``.
Never imported by ``univtg_amd``.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import nn


class _ProjBlock(nn.Module):
    def __init__(self, n_in, n_out, p, relu):
        super().__init__()
        self.LayerNorm = nn.LayerNorm(n_in)
        self.net = nn.Sequential(nn.Dropout(p), nn.Linear(n_in, n_out))
        self.relu = relu

    def forward(self, x):
        y = self.net(self.LayerNorm(x))
        return F.relu(y, inplace=True) if self.relu else y


class _PathDrop(nn.Module):
    """per-sample stochastic depth applied to an (S, B, d) tensor through a (B, S, d) view, as the reference does"""

    def __init__(self, p):
        super().__init__()
        self.p = p

    def forward(self, x):
        if self.p == 0.0 or not self.training:
            return x.permute(1, 0, 2).permute(1, 0, 2)
        xb = x.permute(1, 0, 2)
        keep = 1.0 - self.p
        gate = (keep + torch.rand((xb.shape[0], 1, 1), dtype=x.dtype)).floor_()
        return (xb.div(keep) * gate).permute(1, 0, 2)


class _EncLayer(nn.Module):
    def __init__(self, d, H, F_, p_attn, p_path):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d, H, dropout=p_attn)
        self.linear1, self.linear2 = nn.Linear(d, F_), nn.Linear(F_, d)
        self.norm1, self.norm2 = nn.LayerNorm(d), nn.LayerNorm(d)
        self.droppath1, self.droppath2 = _PathDrop(p_path), _PathDrop(p_path)

    def forward(self, x, pad, pos):
        qk = x + pos
        a = self.self_attn(qk, qk, value=x, key_padding_mask=pad)[0]        # need_weights=True (default), like the reference
        x = self.norm1(x + self.droppath1(a))
        f = self.linear2(F.gelu(self.linear1(x)))
        return self.norm2(x + self.droppath2(f))


class _Stack(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.layers = nn.ModuleList(layers)


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.encoder = _Stack([_EncLayer(cfg.hidden_dim, cfg.nheads, cfg.dim_feedforward, cfg.dropout, cfg.droppath)
                               for _ in range(cfg.enc_layers)])

    def forward(self, src, pad, pos):
        x, p = src.permute(1, 0, 2), pos.permute(1, 0, 2)                  # (S, B, d)
        for lay in self.encoder.layers:
            x = lay(x, pad, p)
        return x.transpose(0, 1)


class _ConvHead(nn.Module):
    def __init__(self, d, n_out):
        super().__init__()
        self.layers = nn.ModuleList([nn.Conv1d(d, d, 3, padding=1), nn.Conv1d(d, d, 3, padding=1), nn.Conv1d(d, n_out, 3, padding=1)])

    def forward(self, x):
        x = x.permute(0, 2, 1)
        for i, c in enumerate(self.layers):
            x = c(x) if i == 2 else F.relu(c(x))
        return x.permute(0, 2, 1)


class _TxtPos(nn.Module):                                    # parameters only (unused unless --use_txt_pos); keeps load_state_dict strict
    def __init__(self, n, d):
        super().__init__()
        self.position_embeddings = nn.Embedding(n, d)
        self.LayerNorm = nn.LayerNorm(d)


class _Pool(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(d, 1))

    def forward(self, x, mask):
        a = torch.tensordot(x, self.weight, dims=1) + (1.0 - mask.unsqueeze(2).float()) * -1e30
        return torch.matmul(x.transpose(1, 2), torch.softmax(a, dim=1)).squeeze(2)


def _sine(mask, d, temperature=10000.0):
    pos = mask.cumsum(1, dtype=torch.float32)
    pos = pos / (pos[:, -1:] + 1e-6) * (2 * math.pi)
    i = torch.arange(d, dtype=torch.float32)
    div = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / d)
    t = pos[:, :, None] / div
    return torch.stack((t[:, :, 0::2].sin(), t[:, :, 1::2].cos()), dim=3).flatten(2)


class NNBaseline(nn.Module):
    """``model(src_txt, src_txt_mask, src_vid, src_vid_mask) -> dict`` with the reference's keys (model/univtg.py:105-155)."""

    def __init__(self, cfg):
        super().__init__()
        d = cfg.hidden_dim
        self.transformer = _Encoder(cfg)
        self.txt_position_embed = _TxtPos(cfg.max_q_l, d)
        self.token_type_embeddings = nn.Embedding(2, d)
        self.span_embed, self.class_embed = _ConvHead(d, 2), _ConvHead(d, 1)
        relu = [True] * 3
        relu[cfg.n_input_proj - 1] = False
        self.input_txt_proj = nn.Sequential(*[_ProjBlock(cfg.t_feat_dim if i == 0 else d, d, cfg.input_dropout, relu[i]) for i in range(cfg.n_input_proj)])
        self.input_vid_proj = nn.Sequential(*[_ProjBlock(cfg.v_feat_dim if i == 0 else d, d, cfg.input_dropout, relu[i]) for i in range(cfg.n_input_proj)])
        self.weightedpool = _Pool(d)
        self.d = d

    def forward(self, src_txt, src_txt_mask, src_vid, src_vid_mask):
        v = self.input_vid_proj(src_vid)
        t = self.input_txt_proj(src_txt)
        v = v + self.token_type_embeddings(torch.full_like(src_vid_mask.long(), 1))
        t = t + self.token_type_embeddings(torch.zeros_like(src_txt_mask.long()))
        src = torch.cat([v, t], dim=1)
        valid = torch.cat([src_vid_mask, src_txt_mask], dim=1).bool()
        pos = torch.cat([_sine(src_vid_mask, self.d), torch.zeros_like(t)], dim=1)
        mem = self.transformer(src, ~valid, pos)[:, : v.shape[1]]
        logits = self.class_embed(mem).sigmoid()
        spans = self.span_embed(mem).sigmoid() * torch.tensor((-1.0, 1.0))
        q = self.weightedpool(t, src_txt_mask).unsqueeze(1)
        sal = F.cosine_similarity(v, q, dim=-1) + (src_vid_mask + 1e-45).log()
        return {"pred_logits": logits, "pred_spans": spans, "src_vid_mask": src_vid_mask, "vid_mem_proj": v, "txt_mem_proj": q,
                "saliency_scores": sal}
