"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the UniVTG hot path (parity pinned, see below).

A functional, plain-torch (CPU, fp32 or fp64) restatement of what the reference computes on the
path SURVEY.md section 8 scopes.  It is NOT the product and is never imported by ``univtg_amd``.
Every function cites the reference file:line it restates (paths relative to the reference
tree).  The arithmetic primitives (LayerNorm, linear, softmax, erf-GELU, conv1d, cosine
similarity, smooth-L1, BCE, log-softmax) live in the third-party dependency ``torch`` (pinned
``torch==2.0.1`` in the reference's requirements.txt:291; this container has 2.10) -- the
oracle uses the same torch CPU primitives, composed by hand instead of through nn.Module.

Parity pin: ``oracle/make_golden.py`` imports the real reference from /root/reference, runs it
on seeded inputs and stores inputs/weights/outputs/losses/gradients under ``tests/golden``;
``tests/test_oracle_golden.py`` checks this restatement against those vectors (and, when
/root/reference is present, against the live reference at full size).

Parameters are a flat ``dict[str, Tensor]`` keyed by the reference's ``state_dict`` names
(model/univtg.py:76-103, SURVEY.md section 8b "Checkpoint layout").
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F

LOG_TINY = 1e-45  # model/univtg.py:147,271 -- fp32 denormal, log() ~= -103.28
NCE_TAU = 0.07    # model/univtg.py:185 -- hard-coded, ignores --temperature


# ----------------------------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------------------------
def make_cfg(**over):
    """Hyper-parameters ``build_model`` reads (model/univtg.py:409-450).  Defaults = the
    production shape of scripts/pretrain.sh:33-58 (config 2 of BASELINE.json)."""
    cfg = dict(
        hidden_dim=1024, nheads=8, dim_feedforward=1024, enc_layers=4,
        v_feat_dim=2818, t_feat_dim=512, n_input_proj=2, max_q_l=75, max_v_l=75,
        input_dropout=0.5, dropout=0.0, droppath=0.1, use_txt_pos=False,
        eos_coef=0.1, b_loss_coef=10.0, g_loss_coef=1.0, f_loss_coef=10.0,
        s_loss_intra_coef=0.1, s_loss_inter_coef=0.1,
        set_cost_span=10.0, set_cost_giou=1.0, set_cost_class=4.0,
        losses=("spans", "labels", "saliency"),
    )
    cfg.update(over)
    return SimpleNamespace(**cfg)


def weight_dict(cfg):
    """model/univtg.py:428-432."""
    return {"loss_b": cfg.b_loss_coef, "loss_g": cfg.g_loss_coef, "loss_f": cfg.f_loss_coef,
            "loss_s_intra": cfg.s_loss_intra_coef, "loss_s_inter": cfg.s_loss_inter_coef}


def param_shapes(cfg):
    """state_dict key -> shape, in the reference's registration order."""
    d, F_, E = cfg.hidden_dim, cfg.dim_feedforward, cfg.enc_layers
    shapes = {}
    for l in range(E):
        p = f"transformer.encoder.layers.{l}."
        shapes[p + "self_attn.in_proj_weight"] = (3 * d, d)
        shapes[p + "self_attn.in_proj_bias"] = (3 * d,)
        shapes[p + "self_attn.out_proj.weight"] = (d, d)
        shapes[p + "self_attn.out_proj.bias"] = (d,)
        shapes[p + "linear1.weight"] = (F_, d)
        shapes[p + "linear1.bias"] = (F_,)
        shapes[p + "linear2.weight"] = (d, F_)
        shapes[p + "linear2.bias"] = (d,)
        shapes[p + "norm1.weight"] = (d,)
        shapes[p + "norm1.bias"] = (d,)
        shapes[p + "norm2.weight"] = (d,)
        shapes[p + "norm2.bias"] = (d,)
    shapes["txt_position_embed.position_embeddings.weight"] = (cfg.max_q_l, d)
    shapes["txt_position_embed.LayerNorm.weight"] = (d,)
    shapes["txt_position_embed.LayerNorm.bias"] = (d,)
    shapes["token_type_embeddings.weight"] = (2, d)
    for head, out in (("span_embed", 2), ("class_embed", 1)):
        for i in range(3):
            o = d if i < 2 else out
            shapes[f"{head}.layers.{i}.weight"] = (o, d, 3)
            shapes[f"{head}.layers.{i}.bias"] = (o,)
    for mod, din in (("input_txt_proj", cfg.t_feat_dim), ("input_vid_proj", cfg.v_feat_dim)):
        for i in range(cfg.n_input_proj):
            k = din if i == 0 else d
            shapes[f"{mod}.{i}.LayerNorm.weight"] = (k,)
            shapes[f"{mod}.{i}.LayerNorm.bias"] = (k,)
            shapes[f"{mod}.{i}.net.1.weight"] = (d, k)
            shapes[f"{mod}.{i}.net.1.bias"] = (d,)
    shapes["weightedpool.weight"] = (d, 1)
    return shapes


def init_params(cfg, seed=2018, dtype=torch.float32):
    """Seeded random weights with the reference's shapes.  The distributions only mimic the
    reference's initialisers in spirit (xavier-ish matrices, N(0, .02) embeddings, LN gamma
    around 1) -- there is no checkpoint to match, parity is tensor-by-tensor on whatever
    weights both sides load.  LN gains/biases and linear biases are perturbed so that no term
    of the computation is trivially 0/1."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith("LayerNorm.weight") or ".norm1.weight" in name or ".norm2.weight" in name:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("bias"):
            t = 0.02 * torch.randn(shape, generator=g)
        elif "embeddings" in name:
            t = 0.02 * torch.randn(shape, generator=g)
        else:
            fan_out, fan_in = shape[0], math.prod(shape[1:])
            bound = math.sqrt(6.0 / (fan_in + fan_out))
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        out[name] = t.to(dtype)
    return out


# ----------------------------------------------------------------------------------------------
# forward pieces
# ----------------------------------------------------------------------------------------------
def _dropout(x, keep_mask, p):
    """nn.Dropout semantics with an explicit Bernoulli keep mask (1=keep)."""
    if keep_mask is None or p == 0.0:
        return x
    return x * keep_mask.to(x.dtype) / (1.0 - p)


def input_projection(params, prefix, x, n_layers, p_drop=0.0, keep_masks=None):
    """model/univtg.py:91-100 (construction) + 384-406 (LinearLayer.forward):
    y = Linear(Dropout(LayerNorm(x))); ReLU after every block except the last."""
    for i in range(n_layers):
        x = F.layer_norm(x, (x.shape[-1],), params[f"{prefix}.{i}.LayerNorm.weight"],
                         params[f"{prefix}.{i}.LayerNorm.bias"], eps=1e-5)
        x = _dropout(x, None if keep_masks is None else keep_masks[i], p_drop)
        x = F.linear(x, params[f"{prefix}.{i}.net.1.weight"], params[f"{prefix}.{i}.net.1.bias"])
        if i != n_layers - 1:
            x = torch.relu(x)
    return x


def sine_position(mask, d):
    """model/position_encoding.py:60-83 with normalize=True, scale=2*pi, temperature=1e4,
    num_pos_feats=d (built at :113-117).  ``mask`` is (B, L) 0/1 float."""
    x = mask.to(torch.float32).cumsum(1, dtype=torch.float32)
    x = x / (x[:, -1:] + 1e-6) * (2 * math.pi)
    i = torch.arange(d, dtype=torch.float32)
    dim_t = 10000 ** (2 * torch.div(i, 2).int() / d)
    ang = x[:, :, None] / dim_t
    pos = torch.stack((ang[:, :, 0::2].sin(), ang[:, :, 1::2].cos()), dim=3).flatten(2)
    return pos


def attention(params, pfx, x, pos, key_valid, nheads, p_attn=0.0, attn_keep=None):
    """model/transformer_encoder_droppath.py:117-118 + torch F.multi_head_attention_forward
    (need_weights path): q = k = x + pos, v = x; q is scaled by hd**-0.5 after its bias;
    padded keys get -inf; softmax over keys; dropout on the probabilities; out-projection.
    x, pos: (B, S, d); key_valid: (B, S) bool."""
    B, S, d = x.shape
    hd = d // nheads
    W, b = params[pfx + "self_attn.in_proj_weight"], params[pfx + "self_attn.in_proj_bias"]
    u = x + pos
    q = F.linear(u, W[:d], b[:d]) * (hd ** -0.5)
    k = F.linear(u, W[d:2 * d], b[d:2 * d])
    v = F.linear(x, W[2 * d:], b[2 * d:])
    q = q.view(B, S, nheads, hd).transpose(1, 2)
    k = k.view(B, S, nheads, hd).transpose(1, 2)
    v = v.view(B, S, nheads, hd).transpose(1, 2)
    scores = q @ k.transpose(-1, -2)                                   # (B, H, S, S)
    scores = scores.masked_fill(~key_valid[:, None, None, :], float("-inf"))
    prob = torch.softmax(scores, dim=-1)
    prob = _dropout(prob, attn_keep, p_attn)
    o = (prob @ v).transpose(1, 2).reshape(B, S, d)
    return F.linear(o, params[pfx + "self_attn.out_proj.weight"], params[pfx + "self_attn.out_proj.bias"])


def encoder_layer(params, l, x, pos, key_valid, cfg, dp_scale=None, attn_keep=None):
    """model/transformer_encoder_droppath.py:112-126 (post-norm), DropPath :154-183.
    dp_scale: None or (2, B) per-sample factors in {0, 1/keep} for the two residual branches."""
    pfx = f"transformer.encoder.layers.{l}."
    d = x.shape[-1]
    a = attention(params, pfx, x, pos, key_valid, cfg.nheads, cfg.dropout, attn_keep)
    if dp_scale is not None:
        a = a * dp_scale[0].to(a.dtype)[:, None, None]
    x = F.layer_norm(x + a, (d,), params[pfx + "norm1.weight"], params[pfx + "norm1.bias"], eps=1e-5)
    h = F.gelu(F.linear(x, params[pfx + "linear1.weight"], params[pfx + "linear1.bias"]))
    f = F.linear(h, params[pfx + "linear2.weight"], params[pfx + "linear2.bias"])
    if dp_scale is not None:
        f = f * dp_scale[1].to(f.dtype)[:, None, None]
    x = F.layer_norm(x + f, (d,), params[pfx + "norm2.weight"], params[pfx + "norm2.bias"], eps=1e-5)
    return x


def conv_head(params, name, m):
    """model/univtg.py:367-382: three Conv1d(k=3, pad=1, zeros) with ReLU between, applied
    along the (batch-padded) clip axis.  m: (B, L, d) -> (B, L, out)."""
    x = m.transpose(1, 2)
    for i in range(3):
        x = F.conv1d(x, params[f"{name}.layers.{i}.weight"], params[f"{name}.layers.{i}.bias"], padding=1)
        if i < 2:
            x = torch.relu(x)
    return x.transpose(1, 2)


def weighted_pool(params, txt, txt_mask):
    """model/univtg.py:22-24,36-49: alpha = softmax_t(x.w + (1-m)*(-1e30)); pooled = sum alpha x."""
    alpha = torch.tensordot(txt, params["weightedpool.weight"], dims=1)      # (B, L_t, 1)
    alpha = alpha + (1.0 - txt_mask.to(torch.float32).to(alpha.dtype)).unsqueeze(2) * (-1e30)
    alpha = torch.softmax(alpha, dim=1)
    return torch.matmul(txt.transpose(1, 2), alpha).squeeze(2)               # (B, d)


def log_mask(mask01, dtype):
    """(mask + 1e-45).log() evaluated in fp32 (model/univtg.py:147) -- 0 or ~-103.2789."""
    return (mask01.to(torch.float32) + LOG_TINY).log().to(dtype)


def forward(params, cfg, src_txt, src_txt_mask, src_vid, src_vid_mask, src_cls=None, src_cls_mask=None, rng=None):
    """model/univtg.py:105-155.  ``rng`` (optional) is a dict of explicit stochastic masks for
    train-mode restatement: 'vid_keep'/'txt_keep' (lists of (B,L,K) 0/1 per projection block),
    'dp_scale' (E,2,B), 'attn_keep' (E,B,H,S,S).  None => eval mode.
    ``src_cls`` / ``src_cls_mask`` (model/univtg.py:109-117,151-153): the class-name token features of the TAL pre-training branch -- they
    take the TEXT projection and type embedding and are pooled to ``cls_mem_proj`` (n_cls, d); they never enter the encoder."""
    rng = rng or {}
    d = cfg.hidden_dim
    dt = src_vid.dtype
    vid = input_projection(params, "input_vid_proj", src_vid, cfg.n_input_proj,
                           cfg.input_dropout, rng.get("vid_keep"))
    txt = input_projection(params, "input_txt_proj", src_txt, cfg.n_input_proj,
                           cfg.input_dropout, rng.get("txt_keep"))
    tt = params["token_type_embeddings.weight"]
    vid = vid + tt[1]                                                   # :114
    txt = txt + tt[0]                                                   # :115
    L_v = vid.shape[1]
    x = torch.cat([vid, txt], dim=1)                                    # :119
    key_valid = torch.cat([src_vid_mask, src_txt_mask], dim=1).bool()   # :120
    pos_v = sine_position(src_vid_mask, d).to(dt)                       # :122
    if cfg.use_txt_pos:
        # TrainablePositionalEncoding.forward (model/position_encoding.py:19-41): Dropout(LayerNorm(txt + E[0..L_t))), built with
        # dropout = input_dropout (position_encoding.py:113-115); the result is used as `pos` of the text rows in EVERY layer
        L_t = txt.shape[1]
        e = params["txt_position_embed.position_embeddings.weight"][:L_t]
        pos_t = F.layer_norm(txt + e, (d,), params["txt_position_embed.LayerNorm.weight"],
                             params["txt_position_embed.LayerNorm.bias"], eps=1e-5)
        pos_t = _dropout(pos_t, rng.get("txtpos_keep"), cfg.input_dropout)
    else:
        pos_t = torch.zeros_like(txt)
    pos = torch.cat([pos_v, pos_t], dim=1)                              # :123-124
    for l in range(cfg.enc_layers):
        x = encoder_layer(params, l, x, pos, key_valid, cfg,
                          None if "dp_scale" not in rng else rng["dp_scale"][l],
                          None if "attn_keep" not in rng else rng["attn_keep"][l])
    vid_mem = x[:, :L_v]                                                # :127
    pred_logits = torch.sigmoid(conv_head(params, "class_embed", vid_mem))      # :129
    sign = torch.tensor([-1.0, 1.0], dtype=dt)
    pred_spans = torch.sigmoid(conv_head(params, "span_embed", vid_mem)) * sign  # :130-136
    pooled = weighted_pool(params, txt, src_txt_mask)                   # :146
    sal = F.cosine_similarity(vid, pooled.unsqueeze(1), dim=-1) + log_mask(src_vid_mask, dt)  # :147
    out = {"pred_logits": pred_logits, "pred_spans": pred_spans, "src_vid_mask": src_vid_mask,
           "vid_mem_proj": vid, "txt_mem_proj": pooled.unsqueeze(1), "saliency_scores": sal,
           "_memory": x}
    if src_cls is not None:
        cls = input_projection(params, "input_txt_proj", src_cls, cfg.n_input_proj, cfg.input_dropout, rng.get("cls_keep")) + tt[0]   # :110-111,116-117
        out["cls_mem_proj"] = weighted_pool(params, cls, src_cls_mask)                                                              # :151-153
    return out


# ----------------------------------------------------------------------------------------------
# criterion
# ----------------------------------------------------------------------------------------------
def paired_giou(a, b):
    """Diagonal of utils/span_utils.py:93-122 (generalized_temporal_iou) -- the reference
    builds the N x N matrix and takes diag (model/univtg.py:209); only pairs (i, i) matter."""
    inter = (torch.minimum(a[:, 1], b[:, 1]) - torch.maximum(a[:, 0], b[:, 0])).clamp(min=0)
    union = (a[:, 1] - a[:, 0]) + (b[:, 1] - b[:, 0]) - inter
    iou = inter / union
    hull = (torch.maximum(a[:, 1], b[:, 1]) - torch.minimum(a[:, 0], b[:, 0])).clamp(min=0)
    return iou - (hull - union) / hull


def giou_matrix(a, b):
    """utils/span_utils.py:46-73,93-122: full (N, M) gIoU (used by the matcher)."""
    inter = (torch.minimum(a[:, None, 1], b[:, 1]) - torch.maximum(a[:, None, 0], b[:, 0])).clamp(min=0)
    union = (a[:, 1] - a[:, 0])[:, None] + (b[:, 1] - b[:, 0]) - inter
    iou = inter / union
    hull = (torch.maximum(a[:, None, 1], b[:, 1]) - torch.minimum(a[:, None, 0], b[:, 0])).clamp(min=0)
    return iou - (hull - union) / hull


def loss_spans(out, tg):
    """model/univtg.py:195-214."""
    src = tg["timestamp"] + out["pred_spans"]
    gt = tg["span_labels_nn"]
    win = tg["timestamp_window"].bool()
    l1 = F.smooth_l1_loss(src, gt, reduction="none") * tg["timestamp_window"].unsqueeze(2)
    return {"loss_b": l1.sum() / win.sum(),
            "loss_g": (1 - paired_giou(src[win], gt[win])).mean()}


def loss_labels(out, tg, eos_coef):
    """model/univtg.py:216-233: weighted BCE, fg weight 1, valid-bg weight eos_coef."""
    p = out["pred_logits"].squeeze(-1)
    valid = tg["timestamp_mask"].bool()
    win = tg["timestamp_window"].bool()
    w = torch.zeros_like(p)
    w[valid] = eos_coef
    w[win] = 1.0
    bce = F.binary_cross_entropy(p, win.to(p.dtype), weight=w, reduction="none") * valid
    return {"loss_f": bce.sum() / valid.sum()}


def cosine_matrix(a, b, eps=1e-8):
    """model/univtg.py:26-34 (sim_matrix)."""
    an = a / torch.clamp(a.norm(dim=1, keepdim=True), min=eps)
    bn = b / torch.clamp(b.norm(dim=1, keepdim=True), min=eps)
    return an @ bn.t()


def loss_saliency(out, tg):
    """model/univtg.py:235-282."""
    if "saliency_pos_labels" not in tg or float(tg["saliency_scores"].sum()) == 0:
        return {"loss_s_inter": 0.0, "loss_s_intra": 0.0}
    vid = out["vid_mem_proj"]
    B = vid.shape[0]
    bi = torch.arange(B)
    pos = tg["saliency_pos_labels"][:, 0].long()
    q = out["txt_mem_proj"].squeeze(1)
    sim = cosine_matrix(vid[bi, pos], q)
    inter = -torch.diag(F.log_softmax(sim / NCE_TAU, dim=1)).sum() / B \
            - torch.diag(F.log_softmax(sim.t() / NCE_TAU, dim=1)).sum() / B
    sal = tg["saliency_scores"]
    neg = sal < sal[bi, pos].unsqueeze(-1)
    neg[bi, pos] = True
    neg = neg * tg["timestamp_mask"].bool()
    z = F.cosine_similarity(vid, q.unsqueeze(1), dim=-1) + log_mask(neg, vid.dtype)
    row = F.log_softmax(z / NCE_TAU, dim=1)[bi, pos]
    col = F.log_softmax(z.t() / NCE_TAU, dim=1)[pos, bi]
    intra = -row.sum() / B - col.sum() / B
    return {"loss_s_inter": inter, "loss_s_intra": intra}


def loss_saliency_cls(out, tg):
    """model/univtg.py:284-326 (the 'saliency_cls' loss of the TAL pre-training branch, selected by 'tal' in train_path, :436-438): the
    inter-video term of loss_saliency, and -- with targets['cls_idx'] (B, n_cls) -- the class term: log-softmax over the classes of the
    cosine similarity between each sample's positive clip and every pooled class-name feature, averaged over the entries cls_idx marks."""
    if "saliency_pos_labels" not in tg or float(tg["saliency_scores"].sum()) == 0:
        return {"loss_s_inter": 0.0, "loss_s_intra": 0.0}
    vid = out["vid_mem_proj"]
    B = vid.shape[0]
    bi = torch.arange(B)
    pos = tg["saliency_pos_labels"][:, 0].long()
    v = vid[bi, pos]
    sim = cosine_matrix(v, out["txt_mem_proj"].squeeze(1))
    inter = -torch.diag(F.log_softmax(sim / NCE_TAU, dim=1)).sum() / B \
            - torch.diag(F.log_softmax(sim.t() / NCE_TAU, dim=1)).sum() / B
    if "cls_idx" not in tg:                        # eval (:312-313)
        return {"loss_s_inter": inter}
    idx = tg["cls_idx"].bool()
    lsm = F.log_softmax(cosine_matrix(v, out["cls_mem_proj"]) / NCE_TAU, dim=1)
    picked = lsm[idx]
    return {"loss_s_inter": inter, "loss_s_intra": -picked.sum() / len(picked)}


def criterion(out, tg, cfg):
    """model/univtg.py:338-351 (SetCriterion.forward; indices=None, matcher unused)."""
    losses = {}
    for name in cfg.losses:
        if name == "spans":
            losses.update(loss_spans(out, tg))
        elif name == "labels":
            losses.update(loss_labels(out, tg, cfg.eos_coef))
        elif name == "saliency":
            losses.update(loss_saliency(out, tg))
        elif name == "saliency_cls":
            losses.update(loss_saliency_cls(out, tg))
        else:
            raise ValueError(name)
    return losses


def total_loss(losses, cfg):
    """main/train_vlp_ddp.py:58-59."""
    w = weight_dict(cfg)
    return sum(losses[k] * w[k] for k in losses if k in w)


# ----------------------------------------------------------------------------------------------
# synthetic batches and dense targets (main/dataset.py:173-230)
# ----------------------------------------------------------------------------------------------
def dense_targets(len_v, L_max, windows_sec, clip_len, rng):
    """Per-sample dense targets as main/dataset.py:173-204 builds them (single- or multi-window
    'qid is not None' branch), padded to L_max like start_end_collate_mr (:1037-1052).
    windows_sec: (G, 2) float tensor of GT windows in seconds."""
    ctx_l = len_v
    ts = ((torch.arange(0, ctx_l) + clip_len / 2) / ctx_l).unsqueeze(1).repeat(1, 2)     # :173
    win = (windows_sec / (ctx_l * clip_len))                                              # :184
    G = win.shape[0]
    winr = win.unsqueeze(0).repeat(ctx_l, 1, 1)
    tsr = ts.unsqueeze(1).repeat(1, G, 1)
    nn_w = torch.zeros_like(ts)
    ok = torch.where(((tsr[..., 0] - winr[..., 0]) >= 0) * ((winr[..., 1] - tsr[..., 1]) >= 0))
    if min(ok[0].shape) == 0:
        nn_w = winr.squeeze(1)                                                            # :194
    else:
        nn_w[ok[0]] = winr[ok[0], ok[1]]                                                  # :196
    tw = 1 * (ts[:, 0] >= nn_w[:, 0]) & (ts[:, 1] <= nn_w[:, 1])                          # :199
    if tw.sum() < 1:                                                                      # :202-205
        idx = max(0, min(int(float(windows_sec[0, 0]) / clip_len), ctx_l - 1))
        tw[idx] = 1
    fg = torch.where(tw)[0].tolist()
    pos = fg[int(torch.randint(len(fg), (1,), generator=rng))]                            # :230

    def pad(t):
        o = torch.zeros((L_max,) + tuple(t.shape[1:]), dtype=torch.float32)
        o[: t.shape[0]] = t.to(torch.float32)
        return o
    m = torch.zeros(L_max)
    m[:ctx_l] = 1
    return dict(timestamp=pad(ts), timestamp_mask=m, timestamp_window=pad(tw),
                span_labels_nn=pad(nn_w), saliency_scores=pad(tw), saliency_pos_labels=pos,
                span_labels=torch.stack([win.sum(-1) * 0.5, win[:, 1] - win[:, 0]], dim=-1).to(torch.float32))  # (cx, w), :302-310


def make_batch(cfg, B, L_v, L_t, seed=0, ragged=False, clip_len=2.0, dtype=torch.float32,
               curve=False):
    """Synthetic batch of SURVEY.md 8d: L2-normalised Gaussian features (per 2304/512 block when
    the layout is SlowFast+CLIP, main/dataset.py:383-389), TEF columns (:206-212), one GT window
    per sample, dense targets as the dataset builds them."""
    g = torch.Generator().manual_seed(seed)
    Dv, Dt = cfg.v_feat_dim, cfg.t_feat_dim
    if ragged:
        lens_v = torch.randint(max(1, L_v // 2), L_v + 1, (B,), generator=g)
        lens_t = torch.randint(max(1, L_t // 4), L_t + 1, (B,), generator=g)
        lens_v[int(torch.randint(B, (1,), generator=g))] = L_v       # collate pads to the batch max
        lens_t[int(torch.randint(B, (1,), generator=g))] = L_t
    else:
        lens_v = torch.full((B,), L_v)
        lens_t = torch.full((B,), L_t)
    src_vid = torch.zeros(B, L_v, Dv)
    src_txt = torch.zeros(B, L_t, Dt)
    vm = torch.zeros(B, L_v)
    tm = torch.zeros(B, L_t)
    tg = {k: [] for k in ("timestamp", "timestamp_mask", "timestamp_window", "span_labels_nn",
                          "saliency_scores", "saliency_pos_labels", "span_labels")}
    for b in range(B):
        lv, lt = int(lens_v[b]), int(lens_t[b])
        feat = torch.randn(lv, Dv - 2, generator=g)
        if Dv - 2 == 2816:
            feat = torch.cat([F.normalize(feat[:, :2304], dim=1), F.normalize(feat[:, 2304:], dim=1)], 1)
        else:
            feat = F.normalize(feat, dim=1)
        st = torch.arange(0, lv, 1.0) / lv
        src_vid[b, :lv] = torch.cat([feat, st[:, None], st[:, None] + 1.0 / lv], dim=1)
        src_txt[b, :lt] = F.normalize(torch.randn(lt, Dt, generator=g), dim=1)
        vm[b, :lv] = 1
        tm[b, :lt] = 1
        dur = lv * clip_len
        w0 = float(torch.rand(1, generator=g)) * 0.7 * dur
        ww = (0.05 + 0.25 * float(torch.rand(1, generator=g))) * dur
        win = torch.tensor([[w0, min(dur, w0 + ww)]])
        t = dense_targets(lv, L_v, win, clip_len, g)
        if curve:   # QVHighlights-style graded saliency (main/dataset.py:214-221)
            sc = torch.zeros(L_v)
            fgi = torch.where(t["timestamp_window"] > 0)[0]
            sc[fgi] = 1.0 + 3.0 * torch.rand(len(fgi), generator=g)
            t["saliency_scores"] = sc
        for k in tg:
            tg[k].append(t[k])
    targets = {k: torch.stack(tg[k]).to(dtype) for k in
               ("timestamp", "timestamp_mask", "timestamp_window", "span_labels_nn", "saliency_scores")}
    targets["saliency_pos_labels"] = torch.tensor(tg["saliency_pos_labels"], dtype=torch.long)[:, None]
    targets["span_labels"] = [dict(spans=s) for s in tg["span_labels"]]
    inputs = dict(src_txt=src_txt.to(dtype), src_txt_mask=tm.to(dtype),
                  src_vid=src_vid.to(dtype), src_vid_mask=vm.to(dtype))
    return inputs, targets
