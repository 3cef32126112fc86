"""TEST INFRASTRUCTURE ONLY -- numpy / pure-Python oracle for the integer-and-index side of the
hot path: span decode + ranking (main/inference_mr.py:109-167), greedy temporal NMS
(utils/temporal_nms.py:6-74), round-to-clip (eval/postprocessing.py:46-51) and the Hungarian
matcher (model/matcher.py:36-100).

The LSAP itself lives in the third-party dependency scipy (``scipy.optimize.linear_sum_assignment``,
pinned scipy==1.9.3 in the reference's requirements.txt:251; rectangular shortest-augmenting-path
algorithm of Crouse 2016).  ``lsap`` below restates that published algorithm; it is pinned against
the reference's own matcher outputs in tests/golden/matcher.npz.
"""
from __future__ import annotations

import numpy as np


# ----------------------------------------------------------------------------------------------
# span decode / ranking  (main/inference_mr.py:109-160)
# ----------------------------------------------------------------------------------------------
def r4(x):
    """float(f"{e:.4f}") -- main/inference_mr.py:159."""
    return float(f"{float(x):.4f}")


def decode_windows(pred_logits, pred_spans, timestamp, timestamp_mask, durations):
    """scores = pred_logits[..., 0] with padded clips zeroed (:112,118-119); windows =
    clamp((timestamp + pred_spans) * duration, 0, duration) (:116-117,152-153); rows sorted by score
    descending with Python's stable sort (:158); every number rounded to 4 decimals (:159).
    Arithmetic is fp32 like the reference's torch tensors.  Returns list (per sample) of [st, ed, score]."""
    pred_logits = np.asarray(pred_logits, np.float32)
    spans = (np.asarray(timestamp, np.float32) + np.asarray(pred_spans, np.float32)).astype(np.float32)
    scores = pred_logits[..., 0].copy()
    scores[~np.asarray(timestamp_mask).astype(bool)] = 0
    out = []
    for b in range(spans.shape[0]):
        dur = np.float32(durations[b])
        w = np.clip((spans[b] * dur).astype(np.float32), np.float32(0), dur)
        rows = np.concatenate([w, scores[b][:, None]], axis=1).tolist()
        rows = sorted(rows, key=lambda x: x[2], reverse=True)
        out.append([[r4(e) for e in row] for row in rows])
    return out


def ranked_clip_indices(pred_logits, timestamp_mask):
    """The clip index behind every ranked row: stable argsort of the masked scores, descending
    (what `sorted(..., reverse=True)` does to row positions)."""
    scores = np.asarray(pred_logits, np.float32)[..., 0].copy()
    scores[~np.asarray(timestamp_mask).astype(bool)] = 0
    out = []
    for b in range(scores.shape[0]):
        idx = sorted(range(scores.shape[1]), key=lambda i: scores[b, i], reverse=True)
        out.append(idx)
    return out


def saliency_for_eval(saliency_scores, pred_logits, vid_mask, mode="add"):
    """main/inference_mr.py:124-136: fp16(saliency) [+ fg prob], truncated to each sample's length."""
    s = np.asarray(saliency_scores, np.float32).astype(np.float16)
    if mode == "add":
        s = s + np.asarray(pred_logits, np.float32)[..., 0]          # fp16 + fp32 -> fp32
    lens = np.asarray(vid_mask).sum(1).astype(int)
    return [s[b, : lens[b]].tolist() for b in range(s.shape[0])]


# ----------------------------------------------------------------------------------------------
# temporal NMS  (utils/temporal_nms.py)
# ----------------------------------------------------------------------------------------------
def hull_iou(a, b):
    """utils/temporal_nms.py:6-22: intersection over the *hull* (not the true union)."""
    inter = max(0, min(a[1], b[1]) - max(a[0], b[0]))
    hull = max(a[1], b[1]) - min(a[0], b[0])
    return 0 if hull == 0 else 1.0 * inter / hull


def temporal_nms(preds, nms_thd, max_after_nms=100):
    """utils/temporal_nms.py:25-74, restated with an alive-flag sweep instead of list pops.
    Quirks kept: a single prediction is returned untouched; the greedy loop only runs while more
    than one candidate is alive; the last survivor is appended if there is still room."""
    if len(preds) == 1:
        return preds
    order = sorted(range(len(preds)), key=lambda i: preds[i][2], reverse=True)
    rows = [preds[i] for i in order]
    alive = [True] * len(rows)
    n_alive = len(rows)
    kept = []
    head = 0
    while n_alive > 1 and len(kept) < max_after_nms:
        while not alive[head]:
            head += 1
        for j in range(head + 1, len(rows)):
            if alive[j] and hull_iou(rows[head][:2], rows[j][:2]) > nms_thd:
                alive[j] = False
                n_alive -= 1
        kept.append(rows[head])
        alive[head] = False
        n_alive -= 1
    if len(kept) < max_after_nms and n_alive >= 1:
        while not alive[head]:
            head += 1
        kept.append(rows[head])
    return [[r[0], r[1], r[2]] for r in kept]


def round_multiple(rows, clip_length):
    """eval/postprocessing.py:26-37,46-51: torch.round(w / clip) * clip in fp32 (half-to-even),
    score re-rounded to 4 decimals."""
    out = []
    for r in rows:
        w = np.asarray(r[:2], np.float32)
        w = (np.round(w / np.float32(clip_length)) * np.float32(clip_length)).astype(np.float32)
        out.append([float(w[0]), float(w[1]), r4(np.float32(r[2]))])
    return out


# ----------------------------------------------------------------------------------------------
# Hungarian matcher  (model/matcher.py)
# ----------------------------------------------------------------------------------------------
def lsap(cost):
    """Rectangular linear sum assignment, shortest augmenting path (Crouse 2016), the algorithm
    behind scipy.optimize.linear_sum_assignment (model/matcher.py:99).  Returns (row_ind, col_ind)
    sorted by row, like scipy."""
    cost = np.asarray(cost, np.float64)
    transposed = cost.shape[1] < cost.shape[0]
    if transposed:
        cost = cost.T
    nr, nc = cost.shape
    u = np.zeros(nr)
    v = np.zeros(nc)
    col4row = -np.ones(nr, int)
    row4col = -np.ones(nc, int)
    for cur in range(nr):
        shortest = np.full(nc, np.inf)
        path = -np.ones(nc, int)
        SR = np.zeros(nr, bool)
        SC = np.zeros(nc, bool)
        remaining = list(range(nc))[::-1]
        min_val = 0.0
        i = cur
        sink = -1
        while sink == -1:
            index = -1
            lowest = np.inf
            SR[i] = True
            for it, j in enumerate(remaining):
                r = min_val + cost[i, j] - u[i] - v[j]
                if r < shortest[j]:
                    path[j] = i
                    shortest[j] = r
                if shortest[j] < lowest or (shortest[j] == lowest and row4col[j] == -1):
                    lowest = shortest[j]
                    index = it
            min_val = lowest
            j = remaining[index]
            if row4col[j] == -1:
                sink = j
            else:
                i = row4col[j]
            SC[j] = True
            remaining[index] = remaining[-1]
            remaining.pop()
        u[cur] += min_val
        for i in range(nr):
            if SR[i] and i != cur:
                u[i] += min_val - shortest[col4row[i]]
        for j in range(nc):
            if SC[j]:
                v[j] -= min_val - shortest[j]
        j = sink
        while True:
            i = path[j]
            row4col[j] = i
            col4row[i], j = j, col4row[i]
            if i == cur:
                break
    if transposed:
        order = np.argsort(col4row)
        return col4row[order], np.arange(nr)[order]
    return np.arange(nr), col4row


def matcher_cost(pred_logits, pred_spans_cxw, tgt_cxw, w_class=4.0, w_span=10.0, w_giou=1.0):
    """model/matcher.py:57-91 cost matrix (fp32): w_span*L1(cxw) + w_giou*(-gIoU(xx)) +
    w_class*(-softmax(logits)[:, 0]).  pred_*: (B*Q, .); tgt: (T, 2)."""
    lg = np.asarray(pred_logits, np.float32)
    e = np.exp(lg - lg.max(-1, keepdims=True))
    prob = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    ps = np.asarray(pred_spans_cxw, np.float32)
    ts = np.asarray(tgt_cxw, np.float32)
    c_class = -prob[:, [0] * len(ts)]
    c_span = np.abs(ps[:, None, :] - ts[None, :, :]).sum(-1)

    def xx(s):
        return np.stack([s[:, 0] - np.float32(0.5) * s[:, 1], s[:, 0] + np.float32(0.5) * s[:, 1]], -1)
    a, b = xx(ps), xx(ts)
    inter = np.clip(np.minimum(a[:, None, 1], b[:, 1]) - np.maximum(a[:, None, 0], b[:, 0]), 0, None)
    union = (a[:, 1] - a[:, 0])[:, None] + (b[:, 1] - b[:, 0]) - inter
    hull = np.clip(np.maximum(a[:, None, 1], b[:, 1]) - np.minimum(a[:, None, 0], b[:, 0]), 0, None)
    giou = inter / union - (hull - union) / hull
    return (np.float32(w_span) * c_span + np.float32(w_giou) * (-giou) + np.float32(w_class) * c_class).astype(np.float32)


def hungarian_match(pred_logits, pred_spans_cxw, tgt_list, **w):
    """model/matcher.py:36-100: per-sample LSAP on that sample's slice of the cost matrix."""
    B, Q = pred_spans_cxw.shape[:2]
    sizes = [len(t) for t in tgt_list]
    tgt = np.concatenate(tgt_list, 0)
    C = matcher_cost(pred_logits.reshape(B * Q, -1), pred_spans_cxw.reshape(B * Q, 2), tgt, **w).reshape(B, Q, -1)
    out, off = [], 0
    for b, n in enumerate(sizes):
        out.append(lsap(C[b, :, off:off + n]))
        off += n
    return out


def detr_criterion(pred_logits, pred_spans_cxw, tgt_list, indices, saliency=None, pos=None, neg=None, proj_q=None,
                   proj_txt=None, eos_coef=0.1, temperature=0.07, margin=1.0, weights=None, dtype=np.float64):
    """model/moment_detr.py:166-365 (span_loss_type 'l1') for one decoder layer: returns (losses[6], grads) with
    losses = (loss_b, loss_g, loss_f, class_error, loss_s_intra, loss_contrastive_align) and, when ``weights`` [6] is
    given, grads = d(sum_k weights[k] * losses[k]) / d(pred_logits, pred_spans, saliency, proj_q, proj_txt).
    ``indices`` is the matcher's list of (pred_idx, tgt_idx) per sample (model/matcher.py:100)."""
    f = dtype
    lg, sp = np.asarray(pred_logits, f), np.asarray(pred_spans_cxw, f)
    B, Q = sp.shape[:2]
    w = np.zeros(6, f) if weights is None else np.asarray(weights, f)
    matched = np.zeros((B, Q), bool)
    tg = np.zeros((B, Q, 2), f)
    for b, (i, j) in enumerate(indices):
        matched[b, np.asarray(i, int)] = True
        tg[b, np.asarray(i, int)] = np.asarray(tgt_list[b], f)[np.asarray(j, int)]
    N = f(matched.sum())
    # spans (:205-230): L1 on (c, w) and 1 - gIoU on (st, ed) over the matched pairs
    src, tgt = sp[matched], tg[matched]
    loss_b = np.abs(src - tgt).sum() / (2 * N)
    x1, x2 = src[:, 0] - f(0.5) * src[:, 1], src[:, 0] + f(0.5) * src[:, 1]
    y1, y2 = tgt[:, 0] - f(0.5) * tgt[:, 1], tgt[:, 0] + f(0.5) * tgt[:, 1]
    inter = np.clip(np.minimum(x2, y2) - np.maximum(x1, y1), 0, None)
    union = (x2 - x1) + (y2 - y1) - inter
    enc = np.clip(np.maximum(x2, y2) - np.minimum(x1, y1), 0, None)
    giou = inter / union - (enc - union) / enc
    loss_g = (1 - giou).sum() / N
    # labels (:234-253): weighted CE, plain mean over B*Q
    cls = np.where(matched, 0, 1)
    mx = lg.max(-1, keepdims=True)
    lse = mx[..., 0] + np.log(np.exp(lg - mx).sum(-1))
    cw = np.where(matched, f(1), f(eos_coef))
    picked = np.take_along_axis(lg, cls[..., None], -1)[..., 0]
    loss_f = (-cw * (picked - lse)).sum() / f(B * Q)
    class_error = f(100) - f(100) * f((matched & (lg[..., 0] >= lg[..., 1])).sum()) / N
    grads = {}
    soft = np.exp(lg - lse[..., None])
    onehot = np.stack([cls == 0, cls == 1], -1).astype(f)
    grads["logits"] = w[2] * cw[..., None] * (soft - onehot) / f(B * Q)
    # span gradients: every term is piecewise linear in the end points
    ind = lambda c: c.astype(f)
    act = ind((np.minimum(x2, y2) - np.maximum(x1, y1)) >= 0)
    di1, di2 = -act * ind(x1 > y1), act * ind(x2 < y2)
    du1, du2 = -1 - di1, 1 - di2
    ea = ind((np.maximum(x2, y2) - np.minimum(x1, y1)) >= 0)
    de1, de2 = -ea * ind(x1 < y1), ea * ind(x2 > y2)
    dg1 = (di1 * union - inter * du1) / union ** 2 + (du1 * enc - union * de1) / enc ** 2
    dg2 = (di2 * union - inter * du2) / union ** 2 + (du2 * enc - union * de2) / enc ** 2
    dsp = np.zeros_like(sp)
    d = np.sign(src - tgt) * w[0] / (2 * N)
    d[:, 0] += -w[1] / N * (dg1 + dg2)
    d[:, 1] += -w[1] / N * f(0.5) * (dg2 - dg1)
    dsp[matched] = d
    grads["spans"] = dsp
    # saliency hinge (:255-270)
    loss_s = f(0)
    if saliency is not None:
        s = np.asarray(saliency, f)
        pos, neg = np.asarray(pos, int), np.asarray(neg, int)
        P = pos.shape[1]
        rows = np.arange(B)[:, None]
        h = f(margin) + s[rows, neg] - s[rows, pos]
        loss_s = np.clip(h, 0, None).sum() / f(B * P) * 2
        ds = np.zeros_like(s)
        gsc = ind(h >= 0) * w[4] * 2 / f(B * P)
        np.add.at(ds, (np.broadcast_to(rows, neg.shape), neg), gsc)
        np.add.at(ds, (np.broadcast_to(rows, pos.shape), pos), -gsc)
        grads["sal"] = ds
    # contrastive alignment (:272-290)
    loss_c = f(0)
    if proj_q is not None:
        q, t = np.asarray(proj_q, f), np.asarray(proj_txt, f)
        tsum = t.sum(1)                                         # (B, D): einsum('bmd,bnd->bmn').sum(2)
        l = np.einsum("bqd,bd->bq", q, tsum) / f(temperature)
        npos = matched.sum(1).astype(f)
        m = l.max(1, keepdims=True)
        lse_q = m[:, 0] + np.log(np.exp(l - m).sum(1))
        loss_c = (-(l * matched).sum(1) / npos + lse_q).mean()
        dl = w[5] / f(B) * (np.exp(l - lse_q[:, None]) - matched / npos[:, None]) / f(temperature)
        grads["pq"] = dl[..., None] * tsum[:, None, :]
        grads["pt"] = np.broadcast_to(np.einsum("bq,bqd->bd", dl, q)[:, None, :], t.shape).copy()
    return np.array([loss_b, loss_g, loss_f, class_error, loss_s, loss_c], f), grads
