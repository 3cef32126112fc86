"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz by executing the REAL reference.

Run in the build container (needs /root/reference; never on the GPU box):

    python oracle/make_golden.py

It imports the reference's own modules (model.univtg, model.matcher, utils.span_utils,
utils.temporal_nms, eval.postprocessing, main.dataset, main.inference_mr -- the last two behind
stubs for the absent h5py / nncore packages), loads seeded weights through
``load_state_dict(strict=True)`` (which also pins the checkpoint key/shape layout), runs forward,
criterion, backward, the matcher and the inference post-processing, and stores every input and
output as fp32/fp64/int64 arrays.  The fixtures are the pin for ``oracle/univtg_oracle.py`` and
``oracle/postproc_oracle.py`` (tests/test_oracle_golden.py) and travel to the GPU box where the
reference does not exist.
"""
from __future__ import annotations

import json
import os
import sys
import types
from argparse import Namespace

import numpy as np
import torch

REF = os.environ.get("UVTG_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import univtg_oracle as O  # noqa: E402


def import_reference():
    """Put the reference on sys.path with stubs for the two packages this image lacks."""
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sys.modules.setdefault("h5py", types.ModuleType("h5py"))
    if "nncore" not in sys.modules:
        nn = types.ModuleType("nncore")
        nn.__path__ = []
        sys.modules["nncore"] = nn
        for sub in ("dataset", "parallel"):
            m = types.ModuleType("nncore." + sub)
            sys.modules["nncore." + sub] = m
            setattr(nn, sub, m)

        class _Registry:
            def register(self, *a, **k):
                if a and isinstance(a[0], type):
                    return a[0]
                return lambda c: c
        sys.modules["nncore.dataset"].DATASETS = _Registry()
        sys.modules["nncore.parallel"].DataContainer = object
    import model.univtg as ref_univtg
    import model.matcher as ref_matcher
    import utils.span_utils as ref_span
    import utils.temporal_nms as ref_nms
    import eval.postprocessing as ref_post
    import main.dataset as ref_dataset
    import main.inference_mr as ref_inf
    return Namespace(univtg=ref_univtg, matcher=ref_matcher, span=ref_span, nms=ref_nms,
                     post=ref_post, dataset=ref_dataset, inf=ref_inf)


def ref_args(cfg, **over):
    """The Namespace build_model reads (model/univtg.py:409-450)."""
    a = dict(device="cpu", hidden_dim=cfg.hidden_dim, dropout=cfg.dropout, droppath=cfg.droppath,
             nheads=cfg.nheads, dim_feedforward=cfg.dim_feedforward, enc_layers=cfg.enc_layers,
             dec_layers=2, pre_norm=False, position_embedding="sine", max_q_l=cfg.max_q_l,
             input_dropout=cfg.input_dropout, t_feat_dim=cfg.t_feat_dim, v_feat_dim=cfg.v_feat_dim,
             span_loss_type="l1", use_txt_pos=False, n_input_proj=cfg.n_input_proj,
             set_cost_span=cfg.set_cost_span, set_cost_giou=cfg.set_cost_giou,
             set_cost_class=cfg.set_cost_class, max_v_l=cfg.max_v_l,
             b_loss_coef=cfg.b_loss_coef, g_loss_coef=cfg.g_loss_coef, f_loss_coef=cfg.f_loss_coef,
             s_loss_intra_coef=cfg.s_loss_intra_coef, s_loss_inter_coef=cfg.s_loss_inter_coef,
             dset_type="vlp", train_path=["synthetic"], eos_coef=cfg.eos_coef, temperature=0.07,
             saliency_margin=0.2)
    a.update(over)
    return Namespace(**a)


def build_reference(ref, cfg, params, **over):
    model, crit = ref.univtg.build_model(ref_args(cfg, use_txt_pos=bool(getattr(cfg, "use_txt_pos", False)), **over))
    missing = model.load_state_dict({k: v.clone() for k, v in params.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return model, crit


def npify(prefix, d, store):
    for k, v in d.items():
        if torch.is_tensor(v):
            store[f"{prefix}{k}"] = v.detach().cpu().numpy()


def make_cls_inputs(cfg, B, n_cls, L_c, seed):
    """Synthetic class-name token features of the TAL branch (main/train_vlp.py:116-122 hands the model `train_dataset.src_cls`, which no
    shipped dataset defines -- shapes follow the text features: (n_cls, L_c, D_t) l2-normalised rows, prefix masks) and a multi-hot
    targets['cls_idx'] (B, n_cls) with at least one class per sample."""
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(1, L_c + 1, (n_cls,), generator=g)
    lens[0] = L_c
    src_cls = torch.zeros(n_cls, L_c, cfg.t_feat_dim)
    mask = torch.zeros(n_cls, L_c)
    for j in range(n_cls):
        src_cls[j, :int(lens[j])] = torch.nn.functional.normalize(torch.randn(int(lens[j]), cfg.t_feat_dim, generator=g), dim=1)
        mask[j, :int(lens[j])] = 1
    idx = torch.zeros(B, n_cls)
    idx[torch.arange(B), torch.randint(n_cls, (B,), generator=g)] = 1
    idx[0, torch.randint(n_cls, (1,), generator=g)] = 1          # one sample with (possibly) two classes
    return src_cls, mask, idx


def run_case(ref, name, cfg, B, L_v, L_t, seed, ragged, curve=False, train_droppath=False,
             real_feats=False, dtype=torch.float32, dset_type="vlp", zero_saliency=False, drop_pos_labels=False, tal=None):
    """dset_type='hl': the reference's loss subset ['labels', 'saliency'] (model/univtg.py:439-440); zero_saliency / drop_pos_labels:
    the two early-outs of loss_saliency (model/univtg.py:237-241) -- the loss dict then holds python floats 0.0."""
    torch.manual_seed(seed)
    params = O.init_params(cfg, seed=seed, dtype=dtype)
    model, crit = build_reference(ref, cfg, params, dset_type=dset_type, **(dict(train_path=["tal"]) if tal else {}))
    assert list(crit.losses) == list(cfg.losses), (crit.losses, cfg.losses)
    inputs, targets = O.make_batch(cfg, B, L_v, L_t, seed=seed + 1, ragged=ragged, dtype=dtype,
                                   curve=curve)
    if tal:      # the TAL pre-training branch (model/univtg.py:109-117,151-153,284-326): class-name features in, cls_idx in the targets
        inputs["src_cls"], inputs["src_cls_mask"], targets["cls_idx"] = make_cls_inputs(cfg, B, tal["n_cls"], tal["L_c"], seed + 2)
    if zero_saliency:
        targets["saliency_scores"] = torch.zeros_like(targets["saliency_scores"])
    if real_feats:
        # config 1 of BASELINE.json: the bundled CLIP features, preprocessed as main_gradio.py:58-80
        vid = np.load(os.path.join(REF, "tmp", "vid.npz"))["features"].astype(np.float32)
        txt = np.load(os.path.join(REF, "tmp", "txt.npz"))["features"].astype(np.float32)
        from utils.basic_utils import l2_normalize_np_array
        vid = torch.from_numpy(l2_normalize_np_array(vid))
        txt = torch.from_numpy(l2_normalize_np_array(txt))
        ctx_l = vid.shape[0]
        tef_st = torch.arange(0, ctx_l, 1.0) / ctx_l
        tef = torch.stack([tef_st, tef_st + 1.0 / ctx_l], dim=1)
        inputs = dict(src_txt=txt[None], src_txt_mask=torch.ones(1, txt.shape[0]),
                      src_vid=torch.cat([vid, tef], dim=1)[None], src_vid_mask=torch.ones(1, ctx_l))
        g = torch.Generator().manual_seed(seed)
        t = O.dense_targets(ctx_l, ctx_l, torch.tensor([[6.0, 14.0]]), 2.0, g)
        targets = {k: t[k][None] for k in ("timestamp", "timestamp_mask", "timestamp_window",
                                            "span_labels_nn", "saliency_scores")}
        targets["saliency_pos_labels"] = torch.tensor([[t["saliency_pos_labels"]]])
        targets["span_labels"] = [dict(spans=t["span_labels"])]
    store = {}
    meta = dict(name=name, cfg=dict(vars(cfg)), B=B, L_v=L_v, L_t=L_t, seed=seed, ragged=ragged,
                torch=torch.__version__, train_droppath=train_droppath, dset_type=dset_type,
                zero_saliency=zero_saliency, drop_pos_labels=drop_pos_labels, tal=tal)
    meta["cfg"]["losses"] = list(meta["cfg"]["losses"])
    npify("param/", params, store)
    npify("in/", inputs, store)
    npify("tg/", {k: v for k, v in targets.items() if torch.is_tensor(v)}, store)
    store["tg/span_labels_sizes"] = np.array([len(s["spans"]) for s in targets["span_labels"]])
    store["tg/span_labels_cat"] = torch.cat([s["spans"] for s in targets["span_labels"]]).numpy()

    if train_droppath:
        # record the per-sample DropPath draws the reference makes (transformer_encoder_droppath.py:163)
        model.train()
        crit.train()
        draws = []
        real_rand = torch.rand

        def rec_rand(*a, **k):
            r = real_rand(*a, **k)
            draws.append(r.detach().clone().flatten())
            return r
        torch.rand = rec_rand
        try:
            out = model(**inputs)
        finally:
            torch.rand = real_rand
        keep = 1.0 - cfg.droppath
        u = torch.stack(draws).view(cfg.enc_layers, 2, B)
        store["rng/dp_scale"] = (torch.floor(keep + u) / keep).numpy()
    else:
        model.eval()
        crit.eval()
        out = model(**inputs)
    crit_targets = {k: v for k, v in targets.items() if not (drop_pos_labels and k == "saliency_pos_labels")}
    # gradients of the weighted total wrt the criterion's INPUTS (a second pass on detached leaves: pins the criterion kernels alone)
    leaves = {k: out[k].detach().clone().requires_grad_(True) for k in ("pred_logits", "pred_spans", "vid_mem_proj", "txt_mem_proj", "cls_mem_proj") if k in out}
    l2 = crit(dict(out, **leaves), crit_targets)
    t2 = sum(l2[k] * crit.weight_dict[k] for k in l2 if k in crit.weight_dict)
    t2.backward()
    for k, v in leaves.items():
        store[f"dout/{k}"] = (v.grad if v.grad is not None else torch.zeros_like(v)).numpy()
    losses = crit(out, crit_targets)
    meta["loss_is_float"] = [k for k, v in losses.items() if not torch.is_tensor(v)]
    wd = crit.weight_dict
    total = sum(losses[k] * wd[k] for k in losses if k in wd)
    total.backward()
    npify("out/", {k: v for k, v in out.items()}, store)
    store["loss/total"] = total.detach().numpy()
    for k, v in losses.items():
        store[f"loss/{k}"] = v.detach().numpy() if torch.is_tensor(v) else np.float32(v)
    for k, p in model.named_parameters():
        if p.grad is not None:
            store[f"grad/{k}"] = p.grad.numpy()
    meta["no_grad_params"] = [k for k, p in model.named_parameters() if p.grad is None]

    # inference glue + post-processing through the reference's own compute_mr_results
    model.eval()
    with torch.no_grad():
        npify("evalout/", model(**inputs), store)
    if tal:      # eval-mode criterion without cls_idx: the inter term only (model/univtg.py:312-313)
        with torch.no_grad():
            le = crit(model(**inputs), {k: v for k, v in crit_targets.items() if k != "cls_idx"})
        meta["eval_loss_keys"] = sorted(le.keys())
        store["evalloss/loss_s_inter"] = le["loss_s_inter"].numpy()
    durations = [float(inputs["src_vid_mask"][b].sum()) * 2.0 for b in range(inputs["src_vid"].shape[0])]
    qmeta = [dict(qid=b, query=f"q{b}", vid=f"v{b}", duration=durations[b]) for b in range(len(durations))]
    m = inputs["src_vid_mask"]
    batched = dict(query_feat=(inputs["src_txt"], inputs["src_txt_mask"]),
                   video_feat=(inputs["src_vid"], m),
                   timestamp=(targets["timestamp"], targets["timestamp_mask"]),
                   timestamp_window=(targets["timestamp_window"], m),
                   span_labels_nn=(targets["span_labels_nn"], m),
                   saliency_scores=(targets["saliency_scores"], m),
                   saliency_pos_labels=targets["saliency_pos_labels"],
                   saliency_neg_labels=targets["saliency_pos_labels"])
    for rm, tag in ((-1, "raw"), (1, "rounded")):
        opt = Namespace(device="cpu", pin_memory=False, span_loss_type="l1", model_id="univtg",
                        eval_mode="add", no_sort_results=False, debug=False, clip_length=2.0,
                        round_multiple=rm)
        mr_res, _ = ref.inf.compute_mr_results(model, [(qmeta, batched)], opt)
        pre = [e["pred_relevant_windows"] for e in mr_res]
        sal = [e["pred_saliency_scores"] for e in mr_res]
        post = ref.inf.post_processing_mr_nms([dict(e) for e in mr_res], nms_thd=0.7,
                                              max_before_nms=1000, max_after_nms=10)
        meta[f"post/{tag}"] = dict(pre=pre, sal=sal, nms=[e["pred_relevant_windows"] for e in post])
    store["meta"] = np.array(json.dumps(meta))
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **store)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB, total loss {float(total):.6f}")


def run_matcher(ref):
    """HungarianMatcher on Moment-DETR-shaped inputs (model/matcher.py:36-100)."""
    g = torch.Generator().manual_seed(7)
    B, Q = 6, 10
    logits = torch.randn(B, Q, 2, generator=g)
    cxw = torch.stack([torch.rand(B, Q, generator=g), 0.05 + 0.4 * torch.rand(B, Q, generator=g)], -1)
    sizes = [1, 2, 5, 3, 1, 4]
    tgt = [dict(spans=torch.stack([torch.rand(n, generator=g), 0.05 + 0.5 * torch.rand(n, generator=g)], -1))
           for n in sizes]
    matcher = ref.matcher.HungarianMatcher(cost_class=4, cost_span=10, cost_giou=1)
    idx = matcher(dict(pred_logits=logits, pred_spans=cxw), dict(span_labels=tgt))
    store = dict(logits=logits.numpy(), spans=cxw.numpy(), sizes=np.array(sizes),
                 tgt=torch.cat([t["spans"] for t in tgt]).numpy())
    for b, (i, j) in enumerate(idx):
        store[f"i{b}"] = i.numpy()
        store[f"j{b}"] = j.numpy()
    # UniVTG-shaped call (pred_logits last dim 1): class cost is constant -1 (SURVEY finding, a17)
    logits1 = torch.rand(B, Q, 1, generator=g)
    idx1 = matcher(dict(pred_logits=logits1, pred_spans=cxw), dict(span_labels=tgt))
    store["logits1"] = logits1.numpy()
    for b, (i, j) in enumerate(idx1):
        store[f"u_i{b}"] = i.numpy()
        store[f"u_j{b}"] = j.numpy()
    np.savez_compressed(os.path.join(OUT, "matcher.npz"), **store)
    print("matcher: ok")


def run_detr_criterion(ref):
    """Moment-DETR SetCriterion (model/moment_detr.py:166-365) on the real matcher's pairs: the six loss values, and the
    autograd gradients of the weighted total with respect to every prediction tensor."""
    import model.moment_detr as ref_detr
    g = torch.Generator().manual_seed(23)
    B, Q, L, T, D, P = 6, 10, 17, 9, 32, 2
    sizes = [1, 2, 5, 3, 1, 4]
    leaf = lambda t: t.clone().requires_grad_(True)
    logits = leaf(torch.randn(B, Q, 2, generator=g))
    spans = leaf(torch.stack([torch.rand(B, Q, generator=g), 0.05 + 0.4 * torch.rand(B, Q, generator=g)], -1))
    sal = leaf(torch.randn(B, L, generator=g))
    pq = leaf(torch.nn.functional.normalize(torch.randn(B, Q, D, generator=g), dim=-1))
    pt = leaf(torch.nn.functional.normalize(torch.randn(B, T, D, generator=g), dim=-1))
    tgt = [dict(spans=torch.stack([torch.rand(n, generator=g), 0.05 + 0.5 * torch.rand(n, generator=g)], -1)) for n in sizes]
    pos = torch.randint(0, L, (B, P), generator=g)
    neg = torch.randint(0, L, (B, P), generator=g)
    matcher = ref.matcher.HungarianMatcher(cost_class=4, cost_span=10, cost_giou=1)
    weight = dict(loss_b=10.0, loss_g=1.0, loss_f=4.0, loss_s_intra=1.0, loss_contrastive_align=0.02)
    crit = ref_detr.SetCriterion(matcher=matcher, weight_dict=weight, eos_coef=0.1,
                                 losses=["spans", "labels", "saliency", "contrastive_align"], temperature=0.07,
                                 span_loss_type="l1", max_v_l=75, saliency_margin=0.2)
    outputs = dict(pred_logits=logits, pred_spans=spans, saliency_scores=sal, proj_queries=pq, proj_txt_mem=pt)
    targets = dict(span_labels=tgt, saliency_pos_labels=pos, saliency_neg_labels=neg)
    losses = crit(outputs, targets)
    total = sum(losses[k] * weight[k] for k in weight)
    total.backward()
    idx = matcher(dict(pred_logits=logits.detach(), pred_spans=spans.detach()), dict(span_labels=tgt))
    store = dict(logits=logits.detach().numpy(), spans=spans.detach().numpy(), sal=sal.detach().numpy(),
                 pq=pq.detach().numpy(), pt=pt.detach().numpy(), sizes=np.array(sizes),
                 tgt=torch.cat([t["spans"] for t in tgt]).numpy(), pos=pos.numpy(), neg=neg.numpy(),
                 weights=np.array([weight["loss_b"], weight["loss_g"], weight["loss_f"], 0.0, weight["loss_s_intra"],
                                   weight["loss_contrastive_align"]], np.float32),
                 hyper=np.array([0.1, 0.07, 0.2], np.float32),
                 losses=np.array([float(losses[k]) for k in ("loss_b", "loss_g", "loss_f", "class_error", "loss_s_intra",
                                                              "loss_contrastive_align")], np.float64),
                 d_logits=logits.grad.numpy(), d_spans=spans.grad.numpy(), d_sal=sal.grad.numpy(), d_pq=pq.grad.numpy(),
                 d_pt=pt.grad.numpy())
    for b, (i, j) in enumerate(idx):
        store[f"i{b}"] = i.numpy()
        store[f"j{b}"] = j.numpy()
    np.savez_compressed(os.path.join(OUT, "detr_criterion.npz"), **store)
    print("detr_criterion: ok", store["losses"])


def run_features(ref):
    """Feature-file readers (main/dataset.py:325-358, :370-390) executed on temporary .npz files: inputs and the tensors the real
    DatasetVLP methods return."""
    import tempfile
    DS = ref.dataset.DatasetVLP
    rng = np.random.RandomState(31)
    sf, clip = rng.randn(23, 40).astype(np.float32) * 3, rng.randn(21, 24).astype(np.float64)
    q_last, q_pool = rng.randn(9, 16).astype(np.float32), rng.randn(16).astype(np.float32)
    store = dict(slowfast=sf, clip=clip, q_last=q_last, q_pool=q_pool)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            for sub, name, arrs in (("vid_slowfast", "v0", dict(features=sf)), ("vid_clip", "v0", dict(features=clip)),
                                    ("txt_clip", "7", dict(last_hidden_state=q_last, pooler_output=q_pool))):
                os.makedirs(os.path.join("data", "synthetic", sub), exist_ok=True)
                np.savez(os.path.join("data", "synthetic", sub, name + ".npz"), **arrs)
            ds = object.__new__(DS)
            ds.use_cache, ds.v_feat_types, ds.v_feat_dirs = 0, ["vid_slowfast", "vid_clip"], ["vid_slowfast", "vid_clip"]
            ds.q_feat_dir, ds.q_feat_dim, ds.txt_drop_ratio = "txt_clip", 16, 0
            meta = dict(vid="v0", qid=7, dset_name="synthetic", v_feat_suffix="", q_feat_suffix="")
            for norm in (True, False):
                ds.normalize_v = ds.normalize_t = norm
                store[f"video_{int(norm)}"] = ds._get_video_feat_by_vid(meta).numpy()
                for ft in ("last_hidden_state", "pooler_output"):
                    ds.q_feat_type = ft
                    store[f"query_{ft}_{int(norm)}"] = ds._get_query_feat_by_qid(meta).numpy()
            ds.q_feat_type, ds.normalize_t = "last_hidden_state", True
            store["query_missing"] = ds._get_query_feat_by_qid(dict(meta, qid=8)).numpy()
        finally:
            os.chdir(cwd)
    np.savez_compressed(os.path.join(OUT, "features.npz"), **store)
    print("features: ok", store["video_1"].shape, store["query_last_hidden_state_1"].shape)


def run_span_utils(ref):
    """Doctest known answers of utils/span_utils.py:13-20,32-39,55-61,106-110 + random matrices."""
    g = torch.Generator().manual_seed(3)
    a = torch.sort(torch.rand(9, 2, generator=g), dim=1).values
    b = torch.sort(torch.rand(5, 2, generator=g), dim=1).values
    iou, union = ref.span.temporal_iou(a, b)
    store = dict(a=a.numpy(), b=b.numpy(), iou=iou.numpy(), union=union.numpy(),
                 giou=ref.span.generalized_temporal_iou(a, b).numpy(),
                 cxw=ref.span.span_xx_to_cxw(a).numpy(),
                 xx=ref.span.span_cxw_to_xx(ref.span.span_xx_to_cxw(a)).numpy())
    d1 = torch.Tensor([[0, 0.2], [0.5, 1.0]])
    d2 = torch.Tensor([[0, 0.3], [0., 1.0]])
    store["doc_giou"] = ref.span.generalized_temporal_iou(d1, d2).numpy()
    store["doc_iou"] = ref.span.temporal_iou(d1, d2)[0].numpy()
    np.savez_compressed(os.path.join(OUT, "span_utils.npz"), **store)
    print("span_utils: ok")


def run_nms(ref):
    """temporal_nms + PostProcessorDETR(round_multiple) on the first rows of the bundled real
    predictions plot/qvhl/univtg.jsonl (SURVEY 8c golden 3)."""
    rows = []
    with open(os.path.join(REF, "plot", "qvhl", "univtg.jsonl")) as f:
        for i, line in enumerate(f):
            if i >= 40:
                break
            rows.append(json.loads(line))
    cases = []
    g = torch.Generator().manual_seed(11)
    for r in rows:
        wins = r["pred_relevant_windows"]
        # densify: jitter copies so that NMS has something to suppress
        dense = []
        for w in wins:
            for _ in range(4):
                j = (torch.rand(3, generator=g) - 0.5).tolist()
                st = max(0.0, w[0] + 6 * j[0])
                ed = max(st, w[1] + 6 * j[1])
                dense.append([float(f"{st:.4f}"), float(f"{ed:.4f}"), float(f"{max(0.0, w[2] + 0.1 * j[2]):.4f}")])
        for thd, mx in ((0.7, 10), (0.5, 5), (0.3, 100)):
            keep = ref.nms.temporal_nms([list(e) for e in dense], nms_thd=thd, max_after_nms=mx)
            cases.append(dict(inp=dense, thd=thd, max_after=mx, out=keep))
    pp = ref.post.PostProcessorDETR(clip_length=2.0, min_ts_val=0, max_ts_val=150, min_w_l=2, max_w_l=150,
                                    move_window_method="left", process_func_names=["round_multiple"])
    lines = [dict(pred_relevant_windows=[list(w) for w in c["inp"]]) for c in cases[:20]]
    rounded = pp([dict(pred_relevant_windows=[list(w) for w in l["pred_relevant_windows"]]) for l in lines])
    with open(os.path.join(OUT, "nms.json"), "w") as f:
        json.dump(dict(cases=cases, round_in=[l["pred_relevant_windows"] for l in lines],
                       round_out=[l["pred_relevant_windows"] for l in rounded]), f)
    print("nms: ok", len(cases))


def run_dense_targets(ref):
    """Pin oracle.dense_targets to DatasetVLP.__getitem__ (main/dataset.py:153-240)."""
    import random
    DS = ref.dataset.DatasetVLP
    ds = object.__new__(DS)
    ds.use_video, ds.use_tef, ds.load_labels = True, True, True
    ds.max_v_l, ds.clip_len, ds.max_windows, ds.span_loss_type = 75, 2, 5, "l1"
    ds.data_path, ds.dset_name = "synthetic", "synthetic"
    ds.add_easy_negative, ds.easy_negative_only = 1, 1
    cases = []
    g = torch.Generator().manual_seed(5)
    for n in range(24):
        lv = int(torch.randint(8, 76, (1,), generator=g))
        dur = lv * 2
        if n % 6 == 5:   # a window shorter than any clip centre spacing: 'not assigned' branch
            st = float(torch.rand(1, generator=g)) * (dur - 1)
            wins = [[st, st + 0.4]]
        else:
            k = 1 + n % 3
            wins = []
            for _ in range(k):
                st = float(torch.rand(1, generator=g)) * 0.7 * dur
                wins.append([st, min(dur, st + (0.05 + 0.25 * float(torch.rand(1, generator=g))) * dur)])
        ds.data = [dict(qid=n, query="q", duration=dur, vid="v", relevant_windows=[list(w) for w in wins],
                        dset_name="synthetic")]
        ds._get_query_feat_by_qid = lambda meta: torch.zeros(3, 4)
        ds._get_video_feat_by_vid = lambda meta, lv=lv: torch.zeros(lv, 6)
        random.seed(n)
        item = DS.__getitem__(ds, 0)["model_inputs"]
        cases.append(dict(lv=lv, wins=wins,
                          timestamp=item["timestamp"].tolist(),
                          span_labels_nn=item["span_labels_nn"].tolist(),
                          timestamp_window=item["timestamp_window"].tolist(),
                          saliency_scores=[float(x) for x in item["saliency_scores"].tolist()],
                          span_labels=item["span_labels"].tolist(),
                          tef=item["video_feat"][:, -2:].tolist(),
                          pos=item["saliency_pos_labels"]))
    with open(os.path.join(OUT, "dense_targets.json"), "w") as f:
        json.dump(cases, f)
    print("dense_targets: ok", len(cases))


def collate_samples(seed=7, B=5, Dv=34, Dt=24):
    """Ragged synthetic samples in the layout DatasetVLP/DatasetMR.__getitem__ returns (main/dataset.py:153-240)."""
    g = torch.Generator().manual_seed(seed)
    batch = []
    for i in range(B):
        lv = int(torch.randint(3, 14, (1,), generator=g)); lt = int(torch.randint(1, 8, (1,), generator=g))
        ng = int(torch.randint(1, 3, (1,), generator=g))
        mi = dict(query_feat=torch.randn(lt, Dt, generator=g), video_feat=torch.randn(lv, Dv, generator=g),
                  timestamp=torch.rand(lv, 2, generator=g), timestamp_window=(torch.rand(lv, generator=g) > 0.5).float(),
                  span_labels_nn=torch.rand(lv, 2, generator=g), saliency_scores=torch.rand(lv, generator=g).double(),
                  span_labels=torch.rand(ng, 2, generator=g), saliency_pos_labels=[int(torch.randint(0, lv, (1,), generator=g))],
                  saliency_neg_labels=[int(torch.randint(0, lv, (1,), generator=g))])
        batch.append(dict(meta=dict(qid=i, duration=2.0 * lv), model_inputs=mi))
    return batch


def run_collate(ref):
    """Golden vectors for the wire-format row of SURVEY 8f: start_end_collate_mr + prepare_batch_inputs_mr
    (main/dataset.py:1037-1052,1071-1100; utils/tensor_utils.py:5-53) on ragged samples."""
    batch = collate_samples()
    meta, batched = ref.dataset.start_end_collate_mr(batch)
    model_inputs, targets = ref.dataset.prepare_batch_inputs_mr(batched, "cpu")
    store = {}
    for i, e in enumerate(batch):
        for k, v in e["model_inputs"].items():
            store[f"sample/{i}/{k}"] = np.asarray(v if not torch.is_tensor(v) else v.numpy())
    for k, v in model_inputs.items():
        store["in/" + k] = v.numpy()
    for k, v in targets.items():
        if k == "span_labels":
            for i, d in enumerate(v):
                store[f"tg/span_labels/{i}"] = d["spans"].numpy()
        else:
            store["tg/" + k] = v.numpy()
    store["n"] = np.asarray(len(batch))
    np.savez_compressed(os.path.join(OUT, "collate.npz"), **store)
    print("collate: ok", {k: tuple(v.shape) for k, v in model_inputs.items()})


def run_feature_cache(ref):
    """SURVEY 8f row 2, cache path: the real ``DatasetVLP._get_video_feat_by_vid`` / ``_get_query_feat_by_qid`` with ``use_cache`` on
    (main/dataset.py:335-340,375-376) over in-memory caches of the shape ``DatasetVLP.__init__`` builds from the hdf5 files
    (main/dataset.py:113-131: ``{feat_type: {vid: array}}``, ``{qid: array}``)."""
    DS = ref.dataset.DatasetVLP
    rng = np.random.RandomState(23)
    sf = rng.randn(39, 24).astype(np.float32)          # stored the way data/create_h5py.py wrote them
    clip = rng.randn(37, 16).astype(np.float32)
    q = rng.randn(9, 16).astype(np.float32)
    ds = object.__new__(DS)
    ds.use_cache, ds.v_feat_types, ds.v_feat_dirs, ds.normalize_v = 1, ["vid_slowfast", "vid_clip"], ["unused_a", "unused_b"], True
    ds.vid_cache = {"vid_slowfast": {7: sf}, "vid_clip": {7: clip}}
    ds.txt_cache, ds.q_feat_dir, ds.q_feat_dim, ds.q_feat_type, ds.normalize_t, ds.txt_drop_ratio = {"q3": q}, "unused", 16, "last_hidden_state", True, 0
    meta = dict(dset_name="d", v_feat_suffix="", q_feat_suffix="", vid=7, qid="q3")
    store = {"slowfast": sf, "clip": clip, "q": q, "video": ds._get_video_feat_by_vid(meta).numpy(), "query": ds._get_query_feat_by_qid(meta).numpy(),
             "query_missing": ds._get_query_feat_by_qid(dict(meta, qid="absent")).numpy()}
    np.savez_compressed(os.path.join(OUT, "features_cache.npz"), **store)
    print("features_cache: ok", {k: v.shape for k, v in store.items()})


def run_branch_cases(ref, tiny):
    """Round 4: the reference branches the earlier fixtures never took -- the criterion's loss subset for dset_type 'hl' / 'vs' and the two
    early-outs of loss_saliency (model/univtg.py:237-241,439-440), --n_input_proj 1 / 3 (model/univtg.py:89-100) and --use_txt_pos
    (model/position_encoding.py:19-41, model/univtg.py:123)."""
    run_case(ref, "tiny_hl", O.make_cfg(**{**tiny, "losses": ("labels", "saliency")}), B=5, L_v=13, L_t=7, seed=21, ragged=True,
             curve=True, dset_type="hl")
    run_case(ref, "tiny_zero_saliency", O.make_cfg(**tiny), B=5, L_v=13, L_t=7, seed=22, ragged=True, zero_saliency=True)
    run_case(ref, "tiny_no_pos_labels", O.make_cfg(**tiny), B=5, L_v=13, L_t=7, seed=23, ragged=True, drop_pos_labels=True)
    run_case(ref, "tiny_nproj1", O.make_cfg(**{**tiny, "n_input_proj": 1}), B=4, L_v=11, L_t=6, seed=24, ragged=True)
    run_case(ref, "tiny_nproj3", O.make_cfg(**{**tiny, "n_input_proj": 3}), B=4, L_v=11, L_t=6, seed=25, ragged=True)
    run_case(ref, "tiny_txt_pos", O.make_cfg(**{**tiny, "use_txt_pos": True}), B=5, L_v=12, L_t=9, seed=26, ragged=True)
    run_tal_case(ref, tiny)


def run_tal_case(ref, tiny):
    """Round 6: the TAL pre-training branch -- src_cls / src_cls_mask through Model.forward and the 'saliency_cls' loss
    (model/univtg.py:109-117,151-153,284-326,436-438; build_model selects it with 'tal' in train_path)."""
    run_case(ref, "tiny_tal", O.make_cfg(**{**tiny, "losses": ("spans", "labels", "saliency_cls")}), B=5, L_v=13, L_t=7, seed=27, ragged=True,
             curve=True, tal=dict(n_cls=6, L_c=4))


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = import_reference()
    tiny = dict(hidden_dim=64, nheads=2, dim_feedforward=96, enc_layers=2, v_feat_dim=34, t_feat_dim=24,
                max_q_l=16, input_dropout=0.0, dropout=0.0, droppath=0.0)
    if sys.argv[1:] == ["branches"]:            # round-4 fixtures only (the others are unchanged)
        run_branch_cases(ref, tiny)
        return
    if sys.argv[1:] == ["tal"]:                 # round-6 fixture only
        run_tal_case(ref, tiny)
        return
    if sys.argv[1:] == ["detr_criterion"]:      # one fixture only (the others are unchanged)
        run_detr_criterion(ref)
        return
    if sys.argv[1:] == ["features"]:
        run_features(ref)
        return
    if sys.argv[1:] == ["feature_cache"]:
        run_feature_cache(ref)
        return
    tiny = dict(hidden_dim=64, nheads=2, dim_feedforward=96, enc_layers=2, v_feat_dim=34, t_feat_dim=24,
                max_q_l=16, input_dropout=0.0, dropout=0.0, droppath=0.0)
    run_case(ref, "tiny_eval_ragged", O.make_cfg(**tiny), B=5, L_v=13, L_t=7, seed=11, ragged=True)
    run_case(ref, "tiny_eval_full", O.make_cfg(**tiny), B=4, L_v=12, L_t=8, seed=12, ragged=False, curve=True)
    run_case(ref, "tiny_train_droppath", O.make_cfg(**{**tiny, "droppath": 0.4}), B=6, L_v=10, L_t=6,
             seed=13, ragged=True, train_droppath=True)
    mid = dict(hidden_dim=128, nheads=4, dim_feedforward=128, enc_layers=2, v_feat_dim=514, t_feat_dim=512,
               max_q_l=32, input_dropout=0.0, dropout=0.0, droppath=0.0)
    run_case(ref, "config1_real_feats", O.make_cfg(**mid), B=1, L_v=15, L_t=12, seed=2018, ragged=False,
             real_feats=True)
    run_branch_cases(ref, tiny)
    run_matcher(ref)
    run_detr_criterion(ref)
    run_features(ref)
    run_feature_cache(ref)
    run_span_utils(ref)
    run_nms(ref)
    run_dense_targets(ref)
    run_collate(ref)


if __name__ == "__main__":
    main()
