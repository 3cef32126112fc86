"""Model-level parity on a real MI355X, through the drop-in boundary (build_model / forward / criterion):
  * HIP path vs the REAL reference's outputs/losses/gradients (tests/golden, made by oracle/make_golden.py)
  * HIP path vs the CPU oracle at the production width on seeded inputs
  * post-NMS index equality, matcher indices, size-independent properties at BASELINE config-2 size."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = ["tiny_eval_ragged", "tiny_eval_full", "config1_real_feats",
         # round 4 fixtures from the real reference: dset_type 'hl' loss subset, the two loss_saliency early-outs (model/univtg.py:237-241,
         # 439-440), n_input_proj 1 / 3 (model/univtg.py:89-100), --use_txt_pos (model/position_encoding.py:19-41)
         "tiny_hl", "tiny_zero_saliency", "tiny_no_pos_labels", "tiny_nproj1", "tiny_nproj3", "tiny_txt_pos",
         # round 6: the TAL pre-training branch -- src_cls / src_cls_mask through Model.forward, the 'saliency_cls' loss (model/univtg.py:109-117,151-153,284-326)
         "tiny_tal"]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def args_from_cfg(cfg, **over):
    a = dict(device="cuda", hidden_dim=cfg.hidden_dim, dropout=cfg.dropout, droppath=cfg.droppath, nheads=cfg.nheads,
             dim_feedforward=cfg.dim_feedforward, enc_layers=cfg.enc_layers, dec_layers=2, pre_norm=False,
             position_embedding="sine", max_q_l=cfg.max_q_l, input_dropout=cfg.input_dropout, t_feat_dim=cfg.t_feat_dim,
             v_feat_dim=cfg.v_feat_dim, span_loss_type="l1", use_txt_pos=bool(getattr(cfg, "use_txt_pos", False)), n_input_proj=cfg.n_input_proj,
             set_cost_span=10, set_cost_giou=1, set_cost_class=4, max_v_l=75, b_loss_coef=cfg.b_loss_coef,
             g_loss_coef=cfg.g_loss_coef, f_loss_coef=cfg.f_loss_coef, s_loss_intra_coef=cfg.s_loss_intra_coef,
             s_loss_inter_coef=cfg.s_loss_inter_coef, dset_type="vlp" if "spans" in cfg.losses else "hl",
             train_path=["tal"] if "saliency_cls" in cfg.losses else ["synthetic"], eos_coef=cfg.eos_coef,
             temperature=0.07, saliency_margin=0.2)
    a.update(over)
    return SimpleNamespace(**a)


def load_case(golden_dir, name):
    from oracle import univtg_oracle as O
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    cfg = O.make_cfg(**meta["cfg"])
    grab = lambda pre: {k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)}
    tg = grab("tg/")
    if meta.get("drop_pos_labels"):                 # the fixture's criterion call had no saliency_pos_labels (model/univtg.py:237-238)
        tg.pop("saliency_pos_labels")
    meta["dout"] = grab("dout/")                    # reference gradients of the weighted total wrt the criterion's inputs
    return meta, cfg, grab("param/"), grab("in/"), tg, grab("out/"), grab("evalout/"), grab("grad/"), \
        {k[5:]: float(z[k]) for k in z.files if k.startswith("loss/")}


def build(cfg, params, dev, precision, proj_precise=True):
    """proj_precise=True keeps the input projections fp32-class even under autograd, so that the saliency losses can be
    pinned tightly; the default of the product ("auto": plain bf16 projections whenever a backward follows) is covered by
    test_auto_projection_mode_training."""
    from univtg_amd.model import build_model
    model, crit = build_model(args_from_cfg(cfg, precision=precision, proj_precise=proj_precise))
    missing = model.load_state_dict(params, strict=True)           # the reference's checkpoint layout loads as is
    assert not missing.missing_keys and not missing.unexpected_keys
    return model.to(dev), crit.to(dev)


def to_dev(d, dev):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()}


@pytest.mark.parametrize("name", CASES)
def test_forward_fp32x3_matches_reference(dev, golden_dir, name):
    meta, cfg, params, inputs, tg, out_ref, eval_ref, *_ = load_case(golden_dir, name)
    model, _ = build(cfg, params, dev, "fp32x3")
    model.eval()
    with torch.no_grad():
        out = model(**to_dev(inputs, dev))
    valid = inputs["src_vid_mask"].bool()
    sal = out["saliency_scores"].cpu()
    # north_star tolerance: saliency logits within 1e-4 of the reference CPU path (valid clips; padded = log-mask constant)
    assert float((sal - eval_ref["saliency_scores"])[valid].abs().max()) < 1e-4
    assert float((sal - eval_ref["saliency_scores"])[~valid].abs().max() if (~valid).any() else 0.0) < 1e-3
    for k, tol in (("pred_logits", 2e-4), ("pred_spans", 2e-4), ("vid_mem_proj", 2e-4), ("txt_mem_proj", 2e-4)) + ((("cls_mem_proj", 2e-4),) if "cls_mem_proj" in eval_ref else ()):
        err = float((out[k].cpu() - eval_ref[k]).abs().max())
        assert err < tol, (k, err)
    assert ("cls_mem_proj" in out) == ("src_cls" in inputs)


@pytest.mark.parametrize("name", CASES)
def test_post_nms_indices_bit_exact(dev, golden_dir, name):
    """north_star: span indices after NMS identical to the reference CPU path (fp32x3 inference mode)."""
    from oracle import postproc_oracle as P
    from univtg_amd import ops
    meta, cfg, params, inputs, tg, out_ref, eval_ref, *_ = load_case(golden_dir, name)
    model, _ = build(cfg, params, dev, "fp32x3")
    model.eval()
    with torch.no_grad():
        out = model(**to_dev(inputs, dev))
    B = inputs["src_vid"].shape[0]
    durations = torch.tensor([float(inputs["src_vid_mask"][b].sum()) * 2.0 for b in range(B)])
    ref_order = P.ranked_clip_indices(eval_ref["pred_logits"].numpy(), tg["timestamp_mask"].numpy())
    win, order, keep, nk = ops.decode_rank_nms(out["pred_logits"], out["pred_spans"], tg["timestamp"].to(dev),
                                               tg["timestamp_mask"].to(dev), durations.to(dev), 0.7, 1000, 10)
    order, keep, nk, win = order.cpu().tolist(), keep.cpu().tolist(), nk.cpu().tolist(), win.cpu()
    ref_pre = meta["post/raw"]["pre"]
    ref_nms = meta["post/raw"]["nms"]
    for b in range(B):
        assert order[b] == ref_order[b], f"ranking differs for sample {b}"
        # reference post-NMS rows -> their rank positions
        ref_keep = []
        rows = [tuple(r) for r in ref_pre[b]]
        used = set()
        for r in ref_nms[b]:
            idx = next(i for i, rr in enumerate(rows) if rr == tuple(r) and i not in used)
            used.add(idx)
            ref_keep.append(idx)
        assert keep[b][: nk[b]] == ref_keep, f"NMS keep-set differs for sample {b}"
        got_rows = np.array([win[b, i].tolist() for i in keep[b][: nk[b]]])
        assert float(np.abs(got_rows - np.array(ref_nms[b])).max()) <= 2e-3      # window seconds (4-dp rounded)


@pytest.mark.parametrize("name", CASES)
def test_forward_bf16_close_to_reference(dev, golden_dir, name):
    meta, cfg, params, inputs, tg, out_ref, eval_ref, *_ = load_case(golden_dir, name)
    model, _ = build(cfg, params, dev, "bf16")
    model.eval()
    with torch.no_grad():
        out = model(**to_dev(inputs, dev))
    valid = inputs["src_vid_mask"].bool()
    # saliency does not pass through the bf16 encoder (SURVEY finding 2): still 1e-4
    assert float((out["saliency_scores"].cpu() - eval_ref["saliency_scores"])[valid].abs().max()) < 1e-4
    for k, tol in (("pred_logits", 3e-2), ("pred_spans", 3e-2)):
        err = float((out[k].cpu() - eval_ref[k]).abs().max())
        assert err < tol, (k, err)


@pytest.mark.parametrize("name", CASES)
def test_losses_and_grads_match_reference(dev, golden_dir, name):
    meta, cfg, params, inputs, tg, out_ref, eval_ref, grads_ref, losses_ref = load_case(golden_dir, name)
    model, crit = build(cfg, params, dev, "bf16")
    model.eval()                                                   # eval = no dropout; gradients still flow
    tgd = to_dev(tg, dev)
    out = model(**to_dev(inputs, dev))
    losses = crit(out, tgd)
    wd = crit.weight_dict
    total = sum(losses[k] * wd[k] for k in losses if k in wd)
    total.backward()
    assert set(losses) == set(losses_ref) - {"total"}, (set(losses), set(losses_ref))      # dset_type 'hl': no loss_b / loss_g
    for k in losses:
        got, ref = float(losses[k]), losses_ref[k]
        tol = 2e-2 * max(1.0, abs(ref)) if k in ("loss_b", "loss_g", "loss_f") else 2e-4 * max(1.0, abs(ref))
        assert abs(got - ref) < tol, (k, got, ref)
    for k in meta.get("loss_is_float", []):         # the reference's early-outs return 0.0
        assert float(losses[k]) == 0.0
    named = dict(model.named_parameters())
    # bf16 operands vs the fp32 reference: per-parameter direction (cosine) and magnitude (norm ratio) --
    # element-wise max error is dominated by bf16 rounding noise at these tiny widths (d=64/128)
    bad = {}
    for k, g in grads_ref.items():
        assert named[k].grad is not None, k
        a, r = named[k].grad.cpu().double().flatten(), g.double().flatten()
        cos = float((a @ r) / (a.norm() * r.norm() + 1e-30))
        ratio = float(a.norm() / (r.norm() + 1e-30))
        # tiny_hl: only the class head feeds the encoder (no span losses), gradients 5x smaller; at d = 64 three biases land at 6-7.5 %
        # (bf16 noise at toy width: the same loss subset at production width is within 0.3 % / cosine 0.9993, test_hl_loss_subset_production_width)
        if cos < 0.985 or abs(ratio - 1) > (0.09 if name == "tiny_hl" else 0.05):
            bad[k] = (cos, ratio)
    assert not bad, bad
    # parameters the reference leaves without a gradient: none here either, or (flat gradient buffer) exactly zero
    for k in meta["no_grad_params"]:
        assert named[k].grad is None or float(named[k].grad.abs().max()) == 0.0, k
    assert {k for k, p in named.items() if p.grad is None} <= set(meta["no_grad_params"])


def test_tal_branch_criterion_kernels_and_early_outs(dev, golden_dir):
    """The 'saliency_cls' criterion alone on the reference's own outputs (tiny_tal fixture): values and the gradients with respect to the
    criterion's inputs incl. cls_mem_proj (pins uvtg_cls_nce_fwd / _bwd apart from the model), the evaluation form without cls_idx (inter
    term only, model/univtg.py:312-313) and both early-outs (:286-290) as zero tensors."""
    meta, cfg, params, inputs, tg, out_ref, eval_ref, grads_ref, losses_ref = load_case(golden_dir, "tiny_tal")
    _, crit = build(cfg, params, dev, "bf16")
    assert list(crit.losses) == ["spans", "labels", "saliency_cls"]
    leaves = {k: out_ref[k].clone().to(dev).requires_grad_(True) for k in ("pred_logits", "pred_spans", "vid_mem_proj", "txt_mem_proj", "cls_mem_proj")}
    tgd = to_dev(tg, dev)
    losses = crit(dict(leaves), tgd)
    for k in ("loss_b", "loss_g", "loss_f", "loss_s_inter", "loss_s_intra"):
        assert abs(float(losses[k]) - losses_ref[k]) <= 2e-5 * max(1.0, abs(losses_ref[k])), (k, float(losses[k]), losses_ref[k])
    sum(losses[k] * crit.weight_dict[k] for k in losses).backward()
    for k, v in leaves.items():
        ref = meta["dout"][k]
        assert float((v.grad.cpu() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-8, k
    # evaluation: no cls_idx in the targets -> the inter-video term only
    z = np.load(os.path.join(golden_dir, "tiny_tal.npz"))
    ev = crit({k: eval_ref[k].to(dev) for k in eval_ref}, {k: v for k, v in tgd.items() if k != "cls_idx"})
    assert sorted(ev) == meta["eval_loss_keys"] and abs(float(ev["loss_s_inter"]) - float(z["evalloss/loss_s_inter"])) <= 2e-5
    # early-outs: both keys, both exactly zero, no gradient
    for tg2 in ({k: v for k, v in tgd.items() if k != "saliency_pos_labels"}, dict(tgd, saliency_scores=torch.zeros_like(tgd["saliency_scores"]))):
        lv = {k: out_ref[k].clone().to(dev).requires_grad_(True) for k in leaves}
        l2 = crit(dict(lv), tg2)
        assert float(l2["loss_s_inter"]) == 0.0 and float(l2["loss_s_intra"]) == 0.0
        (l2["loss_s_inter"] + l2["loss_s_intra"]).backward()
        assert lv["cls_mem_proj"].grad is None or float(lv["cls_mem_proj"].grad.abs().max()) == 0.0


def test_tal_branch_production_width_vs_oracle(dev):
    """The TAL branch at production width (d = 1024, E = 4, D_v = 2818; 24 class names of up to 6 tokens, B = 32 ragged): cls_mem_proj through
    the engine's text path within the saliency-class tolerance (fp32-class projections), the five losses, and the gradients the class term
    sends into the shared text projection / type embedding / pool -- against the oracle (itself pinned to the real reference by tiny_tal)."""
    from oracle import univtg_oracle as O
    from oracle.make_golden import make_cls_inputs
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    cfg = O.make_cfg(input_dropout=0.0, dropout=0.0, droppath=0.0, losses=("spans", "labels", "saliency_cls"))
    params = O.init_params(cfg, seed=81)
    inputs, tg = O.make_batch(cfg, 32, 75, 32, seed=82, ragged=True)
    inputs["src_cls"], inputs["src_cls_mask"], tg["cls_idx"] = make_cls_inputs(cfg, 32, 24, 6, 83)
    p2 = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = O.forward(p2, cfg, **inputs)
    lref = O.criterion(ref, tg, cfg)
    O.total_loss(lref, cfg).backward()
    model, crit = build(cfg, params, dev, "bf16", proj_precise=True)
    model.eval()
    out = model(**to_dev(inputs, dev))
    assert float((out["cls_mem_proj"].detach().cpu() - ref["cls_mem_proj"].detach()).abs().max()) < 1e-4
    ld = crit(out, to_dev(tg, dev))
    sum(ld[k] * crit.weight_dict[k] for k in ld).backward()
    assert set(ld) == set(lref)
    for k in ("loss_s_inter", "loss_s_intra"):
        assert abs(float(ld[k]) - float(lref[k])) < 2e-4 * max(1.0, abs(float(lref[k]))), (k, float(ld[k]), float(lref[k]))
    for k in ("loss_b", "loss_g", "loss_f"):
        assert abs(float(ld[k]) - float(lref[k])) < 3e-2 * max(1.0, abs(float(lref[k]))), (k, float(ld[k]), float(lref[k]))
    named = dict(model.named_parameters())
    for k in ("weightedpool.weight", "token_type_embeddings.weight", "input_txt_proj.0.net.1.weight", "input_txt_proj.1.net.1.weight",
              "input_txt_proj.0.LayerNorm.weight", "input_txt_proj.1.net.1.bias"):
        a, r = named[k].grad.cpu().double().flatten(), p2[k].grad.double().flatten()
        cos, ratio = float((a @ r) / (a.norm() * r.norm() + 1e-30)), float(a.norm() / (r.norm() + 1e-30))
        assert cos > 0.995 and abs(ratio - 1) < 0.04, (k, cos, ratio)


def test_nt_loader_waves_do_not_change_the_train_step(dev):
    """The persistent NT GEMM's staging by one wave per SIMD (default) against every wave staging for itself, through a whole bf16 train-mode
    forward + criterion + backward at production width (the specialised bf16 / FFN / GELU' epilogues only the engine reaches; B = 192 puts
    the launches on 128 - 256-row tiles of the persistent kernel): outputs bit for bit, and every parameter gradient that is bit-stable from
    run to run."""
    from oracle import univtg_oracle as O
    from univtg_amd import _lib
    lib = _lib.load()
    cfg = O.make_cfg(input_dropout=0.5, droppath=0.1, dropout=0.1)
    params = O.init_params(cfg, seed=61)
    inputs, tg = O.make_batch(cfg, 192, 75, 32, seed=62, ragged=True, curve=True)
    ind, tgd = to_dev(inputs, dev), to_dev(tg, dev)
    res = []
    try:
        for mask in (0, 0, 0, 7):
            _lib.check(lib.uvtg_debug_nt_loader_waves(mask))
            model, crit = build(cfg, params, dev, "bf16", proj_precise=False)
            model.train(); crit.train(); model.set_seed(77)
            out = model(**ind)
            ld = crit(out, tgd)
            sum(ld[k] * crit.weight_dict[k] for k in ld).backward()
            res.append(({k: out[k].detach().clone() for k in ("pred_logits", "pred_spans", "saliency_scores")},
                        {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    finally:
        lib.uvtg_debug_nt_loader_waves(7)
    for k, v in res[0][0].items():
        assert torch.equal(v, res[3][0][k]), k
    # gradients: whatever is bit-stable from run to run of the SAME build (everything but the sums that meet through fp32 atomics: LayerNorm
    # gamma / beta and the like) must not move either.  (Three identical runs decide what is stable: with two, a two-element atomic sum --
    # span_embed.layers.2.bias -- once matched by luck and then "moved" in the third run: the round-5 artifact visit.)
    stable = [k for k, v in res[0][1].items() if torch.equal(v, res[1][1][k]) and torch.equal(v, res[2][1][k])]
    assert len(res[0][1]) > 70 and len(stable) >= 40, (len(res[0][1]), len(stable))
    assert any(k.endswith("linear1.weight") for k in stable) and any("in_proj_weight" in k for k in stable)
    for k, v in res[0][1].items():
        if k in stable and k.endswith(("weight",)) and v.numel() >= 1024:      # (matrices: assigned by the weight-gradient launches, no atomics)
            assert torch.equal(v, res[3][1][k]), k
        else:
            assert float((v - res[3][1][k]).abs().max()) <= 1e-4 * float(v.abs().max()) + 1e-9, k


def test_trainstep_survives_a_projection_mode_flip_between_steps(dev):
    """`Model.proj_precise` is a public attribute and changes the workspace / operand-cache LAYOUT at an unchanged
    (B, L_v, L_t) -- split operands take 4 B per element where bf16 takes 2.  A TrainStep that first ran bf16 projections and then precise ones
    must re-allocate (it used to write past both buffers); the flipped step must equal a fresh TrainStep's."""
    from oracle import univtg_oracle as O
    from univtg_amd.trainer import TrainStep
    cfg = O.make_cfg(hidden_dim=256, nheads=4, dim_feedforward=256, enc_layers=2, v_feat_dim=514, t_feat_dim=512,
                     input_dropout=0.5, dropout=0.0, droppath=0.1)
    params = O.init_params(cfg, seed=11)
    inputs, tg = O.make_batch(cfg, 16, 40, 12, seed=12, ragged=True)
    batch, tgd = to_dev(inputs, dev), to_dev(tg, dev)

    def fresh(mode):
        model, crit = build(cfg, params, dev, "auto", proj_precise=mode)
        model.train(); model.set_seed(5)
        return model, TrainStep(model, crit, grad_clip=0.1, packed=False)
    model, step = fresh(False)
    step.step(batch, tgd, optimize=False)
    small = (step.ws.numel(), step.wcache.numel())
    guard = torch.full((1 << 20,), 7, dtype=torch.uint8, device=dev)      # something for an overrun to land in
    model.proj_precise = True
    model.set_seed(5)
    l_flip = step.step(batch, tgd, optimize=False).clone()
    g_flip = step.grads.clone()
    torch.cuda.synchronize()
    assert step.ws.numel() > small[0] or step.wcache.numel() > small[1], "the precise layout is larger: the step must have re-allocated"
    assert bool((guard == 7).all())
    model2, step2 = fresh(True)
    l_new = step2.step(batch, tgd, optimize=False)
    assert torch.allclose(l_flip, l_new, rtol=1e-5, atol=1e-6), (l_flip, l_new)      # (not bitwise: a few loss / gradient sums meet through fp32 atomics)
    gg0, gg1 = g_flip.double(), step2.grads.double()
    assert float((gg0 @ gg1) / (gg0.norm() * gg1.norm())) > 0.999999


@pytest.mark.parametrize("packed", [False, "auto"], ids=["padded", "packed"])
def test_attention_delta_from_the_dgrad_epilogue_matches_its_own_pass(dev, packed):
    """Round 5: attention backward's delta = rowsum_head(dO * O) is produced by the out-projection dgrad GEMM's epilogue (EPI 4: O as the epilogue
    operand, two fp32 atomics per (row, head) at head_dim 128) instead of attn_delta_kernel's pass.  Same rounded dO, same O: the whole train step's
    gradients must agree with the separate pass to fp32 summation order -- padded rows and the packed (ragged) stream with its row tables."""
    from oracle import univtg_oracle as O
    from univtg_amd import _lib
    from univtg_amd.trainer import TrainStep
    lib = _lib.load()
    cfg = O.make_cfg(input_dropout=0.5, dropout=0.0, droppath=0.1)
    params = O.init_params(cfg, seed=21)
    inputs, tg = O.make_batch(cfg, 48, 75, 32, seed=22, ragged=True)
    batch, tgd = to_dev(inputs, dev), to_dev(tg, dev)
    batch["_lens_host"] = (inputs["src_vid_mask"].sum(1).int().tolist(), inputs["src_txt_mask"].sum(1).int().tolist())
    res = []
    try:
        for fuse in (1, 0):
            _lib.check(lib.uvtg_debug_delta_fuse(fuse))
            model, crit = build(cfg, params, dev, "auto", proj_precise=False)
            model.train(); model.set_seed(9)
            step = TrainStep(model, crit, grad_clip=0.1, packed=packed)
            losses = step.step(batch, tgd, optimize=False)
            torch.cuda.synchronize()
            res.append((losses.clone(), step.grads.clone()))
    finally:
        lib.uvtg_debug_delta_fuse(1)
    assert torch.allclose(res[0][0], res[1][0], rtol=1e-6, atol=1e-7)          # (the forward is the same launch sequence)
    g1, g0 = res[0][1].double(), res[1][1].double()
    assert bool(torch.isfinite(g1).all())
    assert float((g1 @ g0) / (g1.norm() * g0.norm())) > 0.999999
    assert float((g1 - g0).abs().max()) <= 2e-3 * float(g0.abs().max())


@pytest.mark.parametrize("variant", ["droppath_attn_dropout", "txt_pos_512", "packed_halo", "packed_all_clips", "mid_batch_short_clip_groups"])
def test_last_layer_ffn_half_on_clip_rows_matches_every_row(dev, variant):
    """Round 5: nobody reads the text rows of the encoder output (`vid_mem = memory[:, :L_v]`, model/univtg.py:127), so the LAST layer's
    LayerNorm 1 -> linear1 -> GELU -> linear2 -> LayerNorm 2 -- and their gradients -- run on the B Lv clip rows only (engine.hip,
    last_layer_clip; LayerNorm 1's backward scatters into the token-major gradient stream whose text rows are zero).  Against the same step
    with every row computed (uvtg_debug_last_layer_clip(0)): the same predictions for the kept rows (row-wise work; a split-K tail tile may sum in another order),
    losses and gradients to fp32 summation order (the weight gradients reduce over fewer, differently grouped rows).  Also the eval call
    with the `memory` output (runs every row) and the refusal of a training call that asks for it."""
    from oracle import univtg_oracle as O
    from univtg_amd import _lib
    from univtg_amd.trainer import TrainStep
    lib = _lib.load()
    packed = "auto" if variant.startswith("packed") else False      # the packed (ragged) stream reads / scatters the clip rows through its row tables
    if variant == "txt_pos_512":
        cfg = O.make_cfg(hidden_dim=512, dim_feedforward=1024, enc_layers=2, input_dropout=0.5, dropout=0.0, droppath=0.0, use_txt_pos=True)
        B = 40
    elif variant == "packed_halo":                                  # loss-only stream: valid clips + 3-clip halo + valid text
        cfg = O.make_cfg(input_dropout=0.5, dropout=0.0, droppath=0.1)
        B = 48
    elif variant == "mid_batch_short_clip_groups":                  # B Lv = 1800 < 2048 <= B S: the clip-row FFN groups are below the hybrid
        cfg = O.make_cfg(input_dropout=0.5, dropout=0.0, droppath=0.1)      # weight-gradient kernel's row floor and leave on their own launches,
        B = 24                                                              # the other 18 groups keep the hybrid launch
    else:                                                           # attention dropout: every clip row (+ valid text when packed)
        cfg = O.make_cfg(input_dropout=0.5, dropout=0.1, droppath=0.1)
        B = 48
    params = O.init_params(cfg, seed=61)
    inputs, tg = O.make_batch(cfg, B, 75, 32, seed=62, ragged=True)
    batch, tgd = to_dev(inputs, dev), to_dev(tg, dev)
    batch["_lens_host"] = (inputs["src_vid_mask"].sum(1).int().tolist(), inputs["src_txt_mask"].sum(1).int().tolist())
    res, preds = [], []
    try:
        for clip in (1, 0):
            _lib.check(lib.uvtg_debug_last_layer_clip(clip))
            model, crit = build(cfg, params, dev, "auto", proj_precise=False)
            model.train(); model.set_seed(9)
            step = TrainStep(model, crit, grad_clip=0.1, packed=packed)
            losses = step.step(batch, tgd, optimize=False)
            torch.cuda.synchronize()
            res.append((losses.clone(), step.grads.clone()))
            preds.append({"pred_logits": step.pred_logits.clone(), "pred_spans": step.pred_spans.clone(), "saliency_scores": step.sal.clone()})
        # the bf16 inference call: clip rows only / every row / every row because `memory` is asked for
        model, _ = build(cfg, params, dev, "bf16")
        model.eval()
        model.packed = bool(packed) and not cfg.use_txt_pos              # (eval: valid clips + one representative padded clip + valid text)
        ebatch = {k: v for k, v in batch.items() if not k.startswith("_")}
        outs = []
        with torch.no_grad():
            for clip, mem in ((1, False), (0, False), (1, True)):
                _lib.check(lib.uvtg_debug_last_layer_clip(clip))
                model.return_memory = mem
                out = model(**ebatch)
                outs.append({k: out[k].clone() for k in out if torch.is_tensor(out[k])})
        assert outs[2]["memory"].shape == (B, 75 + 32, cfg.hidden_dim) and bool(torch.isfinite(outs[2]["memory"]).all())
        model.train()                                   # a training call cannot have the text rows of `memory`: refused, not silently wrong
        with pytest.raises(RuntimeError, match="memory output"):
            model(**ebatch)["pred_logits"].sum().backward()
    finally:
        lib.uvtg_debug_last_layer_clip(1)
    valid = inputs["src_vid_mask"].bool().to(dev)

    def close(a, b, what, tol=5e-3):      # the kept rows run the same row-wise arithmetic; only a split-K tail tile may sum in another order
        a, b = a.reshape(B, 75, -1)[valid].float(), b.reshape(B, 75, -1)[valid].float()
        assert float((a - b).abs().max()) <= tol * float(b.abs().max()) + 1e-6, (what, float((a - b).abs().max()), float(b.abs().max()))

    for k in preds[0]:
        close(preds[0][k], preds[1][k], "train " + k)
        close(outs[0][k], outs[1][k], "eval " + k)
        close(outs[2][k], outs[1][k], "eval+memory " + k, tol=2e-2)      # (the fp32 `memory` output takes the last LayerNorm off the lean kernel: bf16 rounding of another kernel)
    assert torch.allclose(res[0][0], res[1][0], rtol=2e-4, atol=1e-6), (res[0][0], res[1][0])
    g1, g0 = res[0][1].double(), res[1][1].double()
    assert bool(torch.isfinite(g1).all())
    assert float((g1 @ g0) / (g1.norm() * g0.norm())) > 0.99999
    assert float((g1 - g0).abs().max()) <= 4e-3 * float(g0.abs().max())


def test_backward_reads_the_last_layer_rows_the_way_its_forward_wrote_them(dev):
    """the clip-row layout of the last layer's FFN half depends on developer knobs, so uvtg_backward must not re-derive it --
    uvtg_forward records its choice per workspace.  A knob flipped BETWEEN a training forward and its backward (both directions) must leave the
    gradients bit-identical to the undisturbed call's."""
    from oracle import univtg_oracle as O
    from univtg_amd import _lib
    lib = _lib.load()
    cfg = O.make_cfg(input_dropout=0.0, dropout=0.0, droppath=0.0)
    params = O.init_params(cfg, seed=71)
    inputs, tg = O.make_batch(cfg, 24, 75, 32, seed=72, ragged=True)
    batch, tgd = to_dev(inputs, dev), to_dev(tg, dev)

    def grads(fwd_knob, bwd_knob):
        model, crit = build(cfg, params, dev, "bf16", proj_precise=False)
        model.train()
        _lib.check(lib.uvtg_debug_last_layer_clip(fwd_knob))
        out = model(**batch)
        ld = crit(out, tgd)
        _lib.check(lib.uvtg_debug_last_layer_clip(bwd_knob))
        sum(ld[k] * crit.weight_dict[k] for k in ld).backward()
        torch.cuda.synchronize()
        return torch.cat([p.grad.flatten() for _, p in sorted(model.named_parameters()) if p.grad is not None])
    try:
        for knob in (1, 0):
            want, got = grads(knob, knob).double(), grads(knob, 1 - knob).double()
            # (two runs of one build agree to the fp32 atomics' summation order; a backward that read compact clip-row buffers as full ones --
            # or the reverse -- reads stale rows of other tensors: errors of the gradients' own magnitude)
            assert bool(torch.isfinite(got).all()), knob
            assert float((want - got).abs().max()) <= 1e-4 * float(want.abs().max()), (knob, float((want - got).abs().max()), float(want.abs().max()))
            assert float((want @ got) / (want.norm() * got.norm())) > 0.9999999, knob
    finally:
        lib.uvtg_debug_last_layer_clip(1)


@pytest.mark.parametrize("packed", [False, "auto"], ids=["padded", "packed_halo"])
def test_conv_head_weight_gradients_inside_the_hybrid_launch_match_their_own_launch(dev, packed):
    """Round 5: the four Conv1d(k=3) weight gradients of the heads (dW[n][c][tap] = sum_rows dY[row][n] X[row + tap - 1][c] over the zero-framed
    rows, model/univtg.py:375-382) join the encoder's deferred weight-gradient launch: the hybrid kernel reads tap t's rows shifted by t - 1 and
    stores its tiles with the (d, d, 3) layout's stride -- no partial slabs, no reduce pass.  Against the slab + reduce launch
    (uvtg_debug_tn_conv_defer(0)): same operands, so every gradient agrees to fp32 summation order; uniform frames (padded stream) and the ragged
    frames of the loss-only stream."""
    from oracle import univtg_oracle as O
    from univtg_amd import _lib
    from univtg_amd.trainer import TrainStep
    lib = _lib.load()
    cfg = O.make_cfg(input_dropout=0.5, dropout=0.0, droppath=0.1)
    params = O.init_params(cfg, seed=71)
    inputs, tg = O.make_batch(cfg, 48, 75, 32, seed=72, ragged=True)
    batch, tgd = to_dev(inputs, dev), to_dev(tg, dev)
    batch["_lens_host"] = (inputs["src_vid_mask"].sum(1).int().tolist(), inputs["src_txt_mask"].sum(1).int().tolist())
    res = []
    try:
        for on in (1, 0):
            _lib.check(lib.uvtg_debug_tn_conv_defer(on))
            model, crit = build(cfg, params, dev, "auto", proj_precise=False)
            model.train(); model.set_seed(9)
            step = TrainStep(model, crit, grad_clip=0.1, packed=packed)
            losses = step.step(batch, tgd, optimize=False)
            torch.cuda.synchronize()
            res.append((losses.clone(), step.grads.clone()))
    finally:
        lib.uvtg_debug_tn_conv_defer(1)
    assert torch.allclose(res[0][0], res[1][0], rtol=1e-5, atol=1e-7)      # (the forward is the same launch sequence; a few loss sums meet in fp32 atomics)
    g1, g0 = res[0][1].double(), res[1][1].double()
    assert bool(torch.isfinite(g1).all())
    assert float((g1 - g0).abs().max()) <= 1e-4 * float(g0.abs().max()), float((g1 - g0).abs().max())
    assert float((g1 @ g0) / (g1.norm() * g0.norm())) > 0.9999999


def test_hl_loss_subset_production_width(dev):
    """dset_type 'hl' / 'vs' (losses = labels + saliency, model/univtg.py:439-440) at d = 1024, E = 4 against the oracle's fp32 autograd:
    every parameter gradient within 1.5 % in norm, cosine >= 0.998; span_embed gets no gradient."""
    from oracle import univtg_oracle as O
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    cfg = O.make_cfg(input_dropout=0.0, droppath=0.0, dropout=0.0, losses=("labels", "saliency"))
    params = O.init_params(cfg, seed=51)
    inputs, tg = O.make_batch(cfg, 16, 75, 32, seed=52, ragged=True, curve=True)
    p2 = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    lo = O.criterion(O.forward(p2, cfg, **inputs), tg, cfg)
    O.total_loss(lo, cfg).backward()
    model, crit = build(cfg, params, dev, "bf16")
    assert crit.losses == ["labels", "saliency"]
    model.eval()
    losses = crit(model(**to_dev(inputs, dev)), to_dev(tg, dev))
    assert set(losses) == {"loss_f", "loss_s_inter", "loss_s_intra"}
    sum(losses[k] * crit.weight_dict[k] for k in losses).backward()
    named = dict(model.named_parameters())
    bad = {}
    for k, p in p2.items():
        if k.startswith("txt_position_embed"):
            continue
        if p.grad is None or float(p.grad.abs().max()) == 0.0:
            assert named[k].grad is None or float(named[k].grad.abs().max()) == 0.0, k       # span_embed.*: no span loss
            continue
        a, r = named[k].grad.cpu().double().flatten(), p.grad.double().flatten()
        cos, ratio = float((a @ r) / (a.norm() * r.norm() + 1e-30)), float(a.norm() / (r.norm() + 1e-30))
        if cos < 0.998 or abs(ratio - 1) > 0.015:
            bad[k] = (cos, ratio)
    assert not bad, bad


@pytest.mark.parametrize("nproj,txt_pos", [(3, True), (1, False)])
def test_boundary_options_production_width(dev, nproj, txt_pos):
    """--n_input_proj 1 / 3 and --use_txt_pos at d = 1024, E = 4, D_v = 2818 (the persistent GEMM kernels, the wide feature LayerNorm, the
    per-layer position-gradient GEMM) against the oracle: precise forward within 1e-5, bf16 losses / gradients (incl. the text position table
    and its LayerNorm) in direction and norm."""
    from oracle import univtg_oracle as O
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    cfg = O.make_cfg(input_dropout=0.0, droppath=0.0, dropout=0.0, n_input_proj=nproj, use_txt_pos=txt_pos, max_q_l=40)
    params = O.init_params(cfg, seed=71)
    inputs, tg = O.make_batch(cfg, 24, 75, 32, seed=72, ragged=True)
    p2 = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = O.forward(p2, cfg, **inputs)
    lo = O.criterion(ref, tg, cfg)
    O.total_loss(lo, cfg).backward()
    m32, _ = build(cfg, params, dev, "fp32x3")
    m32.eval()
    with torch.no_grad():
        o32 = m32(**to_dev(inputs, dev))
    valid = inputs["src_vid_mask"].bool()
    assert float((o32["saliency_scores"].cpu() - ref["saliency_scores"].detach())[valid].abs().max()) < 1e-5
    for k in ("pred_logits", "pred_spans"):
        assert float((o32[k].cpu() - ref[k].detach()).abs().max()) < 1e-5, k
    model, crit = build(cfg, params, dev, "bf16")
    model.eval()
    losses = crit(model(**to_dev(inputs, dev)), to_dev(tg, dev))
    sum(losses[k] * crit.weight_dict[k] for k in losses).backward()
    for k in lo:
        assert abs(float(losses[k]) - float(lo[k])) < 2e-2 * max(1.0, abs(float(lo[k]))), (k, float(losses[k]), float(lo[k]))
    named, bad = dict(model.named_parameters()), {}
    for k, p in p2.items():
        if p.grad is None or float(p.grad.abs().max()) == 0.0:
            continue
        assert named[k].grad is not None, k
        a, r = named[k].grad.cpu().double().flatten(), p.grad.double().flatten()
        if k == "txt_position_embed.position_embeddings.weight":       # rows >= L_t are never read: zero gradient on both sides
            assert float(a.view(-1, cfg.hidden_dim)[32:].abs().max()) == 0.0
        cos, ratio = float((a @ r) / (a.norm() * r.norm() + 1e-30)), float(a.norm() / (r.norm() + 1e-30))
        if cos < 0.997 or abs(ratio - 1) > (0.04 if k == "input_vid_proj.0.LayerNorm.weight" else 0.02):
            bad[k] = (cos, ratio)
    assert not bad, bad
    assert ("txt_position_embed.LayerNorm.weight" in {k for k, p in p2.items() if p.grad is not None}) == txt_pos


def test_criterion_matches_oracle_fp32(dev, golden_dir):
    """The criterion kernels alone (fp32 math) on the reference's own outputs: losses + input gradients."""
    from oracle import univtg_oracle as O
    from univtg_amd.model import SetCriterion
    for name in CASES:
        meta, cfg, params, inputs, tg, out_ref, *_ = load_case(golden_dir, name)
        outs = {k: out_ref[k].clone().requires_grad_(True) for k in ("pred_logits", "pred_spans", "vid_mem_proj", "txt_mem_proj", "cls_mem_proj") if k in out_ref}
        lo = O.criterion(outs, tg, cfg)
        O.total_loss(lo, cfg).backward()
        crit = SetCriterion(None, O.weight_dict(cfg), cfg.eos_coef, list(cfg.losses), 0.07, "l1", 75).to(dev)
        outs_d = {k: out_ref[k].to(dev).requires_grad_(True) for k in outs}
        ld = crit(outs_d, to_dev(tg, dev))
        total = sum(ld[k] * crit.weight_dict[k] for k in ld)
        total.backward()
        assert set(ld) == set(lo), (name, set(ld), set(lo))
        for k in lo:
            assert abs(float(ld[k]) - float(lo[k])) < 3e-5 * max(1.0, abs(float(lo[k]))), (name, k, float(ld[k]), float(lo[k]))
        for k in outs:
            # against the oracle's autograd AND against the real reference's (fixture "dout/": criterion inputs as leaves)
            for tag, ref in (("oracle", outs[k].grad if outs[k].grad is not None else torch.zeros_like(outs[k])), ("reference", meta["dout"][k])):
                got = outs_d[k].grad.cpu() if outs_d[k].grad is not None else torch.zeros_like(ref)
                err = float((got - ref).abs().max()) / (float(ref.abs().max()) + 1e-12)
                assert err < 2e-4, (name, k, tag, err)


def test_matcher_matches_reference(dev, golden_dir):
    from univtg_amd.model import HungarianMatcher
    z = np.load(os.path.join(golden_dir, "matcher.npz"))
    sizes = z["sizes"].tolist()
    tg, off = [], 0
    for n in sizes:
        tg.append(dict(spans=torch.from_numpy(z["tgt"][off:off + n]).to(dev)))
        off += n
    m = HungarianMatcher(cost_class=4, cost_span=10, cost_giou=1)
    for lg, pre in ((z["logits"], ""), (z["logits1"], "u_")):
        res = m(dict(pred_logits=torch.from_numpy(lg).to(dev), pred_spans=torch.from_numpy(z["spans"]).to(dev)), dict(span_labels=tg))
        for b, (i, j) in enumerate(res):
            assert i.tolist() == z[f"{pre}i{b}"].tolist() and j.tolist() == z[f"{pre}j{b}"].tolist(), (pre, b)


def _detr_targets(z, dev):
    sizes = z["sizes"].tolist()
    off = np.concatenate([[0], np.cumsum(sizes)])
    return dict(span_labels=[dict(spans=torch.from_numpy(z["tgt"][off[b]:off[b + 1]]).to(dev)) for b in range(len(sizes))],
                saliency_pos_labels=torch.from_numpy(z["pos"]).to(dev), saliency_neg_labels=torch.from_numpy(z["neg"]).to(dev))


def test_detr_criterion_matches_reference(dev, golden_dir):
    """the Moment-DETR SetCriterion (ref model/moment_detr.py:166-365) through the device matcher and
    uvtg_detr_criterion -- loss values and every gradient against the real reference's autograd (golden), including the
    aux_outputs repetition against the oracle."""
    from univtg_amd.model import HungarianMatcher
    from univtg_amd.moment_detr import SetCriterion
    from oracle import postproc_oracle as P
    z = np.load(os.path.join(golden_dir, "detr_criterion.npz"))
    eos, temp, margin = (float(x) for x in z["hyper"])
    names = ("loss_b", "loss_g", "loss_f", "class_error", "loss_s_intra", "loss_contrastive_align")
    weight = {n: float(w) for n, w in zip(names, z["weights"]) if n != "class_error"}
    crit = SetCriterion(HungarianMatcher(cost_class=4, cost_span=10, cost_giou=1), weight, eos,
                        ["spans", "labels", "saliency", "contrastive_align"], temp, "l1", 75, saliency_margin=margin).to(dev)
    leaf = lambda k: torch.from_numpy(z[k]).to(dev).requires_grad_(True)
    outs = dict(pred_logits=leaf("logits"), pred_spans=leaf("spans"), saliency_scores=leaf("sal"), proj_queries=leaf("pq"),
                proj_txt_mem=leaf("pt"))
    targets = _detr_targets(z, dev)
    losses = crit(outs, targets)
    assert list(losses) == ["loss_b", "loss_g", "loss_f", "class_error", "loss_s_intra", "loss_contrastive_align"]
    for n, r in zip(names, z["losses"]):
        assert abs(float(losses[n]) - r) <= 2e-5 * max(1.0, abs(r)), (n, float(losses[n]), r)
    sum(losses[k] * weight[k] for k in weight).backward()
    for k, t in (("logits", "pred_logits"), ("spans", "pred_spans"), ("sal", "saliency_scores"), ("pq", "proj_queries"),
                 ("pt", "proj_txt_mem")):
        r = z["d_" + k]
        err = np.abs(outs[t].grad.cpu().numpy() - r).max()
        assert err <= 3e-5 * max(1.0, np.abs(r).max()), (k, err)
    # aux_outputs: same losses again on another layer's predictions, without the saliency term (ref :355-363)
    g = torch.Generator().manual_seed(5)
    aux = dict(pred_logits=torch.randn(6, 10, 2, generator=g).to(dev),
               pred_spans=torch.stack([torch.rand(6, 10, generator=g), 0.05 + 0.4 * torch.rand(6, 10, generator=g)], -1).to(dev),
               proj_queries=outs["proj_queries"].detach(), proj_txt_mem=outs["proj_txt_mem"].detach())
    both = crit({**{k: v.detach() for k, v in outs.items()}, "aux_outputs": [aux]}, targets)
    assert "loss_s_intra_0" not in both and "class_error_0" in both
    sizes = z["sizes"].tolist()
    off = np.concatenate([[0], np.cumsum(sizes)])
    tg = [z["tgt"][off[b]:off[b + 1]] for b in range(len(sizes))]
    lg, sp = aux["pred_logits"].cpu().numpy(), aux["pred_spans"].cpu().numpy()
    idx = P.hungarian_match(lg, sp, tg, w_class=4.0, w_span=10.0, w_giou=1.0)
    want, _ = P.detr_criterion(lg, sp, tg, idx, None, None, None, z["pq"], z["pt"], eos, temp, margin)
    for n, r in zip(names, want):
        if n != "loss_s_intra":
            assert abs(float(both[n + "_0"]) - r) <= 2e-5 * max(1.0, abs(r)), (n, float(both[n + "_0"]), r)


def test_detr_criterion_batch256_against_oracle(dev):
    """Same criterion at a training-size batch (B=256, Q=10, D=64), values and gradients against the fp64 oracle."""
    from univtg_amd.model import HungarianMatcher
    from univtg_amd.moment_detr import SetCriterion
    from oracle import postproc_oracle as P
    g = torch.Generator().manual_seed(41)
    B, Q, L, T, D, Pn = 256, 10, 75, 32, 64, 2
    sizes = torch.randint(1, 6, (B,), generator=g).tolist()
    tg = [torch.stack([torch.rand(n, generator=g), 0.05 + 0.5 * torch.rand(n, generator=g)], -1) for n in sizes]
    arr = dict(logits=torch.randn(B, Q, 2, generator=g),
               spans=torch.stack([torch.rand(B, Q, generator=g), 0.05 + 0.4 * torch.rand(B, Q, generator=g)], -1),
               sal=torch.randn(B, L, generator=g), pq=torch.nn.functional.normalize(torch.randn(B, Q, D, generator=g), dim=-1),
               pt=torch.nn.functional.normalize(torch.randn(B, T, D, generator=g), dim=-1))
    pos, neg = torch.randint(0, L, (B, Pn), generator=g), torch.randint(0, L, (B, Pn), generator=g)
    w = np.array([10.0, 1.0, 4.0, 0.0, 1.0, 0.02])
    names = ("loss_b", "loss_g", "loss_f", "class_error", "loss_s_intra", "loss_contrastive_align")
    crit = SetCriterion(HungarianMatcher(cost_class=4, cost_span=10, cost_giou=1), {}, 0.1,
                        ["spans", "labels", "saliency", "contrastive_align"], 0.07, "l1", 75, saliency_margin=0.2).to(dev)
    leaf = {k: v.to(dev).requires_grad_(True) for k, v in arr.items()}
    outs = dict(pred_logits=leaf["logits"], pred_spans=leaf["spans"], saliency_scores=leaf["sal"], proj_queries=leaf["pq"],
                proj_txt_mem=leaf["pt"])
    targets = dict(span_labels=[dict(spans=t.to(dev)) for t in tg], saliency_pos_labels=pos.to(dev), saliency_neg_labels=neg.to(dev))
    losses = crit(outs, targets)
    sum(losses[n] * float(w[i]) for i, n in enumerate(names) if w[i]).backward()
    npa = {k: v.numpy() for k, v in arr.items()}
    tgn = [t.numpy() for t in tg]
    idx = P.hungarian_match(npa["logits"], npa["spans"], tgn, w_class=4.0, w_span=10.0, w_giou=1.0)
    dev_idx = crit.matcher(dict(pred_logits=leaf["logits"], pred_spans=leaf["spans"]), targets)
    for (a, b), (c, d) in zip(idx, dev_idx):
        assert a.tolist() == c.tolist() and b.tolist() == d.tolist()
    want, G = P.detr_criterion(npa["logits"], npa["spans"], tgn, idx, npa["sal"], pos.numpy(), neg.numpy(), npa["pq"], npa["pt"],
                               0.1, 0.07, 0.2, w)
    for n, r in zip(names, want):
        assert abs(float(losses[n]) - r) <= 3e-5 * max(1.0, abs(r)), (n, float(losses[n]), r)
    for k in ("logits", "spans", "sal", "pq", "pt"):
        err = np.abs(leaf[k].grad.cpu().numpy() - G[k]).max()
        assert err <= 3e-5 * max(1e-3, np.abs(G[k]).max()), (k, err, np.abs(G[k]).max())


def test_device_nms_known_answers(dev, golden_dir):
    """temporal_nms on real QVHighlights predictions (golden from the reference's own utils/temporal_nms.py)."""
    from univtg_amd import ops
    d = json.load(open(os.path.join(golden_dir, "nms.json")))
    for c in d["cases"][:60]:
        rows = c["inp"]
        L = len(rows)
        # feed rows as if they were decoded windows of a duration-1 video: timestamp = 0, spans = st/ed, score = logits
        st = torch.tensor([[r[0], r[1]] for r in rows], dtype=torch.float32)[None] / 200.0
        sc = torch.tensor([r[2] for r in rows], dtype=torch.float32)[None, :, None]
        win, order, keep, nk = ops.decode_rank_nms(sc.to(dev), st.to(dev), torch.zeros(1, L, 2, device=dev), torch.ones(1, L, device=dev),
                                                   torch.tensor([200.0], device=dev), c["thd"], 1000, min(c["max_after"], 64))
        got = [order[0, i].item() for i in keep[0, : nk[0].item()].tolist()]
        # reference keep-set as indices into `rows` (scores in the fixture are unique enough to identify rows)
        srt = sorted(range(L), key=lambda i: rows[i][2], reverse=True)
        ref_rows = [tuple(r) for r in c["out"]][: min(c["max_after"], 64)]
        ref = []
        used = set()
        for r in ref_rows:
            idx = next(i for i in srt if tuple(rows[i]) == r and i not in used)
            used.add(idx)
            ref.append(idx)
        assert got == ref


def test_production_width_vs_oracle(dev):
    """d=1024 / 8 heads / D_v=2818: fp32x3 forward vs the CPU oracle (same seeded weights and batch)."""
    from oracle import univtg_oracle as O
    cfg = O.make_cfg(input_dropout=0.0, droppath=0.0, dropout=0.0, enc_layers=2)
    params = O.init_params(cfg, seed=5)
    inputs, tg = O.make_batch(cfg, 4, 75, 32, seed=6, ragged=True)
    with torch.no_grad():
        ref = O.forward(params, cfg, **inputs)
    model, crit = build(cfg, params, dev, "fp32x3")
    model.eval()
    with torch.no_grad():
        out = model(**to_dev(inputs, dev))
    valid = inputs["src_vid_mask"].bool()
    assert float((out["saliency_scores"].cpu() - ref["saliency_scores"])[valid].abs().max()) < 1e-4
    for k in ("pred_logits", "pred_spans"):
        assert float((out[k].cpu() - ref[k]).abs().max()) < 3e-4, k
    # bf16 training path: losses + a few gradients against oracle autograd
    p2 = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    o2 = O.forward(p2, cfg, **inputs)
    l2 = O.criterion(o2, tg, cfg)
    O.total_loss(l2, cfg).backward()
    model, crit = build(cfg, params, dev, "bf16")
    model.eval()
    out = model(**to_dev(inputs, dev))
    ld = crit(out, to_dev(tg, dev))
    sum(ld[k] * crit.weight_dict[k] for k in ld).backward()
    named = dict(model.named_parameters())
    for k in ("transformer.encoder.layers.0.self_attn.in_proj_weight", "transformer.encoder.layers.1.linear2.weight",
              "input_vid_proj.0.net.1.weight", "input_vid_proj.0.LayerNorm.weight", "input_vid_proj.0.LayerNorm.bias", "span_embed.layers.0.weight",
              "class_embed.layers.2.weight", "weightedpool.weight", "token_type_embeddings.weight",
              "transformer.encoder.layers.0.norm1.bias", "input_txt_proj.1.net.1.bias"):
        a, r = named[k].grad.cpu().double().flatten(), p2[k].grad.double().flatten()
        cos = float((a @ r) / (a.norm() * r.norm() + 1e-30))
        ratio = float(a.norm() / (r.norm() + 1e-30))
        e = float((a - r).abs().max() / (r.abs().max() + 1e-30))
        # input_vid_proj.0.LayerNorm.weight: the two TEF columns (values up to 1 among L2-normalised features of ~0.02) carry 92 % of this
        # vector's norm (measured on the oracle: |g_tef| = 0.39 / 0.11, all other 2816 columns together 0.17), each a cancellation-heavy
        # sum over the rows of a bf16 dgrad product: its norm ratio is the accuracy of TWO scalars, +-1..3 % depending on the batch
        assert cos > 0.998 and abs(ratio - 1) < (0.04 if k == "input_vid_proj.0.LayerNorm.weight" else 0.01) and e < 0.12, (k, cos, ratio, e)


def test_config2_size_properties(dev):
    """BASELINE config 2 (B=256, L_v=75, L_t=32, d=1024, 4 layers): properties that need no CPU run at this size."""
    from oracle import univtg_oracle as O
    cfg = O.make_cfg(input_dropout=0.0, droppath=0.0, dropout=0.0)
    params = O.init_params(cfg, seed=1)
    model, crit = build(cfg, params, dev, "bf16")
    model.eval()
    inputs, tg = O.make_batch(cfg, 256, 75, 32, seed=2, ragged=True)
    ind = to_dev(inputs, dev)
    with torch.no_grad():
        out = model(**ind)
        # (1) batch independence in eval mode: a 32-sample slice and a single sample give the same rows (same padded length).  Small launches
        # may sum K in <= 4 parts (split-K of the single-tile GEMM variant: other fp32 rounding than the one-pass sum of the 256-sample call),
        # so the bf16 stream agrees to bf16 rounding of the intermediates; the precise mode (below) keeps 2e-6
        sub = model(**{k: v[:32] for k, v in ind.items()})
        one = model(**{k: v[:1] for k, v in ind.items()})
    for k in ("pred_logits", "pred_spans", "saliency_scores"):
        assert torch.isfinite(out[k]).all()
        assert float((out[k][:32] - sub[k]).abs().max()) < 3e-2, k
        assert float((out[k][:1] - one[k]).abs().max()) < 3e-2, k
    assert float((out["saliency_scores"][:32] - sub["saliency_scores"]).abs().max()) < 1e-5
    model32, _ = build(cfg, params, dev, "fp32x3")
    model32.eval()
    with torch.no_grad():
        o32 = model32(**ind)
        for n in (32, 1):
            s32 = model32(**{k: v[:n] for k, v in ind.items()})
            for k in ("pred_logits", "pred_spans", "saliency_scores"):
                assert float((o32[k][:n] - s32[k]).abs().max()) < 4e-6, (k, n)
    # (2) ranges: probabilities in (0,1), left offsets <= 0 <= right offsets, padded saliency == log-mask constant
    assert float(out["pred_logits"].min()) > 0 and float(out["pred_logits"].max()) < 1
    assert float(out["pred_spans"][..., 0].max()) <= 0 and float(out["pred_spans"][..., 1].min()) >= 0
    pad = ~inputs["src_vid_mask"].bool()
    assert float((out["saliency_scores"].cpu()[pad] + 103.2789).abs().max()) < 1.01
    # (3) padded text keys do not influence anything: perturb padded text features
    ind2 = dict(ind)
    noise = torch.randn_like(ind["src_txt"]) * (1 - ind["src_txt_mask"])[..., None]
    ind2["src_txt"] = ind["src_txt"] + noise
    with torch.no_grad():
        out2 = model(**ind2)
    assert float((out2["saliency_scores"] - out["saliency_scores"]).abs().max()) < 1e-5
    # (4) gradients exist, are finite, and txt_position_embed gets none (as in the reference)
    outg = model(**ind)
    ld = crit(outg, to_dev(tg, dev))
    sum(ld[k] * crit.weight_dict[k] for k in ld).backward()
    for k, p in model.named_parameters():
        if k.startswith("txt_position_embed"):
            assert p.grad is None
        else:
            assert p.grad is not None and torch.isfinite(p.grad).all(), k


def test_native_train_step_matches_autograd_path(dev):
    """univtg_amd.trainer.TrainStep (compact criterion->model gradient hand-off, flat buffers) computes the same
    losses and parameter gradients as the drop-in autograd path on the same batch."""
    from oracle import univtg_oracle as O
    from univtg_amd.trainer import TrainStep
    cfg = O.make_cfg(hidden_dim=256, nheads=4, dim_feedforward=256, enc_layers=2, v_feat_dim=514, t_feat_dim=512,
                     input_dropout=0.0, dropout=0.0, droppath=0.0)
    params = O.init_params(cfg, seed=9)
    inputs, tg = O.make_batch(cfg, 6, 30, 10, seed=10, ragged=True, curve=True)
    ind, tgd = to_dev(inputs, dev), to_dev(tg, dev)
    model, crit = build(cfg, params, dev, "bf16")
    model.eval()
    out = model(**ind)
    ld = crit(out, tgd)
    sum(ld[k] * crit.weight_dict[k] for k in ld).backward()
    ref = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    model2, crit2 = build(cfg, params, dev, "bf16")
    model2.eval()
    step = TrainStep(model2, crit2, grad_clip=0.1)
    losses = step.step(ind, tgd, optimize=False)
    for i, k in enumerate(("loss_b", "loss_g", "loss_f", "loss_s_inter", "loss_s_intra")):
        assert abs(float(losses[i]) - float(ld[k])) < 1e-5 * max(1.0, abs(float(ld[k]))), k
    offs = model2._offsets(model2._dims(6, 30, 10, 514, 512, False))
    names = {id(p): k for k, p in model2.named_parameters()}
    for i, p in enumerate(model2._ordered_params()):
        g = step.grads[offs[i]: offs[i] + p.numel()].view_as(p)
        r = ref[names[id(p)]]
        err = float((g - r).abs().max()) / (float(r.abs().max()) + 1e-12)
        assert err < 2e-3, (names[id(p)], err)        # same kernels, only fp32 atomic-add order differs
    # one optimizer step moves the parameters and keeps them finite
    before = step.flat.clone()
    step.step(ind, tgd, optimize=True)
    assert torch.isfinite(step.flat).all() and float((step.flat - before).abs().max()) > 0
    # AdamW + clip semantics vs torch.optim.AdamW on the same flat gradient
    p_ref = before.clone().requires_grad_(True)
    p_ref.grad = step.grads.clone()
    torch.nn.utils.clip_grad_norm_([p_ref], 0.1)
    opt = torch.optim.AdamW([p_ref], lr=1e-4, weight_decay=1e-4)
    opt.step()
    assert float((p_ref.detach() - step.flat).abs().max()) < 1e-6


def test_fused_adamw_clip_in_the_reference_training_loop(dev):
    """univtg_amd.optim.FusedAdamWClip swapped into the reference's loop body (main/train_vlp_ddp.py:56-68: model(**inputs) -> criterion ->
    weighted sum -> zero_grad -> backward -> [clip_grad_norm_] -> optimizer.step): after every backward the SAME gradients also go through
    clip_grad_norm_(0.1) + torch.optim.AdamW on shadow parameters (Adam is ill-conditioned across two separately computed backward passes: an
    element whose gradient is rounding noise moves by +-lr on its sign) -- three steps, then the same parameters; autograd's flat gradient
    buffer read in place (no gather), the clip line left in the loop changes nothing, and a state_dict that torch.optim.AdamW loads (the
    reference's checkpoint layout, main/train_vlp_ddp.py:157-195) and that loads back."""
    from oracle import univtg_oracle as O
    from univtg_amd.optim import FusedAdamWClip
    cfg = O.make_cfg(hidden_dim=256, nheads=4, dim_feedforward=256, enc_layers=2, v_feat_dim=514, t_feat_dim=512,
                     input_dropout=0.0, dropout=0.0, droppath=0.0)
    params = O.init_params(cfg, seed=19)
    batches = [O.make_batch(cfg, 6, 30, 10, seed=20 + i, ragged=True) for i in range(3)]

    def loop(clip_line):
        model, crit = build(cfg, params, dev, "bf16", proj_precise=False)
        model.train()
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        opt = FusedAdamWClip([{"params": [p for _, p in named]}], lr=1e-3, weight_decay=1e-4, max_grad_norm=0.1, model=model)     # main/config.py:349
        shadow = [p.detach().clone().requires_grad_(True) for _, p in named]
        sopt = torch.optim.AdamW([{"params": shadow}], lr=1e-3, weight_decay=1e-4)
        for inputs, tg in batches:
            out = model(**to_dev(inputs, dev))
            ld = crit(out, to_dev(tg, dev))
            losses = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
            opt.zero_grad()
            losses.backward()
            for (_, p), sp in zip(named, shadow):
                sp.grad = None if p.grad is None else p.grad.detach().clone()
            torch.nn.utils.clip_grad_norm_(shadow, 0.1)
            sopt.step()
            if clip_line:
                torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1)
            opt.step()
        torch.cuda.synchronize()
        return model, opt, named, shadow, sopt

    for clip_line in (False, True):
        model, opt, named, shadow, sopt = loop(clip_line)
        # in place unless the clip line rescaled p.grad in place first (it still IS the flat buffer then)
        assert opt.in_place_steps == 3, opt.in_place_steps
        moved = 0.0
        for (k, p), sp in zip(named, shadow):
            moved = max(moved, float((p - params[k].to(dev)).abs().max()))
            # (the clip line left in: the second clip is the identity up to one rounding of the norm)
            assert float((p - sp).abs().max()) < (2e-6 if not clip_line else 2e-5), (k, clip_line, float((p - sp).abs().max()))
        assert moved > 1e-3                                         # three lr = 1e-3 steps move a weight by ~3e-3
    # checkpoint layout: torch.optim.AdamW loads the fused optimizer's state, and the fused optimizer loads AdamW's
    sd = opt.state_dict()
    ssd = sopt.state_dict()
    assert sd["param_groups"][0]["params"] == ssd["param_groups"][0]["params"]
    twin = torch.optim.AdamW([{"params": [p for _, p in named]}], lr=1e-3, weight_decay=1e-4)
    twin.load_state_dict(sd)
    for i, ent in twin.state_dict()["state"].items():
        want = ssd["state"][i]
        assert float(ent["step"]) == 3.0
        assert float((ent["exp_avg_sq"] - want["exp_avg_sq"]).abs().max()) <= 1e-4 * float(want["exp_avg_sq"].abs().max()) + 1e-12
    opt2 = FusedAdamWClip([{"params": [p for _, p in named]}], lr=1e-3, weight_decay=1e-4, max_grad_norm=0.1, model=model)
    opt2.load_state_dict(ssd)
    assert opt2.t == 3 and float((opt2.v - opt.v).abs().max()) <= 1e-4 * float(opt.v.abs().max())


def test_nt256_engine_path_matches_nt128(dev):
    """Config-2 size (B=256): the whole forward+criterion+backward with the persistent 256-tile GEMM gives the same
    losses and the same flat gradient as with the 128-tile GEMM forced (identical bf16 products, same K order; only
    the fp32 atomic order of the weight-gradient kernel differs)."""
    from oracle import univtg_oracle as O
    from univtg_amd import _lib
    from univtg_amd.trainer import TrainStep
    lib = _lib.load()
    cfg = O.make_cfg(input_dropout=0.0, droppath=0.0, dropout=0.0)
    params = O.init_params(cfg, seed=21)
    inputs, tg = O.make_batch(cfg, 256, 75, 32, seed=22, ragged=True)
    ind, tgd = to_dev(inputs, dev), to_dev(tg, dev)
    model, crit = build(cfg, params, dev, "bf16")
    model.eval()
    step = TrainStep(model, crit, grad_clip=0.1)
    res = {}
    try:
        for tile, bm in ((128, 0), (256, 0), (256, 320), (256, 256), (256, 192), (256, 128)):     # every tile height of the persistent kernel,
            _lib.check(lib.uvtg_debug_force_nt_tile(tile))                             # with its fused epilogues (residual, act-grad, ...)
            _lib.check(lib.uvtg_debug_force_nt_bm(bm))
            losses = step.step(ind, tgd, optimize=False).clone()
            torch.cuda.synchronize()
            res[(tile, bm)] = (losses, step.grads.clone(), step.pred_logits.clone(), step.pred_spans.clone())
    finally:
        lib.uvtg_debug_force_nt_tile(0)
        lib.uvtg_debug_force_nt_bm(0)
    l1, g1, pl1, ps1 = res[(128, 0)]
    for key in ((256, 320), (256, 256), (256, 192), (256, 128)):
        lk, gk, plk, psk = res[key]
        assert float((pl1 - plk).abs().max()) < 1e-6 and float((ps1 - psk).abs().max()) < 1e-6, key
        assert float((l1 - lk).abs().max()) < 1e-5 * max(1.0, float(l1.abs().max())), key
        assert float((g1 - gk).norm() / g1.norm()) < 1e-4, key
    l2, g2, pl2, ps2 = res[(256, 0)]
    assert torch.isfinite(g2).all()
    assert float((pl1 - pl2).abs().max()) < 1e-6 and float((ps1 - ps2).abs().max()) < 1e-6
    assert float((l1 - l2).abs().max()) < 1e-5 * max(1.0, float(l1.abs().max()))
    offs = model._offsets(model._dims(256, 75, 32, cfg.v_feat_dim, cfg.t_feat_dim, False))
    for i, p in enumerate(model._ordered_params()):
        a, b = g1[offs[i]: offs[i] + p.numel()], g2[offs[i]: offs[i] + p.numel()]
        err = float((a - b).abs().max()) / (float(a.abs().max()) + 1e-12)
        # (round 5: the 256-tile path takes attention backward's delta from the dO GEMM's epilogue, the 128-tile path from attn_delta_kernel -- the
        #  same products in another fp32 summation order; a 1e-7 difference in delta flips single bf16 roundings of dS: measured 2.05e-3 on one bias)
        assert err < 4e-3, (i, err)


@pytest.mark.parametrize("name", CASES)
def test_auto_projection_mode_training(dev, golden_dir, name):
    """Product default (proj_precise="auto"): inference calls keep saliency within 1e-4; calls that will be differentiated run
    the input projections on plain bf16 operands -- losses within bf16 tolerance, gradients still aligned with the reference."""
    meta, cfg, params, inputs, tg, out_ref, eval_ref, grads_ref, losses_ref = load_case(golden_dir, name)
    model, crit = build(cfg, params, dev, "bf16", proj_precise="auto")
    model.eval()
    valid = inputs["src_vid_mask"].bool()
    with torch.no_grad():
        out = model(**to_dev(inputs, dev))
    assert float((out["saliency_scores"].cpu() - eval_ref["saliency_scores"])[valid].abs().max()) < 1e-4
    out = model(**to_dev(inputs, dev))                      # gradient-enabled call -> bf16 projections
    assert float((out["saliency_scores"].detach().cpu() - eval_ref["saliency_scores"])[valid].abs().max()) < 3e-2
    losses = crit(out, to_dev(tg, dev))
    wd = crit.weight_dict
    sum(losses[k] * wd[k] for k in losses if k in wd).backward()
    for k in losses:
        got, ref = float(losses[k]), losses_ref[k]
        assert abs(got - ref) < 3e-2 * max(1.0, abs(ref)), (k, got, ref)
    named = dict(model.named_parameters())
    bad = {}
    for k, g in grads_ref.items():
        a, r = named[k].grad.cpu().double().flatten(), g.double().flatten()
        cos = float((a @ r) / (a.norm() * r.norm() + 1e-30))
        ratio = float(a.norm() / (r.norm() + 1e-30))
        if cos < 0.97 or abs(ratio - 1) > 0.08:
            bad[k] = (cos, ratio)
    assert not bad, bad


@pytest.mark.parametrize("shape", ["toy_width", "production_width"])
def test_overlapped_gradient_exchange_single_rank(dev, shape):
    """The bucketed side-stream exchange (ready events recorded inside uvtg_backward, RCCL all-reduce per range on a comm
    stream, one coalesced collective per readiness group) at world size 1: must leave exactly the gradients / parameters of the plain path.
    production_width: d = 1024, E = 4, where the weight gradients stay in the deferred hybrid launch under the events (the toy width falls back
    to the per-batch launches).  (With UVTG_DEV_ENV=1 the same test covers UVTG_TN_EVENT_GROUPS=1 -- heads + layers E-1 .. 1 behind layer 1,
    layer 0 in a second launch -- and UVTG_TN_EVENTS_PER_LAYER=1, the per-layer-event mode of rounds 2-5.)"""
    import torch.distributed as dist
    from oracle import univtg_oracle as O
    from univtg_amd.trainer import TrainStep
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        created = True
    try:
        if shape == "toy_width":
            cfg = O.make_cfg(hidden_dim=256, nheads=4, dim_feedforward=256, enc_layers=3, v_feat_dim=514, t_feat_dim=512,
                             input_dropout=0.0, dropout=0.0, droppath=0.0)
            B, Lv, Lt = 8, 30, 10
        else:
            cfg = O.make_cfg(input_dropout=0.0, dropout=0.0, droppath=0.0)
            B, Lv, Lt = 64, 75, 32
        params = O.init_params(cfg, seed=31)
        inputs, tg = O.make_batch(cfg, B, Lv, Lt, seed=32, ragged=True)
        ind, tgd = to_dev(inputs, dev), to_dev(tg, dev)
        res = []
        for mode in (False, "force"):
            model, crit = build(cfg, params, dev, "bf16")
            model.eval()
            step = TrainStep(model, crit, grad_clip=0.1, overlap_comm=mode)
            assert bool(step.overlap) == (mode == "force")
            step.step(ind, tgd, optimize=False)             # same parameters in both modes: gradients must agree
            torch.cuda.synchronize()
            g = step.grads.clone()
            for _ in range(3):
                losses = step.step(ind, tgd, optimize=True).clone()
            torch.cuda.synchronize()
            res.append((g, step.flat.clone(), losses))
        assert torch.isfinite(res[1][1]).all() and torch.isfinite(res[1][2]).all()
        assert float((res[0][0] - res[1][0]).abs().max()) <= 2e-3 * float(res[0][0].abs().max())      # fp32 atomic order only
        # after three Adam steps the trajectories may differ by rounding-level noise amplified by the normalised update
        assert float((res[0][1] - res[1][1]).abs().max()) <= 1e-3
        assert float((res[0][2] - res[1][2]).abs().max()) <= 1e-2 * float(res[0][2].abs().max())
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("B,Lv,Lt,d,H,ragged", [(1, 1, 1, 64, 2, False), (2, 3, 2, 64, 2, True), (3, 129, 5, 128, 1, True),
                                                (2, 300, 20, 256, 4, True), (2, 128, 1, 128, 4, False)])
def test_edge_shapes_vs_oracle(dev, B, Lv, Lt, d, H, ragged):
    """Degenerate and boundary shapes against the CPU oracle: a single clip / token, S just over the single-tile attention
    limit (129 + 5 with head_dim 128), a long video (S = 320, the tiled attention kernels), S = 129 exactly one text token."""
    from oracle import univtg_oracle as O
    cfg = O.make_cfg(hidden_dim=d, nheads=H, dim_feedforward=d, enc_layers=2, v_feat_dim=66, t_feat_dim=40, max_q_l=max(Lt, 4),
                     input_dropout=0.0, dropout=0.0, droppath=0.0)
    params = O.init_params(cfg, seed=B + Lv)
    inputs, tg = O.make_batch(cfg, B, Lv, Lt, seed=Lv + Lt, ragged=ragged)
    with torch.no_grad():
        ref = O.forward(params, cfg, **inputs)
    model, _ = build(cfg, params, dev, "fp32x3")
    model.eval()
    with torch.no_grad():
        out = model(**to_dev(inputs, dev))
    valid = inputs["src_vid_mask"].bool()
    assert float((out["saliency_scores"].cpu() - ref["saliency_scores"])[valid].abs().max()) < 1e-4
    for k in ("pred_logits", "pred_spans"):
        assert float((out[k].cpu() - ref[k]).abs().max()) < 3e-4, k
    # training path (bf16 operands) against oracle autograd
    p2 = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    l2 = O.criterion(O.forward(p2, cfg, **inputs), tg, cfg)
    O.total_loss(l2, cfg).backward()
    model, crit = build(cfg, params, dev, "bf16")
    model.eval()
    ld = crit(model(**to_dev(inputs, dev)), to_dev(tg, dev))
    sum(ld[k] * crit.weight_dict[k] for k in ld).backward()
    for k in ("loss_b", "loss_g", "loss_f"):
        assert abs(float(ld[k]) - float(l2[k])) < 3e-2 * max(1.0, abs(float(l2[k]))), k
    named = dict(model.named_parameters())
    worst = 1.0
    for k, p in p2.items():
        if p.grad is None or float(p.grad.abs().max()) == 0.0:
            continue
        a, r = named[k].grad.cpu().double().flatten(), p.grad.double().flatten()
        assert torch.isfinite(a).all(), k
        if float(r.norm()) > 1e-6 * r.numel() ** 0.5:
            worst = min(worst, float((a @ r) / (a.norm() * r.norm() + 1e-30)))
    assert worst > 0.95, worst


@pytest.mark.parametrize("B,Lv,Lt,d,H,E,nproj", [(6, 30, 10, 256, 4, 2, 2), (256, 75, 32, 1024, 8, 4, 2), (3, 150, 12, 128, 2, 2, 2), (5, 128, 32, 256, 2, 2, 2),
                                                (6, 30, 10, 256, 4, 2, 1), (6, 30, 10, 256, 4, 2, 3)])      # the compact projection with 1 / 3 blocks
def test_packed_ragged_stream_matches_padded(dev, B, Lv, Lt, d, H, E, nproj):
    """Packed encoder stream (valid clips + ONE representative padded clip + valid text per sample, include/uvtg.h lens_host)
    against the padded execution of the same ragged batch: same outputs at every clip position (padded ones included), same
    losses, same parameter gradients up to bf16 rounding of the re-associated sums."""
    from oracle import univtg_oracle as O
    from univtg_amd.trainer import TrainStep
    cfg = O.make_cfg(hidden_dim=d, nheads=H, dim_feedforward=d, enc_layers=E, v_feat_dim=66 if d < 1024 else 2818,
                     t_feat_dim=40 if d < 1024 else 512, max_q_l=max(Lt, 4), input_dropout=0.0, dropout=0.0, droppath=0.0, n_input_proj=nproj)
    params = O.init_params(cfg, seed=41)
    inputs, tg = O.make_batch(cfg, B, Lv, Lt, seed=42, ragged=True)
    ind, tgd = to_dev(inputs, dev), to_dev(tg, dev)
    lens = (inputs["src_vid_mask"].sum(1).int().tolist(), inputs["src_txt_mask"].sum(1).int().tolist())
    assert min(lens[0]) < Lv, "the batch must contain padded clips"
    res = {}
    for mode in (False, True):
        model, crit = build(cfg, params, dev, "bf16")
        model.eval()
        step = TrainStep(model, crit, grad_clip=0.1, packed=mode)
        batch = dict(ind)
        if mode:
            batch["_lens_host"] = lens
        losses = step.step(batch, tgd, optimize=False).clone()
        torch.cuda.synchronize()
        res[mode] = (losses, step.grads.clone(), step.pred_logits.clone(), step.pred_spans.clone(), step.sal.clone())
    (l0, g0, pl0, ps0, s0), (l1, g1, pl1, ps1, s1) = res[False], res[True]
    assert torch.isfinite(g1).all()
    assert float((s0 - s1).abs().max()) < 1e-6                       # saliency never touches the encoder
    # all clip positions (padded ones too).  The two executions sum in different orders (attention key tiles, GEMM row tiles), so they
    # differ by bf16 rounding noise; both must be equally close to the fp32-class forward of the same weights
    m32, _ = build(cfg, params, dev, "fp32x3")
    m32.eval()
    with torch.no_grad():
        ref = m32(**ind)
    for a, b, r in ((pl0, pl1, ref["pred_logits"]), (ps0, ps1, ref["pred_spans"])):
        e0, e1 = float((a - r).abs().max()), float((b - r).abs().max())
        assert e1 <= 1.5 * e0 + 1e-3, (e0, e1)
        assert float((a - b).abs().max()) <= 2.0 * max(e0, e1) + 1e-4
    assert float((l0 - l1).abs().max()) < 2e-3 * max(1.0, float(l0.abs().max()))
    offs = model._offsets(model._dims(B, Lv, Lt, cfg.v_feat_dim, cfg.t_feat_dim, False))
    names = {id(p): k for k, p in model.named_parameters()}
    for i, p in enumerate(model._ordered_params()):
        a, b = g0[offs[i]: offs[i] + p.numel()].double(), g1[offs[i]: offs[i] + p.numel()].double()
        if float(a.norm()) == 0.0:
            assert float(b.norm()) == 0.0
            continue
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
        ratio = float(b.norm() / a.norm())
        assert cos > 0.999 and abs(ratio - 1) < 1e-2, (names[id(p)], cos, ratio)


def test_drop_in_model_packed_option(dev):
    """build_model(args with packed=True): the autograd path runs the packed stream (lengths read back from the masks) and matches
    the padded autograd path."""
    from oracle import univtg_oracle as O
    cfg = O.make_cfg(hidden_dim=256, nheads=4, dim_feedforward=256, enc_layers=2, v_feat_dim=514, t_feat_dim=512,
                     input_dropout=0.0, dropout=0.0, droppath=0.0)
    params = O.init_params(cfg, seed=51)
    inputs, tg = O.make_batch(cfg, 5, 40, 12, seed=52, ragged=True)
    ind, tgd = to_dev(inputs, dev), to_dev(tg, dev)
    grads, outs = [], []
    for packed in (False, True):
        from univtg_amd.model import build_model
        model, crit = build_model(args_from_cfg(cfg, precision="bf16", proj_precise=True, packed=packed))
        model.load_state_dict(params, strict=True)
        model.to(dev).eval(); crit.to(dev)
        out = model(**ind)
        ld = crit(out, tgd)
        sum(ld[k] * crit.weight_dict[k] for k in ld).backward()
        grads.append(torch.cat([p.grad.flatten() for p in model._ordered_params()]).double())
        outs.append(out["pred_logits"].detach())
    assert float((outs[0] - outs[1]).abs().max()) < 2e-2
    cos = float((grads[0] @ grads[1]) / (grads[0].norm() * grads[1].norm()))
    assert cos > 0.999 and abs(float(grads[1].norm() / grads[0].norm()) - 1) < 1e-2, cos


def test_pipeline_collate_upload_matches_reference(dev, golden_dir):
    """univtg_amd.pipeline.collate_upload_mr (packed valid rows over PCIe + uvtg_ragged_to_padded on device) reproduces the
    reference's start_end_collate_mr + prepare_batch_inputs_mr bit-exactly (golden collate.npz made by the real reference),
    and its host-side lengths drive the packed encoder stream."""
    from univtg_amd.pipeline import collate_upload_mr
    from tests.test_oracle_golden import _collate_case
    z, batch = _collate_case(golden_dir)
    for e in batch:
        e["model_inputs"] = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in e["model_inputs"].items()}
    meta, model_inputs, targets = collate_upload_mr(batch, dev)
    torch.cuda.synchronize()
    assert [m["qid"] for m in meta] == list(range(len(batch)))
    for k in ("src_txt", "src_txt_mask", "src_vid", "src_vid_mask"):
        assert np.array_equal(model_inputs[k].cpu().numpy(), z["in/" + k]), k
    for k, v in targets.items():
        if k == "span_labels":
            for i, d in enumerate(v):
                assert np.array_equal(d["spans"].cpu().numpy(), z[f"tg/span_labels/{i}"])
        else:
            ref = z["tg/" + k]
            assert np.array_equal(v.cpu().numpy(), ref) and v.cpu().numpy().dtype == ref.dtype, k
    lv, lt = model_inputs["_lens_host"]
    assert lv == [int(x) for x in z["in/src_vid_mask"].sum(1)] and lt == [int(x) for x in z["in/src_txt_mask"].sum(1)]
    # the same split at the PCIe boundary (round 6): pack_batch_host in a loader worker -> upload_packed_batch in the training process,
    # through the prefetcher's side stream
    from univtg_amd.pipeline import DevicePrefetcher, pack_batch_host
    for wire in (torch.float32, torch.bfloat16):
        want = collate_upload_mr(batch, dev, feature_dtype=wire)
        (got,) = list(DevicePrefetcher([pack_batch_host(batch, feature_dtype=wire)], dev))
        torch.cuda.synchronize()
        assert [m["qid"] for m in got[0]] == [m["qid"] for m in want[0]] and got[1]["_lens_host"] == want[1]["_lens_host"]
        for a, b in ((got[1], want[1]), (got[2], want[2])):
            assert a.keys() == b.keys()
            for k in a:
                if torch.is_tensor(a[k]):
                    assert torch.equal(a[k], b[k]) and a[k].dtype == b[k].dtype, k
                elif k == "span_labels":
                    assert all(torch.equal(x["spans"], y["spans"]) for x, y in zip(a[k], b[k]))
    # bf16 wire format: features rounded to bf16, everything else unchanged
    _, mi16, _ = collate_upload_mr(batch, dev, feature_dtype=torch.bfloat16)
    ref16 = torch.from_numpy(z["in/src_vid"]).to(torch.bfloat16).float().numpy()
    assert np.array_equal(mi16["src_vid"].cpu().numpy(), ref16)
    assert np.array_equal(mi16["src_vid_mask"].cpu().numpy(), z["in/src_vid_mask"])


@pytest.mark.parametrize("case", ["tiny_golden", "production_width", "long_sequence", "txt_pos_three_blocks", "tal_branch"])
def test_train_mode_dropout_replayed_through_oracle(dev, golden_dir, case):
    """Train-mode parity: input dropout (p=0.5), attention dropout and DropPath all on.  The kernels' counter-based masks are
    regenerated on the host (tests/philox_ref.py), handed to the CPU oracle as explicit Bernoulli masks, and outputs, losses and
    parameter gradients must agree -- this pins nn.Dropout's 1/(1-p) scaling, drop_path's x/keep*floor(keep+U)
    (model/transformer_encoder_droppath.py:154-183) and that backward re-creates exactly the forward masks."""
    import philox_ref as R
    from oracle import univtg_oracle as O
    if case == "tiny_golden":
        meta, _, params, inputs, tg, *_ = load_case(golden_dir, "tiny_eval_ragged")
        cfg = O.make_cfg(**{**meta["cfg"], "input_dropout": 0.5, "dropout": 0.1, "droppath": 0.25})
    elif case == "production_width":       # d=1024, D_v=2818: the wide-row LayerNorm kernels and their lane-pair Philox sharing
        cfg = O.make_cfg(input_dropout=0.5, dropout=0.1, droppath=0.25, enc_layers=1)
        params = O.init_params(cfg, seed=11)
        inputs, tg = O.make_batch(cfg, 4, 12, 5, seed=12, ragged=True)
    elif case == "tal_branch":              # round 6: src_cls in TRAIN mode -- the class names' projection draws its own input-dropout masks (the
        # second engine call of Model.forward: the next step seed), and the 'saliency_cls' class term sends gradients through them
        meta, _, params, inputs, tg, *_ = load_case(golden_dir, "tiny_tal")
        cfg = O.make_cfg(**{**meta["cfg"], "input_dropout": 0.5, "dropout": 0.1, "droppath": 0.25})
    elif case == "txt_pos_three_blocks":    # --use_txt_pos + --n_input_proj 3 in TRAIN mode: the text positions' own dropout (p = input_dropout,
        # position_encoding.py:113-115), three dropout streams per modality, the position table / LayerNorm gradients
        meta, _, params, inputs, tg, *_ = load_case(golden_dir, "tiny_txt_pos")
        cfg = O.make_cfg(**{**meta["cfg"], "input_dropout": 0.5, "dropout": 0.1, "droppath": 0.25, "n_input_proj": 3})
        params = O.init_params(cfg, seed=17)
    else:       # S = 300 + 12 > 256 at head_dim 128: the tiled attention kernels' DROPOUT instantiations over many query / key blocks (the
                # pipelined dK/dV loop, the LDS-DMA dQ kernel; attention dropout never reaches the fused S <= 256 kernels)
        cfg = O.make_cfg(hidden_dim=256, nheads=2, dim_feedforward=256, v_feat_dim=514, max_v_l=300, input_dropout=0.5, dropout=0.1, droppath=0.25,
                         enc_layers=2)
        params = O.init_params(cfg, seed=13)
        inputs, tg = O.make_batch(cfg, 2, 300, 12, seed=14, ragged=True)
    model, crit = build(cfg, params, dev, "bf16")
    model.train()
    model.set_seed(20240917)
    out = model(**to_dev(inputs, dev))
    losses = crit(out, to_dev(tg, dev))
    total = sum(losses[k] * crit.weight_dict[k] for k in losses if k in crit.weight_dict)
    total.backward()

    seed = (20240917 * 1000003 + 1) & 0xFFFFFFFFFFFFFFFF              # first training call after set_seed (model.py: _dims)
    B, Lv, Dv = inputs["src_vid"].shape
    Lt, Dt = inputs["src_txt"].shape[1:]
    d, H, E, S = cfg.hidden_dim, cfg.nheads, cfg.enc_layers, Lv + Lt
    t = lambda a: torch.from_numpy(a)
    nb = cfg.n_input_proj
    rng = {"vid_keep": [t(R.row_keep(seed, R.RNG_IN_VID + b, B * Lv, Dv if b == 0 else d, 0.5)).view(B, Lv, Dv if b == 0 else d) for b in range(nb)],
           "txt_keep": [t(R.row_keep(seed, R.RNG_IN_TXT + b, B * Lt, Dt if b == 0 else d, 0.5)).view(B, Lt, Dt if b == 0 else d) for b in range(nb)],
           "dp_scale": t(R.droppath_scales(seed, E, B, 0.25)),
           "attn_keep": torch.stack([t(R.attn_keep(seed, l, B, H, S, 0.1)) for l in range(E)])}
    if getattr(cfg, "use_txt_pos", False):      # counters keyed by the token's row b * S + L_v + t of the padded layout (engine.hip, text_positions)
        rng["txtpos_keep"] = t(R.row_keep(seed, R.RNG_TXT_POS, B * S, d, 0.5)).view(B, S, d)[:, Lv:]
    if "src_cls" in inputs:      # the class names go through the engine as a SECOND training call: step seed + 1, text-stream counters over n_cls x L_c rows
        seed_cls = (20240917 * 1000003 + 2) & 0xFFFFFFFFFFFFFFFF
        nc, Lc = inputs["src_cls"].shape[:2]
        rng["cls_keep"] = [t(R.row_keep(seed_cls, R.RNG_IN_TXT + b, nc * Lc, Dt if b == 0 else d, 0.5)).view(nc, Lc, Dt if b == 0 else d) for b in range(nb)]
    assert 0 < float((rng["dp_scale"] == 0).float().mean()) < 1           # the case exercises dropped AND kept branches
    ref_params = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref_out = O.forward(ref_params, cfg, inputs["src_txt"], inputs["src_txt_mask"], inputs["src_vid"], inputs["src_vid_mask"],
                        inputs.get("src_cls"), inputs.get("src_cls_mask"), rng=rng)
    ref_losses = O.criterion(ref_out, tg, cfg)
    O.total_loss(ref_losses, cfg).backward()

    valid = inputs["src_vid_mask"].bool()
    assert float((out["saliency_scores"].detach().cpu() - ref_out["saliency_scores"].detach())[valid].abs().max()) < 2e-2
    for k, tol in (("pred_logits", 4e-2), ("pred_spans", 4e-2), ("vid_mem_proj", 4e-2), ("txt_mem_proj", 4e-2)) + ((("cls_mem_proj", 4e-2),) if "cls_mem_proj" in ref_out else ()):
        err = float((out[k].detach().cpu() - ref_out[k].detach()).abs().max())
        assert err < tol, (k, err)
    for k in ("loss_b", "loss_g", "loss_f", "loss_s_inter", "loss_s_intra"):
        got, ref = float(losses[k].detach()), float(ref_losses[k].detach())
        assert abs(got - ref) < 3e-2 * max(1.0, abs(ref)), (k, got, ref)
    named, bad = dict(model.named_parameters()), {}
    for k, p in ref_params.items():
        if p.grad is None or float(p.grad.abs().max()) == 0.0:
            continue
        a, r = named[k].grad.cpu().double().flatten(), p.grad.double().flatten()
        cos = float((a @ r) / (a.norm() * r.norm() + 1e-30))
        ratio = float(a.norm() / (r.norm() + 1e-30))
        if cos < 0.98 or abs(ratio - 1) > 0.06:
            bad[k] = (cos, ratio)
    assert not bad, bad


def test_hip_graph_inference_replays_the_eager_call(dev):
    """univtg_amd.graph.GraphedInference: forward (precise arithmetic under no_grad) + device post-processing captured into a HIP graph per
    input shape.  The replay must give the eager call's bits for every batch of that shape, a second shape gets its own graph, and a
    parameter update invalidates the captured operand cache (re-capture)."""
    from oracle import univtg_oracle as O
    from univtg_amd import ops
    from univtg_amd.graph import GraphedInference
    cfg = O.make_cfg(hidden_dim=256, nheads=4, dim_feedforward=256, enc_layers=2, v_feat_dim=514, t_feat_dim=512, input_dropout=0.0, dropout=0.0,
                     droppath=0.0)
    params = O.init_params(cfg, seed=61)
    model, _ = build(cfg, params, dev, "auto", proj_precise="auto")
    model.eval()
    run = GraphedInference(model, clip_length=2.0)

    def batch(B, Lv, Lt, seed):
        inputs, tg = O.make_batch(cfg, B, Lv, Lt, seed=seed, ragged=True)
        dur = torch.tensor([float(inputs["src_vid_mask"][b].sum()) * 2.0 for b in range(B)])
        return to_dev(inputs, dev), tg["timestamp"].to(dev), tg["timestamp_mask"].to(dev), dur.to(dev)

    def eager(inp, ts, tm, dur):
        with torch.no_grad():
            out = model(**inp)
            win, order, keep, nk, sal = ops.postprocess_mr(out["pred_logits"], out["pred_spans"], out["saliency_scores"], ts, tm, dur, clip_length=2.0)
        return dict(pred_logits=out["pred_logits"], pred_spans=out["pred_spans"], windows=win, order=order, keep=keep, n_keep=nk, saliency=sal)

    def same(got, want):
        for k, v in want.items():
            assert torch.equal(got[k], v), k
    for seed in (1, 2, 3):                                    # three batches of one shape through ONE graph
        b = batch(8, 30, 10, seed)
        same(run(b[0]["src_txt"], b[0]["src_txt_mask"], b[0]["src_vid"], b[0]["src_vid_mask"], *b[1:]), eager(*b))
    assert len(run._graphs) == 1
    b2 = batch(3, 75, 12, 4)                                  # another shape: its own graph
    same(run(b2[0]["src_txt"], b2[0]["src_txt_mask"], b2[0]["src_vid"], b2[0]["src_vid_mask"], *b2[1:]), eager(*b2))
    assert len(run._graphs) == 2
    with torch.no_grad():                                     # parameter update -> the graph is re-captured with fresh operands
        model.weightedpool.weight.mul_(1.5)
        model.class_embed.layers[2].bias.add_(0.25)
    b = batch(8, 30, 10, 5)
    same(run(b[0]["src_txt"], b[0]["src_txt_mask"], b[0]["src_vid"], b[0]["src_vid_mask"], *b[1:]), eager(*b))
    # the shape cache is an LRU (the evaluation loop pads every batch to its own maximum length)
    small = GraphedInference(model, clip_length=2.0, max_graphs=2, clone_outputs=True)
    shapes = [(4, 20, 8), (4, 24, 8), (4, 28, 8), (4, 20, 8)]
    for i, (B, Lv, Lt) in enumerate(shapes):
        bb = batch(B, Lv, Lt, 10 + i)
        same(small(bb[0]["src_txt"], bb[0]["src_txt_mask"], bb[0]["src_vid"], bb[0]["src_vid_mask"], *bb[1:]), eager(*bb))
        assert len(small._graphs) <= 2
    # four calls over three shapes through two slots: the first shape was evicted before it came back -- four captures, two evictions, no hit
    assert small.stats == dict(hits=0, captures=4, evictions=2, recaptures_after_parameter_update=0), small.stats
    bb = batch(4, 20, 8, 13)
    small(bb[0]["src_txt"], bb[0]["src_txt_mask"], bb[0]["src_vid"], bb[0]["src_vid_mask"], *bb[1:])
    assert small.stats["hits"] == 1


def test_bench_two_rank_control_flow(dev):
    """bench.py launched the way the driver launches N>1 (torch.distributed.run, one process per rank), with gloo and both ranks
    on cuda:0 so that it runs on a 1-GPU box: every rank must take part in every collective (an instrumented step that only
    rank 0 ran once deadlocked this path) and rank 0 prints the one JSON line."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, UVTG_BENCH_BACKEND="gloo", UVTG_BENCH_SAME_DEVICE="1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--profile-steps", "1", "--batch", "32"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 64 and out["value"] > 0 and out["roofline"]["achieved"] > 0
    # the self-verification fields of a multi-rank run: the group really has N ranks, the replicas hold bit-identical
    # parameters after the timed loop, and the exposed-communication median is over EVERY step (no sample dropped, no sync in the loop)
    assert out["comm_world_size"] == 2 and out["world_size"] == 2
    assert out["replicas"]["flat_parameter_checksums_equal"] is True and out["replicas"]["ranks"] == 2
    assert out["exposed_comm_samples"] == 3 and out["exposed_comm_ms_per_step"] is not None
    assert out["median_over_steps"] >= 50 and out["config"]["variant"] == "A" and out["config"]["projections"] == "precise"
