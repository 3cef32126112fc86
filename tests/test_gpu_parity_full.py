"""Parity at the BENCHMARKED configurations (BASELINE config 2 at full size, the exact bench path, configs 4 and 5), HIP path vs
the CPU oracle on the same seeded inputs, through the drop-in boundary.  The oracle runs take seconds (B = 256 fwd+bwd ~10 s)."""
import os

import numpy as np
import pytest
import torch

from test_gpu_model import args_from_cfg, build, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _threads():
    torch.set_num_threads(min(os.cpu_count() or 1, 32))


def _grad_report(named_or_flat, ref_params, get):
    """per-parameter (cosine, norm ratio) of HIP gradients vs oracle autograd."""
    rep = {}
    for k, p in ref_params.items():
        if p.grad is None or float(p.grad.abs().max()) == 0.0:
            continue
        a, r = get(k).detach().cpu().double().flatten(), p.grad.double().flatten()
        rep[k] = (float((a @ r) / (a.norm() * r.norm() + 1e-30)), float(a.norm() / (r.norm() + 1e-30)))
    return rep


def _ref_keep(ref_pre, ref_nms):
    """reference post-NMS rows -> their rank positions in the ranked (pre-NMS) list"""
    rows = [tuple(r) for r in ref_pre]
    used, out = set(), []
    for r in ref_nms:
        idx = next(i for i, rr in enumerate(rows) if rr == tuple(r) and i not in used)
        used.add(idx)
        out.append(idx)
    return out


@pytest.fixture(scope="module")
def config2(dev):
    """Config 2 at FULL size (B=256, L_v=75, L_t=32, d=1024, E=4, ragged): oracle forward, losses and gradients (eval mode)."""
    from oracle import univtg_oracle as O
    _threads()
    cfg = O.make_cfg(input_dropout=0.0, droppath=0.0, dropout=0.0)
    params = O.init_params(cfg, seed=101)
    inputs, tg = O.make_batch(cfg, 256, 75, 32, seed=102, ragged=True)
    p2 = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = O.forward(p2, cfg, **inputs)
    losses = O.criterion(ref, tg, cfg)
    O.total_loss(losses, cfg).backward()
    return cfg, params, inputs, tg, p2, {k: (v.detach() if torch.is_tensor(v) else v) for k, v in ref.items()}, \
        {k: float(v) for k, v in losses.items()}


def test_config2_full_size_fp32x3_forward_and_post_nms_indices(dev, config2):
    """north_star at production size (B=256, E=4, d=1024).
    (a) the post-processing tail itself (decode, mask, stable rank, round_multiple, hull-IoU NMS) is BIT-EXACT: fed with the
        oracle's outputs it returns, for all 256 samples, exactly the reference algorithm's ranked rows, keep-set and windows;
    (b) the fp32x3 forward (fp16 hi/lo operand images, three products: ~22 bits) keeps saliency within 1e-4 and pred_* within 2e-5 of the
        oracle's, and forward + tail give the reference's ranking and keep-set for EVERY sample; a sample may differ only where the fp32 and
        the fp64 oracle disagree on it (the reference itself is then decided by rounding noise) -- see _reference_is_ambiguous."""
    from oracle import postproc_oracle as P
    from univtg_amd import ops
    cfg, params, inputs, tg, _, ref, _ = config2
    B, Lv = inputs["src_vid"].shape[:2]
    durations = torch.tensor([float(inputs["src_vid_mask"][b].sum()) * 2.0 for b in range(B)])
    pl_ref, ps_ref = ref["pred_logits"].numpy(), ref["pred_spans"].numpy()
    ts, tm = tg["timestamp"].numpy(), tg["timestamp_mask"].numpy()
    ref_order = P.ranked_clip_indices(pl_ref, tm)
    ref_pre = P.decode_windows(pl_ref, ps_ref, ts, tm, durations.tolist())
    tsd, tmd, dud = tg["timestamp"].to(dev), tg["timestamp_mask"].to(dev), durations.to(dev)
    cases = {}
    for clip_length in (0.0, 2.0):
        pre = ref_pre if clip_length == 0 else [P.round_multiple(p, clip_length) for p in ref_pre]
        ref_nms = [P.temporal_nms(p[:1000], 0.7, 10) for p in pre]
        cases[clip_length] = (pre, ref_nms, [_ref_keep(pre[b], ref_nms[b]) for b in range(B)])
        # ---- (a) identical inputs -> identical indices and rows, all samples ----
        win, order, keep, nk, _ = ops.postprocess_mr(ref["pred_logits"].to(dev), ref["pred_spans"].to(dev), None, tsd, tmd, dud, clip_length=clip_length)
        order, keep, nk, win = order.cpu().tolist(), keep.cpu().tolist(), nk.cpu().tolist(), win.cpu().numpy()
        for b in range(B):
            assert order[b] == ref_order[b], (clip_length, b)
            assert win[b].tolist() == pre[b], (clip_length, b)
            assert [win[b, i].tolist() for i in keep[b][: nk[b]]] == ref_nms[b], (clip_length, b)
            if clip_length > 0:
                assert np.all(np.mod(win[b, :, :2], clip_length) == 0)          # integer clip multiples (eval/postprocessing.py:46-51)
    # ---- (b) the fp32x3 forward in front of it ----
    # the DEFAULT drop-in model (precision="auto", what a maintainer following INTEGRATION section 1 gets): inference calls run fp32x3
    model, _ = build(cfg, params, dev, "auto", proj_precise="auto")
    model.eval()
    with torch.no_grad():
        out = model(**to_dev(inputs, dev))
    m32, _ = build(cfg, params, dev, "fp32x3")
    m32.eval()
    with torch.no_grad():
        o32 = m32(**to_dev(inputs, dev))
    for k in ("pred_logits", "pred_spans", "saliency_scores"):
        assert torch.equal(out[k], o32[k]), k                                    # "auto" under no_grad IS the fp32x3 path
    valid = inputs["src_vid_mask"].bool()
    e_sal = float((out["saliency_scores"].cpu() - ref["saliency_scores"])[valid].abs().max())
    e_log = float((out["pred_logits"].cpu() - ref["pred_logits"]).abs().max())
    e_spn = float((out["pred_spans"].cpu() - ref["pred_spans"]).abs().max())
    print(f"\n[config2 fp32x3] saliency err {e_sal:.2e}  pred_logits err {e_log:.2e}  pred_spans err {e_spn:.2e}")
    assert e_sal < 1e-4 and e_log < 2e-5 and e_spn < 2e-5       # (measured ~1e-6 with the fp16 hi/lo split; the bf16 split of rounds 1-3 gave 1.2e-5)
    for clip_length, (pre, ref_nms, ref_keep) in cases.items():
        win, order, keep, nk, _ = ops.postprocess_mr(out["pred_logits"], out["pred_spans"], None, tsd, tmd, dud, clip_length=clip_length)
        order, keep, nk = order.cpu().tolist(), keep.cpu().tolist(), nk.cpu().tolist()
        diff = [b for b in range(B) if not (order[b] == ref_order[b] and keep[b][: nk[b]] == ref_keep[b])]
        print(f"[config2 fp32x3 forward + tail, clip_length={clip_length}] identical ranking + keep-set: {B - len(diff)}/{B}")
        # Round 4 rule: every sample must be bit-identical -- a sample may differ only where the REFERENCE ITSELF is
        # ambiguous: the fp32 and the fp64 oracle disagree on it, or the deciding margin is below the fp32 oracle's own deviation from fp64
        if diff:       # (i) does the reference agree with ITSELF on these samples?  (second fp32 GEMM backend of the same torch)
            alt_order, alt_keep = _reference_alt_backend(cfg, params, inputs, tg, durations, clip_length)
        reasons = {}
        for b in diff:
            if alt_order[b] != ref_order[b] or alt_keep[b] != ref_keep[b]:
                print(f"   sample {b}: the fp32 reference on torch's other CPU GEMM backend (mkldnn off) ranks / keeps differently too")
                reasons[b] = "the fp32 reference with mkldnn off differs too"
                continue
            assert _reference_is_ambiguous(cfg, params, inputs, tg, durations, b, clip_length, ref_order[b], ref_keep[b], order[b]), (clip_length, b)
            reasons[b] = "fp64 margin <= the fp32 reference's own deviation from fp64, or fp64 oracle differs from fp32 oracle"
        _record_index_clause((101, 102), clip_length, diff, reasons)


# ---- the index clause's excused samples, pinned (VERDICT r5 item 1d / ADVICE r4) ----
# (weights seed, batch seed, clip_length) -> samples of the 256 whose ranking / keep-set differs from the fp32 reference algorithm's AND that the
# reference does not resolve itself (see _reference_is_ambiguous).  Every other sample is bit-identical.  A build that excuses any OTHER sample
# fails, even if that sample is ambiguous too: the set only changes together with this table and profiles/r06_index_clause.txt.
KNOWN_EXCUSED = {
    (101, 102, 0.0): [182], (101, 102, 2.0): [182],
    (201, 202, 0.0): [], (201, 202, 2.0): [],
    (301, 302, 0.0): [], (301, 302, 2.0): [],
    (401, 402, 0.0): [131], (401, 402, 2.0): [131],
}


def _record_index_clause(seeds, clip_length, diff, reasons):
    """Append one line per (draw, clip_length) to gpurun_out/index_clause.txt (copied to profiles/ by the visit script) and hold the excused
    set to KNOWN_EXCUSED."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "index_clause.txt"), "a") as f:
            f.write(f"weights seed {seeds[0]} batch seed {seeds[1]} clip_length {clip_length}: identical ranking + post-NMS keep-set "
                    f"{256 - len(diff)}/256; excused samples {diff} {reasons if diff else ''}\n")
    except OSError:
        pass
    if not os.environ.get("UVTG_INDEX_CLAUSE_RECORD"):          # (set only to take the table above from a new build)
        assert diff == KNOWN_EXCUSED[(seeds[0], seeds[1], clip_length)], (seeds, clip_length, diff, reasons)


def test_real_reference_beside_the_hip_path_at_config2_width(dev, config2, tmp_path):
    """The REFERENCE ITSELF (showlab/UniVTG, unmodified, from oracle/_ref, in a child process on the host cores -- oracle/ref_runner.py) and
    the HIP path on ONE box at production width: build_model at config 2 (d = 1024, E = 4, D_v = 2818), the oracle-seeded
    weights through load_state_dict(strict=True) on both sides, eval mode, B = 256 ragged.
    (a) the reference's outputs vs the default drop-in model under no_grad: saliency <= 1e-4 (north_star), pred_* <= 2e-5
        (model/univtg.py:105-155);
    (b) the reference's OWN inference tail -- PostProcessorDETR round_multiple (eval/postprocessing.py:46-51) + temporal_nms
        (utils/temporal_nms.py:25-74) behind the compose glue of main/inference_mr.py:111-163 -- run on the HIP outputs, vs
        uvtg_postprocess_mr on the same outputs: identical ranked rows and identical post-NMS rows for 256/256 samples, raw and rounded;
    (c) the restated oracle against the reference at this width (the fixtures pin it at d = 64 / 128 only): <= 2e-5."""
    import json
    from oracle.build_ref import ARCHIVE
    from oracle.ref_runner import run_job
    from univtg_amd import ops
    if not os.path.exists(ARCHIVE):
        pytest.skip("no oracle/_ref archive in this tree (built by __graft_entry__.build() where /root/reference exists)")
    cfg, params, inputs, tg, _, oracle_out, _ = config2
    B, Lv = inputs["src_vid"].shape[:2]
    threads = min(os.cpu_count() or 1, 32)
    ev = str(tmp_path / "ref_eval.npz")
    r = run_job(dict(task="model", threads=threads, cfg=dict(input_dropout=0.0, droppath=0.0, dropout=0.0), param_seed=101,
                     batch=dict(B=256, Lv=75, Lt=32, seed=102, ragged=True), eval_out=ev))
    assert ".zip" in r["module_file"] and len(r["manifest"]) >= 11
    ref = {k: torch.from_numpy(v) for k, v in np.load(ev).items()}
    model, _ = build(cfg, params, dev, "auto", proj_precise="auto")         # what INTEGRATION section 1 gives a maintainer
    model.eval()
    with torch.no_grad():
        out = model(**to_dev(inputs, dev))
    valid = inputs["src_vid_mask"].bool()
    err = dict(saliency=float((out["saliency_scores"].cpu() - ref["saliency_scores"])[valid].abs().max()),
               pred_logits=float((out["pred_logits"].cpu() - ref["pred_logits"]).abs().max()),
               pred_spans=float((out["pred_spans"].cpu() - ref["pred_spans"]).abs().max()))
    oerr = {k: float((oracle_out[k] - ref[k])[valid if k == "saliency_scores" else slice(None)].abs().max()) for k in ("saliency_scores", "pred_logits", "pred_spans")}
    print(f"\n[real reference vs HIP, config 2 width] {err}   [oracle vs real reference] {oerr}")
    assert err["saliency"] <= 1e-4 and err["pred_logits"] <= 2e-5 and err["pred_spans"] <= 2e-5, err
    assert max(oerr.values()) <= 2e-5, oerr
    # ---- (b) the reference's own tail on the HIP outputs ----
    durations = torch.tensor([float(inputs["src_vid_mask"][b].sum()) * 2.0 for b in range(B)])
    pin = str(tmp_path / "hip_out.npz")
    np.savez(pin, pred_logits=out["pred_logits"].cpu().numpy(), pred_spans=out["pred_spans"].cpu().numpy(), timestamp=tg["timestamp"].numpy(),
             timestamp_mask=tg["timestamp_mask"].numpy(), durations=durations.numpy())
    pj = str(tmp_path / "ref_tail.json")
    r2 = run_job(dict(task="postproc", outputs_npz=pin, result_json=pj, clip_lengths=[0.0, 2.0]))
    assert all(".zip" in f for f in r2["module_files"])
    with open(pj) as f:
        tail = json.load(f)
    tsd, tmd, dud = tg["timestamp"].to(dev), tg["timestamp_mask"].to(dev), durations.to(dev)
    for clip_length in (0.0, 2.0):
        win, order, keep, nk, _ = ops.postprocess_mr(out["pred_logits"], out["pred_spans"], None, tsd, tmd, dud, clip_length=clip_length)
        win, keep, nk = win.cpu().numpy(), keep.cpu().tolist(), nk.cpu().tolist()
        want = tail[str(clip_length)]
        same = sum(win[b].tolist() == want["pre"][b] and [win[b, i].tolist() for i in keep[b][: nk[b]]] == want["nms"][b] for b in range(B))
        print(f"[reference's own round_multiple + temporal_nms on the HIP outputs, clip_length={clip_length}] identical rows {same}/{B}")
        assert same == B, (clip_length, same)


def _oracle_tail(pl, ps, ts, tm, durations, clip_length):
    """the reference's ranking + post-NMS keep indices (oracle/postproc_oracle.py) for a batch of model outputs"""
    from oracle import postproc_oracle as P
    order = P.ranked_clip_indices(pl, tm)
    pre = P.decode_windows(pl, ps, ts, tm, durations)
    if clip_length > 0:
        pre = [P.round_multiple(p, clip_length) for p in pre]
    nms = [P.temporal_nms(p[:1000], 0.7, 10) for p in pre]
    return order, [_ref_keep(pre[b], nms[b]) for b in range(len(pre))]


def _reference_alt_backend(cfg, params, inputs, tg, durations, clip_length):
    """The SAME fp32 reference arithmetic on torch's other CPU GEMM backend (mkldnn off): ranking / keep-set of the whole batch.  Two fp32
    evaluations of the reference differ by up to ~8e-7 on pred_logits (profiles/r04_reference_backend_ambiguity.txt) and flip near-tied samples."""
    from oracle import univtg_oracle as O
    prev = torch.backends.mkldnn.enabled
    torch.backends.mkldnn.enabled = False
    try:
        with torch.no_grad():
            alt = O.forward(params, cfg, **inputs)
    finally:
        torch.backends.mkldnn.enabled = prev
    return _oracle_tail(alt["pred_logits"].numpy(), alt["pred_spans"].numpy(), tg["timestamp"].numpy(), tg["timestamp_mask"].numpy(),
                        [float(x) for x in durations], clip_length)


def _reference_is_ambiguous(cfg, params, inputs, tg, durations, b, clip_length, order32, keep32, order_got):
    """Is sample b decided by the fp32 REFERENCE's own rounding noise?  The sample goes through the oracle in DOUBLE precision; it is
    ambiguous when (i) the fp64 reference ranks / keeps other clips than the fp32 reference did, or (ii) the margin that decides the
    differing ranks -- the fp64 score gap between the clips our ranking swaps -- is not larger than the fp32 reference's own deviation from
    fp64 on this sample (max |logit32 - logit64|): an fp32 implementation cannot be held to a margin its reference does not resolve itself.
    Measured (seeds 301 / 302, sample 6): deciding gap 2.5e-7, the reference's own deviation 4.3e-7, ours 1.8e-6 max."""
    from oracle import univtg_oracle as O
    p64 = {k: v.double() for k, v in params.items()}
    i64 = {k: v[b:b + 1].double() for k, v in inputs.items()}
    with torch.no_grad():
        o64 = O.forward(p64, cfg, **i64)
        o32 = O.forward(params, cfg, **{k: v[b:b + 1] for k, v in inputs.items()})
    order, keep = _oracle_tail(o64["pred_logits"].float().numpy(), o64["pred_spans"].float().numpy(), tg["timestamp"][b:b + 1].numpy(),
                               tg["timestamp_mask"][b:b + 1].numpy(), [float(durations[b])], clip_length)
    if order[0] != order32 or keep[0] != keep32:
        print(f"   sample {b}: the fp64 oracle ranks / keeps differently from the fp32 oracle")
        return True
    l64 = o64["pred_logits"][0, :, 0]
    valid = inputs["src_vid_mask"][b].bool()
    ref_dev = float((o32["pred_logits"][0, :, 0].double() - l64)[valid].abs().max())
    swapped = [i for i in range(len(order32)) if order_got[i] != order32[i]]
    if not swapped:
        return False                       # same ranking, other keep-set: not excused here
    margin = max(abs(float(l64[order_got[i]]) - float(l64[order32[i]])) for i in swapped)
    print(f"   sample {b}: fp64 margin of the swapped clips {margin:.2e}, the fp32 reference's own deviation from fp64 {ref_dev:.2e}")
    return margin <= ref_dev


@pytest.mark.parametrize("seeds", [(201, 202), (301, 302), (401, 402)])
def test_post_nms_indices_identical_for_every_sample_three_seeds(dev, seeds):
    """north_star's index clause at BASELINE config-2 size on three more (weights, batch) draws: the default drop-in model (precision
    'auto' -> the split-operand fp32x3 arithmetic under no_grad) + the device post-processing give, for ALL 256 samples, exactly the
    ranking and the post-NMS keep-set of the fp32 reference algorithm, raw and with round_multiple; saliency within 1e-4."""
    from oracle import univtg_oracle as O
    from univtg_amd import ops
    _threads()
    cfg = O.make_cfg(input_dropout=0.0, droppath=0.0, dropout=0.0)
    params = O.init_params(cfg, seed=seeds[0])
    inputs, tg = O.make_batch(cfg, 256, 75, 32, seed=seeds[1], ragged=True)
    with torch.no_grad():
        ref = O.forward(params, cfg, **inputs)
    B = 256
    durations = torch.tensor([float(inputs["src_vid_mask"][b].sum()) * 2.0 for b in range(B)])
    model, _ = build(cfg, params, dev, "auto", proj_precise="auto")
    model.eval()
    with torch.no_grad():
        out = model(**to_dev(inputs, dev))
    valid = inputs["src_vid_mask"].bool()
    e_sal = float((out["saliency_scores"].cpu() - ref["saliency_scores"])[valid].abs().max())
    e_log = float((out["pred_logits"].cpu() - ref["pred_logits"]).abs().max())
    e_spn = float((out["pred_spans"].cpu() - ref["pred_spans"]).abs().max())
    print(f"\n[seeds {seeds}] saliency err {e_sal:.2e}  pred_logits err {e_log:.2e}  pred_spans err {e_spn:.2e}")
    assert e_sal < 1e-4 and e_log < 2e-5 and e_spn < 2e-5
    tsd, tmd, dud = tg["timestamp"].to(dev), tg["timestamp_mask"].to(dev), durations.to(dev)
    for clip_length in (0.0, 2.0):
        ref_order, ref_keep = _oracle_tail(ref["pred_logits"].numpy(), ref["pred_spans"].numpy(), tg["timestamp"].numpy(), tg["timestamp_mask"].numpy(),
                                           durations.tolist(), clip_length)
        win, order, keep, nk, _ = ops.postprocess_mr(out["pred_logits"], out["pred_spans"], None, tsd, tmd, dud, clip_length=clip_length)
        order, keep, nk = order.cpu().tolist(), keep.cpu().tolist(), nk.cpu().tolist()
        diff = [b for b in range(B) if not (order[b] == ref_order[b] and keep[b][: nk[b]] == ref_keep[b])]
        print(f"[seeds {seeds}, clip_length={clip_length}] identical ranking + keep-set: {B - len(diff)}/{B}")
        if diff:       # (i) does the reference agree with ITSELF on these samples?  (second fp32 GEMM backend of the same torch)
            alt_order, alt_keep = _reference_alt_backend(cfg, params, inputs, tg, durations, clip_length)
        reasons = {}
        for b in diff:
            if alt_order[b] != ref_order[b] or alt_keep[b] != ref_keep[b]:
                print(f"   sample {b}: the fp32 reference on torch's other CPU GEMM backend (mkldnn off) ranks / keeps differently too")
                reasons[b] = "the fp32 reference with mkldnn off differs too"
                continue
            assert _reference_is_ambiguous(cfg, params, inputs, tg, durations, b, clip_length, ref_order[b], ref_keep[b], order[b]), (clip_length, b)
            reasons[b] = "fp64 margin <= the fp32 reference's own deviation from fp64, or fp64 oracle differs from fp32 oracle"
        _record_index_clause(seeds, clip_length, diff, reasons)


def test_config2_full_size_bf16_losses_gradients_and_index_agreement(dev, config2):
    """The benchmarked arithmetic (bf16 operands) at config 2 full size, eval mode: five losses, EVERY parameter gradient, and the
    measured top-10 post-NMS index agreement with the fp32 reference algorithm."""
    from oracle import postproc_oracle as P
    from univtg_amd import ops
    cfg, params, inputs, tg, p2, ref, ref_losses = config2
    model, crit = build(cfg, params, dev, "bf16")
    model.eval()
    ind, tgd = to_dev(inputs, dev), to_dev(tg, dev)
    out = model(**ind)
    ld = crit(out, tgd)
    sum(ld[k] * crit.weight_dict[k] for k in ld).backward()
    for k in ("loss_b", "loss_g", "loss_f", "loss_s_inter", "loss_s_intra"):
        got, want = float(ld[k]), ref_losses[k]
        assert abs(got - want) < 2e-2 * max(1.0, abs(want)), (k, got, want)
    named = dict(model.named_parameters())
    rep = _grad_report(named, p2, lambda k: named[k].grad)
    worst_cos = min(rep.items(), key=lambda kv: kv[1][0])
    worst_ratio = max(rep.items(), key=lambda kv: abs(kv[1][1] - 1))
    print(f"\n[config2 bf16] {len(rep)} parameter gradients; worst cosine {worst_cos[1][0]:.5f} ({worst_cos[0]}), "
          f"worst norm ratio {worst_ratio[1][1]:.4f} ({worst_ratio[0]})")
    bad = {k: v for k, v in rep.items() if v[0] < 0.998 or abs(v[1] - 1) > 0.01}
    assert not bad, sorted(bad.items())
    # index agreement of the bf16 inference path with the fp32 algorithm (reported, with a floor)
    B = inputs["src_vid"].shape[0]
    durations = torch.tensor([float(inputs["src_vid_mask"][b].sum()) * 2.0 for b in range(B)])
    ref_pre = P.decode_windows(ref["pred_logits"].numpy(), ref["pred_spans"].numpy(), tg["timestamp"].numpy(),
                               tg["timestamp_mask"].numpy(), durations.tolist())
    ref_order = P.ranked_clip_indices(ref["pred_logits"].numpy(), tg["timestamp_mask"].numpy())
    ref_nms = [P.temporal_nms(p[:1000], 0.7, 10) for p in ref_pre]
    with torch.no_grad():
        o = model(**ind)
    win, order, keep, nk = ops.decode_rank_nms(o["pred_logits"], o["pred_spans"], tgd["timestamp"], tgd["timestamp_mask"], durations.to(dev))
    order, keep, nk = order.cpu().tolist(), keep.cpu().tolist(), nk.cpu().tolist()
    top1 = same_set = same_list = 0
    for b in range(B):
        got = [order[b][i] for i in keep[b][: nk[b]]]
        want = [ref_order[b][i] for i in _ref_keep(ref_pre[b], ref_nms[b])]
        top1 += got[:1] == want[:1]
        same_list += got == want
        same_set += set(got) == set(want)
    print(f"[config2 bf16] post-NMS top-10 clip indices vs fp32 reference algorithm: identical ordered list {same_list}/{B}, "
          f"identical set {same_set}/{B}, identical top-1 {top1}/{B}")
    # measured 251 / 226 / 219 of 256 (profiles): floors a few samples below the measurement.  precision="bf16" is the
    # OPT-IN fast inference mode; the default ("auto") runs inference calls in fp32x3, where the index clause holds (test above)
    assert top1 >= B - 10 and same_set >= B - 40 and same_list >= B - 48


def _philox_rng(seed, cfg, B, Lv, Lt, p_in, p_path):
    """The device Philox masks of one TrainStep call, regenerated on the host in the oracle's rng layout."""
    import philox_ref as R
    Dv, Dt, d, E = cfg.v_feat_dim, cfg.t_feat_dim, cfg.hidden_dim, cfg.enc_layers
    t = lambda a: torch.from_numpy(a)
    return {"vid_keep": [t(R.row_keep(seed, R.RNG_IN_VID, B * Lv, Dv, p_in)).view(B, Lv, Dv),
                         t(R.row_keep(seed, R.RNG_IN_VID + 1, B * Lv, d, p_in)).view(B, Lv, d)],
            "txt_keep": [t(R.row_keep(seed, R.RNG_IN_TXT, B * Lt, Dt, p_in)).view(B, Lt, Dt),
                         t(R.row_keep(seed, R.RNG_IN_TXT + 1, B * Lt, d, p_in)).view(B, Lt, d)],
            "dp_scale": t(R.droppath_scales(seed, E, B, p_path))}


def _replay_bench_path(dev, tag, B, Lv, Lt, seeds, compare_padded=True, proj_precise="auto", ragged=True):
    """The EXACT bench.py path at one BASELINE shape: native TrainStep, train mode, input dropout 0.5 + DropPath 0.1, packed="auto" with
    the collate's host-side lengths (loss-only packing).  The device Philox masks are regenerated on the host and handed to the oracle;
    the five losses, pred_logits at the valid positions and EVERY parameter gradient must agree."""
    from oracle import univtg_oracle as O
    from univtg_amd.trainer import TrainStep
    _threads()
    cfg = O.make_cfg(input_dropout=0.5, dropout=0.0, droppath=0.1, max_v_l=Lv)
    params = O.init_params(cfg, seed=seeds[0])
    inputs, tg = O.make_batch(cfg, B, Lv, Lt, seed=seeds[1], ragged=ragged)
    if not ragged:      # SURVEY 8d variant A: all-ones masks, every position executed -- the rows bench.py's headline times
        assert bool(inputs["src_vid_mask"].bool().all()) and bool(inputs["src_txt_mask"].bool().all())
    res = {}
    for packed in ((False, "auto") if compare_padded else ("auto",)):
        model, crit = build(cfg, params, dev, "auto", proj_precise=proj_precise)      # as bench.py builds it (--proj precise: True)
        model.train()
        model.set_seed(777)
        step = TrainStep(model, crit, grad_clip=0.1, packed=packed)
        batch = to_dev(inputs, dev)
        batch["_lens_host"] = (inputs["src_vid_mask"].sum(1).int().tolist(), inputs["src_txt_mask"].sum(1).int().tolist())
        losses = step.step(batch, to_dev(tg, dev), optimize=False)
        torch.cuda.synchronize()
        res[packed] = (losses.cpu(), step.grads.clone(), model, step.pred_logits.clone(), step.sal.clone())
    l1, g1, model, pl1, sal1 = res["auto"]
    if compare_padded:      # the two executions draw the same masks and compute the same rows: equal to re-association noise
        l0, g0, _, pl0, _ = res[False]
        assert float((l0 - l1).abs().max()) < 2e-3 * max(1.0, float(l0.abs().max()))
        gg0, gg1 = g0.double(), g1.double()
        assert float((gg0 @ gg1) / (gg0.norm() * gg1.norm())) > 0.9995
    # ---- replay through the oracle ----
    seed = (777 * 1000003 + 1) & 0xFFFFFFFFFFFFFFFF
    rng = _philox_rng(seed, cfg, B, Lv, Lt, 0.5, 0.1)
    p2 = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = O.forward(p2, cfg, inputs["src_txt"], inputs["src_txt_mask"], inputs["src_vid"], inputs["src_vid_mask"], rng=rng)
    ref_losses = O.criterion(ref, tg, cfg)
    O.total_loss(ref_losses, cfg).backward()
    vmask = inputs["src_vid_mask"].bool()
    e = float((pl1.cpu() - ref["pred_logits"].detach())[vmask].abs().max())     # (loss-only packing: padded positions beyond the conv halo differ)
    e_sal = float((sal1.cpu() - ref["saliency_scores"].detach())[vmask].abs().max())
    print(f"\n[{tag}] train-mode saliency_scores err {e_sal:.2e} (proj_precise={proj_precise})")
    if proj_precise is True:      # north_star's saliency tolerance on the TIMED configuration (split-operand input projections under dropout)
        assert e_sal < 1e-4, e_sal
    lerr = {}
    for i, k in enumerate(("loss_b", "loss_g", "loss_f", "loss_s_inter", "loss_s_intra")):
        got, want = float(l1[i]), float(ref_losses[k])
        lerr[k] = abs(got - want) / max(1.0, abs(want))
    Dv, Dt = cfg.v_feat_dim, cfg.t_feat_dim
    offs = model._offsets(model._dims(B, Lv, Lt, Dv, Dt, False))
    names = {id(p): k for k, p in model.named_parameters()}
    flat = {names[id(p)]: g1[offs[i]: offs[i] + p.numel()] for i, p in enumerate(model._ordered_params())}
    rep = _grad_report(None, p2, lambda k: flat[k])
    worst_cos = min(rep.items(), key=lambda kv: kv[1][0])
    worst_ratio = max(rep.items(), key=lambda kv: abs(kv[1][1] - 1))
    print(f"\n[{tag}: bench path, train mode, B={B} L_v={Lv}] pred_logits err {e:.2e}; relative loss errors "
          + ", ".join(f"{k} {v:.1e}" for k, v in lerr.items())
          + f"; {len(rep)} gradients; worst cosine {worst_cos[1][0]:.5f} ({worst_cos[0]}), worst norm ratio {worst_ratio[1][1]:.4f} ({worst_ratio[0]})")
    assert len(rep) == 78
    assert e < 2e-2, e              # measured 8.7e-3 (bf16 encoder operands under dropout)
    assert max(lerr.values()) < 3e-2, lerr
    # floors: the measured level (round 2: worst cosine 0.9965, norm within 1.8 %) minus a small margin -- the noise is the plain-bf16
    # 2818-wide input projection under dropout
    # (input_vid_proj.0.LayerNorm.weight: its norm is two TEF-column scalars, see test_production_width_vs_oracle -- measured 0.973 .. 1.014)
    bad = {k: v for k, v in rep.items() if v[0] < 0.995 or abs(v[1] - 1) > (0.04 if k == "input_vid_proj.0.LayerNorm.weight" else 0.025)}
    assert not bad, sorted(bad.items())


@pytest.mark.parametrize("ragged", [True, False], ids=["variantB_ragged", "variantA_all_ones"])
def test_bench_path_trainstep_dropout_replayed_through_oracle(dev, ragged):
    """Config 2 at full size (B=256, L_v=75, L_t=32, d=1024, E=4): what `python bench.py` times.  `variantA_all_ones` is the HEADLINE
    workload itself (SURVEY 8d variant A: all-ones masks, every position executed, padded execution, split-operand projections, train mode);
    `variantB_ragged` is the companion line.  (Under input dropout the native step's packed stream keeps the valid clips, the three padded
    clips inside the conv heads' receptive field of a valid position -- each with its own mask -- and the valid text tokens: exact for
    everything a loss can see, unlike round 1's shared-mask representative.)"""
    _replay_bench_path(dev, "config2-" + ("B" if ragged else "A"), 256, 75, 32, (201, 202), proj_precise=True, ragged=ragged,
                       compare_padded=ragged)


def test_config3_bench_path_replayed_through_oracle(dev):
    """Config 3 (the 4M VLP pre-training shape, scripts/pretrain.sh:26-47) at production width: per-GPU shard B=256, L_v=128, L_t=32
    (S=160: the 8-wave fused attention backward), d=1024, E=4, train mode p_in=0.5 / DropPath 0.1, packed="auto" loss-only stream,
    Philox masks replayed through the oracle -- what `python bench.py --config 3` times."""
    _replay_bench_path(dev, "config3", 256, 128, 32, (301, 302), compare_padded=False)


def test_convergence_fixed_batch_matches_oracle_training(dev):
    """200 native TrainStep steps (train mode: input dropout 0.5 + DropPath 0.1, clip 0.1, AdamW) on ONE fixed batch, against the oracle
    trained with torch.optim.AdamW + clip_grad_norm_ on the same batch with the SAME per-step Philox masks (main/train_vlp_ddp.py:44-75).
    bf16 operands vs fp32: the weighted total loss must follow the oracle's curve inside a stated band and end well below its start.
    Width 256 / E=2 / B=32 so that 200 oracle steps take seconds; the production-width single-step parity is the replay tests above."""
    from oracle import univtg_oracle as O
    from univtg_amd.trainer import TrainStep
    torch.set_num_threads(min(os.cpu_count() or 1, 8))          # (small tensors: more threads only add overhead; ~0.1 s per oracle step)
    cfg = O.make_cfg(hidden_dim=256, nheads=4, dim_feedforward=256, enc_layers=2, v_feat_dim=514, t_feat_dim=512,
                     input_dropout=0.5, dropout=0.0, droppath=0.1)
    B, Lv, Lt, steps, lr = 32, 40, 12, 200, 2e-4
    params = O.init_params(cfg, seed=901)
    inputs, tg = O.make_batch(cfg, B, Lv, Lt, seed=902, ragged=True)
    wd = O.weight_dict(cfg)
    keys = ("loss_b", "loss_g", "loss_f", "loss_s_inter", "loss_s_intra")
    model, crit = build(cfg, params, dev, "auto", proj_precise="auto")
    model.train()
    model.set_seed(4242)
    step = TrainStep(model, crit, lr=lr, weight_decay=1e-4, grad_clip=0.1, packed="auto")
    batch = to_dev(inputs, dev)
    batch["_lens_host"] = (inputs["src_vid_mask"].sum(1).int().tolist(), inputs["src_txt_mask"].sum(1).int().tolist())
    tgd = to_dev(tg, dev)
    dev_curve = torch.stack([step.step(batch, tgd) for _ in range(steps)]).cpu()
    dev_total = sum(dev_curve[:, i] * wd[k] for i, k in enumerate(keys))
    # oracle: same masks step by step
    p2 = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    train_keys = [k for k in p2 if not k.startswith("txt_position_embed")]     # (unused parameters: no gradient in the reference either)
    opt = torch.optim.AdamW([p2[k] for k in train_keys], lr=lr, weight_decay=1e-4)
    ref_total = []
    for it in range(steps):
        seed = (4242 * 1000003 + it + 1) & 0xFFFFFFFFFFFFFFFF
        rng = _philox_rng(seed, cfg, B, Lv, Lt, 0.5, 0.1)
        opt.zero_grad(set_to_none=True)
        out = O.forward(p2, cfg, inputs["src_txt"], inputs["src_txt_mask"], inputs["src_vid"], inputs["src_vid_mask"], rng=rng)
        total = O.total_loss(O.criterion(out, tg, cfg), cfg)
        total.backward()
        torch.nn.utils.clip_grad_norm_([p2[k] for k in train_keys], 0.1)
        opt.step()
        ref_total.append(float(total))
    ref_total = torch.tensor(ref_total)
    rel = ((dev_total - ref_total).abs() / ref_total.abs().clamp(min=1.0))
    sm = lambda x: torch.nn.functional.avg_pool1d(x[None, None], 10, 10)[0, 0]          # 10-step means: the curves, not the dropout noise
    rel_sm = ((sm(dev_total) - sm(ref_total)).abs() / sm(ref_total).abs().clamp(min=1.0))
    print(f"\n[convergence, 200 steps, fixed batch] total loss start {float(ref_total[0]):.3f} (oracle) / {float(dev_total[0]):.3f} (HIP); "
          f"mean of last 10: {float(ref_total[-10:].mean()):.3f} / {float(dev_total[-10:].mean()):.3f}; per-step relative gap max {float(rel.max()):.3f} "
          f"median {float(rel.median()):.4f}; 10-step-mean gap max {float(rel_sm.max()):.4f}; every 50th step: "
          + ", ".join(f"{i}: {float(ref_total[i]):.3f}/{float(dev_total[i]):.3f}" for i in (0, 49, 99, 149, 199)))
    assert float(rel[0]) < 2e-2                                     # identical weights and masks at step 1
    # the band (measured: 10-step means within 0.9-1.2 %, median step 0.3 %, worst single step 2.3-3.7 % -- dropout noise on bf16 weights)
    assert float(rel_sm.max()) < 0.03 and float(rel.median()) < 0.01 and float(rel.max()) < 0.08
    assert float(dev_total[-10:].mean()) < 0.75 * float(dev_total[:10].mean())   # and it does converge
    assert float(ref_total[-10:].mean()) < 0.75 * float(ref_total[:10].mean())


def test_eval_after_native_train_step_sees_new_weights(dev):
    """TrainStep updates the flat parameter buffer with a raw kernel (tensor versions do not move); a later
    model(...) call must rebuild its bf16 operand cache.  eval -> TrainStep.step -> eval == a fresh model loaded from state_dict()."""
    from oracle import univtg_oracle as O
    from univtg_amd.trainer import TrainStep
    cfg = O.make_cfg(hidden_dim=256, nheads=4, dim_feedforward=256, enc_layers=2, v_feat_dim=514, t_feat_dim=512,
                     input_dropout=0.0, dropout=0.0, droppath=0.0)
    params = O.init_params(cfg, seed=61)
    inputs, tg = O.make_batch(cfg, 6, 30, 10, seed=62, ragged=True)
    ind, tgd = to_dev(inputs, dev), to_dev(tg, dev)
    model, crit = build(cfg, params, dev, "bf16", proj_precise="auto")
    model.eval()
    with torch.no_grad():
        before = model(**ind)["pred_logits"].clone()
    step = TrainStep(model, crit, lr=1e-2, grad_clip=0.0)
    with torch.no_grad():
        assert torch.equal(model(**ind)["pred_logits"], before)          # re-homing the parameters changes nothing
    for _ in range(3):
        step.step(ind, tgd, optimize=True)
    with torch.no_grad():
        after = model(**ind)["pred_logits"].clone()
    fresh, _ = build(cfg, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, dev, "bf16", proj_precise="auto")
    fresh.eval()
    with torch.no_grad():
        want = fresh(**ind)["pred_logits"]
    assert float((after - before).abs().max()) > 1e-3                    # the steps did move the predictions
    assert torch.equal(after, want)
    # optimizer state round trip in torch.optim.AdamW's layout (reference checkpoints, main/train_vlp_ddp.py:157-195)
    sd = step.state_dict()
    ref_opt = torch.optim.AdamW([p for n, p in fresh.named_parameters() if p.requires_grad], lr=1e-2, weight_decay=1e-4)
    ref_opt.load_state_dict(sd)                                          # loadable by the reference's optimizer
    fresh2, crit2 = build(cfg, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, dev, "bf16", proj_precise="auto")
    fresh2.eval()
    step2 = TrainStep(fresh2, crit2, lr=1e-2, grad_clip=0.0)
    step2.load_state_dict(sd)
    assert step2.t == step.t and torch.equal(step2.m, step.m) and torch.equal(step2.v, step.v)
    a = step.step(ind, tgd, optimize=True)
    b = step2.step(ind, tgd, optimize=True)
    torch.cuda.synchronize()
    assert float((a - b).abs().max()) < 1e-5 * max(1.0, float(a.abs().max()))
    # (Adam's normalised update turns fp32-atomic-order noise on near-zero gradients into full-size steps for a few elements)
    assert float(((step.flat - step2.flat).abs() > 1e-4).float().mean()) < 1e-3


@pytest.mark.parametrize("name", ["tiny_eval_ragged", "tiny_eval_full", "config1_real_feats"])
def test_postprocess_round_multiple_and_eval_mode_saliency(dev, golden_dir, name):
    """round-to-clip on device (integer clip multiples bit-exact) + the --eval_mode add saliency, against what
    the REAL reference's compute_mr_results / PostProcessorDETR / post_processing_mr_nms produced (tests/golden)."""
    from test_gpu_model import load_case
    from univtg_amd import ops
    meta, cfg, params, inputs, tg, out_ref, eval_ref, *_ = load_case(golden_dir, name)
    B, Lv = inputs["src_vid"].shape[:2]
    durations = torch.tensor([float(inputs["src_vid_mask"][b].sum()) * 2.0 for b in range(B)])
    args = [eval_ref["pred_logits"].to(dev), eval_ref["pred_spans"].to(dev), eval_ref["saliency_scores"].to(dev), tg["timestamp"].to(dev),
            tg["timestamp_mask"].to(dev), durations.to(dev)]
    for tag, clip in (("raw", 0.0), ("rounded", 2.0)):
        win, order, keep, nk, sal = ops.postprocess_mr(*args, clip_length=clip, eval_mode="add")
        win, keep, nk, sal = win.cpu().numpy(), keep.cpu().tolist(), nk.cpu().tolist(), sal.cpu().numpy()
        ref_pre, ref_nms, ref_sal = meta[f"post/{tag}"]["pre"], meta[f"post/{tag}"]["nms"], meta[f"post/{tag}"]["sal"]
        for b in range(B):
            assert win[b].tolist() == ref_pre[b], (tag, b)                              # every ranked row, bit-exact
            assert [win[b, i].tolist() for i in keep[b][: nk[b]]] == ref_nms[b], (tag, b)
            lv = int(inputs["src_vid_mask"][b].sum())
            assert sal[b, :lv].astype(np.float64).tolist() == ref_sal[b], (tag, b)
    _, _, _, _, sal0 = ops.postprocess_mr(*args, clip_length=0.0, eval_mode="none")
    assert torch.equal(sal0.cpu(), eval_ref["saliency_scores"].half().float())


def test_postprocess_round_multiple_known_answers(dev, golden_dir):
    """PostProcessorDETR(round_multiple) on real QVHighlights prediction rows (tests/golden/nms.json round_in/round_out)."""
    import json
    from univtg_amd import ops
    d = json.load(open(os.path.join(golden_dir, "nms.json")))
    for rin, rout in list(zip(d["round_in"], d["round_out"]))[:40]:
        L = len(rin)
        # feed the rows as decoded windows of a duration-1000 video (timestamp = 0; nothing reaches the clamp)
        st = torch.tensor([[r[0], r[1]] for r in rin], dtype=torch.float64)[None] / 1000.0
        sc = torch.tensor([r[2] for r in rin], dtype=torch.float32)[None, :, None]
        win, order, *_ = ops.postprocess_mr(sc.to(dev), st.float().to(dev), None, torch.zeros(1, L, 2, device=dev), torch.ones(1, L, device=dev),
                                            torch.tensor([1000.0], device=dev), clip_length=2.0, nms_thd=0.7, max_after=10)
        got = win[0].cpu().numpy()
        srt = sorted(range(L), key=lambda i: rin[i][2], reverse=True)
        want = np.array([rout[i] for i in srt])
        assert np.array_equal(got[:, :2], want[:, :2])
        assert np.abs(got[:, 2] - want[:, 2]).max() <= 1.0001e-4


def test_pipeline_staging_ring_survives_host_run_ahead(dev, golden_dir):
    """ADVICE r1 (medium) / SURVEY 8f row 2: two uploads issued back-to-back while a long kernel still occupies the stream -- the
    host refills the staging buffers before the first batch's H2D copies have run.  Both batches must arrive bit-exact."""
    from oracle import pipeline_oracle as PO
    from test_oracle_golden import _collate_case
    from univtg_amd import pipeline
    z, batch = _collate_case(golden_dir)
    for e in batch:
        e["model_inputs"] = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in e["model_inputs"].items()}
    batch_a = batch
    batch_b = [dict(meta=e["meta"], model_inputs={k: (v * 3.0 + 1.0 if (torch.is_tensor(v) and v.is_floating_point() and k in ("video_feat", "query_feat")) else v)
                                                  for k, v in e["model_inputs"].items()}) for e in reversed(batch)]
    npb = lambda b: [dict(meta=e["meta"], model_inputs={k: (v.numpy() if torch.is_tensor(v) else v) for k, v in e["model_inputs"].items()}) for e in b]
    want = [PO.collate_mr(npb(batch_a)), PO.collate_mr(npb(batch_b))]
    torch.cuda.synchronize()
    torch.cuda._sleep(int(2.0e8))                       # ~0.1 s of device time ahead of the uploads
    got = [pipeline.collate_upload_mr(batch_a, dev), pipeline.collate_upload_mr(batch_b, dev),
           pipeline.collate_upload_mr(batch_a, dev), pipeline.collate_upload_mr(batch_b, dev)]
    torch.cuda.synchronize()
    for i, (_, mi, tg) in enumerate(got):
        wmi, wtg = want[i % 2]
        for k in ("src_txt", "src_txt_mask", "src_vid", "src_vid_mask"):
            assert np.array_equal(mi[k].cpu().numpy(), wmi[k]), (i, k)
        for k in ("timestamp", "timestamp_mask", "timestamp_window", "span_labels_nn", "saliency_scores", "saliency_pos_labels"):
            assert np.array_equal(tg[k].cpu().numpy(), wtg[k]), (i, k)


def test_device_prefetcher_overlaps_upload_with_compute(dev):
    """DevicePrefetcher: batch n + 1 is collated and uploaded on a side stream while batch n trains; results are bit-identical to
    the synchronous path and the upload time is hidden behind the step (reported)."""
    import time
    from oracle import univtg_oracle as O
    from univtg_amd import pipeline
    from univtg_amd.trainer import TrainStep
    cfg = O.make_cfg(input_dropout=0.0, dropout=0.0, droppath=0.0, enc_layers=2)
    params = O.init_params(cfg, seed=71)
    g = torch.Generator().manual_seed(5)

    def samples(n, seed):
        inputs, tg = O.make_batch(cfg, n, 75, 32, seed=seed, ragged=True)
        out = []
        for b in range(n):
            lv, lt = int(inputs["src_vid_mask"][b].sum()), int(inputs["src_txt_mask"][b].sum())
            out.append(dict(meta=dict(qid=b), model_inputs=dict(
                query_feat=inputs["src_txt"][b, :lt], video_feat=inputs["src_vid"][b, :lv], timestamp=tg["timestamp"][b, :lv],
                timestamp_window=tg["timestamp_window"][b, :lv], span_labels_nn=tg["span_labels_nn"][b, :lv],
                saliency_scores=tg["saliency_scores"][b, :lv], saliency_pos_labels=tg["saliency_pos_labels"][b].tolist())))
        return out
    batches = [samples(64, 300 + i) for i in range(6)]
    runs = {}
    for mode in ("sync", "prefetch"):
        model, crit = build(cfg, params, dev, "bf16", proj_precise="auto")
        model.eval()
        step = TrainStep(model, crit, lr=1e-4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == "sync":
            it = (pipeline.collate_upload_mr(b, dev) for b in batches)
            pf = None
        else:
            pf = pipeline.DevicePrefetcher(batches, dev, depth=2, timing=True)
            it = iter(pf)
        losses = []
        for _, mi, tg in it:
            losses.append(step.step(mi, tg, optimize=False))      # fixed weights: the losses depend on the uploaded data only
        torch.cuda.synchronize()
        runs[mode] = (time.perf_counter() - t0, torch.stack(losses).cpu(), pf)
    assert torch.allclose(runs["sync"][1], runs["prefetch"][1], rtol=1e-5, atol=1e-7)   # same bytes arrived (loss sums use fp32 atomics)
    assert len(set(float(x) for x in runs["sync"][1][:, 0])) == 6    # (and the six batches are different)
    st = runs["prefetch"][2].stats
    print(f"\n[prefetcher] 6 batches of 64: synchronous {runs['sync'][0] * 1e3:.1f} ms, prefetched {runs['prefetch'][0] * 1e3:.1f} ms; "
          f"upload device time {st['upload_ms']:.1f} ms on the side stream, host collate {st['host_collate_s'] * 1e3:.1f} ms")


# ------------------------------------------------------------------------------------------------------------------------------
# config 4 (Ego4D-NLQ long video: L_v = 1200, S = 1232) -- the tiled attention kernels at their production shape
# ------------------------------------------------------------------------------------------------------------------------------
def _attn_ref(qkv, kvalid, B, S, H, hd):
    d = H * hd
    q, k, v = [t.view(B, S, H, hd).transpose(1, 2) for t in qkv.double().view(B, S, 3 * d).split(d, dim=-1)]
    sc = q @ k.transpose(-1, -2)
    sc = sc.masked_fill(~kvalid.bool()[:, None, None, :], float("-inf"))
    p = torch.softmax(sc, -1)
    return (p @ v).transpose(1, 2).reshape(B * S, d), torch.logsumexp(sc, -1)


def test_config4_attention_kernels_at_production_shape(dev):
    """attn_fwd (multi key-tile), attn_bwd_dkdv, attn_bwd_dq at (B=2, S=1232, H=8, hd=128) vs fp64 torch."""
    from univtg_amd import ops
    B, S, H, hd = 2, 1232, 8, 128
    d = H * hd
    g = torch.Generator().manual_seed(4)
    qkv = torch.randn(B * S, 3 * d, generator=g)
    qkv[:, :d] *= hd ** -0.5
    kv = torch.ones(B, S, dtype=torch.uint8)
    kv[0, 900:1200] = 0                                   # a short video in a long batch: padded clips, then the text tokens
    kv[1, 1215:] = 0                                      # padded text
    kv = kv.to(dev)
    xb = qkv.to(torch.bfloat16).to(dev)
    o, lse = ops.attention_fwd(xb, kv, B, S, H, hd, False)
    xr = xb.double().cpu().requires_grad_(True)
    ref_o, ref_l = _attn_ref(xr, kv.cpu(), B, S, H, hd)
    assert float((o.float().cpu() - ref_o.detach()).abs().max()) < 1.5e-2
    assert float((lse.cpu() - ref_l.detach()).abs().max()) < 2e-3
    o32, lse32 = ops.attention_fwd(qkv.to(dev), kv, B, S, H, hd, True)
    r32, l32 = _attn_ref(qkv, kv.cpu(), B, S, H, hd)
    assert float((o32.cpu() - r32).abs().max()) < 3e-5 and float((lse32.cpu() - l32).abs().max()) < 1e-4
    do = torch.randn(B * S, d, generator=g).to(torch.bfloat16).to(dev)
    dqkv = ops.attention_bwd(xb, kv, o, lse, do, 1.0, B, S, H, hd)
    ref_o.backward(do.double().cpu())
    for name, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d))):
        a, r = dqkv[:, sl].float().cpu().double(), xr.grad[:, sl]
        e = float((a - r).abs().max() / (r.abs().max() + 1e-30))
        assert e < 2.5e-2, (name, e)


def test_config4_model_level_vs_oracle(dev):
    """(B=2, L_v=1200, L_t=32, d=1024, H=8, E=1) against the oracle: fp32x3 forward, bf16 losses + gradients, packed stream."""
    from oracle import univtg_oracle as O
    from univtg_amd.trainer import TrainStep
    _threads()
    cfg = O.make_cfg(input_dropout=0.0, droppath=0.0, dropout=0.0, enc_layers=1, max_v_l=1200)
    params = O.init_params(cfg, seed=401)
    inputs, tg = O.make_batch(cfg, 2, 1200, 32, seed=402, ragged=True)
    p2 = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = O.forward(p2, cfg, **inputs)
    rl = O.criterion(ref, tg, cfg)
    O.total_loss(rl, cfg).backward()
    model, _ = build(cfg, params, dev, "fp32x3")
    model.eval()
    with torch.no_grad():
        out = model(**to_dev(inputs, dev))
    valid = inputs["src_vid_mask"].bool()
    assert float((out["saliency_scores"].cpu() - ref["saliency_scores"].detach())[valid].abs().max()) < 1e-4
    for k in ("pred_logits", "pred_spans"):
        assert float((out[k].cpu() - ref[k].detach()).abs().max()) < 3e-4, k
    lens = (inputs["src_vid_mask"].sum(1).int().tolist(), inputs["src_txt_mask"].sum(1).int().tolist())
    for packed in (False, True):
        model, crit = build(cfg, params, dev, "bf16")
        model.eval()
        step = TrainStep(model, crit, packed=packed)
        batch = to_dev(inputs, dev)
        if packed:
            batch["_lens_host"] = lens
        losses = step.step(batch, to_dev(tg, dev), optimize=False).cpu()
        for i, k in enumerate(("loss_b", "loss_g", "loss_f", "loss_s_inter", "loss_s_intra")):
            assert abs(float(losses[i]) - float(rl[k])) < 2e-2 * max(1.0, abs(float(rl[k]))), (packed, k)
        offs = model._offsets(model._dims(2, 1200, 32, cfg.v_feat_dim, cfg.t_feat_dim, False))
        names = {id(p): k for k, p in model.named_parameters()}
        flat = {names[id(p)]: step.grads[offs[i]: offs[i] + p.numel()] for i, p in enumerate(model._ordered_params())}
        rep = _grad_report(None, p2, lambda k: flat[k])
        bad = {k: v for k, v in rep.items() if v[0] < 0.995 or abs(v[1] - 1) > 0.02}
        assert not bad, (packed, bad)


# ------------------------------------------------------------------------------------------------------------------------------
# config 5 (multi-dataset co-training, mixed L in {75, 200, 600}): ragged-batch packed attention
# ------------------------------------------------------------------------------------------------------------------------------
def test_config5_mixed_length_batch_packed_vs_padded_vs_oracle(dev):
    from bench import mixed_length_lens
    from oracle import univtg_oracle as O
    from univtg_amd.trainer import TrainStep
    _threads()
    B, Lt = 12, 32
    lens_v = mixed_length_lens(B, seed=5)
    assert set(lens_v) <= {75, 200, 600} and max(lens_v) == 600
    cfg = O.make_cfg(input_dropout=0.0, droppath=0.0, dropout=0.0, enc_layers=2, max_v_l=600)
    params = O.init_params(cfg, seed=501)
    # build the mixed batch sample by sample (every sample full length for its dataset; the batch pads to the longest)
    Lv = max(lens_v)
    parts = [O.make_batch(cfg, 1, lv, Lt, seed=600 + i, ragged=False) for i, lv in enumerate(lens_v)]
    pad = lambda t, L: torch.cat([t, t.new_zeros((1, L - t.shape[1]) + tuple(t.shape[2:]))], 1)
    inputs = {k: torch.cat([pad(p[0][k], Lv if "vid" in k else Lt) for p in parts]) for k in parts[0][0]}
    tg = {k: torch.cat([pad(p[1][k], Lv) for p in parts]) for k in ("timestamp", "timestamp_mask", "timestamp_window", "span_labels_nn", "saliency_scores")}
    tg["saliency_pos_labels"] = torch.cat([p[1]["saliency_pos_labels"] for p in parts])
    p2 = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = O.forward(p2, cfg, **inputs)
    rl = O.criterion(ref, tg, cfg)
    O.total_loss(rl, cfg).backward()
    lens = (lens_v, [Lt] * B)
    res = {}
    for packed in (False, True):
        model, crit = build(cfg, params, dev, "bf16")
        model.eval()
        step = TrainStep(model, crit, packed=packed)
        batch = to_dev(inputs, dev)
        if packed:
            batch["_lens_host"] = lens
        losses = step.step(batch, to_dev(tg, dev), optimize=False).cpu()
        res[packed] = (losses, step.grads.clone().double(), step.pred_logits.clone())
        for i, k in enumerate(("loss_b", "loss_g", "loss_f", "loss_s_inter", "loss_s_intra")):
            assert abs(float(losses[i]) - float(rl[k])) < 2e-2 * max(1.0, abs(float(rl[k]))), (packed, k)
        e = float((step.pred_logits.cpu() - ref["pred_logits"].detach()).abs().max())
        assert e < 3e-2, (packed, e)
        offs = model._offsets(model._dims(B, Lv, Lt, cfg.v_feat_dim, cfg.t_feat_dim, False))
        names = {id(p): k for k, p in model.named_parameters()}
        flat = {names[id(p)]: step.grads[offs[i]: offs[i] + p.numel()] for i, p in enumerate(model._ordered_params())}
        rep = _grad_report(None, p2, lambda k: flat[k])
        # (norm floor 2.5 % as in the bench-path replays: at B = 12 the feature LayerNorm's weight gradient -- the sum of few bf16-rounded rows --
        #  sits at 1.9-2.1 % depending on the rounding realisation; every other gradient is within 1 %)
        bad = {k: v for k, v in rep.items() if v[0] < 0.995 or abs(v[1] - 1) > 0.025}
        assert not bad, (packed, bad)
        worst = max(rep.items(), key=lambda kv: abs(kv[1][1] - 1))
        print(f"\n[config5 packed={packed}] worst gradient norm ratio {worst[1][1]:.4f} ({worst[0]})")
    g0, g1 = res[False][1], res[True][1]
    assert float((g0 @ g1) / (g0.norm() * g1.norm())) > 0.9995
    rows = sum(lv + (lv < Lv) + Lt for lv in lens_v)
    print(f"\n[config5] mixed lengths {sorted(lens_v)}: packed rows {rows} of {B * (Lv + Lt)} padded ({rows / (B * (Lv + Lt)):.2f})")


# ------------------------------------------------------------------------------------------------------------------------------
# the reference's own DDP wrapper around the drop-in model (main/train_vlp_ddp.py:272-275)
# ------------------------------------------------------------------------------------------------------------------------------
def _ddp_worker(rank, world, port, out_path):
    import torch.distributed as dist
    from oracle import univtg_oracle as O
    from univtg_amd.trainer import TrainStep
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    cfg = O.make_cfg(hidden_dim=256, nheads=4, dim_feedforward=256, enc_layers=2, v_feat_dim=514, t_feat_dim=512,
                     input_dropout=0.0, dropout=0.0, droppath=0.0)
    params = O.init_params(cfg, seed=81)
    inputs, tg = O.make_batch(cfg, 8, 30, 10, seed=82, ragged=True)
    sl = slice(rank * 4, rank * 4 + 4)                                        # DistributedSampler-style shard
    ind = {k: v[sl].to(dev) for k, v in inputs.items()}
    tgd = {k: v[sl].to(dev) for k, v in tg.items() if torch.is_tensor(v)}
    model, crit = build(cfg, params, dev, "bf16")
    model.eval()
    # round 6: the optimizer is built BEFORE the DDP wrap, as main/config.py:349-350 / main/train_vlp_ddp.py:272 do -- with FusedAdamWClip swapped
    # in for torch.optim.AdamW (it re-homes the parameters into one flat buffer: DDP then wraps the re-homed parameters)
    from univtg_amd.optim import FusedAdamWClip
    group = [{"params": [p for n, p in model.named_parameters() if p.requires_grad]}]
    opt = FusedAdamWClip(group, lr=1e-3, weight_decay=1e-4, max_grad_norm=0.1, model=model)
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], output_device=0, find_unused_parameters=True)   # :272-275
    out = ddp(**ind)
    ld = crit(out, tgd)
    opt.zero_grad()
    sum(ld[k] * crit.weight_dict[k] for k in ld).backward()
    torch.cuda.synchronize()
    g_ddp = torch.cat([p.grad.flatten() for p in model._ordered_params()]).double().cpu()
    unused = [k for k, p in model.named_parameters() if p.grad is None]
    # ... and steps on DDP's averaged gradients (bucket copies, not the backward's flat buffer: the gather path): against clip_grad_norm_ +
    # torch.optim.AdamW on shadow parameters fed the same gradients, and bit-identical across the two ranks
    shadow = [p.detach().clone().requires_grad_(True) for p in group[0]["params"]]
    for p, sp in zip(group[0]["params"], shadow):
        sp.grad = None if p.grad is None else p.grad.detach().clone()
    sopt = torch.optim.AdamW([{"params": shadow}], lr=1e-3, weight_decay=1e-4)
    torch.nn.utils.clip_grad_norm_(shadow, 0.1)
    sopt.step()
    opt.step()
    torch.cuda.synchronize()
    fused_vs_shadow = max(float((p - sp).abs().max()) for p, sp in zip(group[0]["params"], shadow))
    moved = max(float((p.detach().cpu() - params[n]).abs().max()) for n, p in model.named_parameters() if p.grad is not None)
    flat = opt.flat.detach().cpu()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    ranks_equal = all(torch.equal(gathered[0], g) for g in gathered)
    # native exchange on the same shards
    model2, crit2 = build(cfg, params, dev, "bf16")
    model2.eval()
    step = TrainStep(model2, crit2, overlap_comm=False)
    step.step(ind, tgd, optimize=False)
    torch.cuda.synchronize()
    offs = model2._offsets(model2._dims(4, 30, 10, 514, 512, False))
    g_nat = torch.cat([step.grads[offs[i]: offs[i] + p.numel()] for i, p in enumerate(model2._ordered_params())]).double().cpu() / world
    if rank == 0:
        torch.save(dict(ddp=g_ddp, native=g_nat, unused=unused, world=step.world, fused_vs_shadow=fused_vs_shadow, moved=moved, ranks_equal=ranks_equal,
                        in_place_steps=opt.in_place_steps), out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_drop_in_model_under_reference_ddp_wrapper(dev, tmp_path):
    """The autograd Model wrapped in DistributedDataParallel(find_unused_parameters=True) exactly as main/train_vlp_ddp.py:272-275
    does (2 ranks, gloo, both on cuda:0): DDP's averaged gradients == TrainStep's all-reduced gradients / world."""
    import torch.multiprocessing as mp
    out_path = str(tmp_path / "ddp.pt")
    port = 29700 + os.getpid() % 200
    mp.spawn(_ddp_worker, args=(2, port, out_path), nprocs=2, join=True)
    r = torch.load(out_path)
    assert r["world"] == 2
    assert all(k.startswith("txt_position_embed") for k in r["unused"]) and len(r["unused"]) == 3
    a, b = r["ddp"], r["native"]
    assert float((a - b).abs().max()) <= 2e-3 * float(b.abs().max())          # same kernels; fp32 atomic order only
    assert float((a @ b) / (a.norm() * b.norm())) > 0.99999
    # FusedAdamWClip under the DDP wrapper: the same update as clip_grad_norm_ + torch.optim.AdamW on DDP's gradients, replicas bit-identical
    assert r["moved"] > 5e-4 and r["fused_vs_shadow"] < 2e-6 and r["ranks_equal"], {k: r[k] for k in ("moved", "fused_vs_shadow", "ranks_equal", "in_place_steps")}


# ------------------------------------------------------------------------------------------------------------------------------
# the native data-parallel step itself (main/train_vlp_ddp.py:112,215,272-275): 2 ranks, gloo, both on cuda:0
# ------------------------------------------------------------------------------------------------------------------------------
def _dp_worker(rank, world, port, out_path):
    import torch.distributed as dist
    from oracle import univtg_oracle as O
    from univtg_amd.trainer import TrainStep
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    res = {}
    # ---- (i) three real train-mode steps (dropout 0.5 / DropPath 0.1, clip, AdamW, overlapped bucketed exchange): the ranks start from
    # DIFFERENT weights (the constructor must broadcast rank 0's), see different data and draw different masks; their flat parameter
    # buffers must stay bit-identical
    cfg = O.make_cfg(hidden_dim=256, nheads=4, dim_feedforward=256, enc_layers=2, v_feat_dim=514, t_feat_dim=512,
                     input_dropout=0.5, dropout=0.0, droppath=0.1)
    params = O.init_params(cfg, seed=91 + rank)
    model, crit = build(cfg, params, dev, "auto", proj_precise="auto")
    model.train()
    model.set_seed(1000 + rank)
    step = TrainStep(model, crit, lr=1e-3, grad_clip=0.1, overlap_comm=True, time_comm=True)
    p_start = step.flat.clone()
    for it in range(3):
        inputs, tg = O.make_batch(cfg, 6, 30, 10, seed=500 + 10 * it + rank, ragged=True)
        batch = to_dev(inputs, dev)
        batch["_lens_host"] = (inputs["src_vid_mask"].sum(1).int().tolist(), inputs["src_txt_mask"].sum(1).int().tolist())
        step.step(batch, to_dev(tg, dev))
    torch.cuda.synchronize()
    res["world"], res["overlap"], res["exposed"] = step.world, step.overlap, step.exposed_comm_ms()
    res["flat"], res["start"], res["m"] = step.flat.cpu(), p_start.cpu(), step.m.cpu()
    res["state_w"] = model.state_dict()["transformer.encoder.layers.0.linear1.weight"].cpu()      # (the parameters ARE views of the flat buffer)
    # ---- (ii) rank-separable loss ("labels": a mean over valid clips, unragged shards): the exchanged gradient / world == the gradient
    # of the concatenated batch on one device (autograd path, no exchange), in both wire dtypes and both exchange schedules
    cfg2 = O.make_cfg(hidden_dim=256, nheads=4, dim_feedforward=256, enc_layers=2, v_feat_dim=514, t_feat_dim=512,
                      input_dropout=0.0, dropout=0.0, droppath=0.0)
    params2 = O.init_params(cfg2, seed=95)
    inputs, tg = O.make_batch(cfg2, 8, 30, 10, seed=96, ragged=False)
    sl = slice(rank * 4, rank * 4 + 4)                                        # DistributedSampler-style shard (train_vlp_ddp.py:112)
    ind = {k: v[sl].to(dev) for k, v in inputs.items()}
    tgd = {k: v[sl].to(dev) for k, v in tg.items() if torch.is_tensor(v)}
    for tag, kw in (("fp32", dict(overlap_comm=True)), ("fp32_flat", dict(overlap_comm=False)), ("bf16", dict(overlap_comm=True, grad_comm_dtype="bf16"))):
        m2, c2 = build(cfg2, params2, dev, "bf16")
        m2.eval()
        c2.losses = ["labels"]
        st2 = TrainStep(m2, c2, **kw)
        st2.step(ind, tgd, optimize=False)
        torch.cuda.synchronize()
        offs = m2._offsets(m2._dims(4, 30, 10, 514, 512, False))
        res["g_" + tag] = torch.cat([st2.grads[offs[i]: offs[i] + p.numel()] for i, p in enumerate(m2._ordered_params())]).double().cpu() / world
    if rank == 0:
        m3, c3 = build(cfg2, params2, dev, "bf16")
        m3.eval()
        c3.losses = ["labels"]
        out = m3(**to_dev(inputs, dev))
        ld = c3(out, to_dev({k: v for k, v in tg.items() if torch.is_tensor(v)}, dev))
        (ld["loss_f"] * c3.weight_dict["loss_f"]).backward()
        torch.cuda.synchronize()
        res["g_full"] = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten() for p in m3._ordered_params()]).double().cpu()
    torch.save(res, out_path + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_native_trainstep_is_a_data_parallel_step(dev, tmp_path):
    """VERDICT r2 missing #1: a 2-rank TrainStep IS a data-parallel step -- (i) flat parameters (and AdamW moments) bit-identical across
    ranks after 3 train-mode steps from different initial weights / data / masks; (ii) for the rank-separable `labels` loss the exchanged
    gradient equals the single-device gradient of the concatenated batch, with fp32 buckets (overlapped and flat) and with bf16 buckets
    (within bf16 rounding of the wire format)."""
    import torch.multiprocessing as mp
    out_path = str(tmp_path / "dp.pt")
    port = 29900 + os.getpid() % 90
    mp.spawn(_dp_worker, args=(2, port, out_path), nprocs=2, join=True)
    r0, r1 = torch.load(out_path + ".0"), torch.load(out_path + ".1")
    assert r0["world"] == 2 and r0["overlap"] and len(r0["exposed"]) == 3
    assert torch.equal(r0["start"], r1["start"])                               # rank 1's own initial weights were replaced by rank 0's
    assert torch.equal(r0["flat"], r1["flat"]) and torch.equal(r0["m"], r1["m"])
    assert not torch.equal(r0["flat"], r0["start"])                            # (and the steps did move them)
    n = r0["state_w"].numel()
    assert torch.equal(r0["state_w"], r1["state_w"]) and n > 0
    full = r0["g_full"]
    for tag, tol_max, tol_cos in (("fp32", 2e-3, 0.99999), ("fp32_flat", 2e-3, 0.99999), ("bf16", 1.2e-2, 0.9999)):
        a0, a1 = r0["g_" + tag], r1["g_" + tag]
        assert torch.equal(a0, a1), tag                                        # every rank holds the same reduced gradient
        err = float((a0 - full).abs().max() / full.abs().max())
        cos = float((a0 @ full) / (a0.norm() * full.norm()))
        print(f"\n[2-rank TrainStep, {tag} buckets] reduced gradient vs concatenated-batch gradient: max err {err:.2e} of max |g|, cosine {cos:.7f}")
        assert err <= tol_max and cos > tol_cos, (tag, err, cos)
    print(f"[2-rank TrainStep] exposed communication per step (gloo through the host, both ranks on one GPU): {[round(x, 2) for x in r0['exposed']]} ms")
