"""Pins the CPU oracle (oracle/) to vectors produced by the REAL reference (tests/golden, made by
oracle/make_golden.py) -- CPU only, no GPU, no reference tree needed."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import postproc_oracle as P
from oracle import univtg_oracle as O

CASES = ["tiny_eval_ragged", "tiny_eval_full", "tiny_train_droppath", "config1_real_feats",
         # round 4: dset_type 'hl' loss subset, the two loss_saliency early-outs, n_input_proj 1 / 3, use_txt_pos
         "tiny_hl", "tiny_zero_saliency", "tiny_no_pos_labels", "tiny_nproj1", "tiny_nproj3", "tiny_txt_pos",
         # round 6: the TAL pre-training branch -- src_cls through the text projection + pool, the 'saliency_cls' loss (model/univtg.py:109-117,151-153,284-326)
         "tiny_tal"]


def load_case(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    cfg = O.make_cfg(**meta["cfg"])
    grab = lambda pre: {k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)}
    params, inputs, tg, out, grads = grab("param/"), grab("in/"), grab("tg/"), grab("out/"), grab("grad/")
    losses = {k[5:]: float(z[k]) for k in z.files if k.startswith("loss/")}
    rng = {"dp_scale": torch.from_numpy(z["rng/dp_scale"])} if "rng/dp_scale" in z.files else None
    if meta.get("drop_pos_labels"):
        tg.pop("saliency_pos_labels")
    return meta, cfg, params, inputs, tg, out, grads, losses, rng


@pytest.mark.parametrize("name", CASES)
def test_forward_losses_grads_match_reference(golden_dir, name):
    meta, cfg, params, inputs, tg, out_ref, grads_ref, losses_ref, rng = load_case(golden_dir, name)
    params = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    out = O.forward(params, cfg, rng=rng, **inputs)
    for k in ("pred_logits", "pred_spans", "vid_mem_proj", "txt_mem_proj", "saliency_scores") + (("cls_mem_proj",) if "cls_mem_proj" in out_ref else ()):
        torch.testing.assert_close(out[k], out_ref[k], rtol=2e-5, atol=2e-6, msg=lambda m: f"{k}: {m}")
    if meta.get("tal"):      # eval-mode criterion without cls_idx: the inter-video term only (model/univtg.py:312-313)
        z = np.load(os.path.join(golden_dir, name + ".npz"))
        with torch.no_grad():
            le = O.criterion(O.forward({k: v.detach() for k, v in params.items()}, cfg, **inputs), {k: v for k, v in tg.items() if k != "cls_idx"}, cfg)
        assert sorted(le) == meta["eval_loss_keys"] and abs(float(le["loss_s_inter"]) - float(z["evalloss/loss_s_inter"])) <= 2e-5
    losses = O.criterion(out, tg, cfg)
    assert set(losses) == set(losses_ref) - {"total"}, (set(losses), set(losses_ref))        # 'hl': no loss_b / loss_g
    for k in meta.get("loss_is_float", []):                 # the early-outs hand out python floats 0.0 (model/univtg.py:237-241)
        assert not torch.is_tensor(losses[k]) and losses[k] == 0.0 and losses_ref[k] == 0.0
    for k, v in losses.items():
        assert abs(float(v.detach() if torch.is_tensor(v) else v) - losses_ref[k]) <= 2e-5 * max(1.0, abs(losses_ref[k])), (k, float(v), losses_ref[k])
    total = O.total_loss(losses, cfg)
    assert abs(float(total) - losses_ref["total"]) <= 2e-5 * abs(losses_ref["total"])
    total.backward()
    assert set(meta["no_grad_params"]) == {k for k, p in params.items() if p.grad is None}
    for k, g in grads_ref.items():
        scale = float(g.abs().max()) + 1e-12
        err = float((params[k].grad - g).abs().max())
        assert err <= 2e-4 * scale + 1e-7, (k, err, scale)


def test_checkpoint_layout_matches_reference(golden_dir):
    """param_shapes == the reference's state_dict (names, shapes, order); the fixture weights went
    through load_state_dict(strict=True) when it was made."""
    z = np.load(os.path.join(golden_dir, "tiny_eval_full.npz"))
    meta = json.loads(str(z["meta"]))
    shapes = O.param_shapes(O.make_cfg(**meta["cfg"]))
    stored = {k[6:]: tuple(z[k].shape) for k in z.files if k.startswith("param/")}
    assert stored == {k: tuple(v) for k, v in shapes.items()}


def test_log_mask_denormal():
    v = O.log_mask(torch.tensor([0.0, 1.0]), torch.float32)
    assert abs(float(v[0]) + 103.2789) < 1e-3 and float(v[1]) == 0.0


def test_span_utils_known_answers(golden_dir):
    z = np.load(os.path.join(golden_dir, "span_utils.npz"))
    a, b = torch.from_numpy(z["a"]), torch.from_numpy(z["b"])
    np.testing.assert_allclose(O.giou_matrix(a, b).numpy(), z["giou"], rtol=1e-6, atol=1e-7)
    d1 = torch.tensor([[0, 0.2], [0.5, 1.0]])
    d2 = torch.tensor([[0, 0.3], [0.0, 1.0]])
    np.testing.assert_allclose(O.giou_matrix(d1, d2).numpy(), z["doc_giou"], rtol=1e-6)
    # doctest values printed in utils/span_utils.py:106-110
    np.testing.assert_allclose(z["doc_giou"], [[0.6667, 0.2], [-0.2, 0.5]], atol=1e-4)
    np.testing.assert_allclose(np.diag(O.giou_matrix(a[:5], b).numpy()), O.paired_giou(a[:5], b).numpy(), rtol=1e-6)


def test_postprocessing_matches_reference(golden_dir):
    for name in CASES:
        meta, cfg, params, inputs, tg, _, *_ = load_case(golden_dir, name)
        z = np.load(os.path.join(golden_dir, name + ".npz"))
        out_ref = {k[8:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("evalout/")}
        durations = [float(inputs["src_vid_mask"][b].sum()) * 2.0 for b in range(inputs["src_vid"].shape[0])]
        pre = P.decode_windows(out_ref["pred_logits"].numpy(), out_ref["pred_spans"].numpy(),
                               tg["timestamp"].numpy(), tg["timestamp_mask"].numpy(), durations)
        assert pre == meta["post/raw"]["pre"]
        nms = [P.temporal_nms(p[:1000], 0.7, 10) for p in pre]
        assert nms == meta["post/raw"]["nms"]
        rounded = [P.round_multiple(p, 2.0) for p in pre]
        assert rounded == meta["post/rounded"]["pre"]
        sal = P.saliency_for_eval(out_ref["saliency_scores"].numpy(), out_ref["pred_logits"].numpy(),
                                  inputs["src_vid_mask"].numpy())
        for a, b in zip(sal, meta["post/raw"]["sal"]):
            np.testing.assert_allclose(a, b, rtol=0, atol=0)


def test_temporal_nms_known_answers(golden_dir):
    d = json.load(open(os.path.join(golden_dir, "nms.json")))
    for c in d["cases"]:
        assert P.temporal_nms([list(r) for r in c["inp"]], c["thd"], c["max_after"]) == c["out"]
    for rin, rout in zip(d["round_in"], d["round_out"]):
        assert P.round_multiple(rin, 2.0) == rout
    assert P.temporal_nms([[0.0, 1.0, 0.5]], 0.7) == [[0.0, 1.0, 0.5]]


def test_matcher_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "matcher.npz"))
    sizes = z["sizes"].tolist()
    tg, off = [], 0
    for n in sizes:
        tg.append(z["tgt"][off:off + n])
        off += n
    got = P.hungarian_match(z["logits"], z["spans"], tg)
    for b, (i, j) in enumerate(got):
        assert i.tolist() == z[f"i{b}"].tolist() and j.tolist() == z[f"j{b}"].tolist()
    got = P.hungarian_match(z["logits1"], z["spans"], tg)
    for b, (i, j) in enumerate(got):
        assert i.tolist() == z[f"u_i{b}"].tolist() and j.tolist() == z[f"u_j{b}"].tolist()


def _detr_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, "detr_criterion.npz"))
    sizes = z["sizes"].tolist()
    off = np.concatenate([[0], np.cumsum(sizes)])
    tg = [z["tgt"][off[b]:off[b + 1]] for b in range(len(sizes))]
    idx = [(z[f"i{b}"], z[f"j{b}"]) for b in range(len(sizes))]
    return z, tg, idx


def test_detr_criterion_matches_reference(golden_dir):
    """Moment-DETR SetCriterion (model/moment_detr.py:166-365): the six losses and the autograd gradients of the real reference."""
    z, tg, idx = _detr_fixture(golden_dir)
    got = P.hungarian_match(z["logits"], z["spans"], tg, w_class=4.0, w_span=10.0, w_giou=1.0)
    for (a, b), (c, d) in zip(got, idx):
        assert a.tolist() == c.tolist() and b.tolist() == d.tolist()
    eos, temp, margin = (float(x) for x in z["hyper"])
    for dt, tol in ((np.float64, 2e-7), (np.float32, 2e-5)):
        L, G = P.detr_criterion(z["logits"], z["spans"], tg, idx, z["sal"], z["pos"], z["neg"], z["pq"], z["pt"], eos, temp, margin,
                                z["weights"], dtype=dt)
        assert np.allclose(L, z["losses"], rtol=tol * 10, atol=tol), (dt, L, z["losses"])
        for k in ("logits", "spans", "sal", "pq", "pt"):
            r = z["d_" + k]
            assert np.abs(G[k] - r).max() <= tol * 10 * max(1.0, np.abs(r).max()), (dt, k)


def test_feature_file_readers_match_reference(golden_dir, tmp_path):
    """the .npz readers of main/dataset.py:325-358,370-390 (golden = the tensors the real DatasetVLP methods returned
    on the same arrays): l2 normalisation per row, truncation to the shortest feature type, fp64 -> fp32, zeros((10, D)) placeholder."""
    from univtg_amd import pipeline
    z = np.load(os.path.join(golden_dir, "features.npz"))
    np.savez(tmp_path / "sf.npz", features=z["slowfast"])
    np.savez(tmp_path / "clip.npz", features=z["clip"])
    np.savez(tmp_path / "q.npz", last_hidden_state=z["q_last"], pooler_output=z["q_pool"])
    for norm in (True, False):
        v = pipeline.read_video_features([tmp_path / "sf.npz", tmp_path / "clip.npz"], normalize=norm)
        assert v.dtype == torch.float32 and np.array_equal(v.numpy(), z[f"video_{int(norm)}"])
        for ft in ("last_hidden_state", "pooler_output"):
            q = pipeline.read_query_features(tmp_path / "q.npz", ft, normalize=norm, feat_dim=16)
            assert np.array_equal(q.numpy(), z[f"query_{ft}_{int(norm)}"]), (ft, norm)
    q = pipeline.read_query_features(tmp_path / "missing.npz", "last_hidden_state", normalize=True, feat_dim=16)
    assert np.array_equal(q.numpy(), z["query_missing"])


def test_feature_cache_readers_match_reference(golden_dir):
    """SURVEY 8f row 2, hdf5 cache path (main/dataset.py:113-131,335-340,375-376): the store is any mapping whose items slice to arrays
    (an open h5py.File where h5py exists -- emulated here by a dict behind a str-keyed view), entries are used as stored."""
    from univtg_amd import pipeline
    z = np.load(os.path.join(golden_dir, "features_cache.npz"))

    class Store(dict):                      # what an h5py.File looks like to the loader: str keys, datasets that slice to ndarrays
        def __getitem__(self, k):
            assert isinstance(k, str)
            return dict.__getitem__(self, k)
    caches = [pipeline.load_feature_cache(Store({"7": z["slowfast"]}), [7]), pipeline.load_feature_cache(Store({"7": z["clip"]}), [7])]
    v = pipeline.read_video_features_cached(caches, 7)
    assert v.dtype == torch.float32 and np.array_equal(v.numpy(), z["video"])
    tc = pipeline.load_feature_cache(Store({"q3": z["q"]}), ["q3", "absent"], optional=True)
    assert set(tc) == {"q3"}
    assert np.array_equal(pipeline.read_query_features_cached(tc, "q3", feat_dim=16).numpy(), z["query"])
    assert np.array_equal(pipeline.read_query_features_cached(tc, "absent", feat_dim=16).numpy(), z["query_missing"])
    with pytest.raises(KeyError):
        pipeline.load_feature_cache(Store({}), [1])


def test_lsap_against_scipy_random():
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(0)
    for _ in range(200):
        n, m = rng.integers(1, 12), rng.integers(1, 12)
        c = rng.standard_normal((n, m))
        i, j = P.lsap(c)
        ri, rj = linear_sum_assignment(c)
        assert i.tolist() == ri.tolist() and j.tolist() == rj.tolist()


def test_dense_targets_match_dataset(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "dense_targets.json")))
    for c in cases:
        g = torch.Generator().manual_seed(0)
        lv = c["lv"]
        t = O.dense_targets(lv, lv, torch.tensor(c["wins"]), 2, g)
        np.testing.assert_allclose(t["timestamp"].numpy(), np.array(c["timestamp"], np.float32), rtol=1e-6)
        np.testing.assert_allclose(t["span_labels_nn"].numpy(), np.array(c["span_labels_nn"], np.float32), rtol=1e-6, atol=1e-7)
        assert t["timestamp_window"].int().tolist() == [int(x) for x in c["timestamp_window"]]
        assert t["saliency_scores"].tolist() == [float(x) for x in c["saliency_scores"]]
        np.testing.assert_allclose(t["span_labels"].numpy(), np.array(c["span_labels"], np.float32), rtol=1e-6)
        assert c["timestamp_window"][c["pos"][0]] == 1 and t["timestamp_window"][t["saliency_pos_labels"]] == 1


@pytest.mark.reference
def test_oracle_vs_live_reference_full_width():
    """Build container only: d=1024 production width, live reference vs oracle (no fixture)."""
    from oracle.make_golden import build_reference, import_reference
    ref = import_reference()
    cfg = O.make_cfg(input_dropout=0.0, droppath=0.0, dropout=0.0, enc_layers=2)
    params = O.init_params(cfg, seed=5)
    model, crit = build_reference(ref, cfg, params)
    model.eval()
    inputs, tg = O.make_batch(cfg, 3, 20, 9, seed=6, ragged=True)
    with torch.no_grad():
        a = model(**inputs)
        b = O.forward(params, cfg, **inputs)
    for k in ("pred_logits", "pred_spans", "saliency_scores", "vid_mem_proj", "txt_mem_proj"):
        torch.testing.assert_close(b[k], a[k], rtol=1e-4, atol=1e-5)
    la, lb = crit(a, tg), O.criterion(b, tg, cfg)
    for k in la:
        assert abs(float(la[k]) - float(lb[k])) < 1e-4 * max(1, abs(float(la[k])))


def _collate_case(golden_dir):
    z = np.load(os.path.join(golden_dir, "collate.npz"))
    n = int(z["n"])
    batch = []
    for i in range(n):
        mi = {k.split("/", 2)[2]: z[k] for k in z.files if k.startswith(f"sample/{i}/")}
        for k in ("saliency_pos_labels", "saliency_neg_labels"):
            mi[k] = [int(x) for x in mi[k]]
        batch.append(dict(meta=dict(qid=i), model_inputs=mi))
    return z, batch


def test_pipeline_oracle_matches_reference_collate(golden_dir):
    """oracle/pipeline_oracle.py vs the real start_end_collate_mr + prepare_batch_inputs_mr (golden collate.npz): bit-exact."""
    from oracle import pipeline_oracle as PO
    z, batch = _collate_case(golden_dir)
    model_inputs, targets = PO.collate_mr(batch)
    for k, v in model_inputs.items():
        assert np.array_equal(v, z["in/" + k]), k
    for k, v in targets.items():
        if k == "span_labels":
            for i, sp in enumerate(v):
                assert np.array_equal(sp, z[f"tg/span_labels/{i}"])
        else:
            assert np.array_equal(v, z["tg/" + k]) and v.dtype == z["tg/" + k].dtype, k


def test_nn_baseline_matches_oracle():
    """oracle/nn_baseline.py (bench.py's CPU baseline, composed of the torch.nn modules the reference composes) loads the reference's
    state_dict layout strictly and computes what the oracle computes: outputs, losses and every parameter gradient (eval mode)."""
    from oracle.nn_baseline import NNBaseline
    cfg = O.make_cfg(hidden_dim=64, nheads=2, dim_feedforward=96, enc_layers=2, v_feat_dim=34, t_feat_dim=24, max_q_l=16,
                     input_dropout=0.0, dropout=0.0, droppath=0.0)
    params = O.init_params(cfg, seed=11)
    inputs, tg = O.make_batch(cfg, 5, 14, 6, seed=12, ragged=True)
    m = NNBaseline(cfg)
    res = m.load_state_dict(params, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    m.eval()
    p2 = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = O.forward(p2, cfg, **inputs)
    out = m(**inputs)
    for k in ("pred_logits", "pred_spans", "saliency_scores", "vid_mem_proj", "txt_mem_proj"):
        torch.testing.assert_close(out[k], ref[k], rtol=1e-5, atol=1e-6)
    O.total_loss(O.criterion(ref, tg, cfg), cfg).backward()
    O.total_loss(O.criterion(out, tg, cfg), cfg).backward()
    named = dict(m.named_parameters())
    n = 0
    for k, p in p2.items():
        if p.grad is None:
            assert named[k].grad is None or float(named[k].grad.abs().max()) == 0.0, k
            continue
        torch.testing.assert_close(named[k].grad, p.grad, rtol=2e-4, atol=1e-6)
        n += 1
    assert n == 12 * cfg.enc_layers + 30
    # train mode: the dropouts are live (two calls differ) and DropPath scales whole samples
    cfg_t = O.make_cfg(**{**vars(cfg), "input_dropout": 0.5, "droppath": 0.3})
    mt = NNBaseline(cfg_t)
    mt.load_state_dict(params, strict=True)
    mt.train()
    torch.manual_seed(0)
    a = mt(**inputs)["pred_logits"]
    b = mt(**inputs)["pred_logits"]
    assert not torch.equal(a, b)
