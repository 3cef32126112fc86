#!/usr/bin/env python
"""Benchmark of the UniVTG hot path on MI355X (BASELINE.json: clips/sec fwd+bwd, L=75, d=1024).

    python bench.py --gpus 1 --steps 50 --warmup 10 [--config 2|3|4|5] [--variant A|B] [--proj precise|bf16] [--comm-cus k] [--grad-comm-dtype fp32|bf16]
    python bench.py --mode infer                                       # inference line (main/inference_mr.py path), see run_infer()
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one full training step on synthetic features resident in HBM: input projections -> 4-layer encoder -> conv heads +
saliency -> dense criterion -> backward -> (RCCL gradient all-reduce) -> global-norm clip + AdamW.  bf16 MFMA operands / fp32
accumulation, the reference's training dropouts (input 0.5 / attention 0 / DropPath 0.1, scripts/pretrain.sh:33-35).
Per-GPU batch is fixed (weak scaling).  Rank 0 prints ONE JSON line.

`value` counts clips the step EXECUTES.  Default (config 2): SURVEY 8d variant A -- all-ones masks, the shape "L=75, d=1024, batch=256"
literally names: every clip position is executed and counted, `roofline_encoder.frac` is the hardware fraction of the >= 40 % target with no
arithmetic left to do.  The timed step runs the forward input projections on fp32-class split operands (`--proj precise`, default): the
configuration whose saliency_scores meet north_star's 1e-4 tolerance.  `companions` (single GPU) times, in the same run, the ragged-batch
packed stream (variant B: valid clips + 3-clip conv halo + valid text; `executed_or_valid_clips_per_sec` counts VALID clips only) with its
padded / all-clip-rows executions, and the plain-bf16-projection step.  With N > 1 the line also carries `replicas` (bit-equality of the
flat parameter buffers across ranks after the timed loop), `exposed_comm_ms_per_step` and `comm_world_size`.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

MODEL = dict(D_v=2818, D_t=512, d=1024, F=1024, H=8, E=4)
# BASELINE.json configs that run on one GPU (config 3 = its per-GPU shard; config 1 is the CPU plumbing case)
CONFIGS = {
    2: dict(B=256, L_v=75, L_t=32, what="QVHighlights training shape (BASELINE config 2)", lens="len_v~U{38..75}, len_t~U{8..32} (SURVEY 8d variant B)"),
    3: dict(B=256, L_v=128, L_t=32, what="4M VLP pretraining shape, per-GPU shard of the global batch 2048 (BASELINE config 3)",
            lens="len_v~U{64..128}, len_t~U{8..32}"),
    4: dict(B=32, L_v=1200, L_t=32, what="Ego4D-NLQ long-video shape (BASELINE config 4), tiled attention S=1232", lens="len_v~U{600..1200}, len_t~U{8..32}"),
    5: dict(B=64, L_v=600, L_t=32, what="multi-dataset co-training batch (BASELINE config 5), mixed L_v in {75,200,600}",
            lens="len_v drawn from {75: 0.45, 200: 0.40, 600: 0.15} (QVHighlights / Charades-STA+ANet / TACoS-length videos; assumed mix, "
                 "the co-training lists are not shipped), len_t~U{8..32}"),
}
WORKLOAD = dict(MODEL, **{k: CONFIGS[2][k] for k in ("B", "L_v", "L_t")})       # (kept for the tools that import it)


def model_args(**over):
    from types import SimpleNamespace
    a = dict(device="cuda", hidden_dim=MODEL["d"], dropout=0.0, droppath=0.1, nheads=MODEL["H"],
             dim_feedforward=MODEL["F"], enc_layers=MODEL["E"], dec_layers=2, pre_norm=False, position_embedding="sine",
             max_q_l=75, input_dropout=0.5, t_feat_dim=MODEL["D_t"], v_feat_dim=MODEL["D_v"], span_loss_type="l1",
             use_txt_pos=False, n_input_proj=2, set_cost_span=10, set_cost_giou=1, set_cost_class=4, max_v_l=75,
             b_loss_coef=10, g_loss_coef=1, f_loss_coef=10, s_loss_intra_coef=0.1, s_loss_inter_coef=0.1,
             dset_type="vlp", train_path=["synthetic"], eos_coef=0.1, temperature=0.07, saliency_margin=0.2)
    a.update(over)
    return SimpleNamespace(**a)


def mixed_length_lens(B, seed=0, lengths=(75, 200, 600), probs=(0.45, 0.40, 0.15)):
    """Config 5: per-sample clip counts of a co-training batch (one dataset length per sample); the longest length is always present
    (the collate pads to the batch maximum, utils/tensor_utils.py:34-53)."""
    g = torch.Generator().manual_seed(seed)
    idx = torch.multinomial(torch.tensor(probs), B, replacement=True, generator=g).tolist()
    lens = [lengths[i] for i in idx]
    lens[0] = max(lengths)
    return lens


def synth_batch(B, Lv, Lt, Dv, Dt, seed, dev, lens_v=None, full=False):
    """Synthetic batch generated ON DEVICE (shape/statistics of SURVEY 8d: L2-normalised feature blocks, TEF columns, ragged valid
    lengths, one GT window per sample with dense targets as main/dataset.py:173-230)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    if lens_v is None:
        lens_v = torch.randint((Lv + 1) // 2, Lv + 1, (B,), generator=g, device=dev)
        lens_v[0] = Lv
    else:
        lens_v = torch.tensor(lens_v, device=dev)
    lens_t = torch.randint(8, Lt + 1, (B,), generator=g, device=dev)
    lens_t[0] = Lt
    if full:                                                                    # SURVEY 8d variant A: all-ones masks
        lens_v, lens_t = torch.full_like(lens_v, Lv), torch.full_like(lens_t, Lt)
    tv = torch.arange(Lv, device=dev)[None]
    vm = (tv < lens_v[:, None]).float()
    tm = (torch.arange(Lt, device=dev)[None] < lens_t[:, None]).float()
    feat = torch.randn(B, Lv, Dv - 2, generator=g, device=dev)
    if Dv - 2 == 2816:
        feat = torch.cat([torch.nn.functional.normalize(feat[..., :2304], dim=-1), torch.nn.functional.normalize(feat[..., 2304:], dim=-1)], -1)
    else:
        feat = torch.nn.functional.normalize(feat, dim=-1)
    st = tv.float() / lens_v[:, None]
    vid = torch.cat([feat, st[..., None], (st + 1.0 / lens_v[:, None])[..., None]], -1) * vm[..., None]
    txt = torch.nn.functional.normalize(torch.randn(B, Lt, Dt, generator=g, device=dev), dim=-1) * tm[..., None]
    clip_len = 2.0
    ts = ((tv.float() + clip_len / 2) / lens_v[:, None]) * vm                    # dataset.py:173
    w0 = torch.rand(B, generator=g, device=dev) * 0.7
    ww = 0.05 + 0.25 * torch.rand(B, generator=g, device=dev)
    win = torch.stack([w0, torch.clamp(w0 + ww, max=1.0)], -1)                  # normalised GT window
    inside = ((ts >= win[:, None, 0]) & (ts <= win[:, None, 1]) & (vm > 0)).float()
    empty = inside.sum(1) == 0
    if bool(empty.any()):                                                       # dataset.py:202-205
        idx = torch.clamp((win[:, 0] * lens_v).long(), max=Lv - 1).clamp(min=0)
        idx = torch.minimum(idx, lens_v - 1)
        inside[empty, idx[empty]] = 1.0
    span_nn = win[:, None, :] * inside[..., None]
    pos = torch.multinomial(inside + 1e-9 * vm, 1, generator=g)                 # dataset.py:230 (random fg clip)
    targets = dict(timestamp=torch.stack([ts, ts], -1).contiguous(), timestamp_mask=vm.contiguous(),
                   timestamp_window=inside.contiguous(), span_labels_nn=span_nn.contiguous(),
                   saliency_scores=inside.clone(), saliency_pos_labels=pos, _pos_idx=pos[:, 0].contiguous())
    inputs = dict(src_txt=txt.contiguous(), src_txt_mask=tm.contiguous(), src_vid=vid.contiguous(), src_vid_mask=vm.contiguous())
    # the valid lengths a collate knows on the host anyway (utils/tensor_utils.py:34-53 computes them to build the masks):
    # handing them over lets the engine run the packed (ragged) encoder stream without a device->host sync
    inputs["_lens_host"] = (lens_v.cpu().tolist(), lens_t.cpu().tolist())
    return inputs, targets


def reference_parity(batch, dev, ref_eval_npz, run_job, tmp):
    """The reference's own eval-mode outputs (child process, oracle/ref_runner.py) beside the HIP path on THIS box at the headline batch
    (VERDICT r5 item 1c): the default drop-in model (precision 'auto' -> fp32x3 under no_grad) with the same oracle-seeded weights through
    load_state_dict(strict=True); then the REFERENCE's round_multiple + temporal_nms on the HIP outputs against uvtg_postprocess_mr, and the
    end-to-end ranking + post-NMS keep-set (reference forward + reference tail vs HIP forward + device tail)."""
    import numpy as np
    from oracle import univtg_oracle as O
    from univtg_amd import ops
    from univtg_amd.model import build_model
    inputs, tg = batch
    B, Lv = inputs["src_vid"].shape[:2]
    cfg = O.make_cfg(input_dropout=0.5, dropout=0.0, droppath=0.1)
    params = O.init_params(cfg, seed=0)
    model, _ = build_model(model_args(max_v_l=Lv, precision="auto", packed=False))
    model.load_state_dict(params, strict=True)
    model.to(dev).eval()
    with torch.no_grad():
        out = model(**{k: v for k, v in inputs.items() if not k.startswith("_")})
    ref = {k: torch.from_numpy(v) for k, v in np.load(ref_eval_npz).items()}
    valid = inputs["src_vid_mask"].bool().cpu()
    rep = dict(saliency_max_err=float((out["saliency_scores"].cpu() - ref["saliency_scores"])[valid].abs().max()),
               pred_logits_max_err=float((out["pred_logits"].cpu() - ref["pred_logits"]).abs().max()),
               pred_spans_max_err=float((out["pred_spans"].cpu() - ref["pred_spans"]).abs().max()),
               tolerances=dict(saliency=1e-4, pred=2e-5))
    durations = (inputs["src_vid_mask"].sum(1) * 2.0).float().contiguous()
    common = dict(timestamp=tg["timestamp"].cpu().numpy(), timestamp_mask=tg["timestamp_mask"].cpu().numpy(), durations=durations.cpu().numpy())
    tails = {}
    for tag, o in (("hip", out), ("ref", ref)):
        pin, pj = os.path.join(tmp, tag + "_out.npz"), os.path.join(tmp, tag + "_tail.json")
        np.savez(pin, pred_logits=o["pred_logits"].cpu().numpy(), pred_spans=o["pred_spans"].cpu().numpy(), **common)
        run_job(dict(task="postproc", outputs_npz=pin, result_json=pj, clip_lengths=[0.0, 2.0]))
        with open(pj) as f:
            tails[tag] = json.load(f)
    same_inputs, end_to_end, excused, gaps = {}, {}, {}, {}
    for cl in (0.0, 2.0):
        win, order, keep, nk, _ = ops.postprocess_mr(out["pred_logits"], out["pred_spans"], None, tg["timestamp"], tg["timestamp_mask"], durations, clip_length=cl)
        win, order, keep, nk = win.cpu().numpy(), order.cpu().tolist(), keep.cpu().tolist(), nk.cpu().tolist()
        th, tr = tails["hip"][str(cl)], tails["ref"][str(cl)]
        same_inputs[str(cl)] = sum(win[b].tolist() == th["pre"][b] and [win[b, i].tolist() for i in keep[b][: nk[b]]] == th["nms"][b] for b in range(B))
        diff = [b for b in range(B) if not (order[b] == tr["order"][b] and keep[b][: nk[b]] == tr["keep"][b])]
        end_to_end[str(cl)], excused[str(cl)] = B - len(diff), diff
        # how close a call each differing sample is IN THE REFERENCE: the largest reference-score gap between the clips the two rankings swap
        rs = ref["pred_logits"][..., 0]
        for b in diff:
            sw = [i for i in range(Lv) if order[b][i] != tr["order"][b][i]]
            gaps[b] = max([abs(float(rs[b, order[b][i]]) - float(rs[b, tr["order"][b][i]])) for i in sw] or [0.0])
    rep.update(samples=B, post_nms_identical=min(end_to_end.values()), post_nms_excused=max(len(v) for v in excused.values()),
               post_nms_identical_by_clip_length=end_to_end, differing_samples=excused,
               differing_samples_reference_score_gap_of_swapped_clips={str(b): g for b, g in gaps.items()},
               device_tail_equals_reference_tail_on_same_inputs=same_inputs,
               what="the reference itself (child process, eval mode, fp32, weights = oracle seed 0) vs the default drop-in model under no_grad on the headline "
                    "batch; post_nms_identical = samples whose ranked clip indices AND post-NMS keep-set (nms_thd 0.7, max 10; raw and round_multiple 2 s) "
                    "equal the reference forward + the reference's own temporal_nms / PostProcessorDETR; for every differing sample the line carries "
                    "the largest gap, in the REFERENCE's own scores, between the clips the two rankings swap (a gap of the order of pred_logits_max_err is a "
                    "tie decided by rounding; the test suite applies the stricter fp64 rule of tests/test_gpu_parity_full.py::_reference_is_ambiguous "
                    "to its four pinned draws, this synthetic all-ones-mask batch is not one of them); "
                    "device_tail_equals_reference_tail_on_same_inputs = uvtg_postprocess_mr vs the reference's tail, both fed the HIP outputs")
    del model
    return rep


def cpu_baseline(batch, dev, steps=2):
    """CPU baseline (reported, not optimised against; SURVEY 8d, north_star: "the reference's own PyTorch CPU forward is timed on the host
    cores of the same box in the same run").  kind = "reference": the REAL showlab/UniVTG model path -- build_model() -> Model.forward +
    SetCriterion + backward (model/univtg.py:105-155,195-351,409-450), imported from oracle/_ref/uvtg_reference_model.zip, the archive
    __graft_entry__.build() packs from /root/reference where that tree exists (oracle/build_ref.py; git-ignored, ships with the built tree
    like the .so; the reference's LICENSE inside) -- in a CHILD process (oracle/ref_runner.py: the archive's generic `model` / `utils` / `eval`
    packages never enter this process, and the child refuses modules that did not come out of the archive) -- fp32, TRAIN mode, the GPU run's
    own synthetic batch at the full B: 1 warm-up + `steps` timed steps.  The same child then runs ONE eval-mode forward whose outputs are
    compared with the HIP path here (`reference_parity`).  The port (oracle/nn_baseline.py: the same torch.nn modules composed the same way +
    the oracle's criterion) is timed once beside it so that the ratio of the two is on the same box; it is the fallback (kind = "port") only
    when the archive is absent."""
    import shutil
    import tempfile
    import numpy as np
    from oracle import univtg_oracle as O
    from oracle.nn_baseline import NNBaseline
    threads = min(os.cpu_count() or 1, 32)                       # more threads than this only adds contention on the 2-socket host
    torch.set_num_threads(threads)
    inputs, tg = batch
    cfg = O.make_cfg(input_dropout=0.5, dropout=0.0, droppath=0.1)
    params = O.init_params(cfg, seed=0)
    cpu_in = {k: v.cpu() for k, v in inputs.items() if torch.is_tensor(v)}
    cpu_tg = {k: v.cpu() for k, v in tg.items() if torch.is_tensor(v) and not k.startswith("_")}
    B, Lv = cpu_in["src_vid"].shape[:2]

    ref, ref_note, parity = None, None, None
    tmp = tempfile.mkdtemp(prefix="uvtg_ref_")
    try:
        from oracle.build_ref import ARCHIVE
        from oracle.ref_runner import run_job
        if not os.path.exists(ARCHIVE):
            raise ImportError("oracle/_ref/uvtg_reference_model.zip is absent (built by __graft_entry__.build() where /root/reference exists)")
        bn, ev = os.path.join(tmp, "batch.npz"), os.path.join(tmp, "ref_eval.npz")
        np.savez(bn, **{"in/" + k: v.numpy() for k, v in cpu_in.items()}, **{"tg/" + k: v.numpy() for k, v in cpu_tg.items()})
        ref = run_job(dict(task="model", threads=threads, cfg=dict(input_dropout=0.5, dropout=0.0, droppath=0.1), param_seed=0, batch_npz=bn,
                           train_steps=steps, eval_out=ev))
        try:
            parity = reference_parity(batch, dev, ev, run_job, tmp)
        except Exception as e:                                   # the baseline number must not die with the comparison
            parity = dict(error=f"{type(e).__name__}: {e}")
    except (ImportError, RuntimeError) as e:
        ref_note = f"reference archive not usable: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

    port = NNBaseline(cfg)
    port.load_state_dict(params, strict=True)
    port.train()

    def port_one():
        port.zero_grad(set_to_none=True)
        t0 = time.perf_counter()
        out = port(**cpu_in)
        O.total_loss(O.criterion(out, cpu_tg, cfg), cfg).backward()
        return time.perf_counter() - t0
    port_one()                                                   # warm-up
    port_times = [port_one() for _ in range(1 if ref else max(steps, 3))]
    port_rate = len(port_times) * B * Lv / sum(port_times)
    common = dict(unit="clips/s", cores=torch.get_num_threads(), torch=torch.__version__, host_cpus=os.cpu_count())
    port_line = dict(value=round(port_rate, 1), step_s=[round(t, 2) for t in port_times], flavour="port-nn-modules",
                     what="oracle/nn_baseline.py (the reference's module composition rebuilt from torch.nn, pinned to the oracle) + the oracle's criterion, same batch, same threads")
    if ref:
        ref_times, ref_fwd = ref["train_step_s"], ref["train_forward_s"]
        t = sum(ref_times)
        rate = len(ref_times) * B * Lv / t
        return dict(value=rate, kind="reference", **dict(common, cores=ref["threads"]), process="child (oracle/ref_runner.py)",
                    forward_only_clips_per_sec=round(len(ref_fwd) * B * Lv / sum(ref_fwd), 1),
                    reference_archive_sha256=ref["manifest"], port_beside_it=port_line, port_over_reference=round(port_rate / rate, 3),
                    reference_parity=parity,
                    sample=f"the reference itself (showlab/UniVTG model/univtg.py build_model -> Model.forward + SetCriterion + backward, imported unmodified from "
                           f"oracle/_ref/uvtg_reference_model.zip in a child process): fp32, TRAIN mode (input dropout 0.5 + DropPath 0.1), the GPU run's own synthetic "
                           f"batch at the full B={B} (L_v={Lv}, all-ones masks): {len(ref_times)} timed steps after 1 warm-up ({t:.2f} s total, "
                           f"{t / len(ref_times):.2f} s per step, min {min(ref_times):.2f} s; forward alone {sum(ref_fwd) / len(ref_fwd):.2f} s)")
    t = sum(port_times)
    return dict(value=port_rate, kind="port", flavour="port-nn-modules", **common, fallback_reason=ref_note,
                sample=f"FALLBACK (no reference archive in this tree): the reference's module composition rebuilt from torch.nn (oracle/nn_baseline.py, pinned to the "
                       f"oracle by tests/test_oracle_golden.py) + the oracle's criterion: fwd+criterion+bwd, fp32, TRAIN mode, the GPU run's own batch at the full "
                       f"B={B}: {len(port_times)} timed steps after 1 warm-up ({t:.2f} s total, {t / len(port_times):.2f} s per step)")


def kernel_src_sha():
    """Identity of the GEMM kernels' source: profiles/*_pmc_nt256.json carries it, and a PMC file taken from other kernels is stale."""
    import hashlib
    h = hashlib.sha256()
    for f in ("gemm.hip", "uvtg_kernels.h", "uvtg_common.h"):
        with open(os.path.join(ROOT, "univtg_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def infer_batch(B, Lv, Lt, Dv, Dt, seed, dev):
    inputs, tg = synth_batch(B, Lv, Lt, Dv, Dt, seed, dev)
    lens_v = torch.tensor(inputs["_lens_host"][0], device=dev, dtype=torch.float32)
    return {k: v for k, v in inputs.items() if not k.startswith("_")}, tg["timestamp"], tg["timestamp_mask"], (lens_v * 2.0).contiguous()


def run_infer(args, dev):
    """Inference line: the drop-in model under torch.no_grad() + uvtg_postprocess_mr, i.e. the body of
    compute_mr_results + post-processing (main/inference_mr.py:88-193) on synthetic features resident in HBM.  Shapes: config 2 eval batches
    B=32 (scripts/qvhl_inference.sh:26 --eval_bsz 32) and B=256, and config 1 (batch 1, CLIP-only features D_v=514) as a latency.
    Precisions: the default ("auto" -> fp32x3 under no_grad: split-bf16 operands, 3 MFMAs per product -- the mode in which post-NMS indices
    equal the fp32 reference's) and the opt-in "bf16" (padded and packed row stream)."""
    from univtg_amd import _lib, ops
    from univtg_amd.model import build_model
    lib = _lib.load()
    cases = [("config2_B32", 32, 75, 32, MODEL["D_v"]), ("config2_B256", 256, 75, 32, MODEL["D_v"]), ("config1_B1", 1, 75, 32, 514)]
    res, roof = {}, None
    for name, B, Lv, Lt, Dv in cases:
        batches = [infer_batch(B, Lv, Lt, Dv, MODEL["D_t"], 50 + i, dev) for i in range(2)]
        for prec, packed in (("auto", False), ("bf16", False), ("bf16", True)):
            torch.manual_seed(2018)
            model, _ = build_model(model_args(max_v_l=Lv, v_feat_dim=Dv, precision=prec, packed=packed))
            model.to(dev).eval()

            def call(i):
                inp, ts, tm, dur = batches[i % 2]
                with torch.no_grad():
                    out = model(**inp)
                    return ops.postprocess_mr(out["pred_logits"], out["pred_spans"], out["saliency_scores"], ts, tm, dur, clip_length=2.0, eval_mode="add")
            for i in range(max(3, args.warmup)):
                call(i)
            torch.cuda.synchronize()
            n = max(10, args.steps)
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
            t0 = time.perf_counter()
            evs[0].record()
            for i in range(n):
                call(i)
                evs[i + 1].record()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / n * 1e3
            per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(n))
            key = f"{name}/{'fp32x3' if prec == 'auto' else 'bf16'}{'_packed' if packed else ''}"
            res[key] = dict(ms_per_batch=round(wall, 3), ms_event_median=round(per[n // 2], 3), clips_per_sec=round(B * Lv / wall * 1e3, 1),
                            samples_per_sec=round(B / wall * 1e3, 1))
            if prec == "auto" and not packed:                       # the same call replayed from a HIP graph (univtg_amd/graph.py)
                from univtg_amd.graph import GraphedInference
                run = GraphedInference(model, clip_length=2.0, eval_mode="add")

                def gcall(i):
                    inp, ts, tm, dur = batches[i % 2]
                    return run(inp["src_txt"], inp["src_txt_mask"], inp["src_vid"], inp["src_vid_mask"], ts, tm, dur)
                for i in range(4):
                    gcall(i)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(n):
                    gcall(i)
                torch.cuda.synchronize()
                gw = (time.perf_counter() - t0) / n * 1e3
                res[key + "_hipgraph"] = dict(ms_per_batch=round(gw, 3), clips_per_sec=round(B * Lv / gw * 1e3, 1), samples_per_sec=round(B / gw * 1e3, 1))
                del run
            if name == "config2_B256" and prec == "auto":           # roofline of the split-operand GEMM (family 1)
                lib.uvtg_profile_start()
                for i in range(3):
                    call(i)
                ms, fl, cnt = (C.c_double * 8)(), (C.c_double * 8)(), (C.c_longlong * 8)()
                _lib.check(lib.uvtg_profile_stop(ms, fl, cnt), "uvtg_profile_stop")
                floor = lib.uvtg_profile_event_floor_ms()
                raw = ms[1] + floor * cnt[1]
                ach = fl[1] / (raw * 1e-3) / 1e12 if raw > 0 else 0.0
                roof = dict(bound="mfma", kernel="split-operand GEMM (fp32x3: fp16 hi / lo images, hi*hi + hi*lo + lo*hi; gemm_nt256_kernel<..., HALF> / gemm_nt_kernel<true>)", achieved=round(ach, 2), peak=round(2500.0 / 3, 1),
                            unit="TFLOP/s", frac=round(ach / (2500.0 / 3), 4), traffic=None,
                            note="achieved = sum(2MNK) ALGORITHMIC flops / event-pair time over the launches; every product costs 3 fp16 MFMAs, so the "
                                 "effective peak is 2.5 PFLOP/s / 3; executed MFMA rate = 3 x achieved",
                            executed_mfma_tflops=round(3 * ach, 1), launches_per_batch=int(cnt[1] // 3), avg_launch_us=round(raw * 1e3 / max(1, cnt[1]), 2))
            del model
    head = res["config2_B32/fp32x3"]
    out = dict(metric="clips/sec inference (L=75,d=1024) forward + post-processing", value=head["clips_per_sec"], unit="clips/s", n_gpus=1,
               steps=max(10, args.steps), warmup=max(3, args.warmup), ms_per_step=head["ms_per_batch"], higher_is_better=True, scaling="weak",
               vs_baseline=None, dtype="fp32x3 (fp16 hi + lo operand images, three products: fp32-class)", data="synthetic",
               config=dict(workload="QVHighlights inference shape (BASELINE config 2 eval, scripts/qvhl_inference.sh: eval batch 32): model(...) under "
                                    "torch.no_grad() with the DEFAULT precision ('auto' -> fp32x3) + uvtg_postprocess_mr (decode, rank, round_multiple 2 s, "
                                    "hull-IoU NMS 0.7, eval_mode add saliency), padded execution, ragged lengths len_v~U{38..75}; timing = host wall clock per "
                                    "batch incl. the Python boundary and per-call allocations", baseline_config=2, per_gpu_batch=32, mode="infer"),
               cases=res, roofline=roof, cpu_baseline=None,
               note="cases: <shape>/<precision>[_packed][_hipgraph]; _hipgraph = the same call (forward + post-processing) captured once per shape and "
                    "replayed as a HIP graph (univtg_amd.graph.GraphedInference; inputs copied into the graph's static buffers per call); "
                    "config1_B1 = batch-1 latency with CLIP-only features (D_v=514); bf16 = opt-in fast mode "
                    "(post-NMS top-1 identical to fp32 for ~98 % of the samples); packed = Model(packed=True): encoder on valid rows + one "
                    "representative padded clip per sample, one device->host read of the mask sums per call")
    print(json.dumps(out))


def timed_steps(step, batches, n, barrier=None):
    """n steps bracketed by (barrier +) device synchronisation; returns (elapsed seconds, sorted per-step event times in ms)."""
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    (barrier or torch.cuda.synchronize)()
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(n):
        step.step(*batches[i % len(batches)])
        evs[i + 1].record()                                   # on the launch stream (torch's current stream)
    (barrier or torch.cuda.synchronize)()
    elapsed = time.perf_counter() - t0
    return elapsed, sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(n))


def encoder_flops(lens, B, Lv, Lt, packed_halo):
    """(algorithmic, executed) encoder fwd+bwd FLOPs per step: SURVEY 8d's 3*E*B*(8Sd^2+4SdF+4S^2d) with padded positions counted, and
    the same sum over the rows the step really runs (packed loss-only stream: kept clips = valid + 3-clip conv halo, valid text tokens;
    every stream: no text rows in the last layer's FFN)."""
    S, d, F_, E = Lv + Lt, MODEL["d"], MODEL["F"], MODEL["E"]
    alg = 3 * E * B * (8 * S * d * d + 4 * S * d * F_ + 4 * S * S * d)
    # the LAST layer's FFN runs on the clip rows only (engine.hip, last_layer_clip): its text rows' 3 x 4 d F are not executed
    clip = 0 if os.environ.get("UVTG_LAST_CLIP_OFF") else 3 * 4 * d * F_
    # the encoder SECTION also holds the four conv-head weight gradients (2 R_f d 3d each over the zero-framed rows): they ride in the section's
    # deferred weight-gradient launch (engine.hip, tn_flush; N > 1: in the group behind layer 1) -- counted as executed work of the section, not as
    # encoder FLOPs; only the per-layer-event mode of rounds 2-5 (UVTG_TN_EVENTS_PER_LAYER) keeps them outside
    conv = 0 if (os.environ.get("UVTG_TN_CONV_DEFER_OFF") or os.environ.get("UVTG_TN_DEFER_OFF")
                 or ((int(os.environ.get("WORLD_SIZE", "1")) > 1 or "force" in sys.argv) and os.environ.get("UVTG_TN_EVENTS_PER_LAYER"))) else 4 * 2 * d * 3 * d
    if not packed_halo:
        return alg, alg - clip * B * Lt + conv * B * (Lv + 2)
    per = [[min(Lv, x + 3) + y for x, y in zip(a, b)] for a, b in lens]
    exe = sum(3 * E * sum(r * (8 * d * d + 4 * d * F_) + 4 * r * r * d for r in rows) - clip * sum(b) + conv * sum(min(Lv, x + 3) + 2 for x in a)
              for rows, (a, b) in zip(per, lens)) / len(per)
    return alg, exe


def encoder_roofline(alg_flops, exe_flops, t_enc, B, Lv, halo, lens, roof):
    """north_star's number AND the executed one.  `frac_survey` = SURVEY 8d's "Target translation": the ALGORITHMIC encoder
    fwd+bwd FLOPs 3*E*B*(8Sd^2+4SdF+4S^2d) (padded positions counted as the reference computes them; 4.280 TFLOP at config 2) / t_encoder /
    2.5 PFLOP/s -- t_encoder as measured, i.e. INCLUDING the four conv-head weight gradients that ride in the section's deferred
    weight-gradient launch on a single rank (they cannot be bracketed by events: one launch).  `frac_executed` divides the FLOPs the section
    really runs.  `frac_survey_conv_wgrads_priced_out` removes the conv gradients' share of the section time at the weight-gradient family's
    own measured rate (derived, not an event pair)."""
    d = MODEL["d"]
    conv_on = not (os.environ.get("UVTG_TN_CONV_DEFER_OFF") or os.environ.get("UVTG_TN_DEFER_OFF")
                   or ((int(os.environ.get("WORLD_SIZE", "1")) > 1 or "force" in sys.argv) and os.environ.get("UVTG_TN_EVENTS_PER_LAYER")))
    rows_f = (sum(sum(min(Lv, x + 3) + 2 for x in a) for a, _ in lens) / len(lens)) if halo else B * (Lv + 2)
    conv_flops = 4 * 2 * d * 3 * d * rows_f if conv_on else 0.0
    tn = ((roof or {}).get("all_gemm_kernels") or {}).get("gemm_tn_kernel") or {}
    t_conv = conv_flops / (tn["tflops"] * 1e12) if tn.get("tflops") else 0.0
    out = dict(achieved=round(exe_flops / t_enc / 1e12, 1), peak=2500.0, unit="TFLOP/s",
               frac=round(exe_flops / t_enc / 2.5e15, 4), frac_executed=round(exe_flops / t_enc / 2.5e15, 4),
               frac_survey=round(alg_flops / t_enc / 2.5e15, 4), algorithmic_tflop_per_step=round(alg_flops / 1e12, 3),
               executed_tflop_per_step=round(exe_flops / 1e12, 3), target_frac=0.40,
               conv_head_wgrads_in_section_tflop=round(conv_flops / 1e12, 3),
               frac_survey_conv_wgrads_priced_out=(round(alg_flops / max(t_enc - t_conv, 1e-9) / 2.5e15, 4) if t_conv else None),
               note="frac_survey = SURVEY 8d's algorithmic encoder FLOPs (3*E*B*(8Sd^2+4SdF+4S^2d), padded positions counted) / t_encoder / 2.5 PFLOP/s: the "
                    "number north_star's >= 40 % target is written against; t_encoder (HIP events on the launch stream around the E layers forward + "
                    "their backward incl. the deferred weight-gradient launch) INCLUDES the four conv-head weight gradients that ride in that launch on "
                    "a single rank.  frac_executed = FLOPs of the rows the section really runs (algorithmic - the last layer's FFN on text rows, which no "
                    "longer run, + those conv-head weight gradients; packed streams: the kept rows only) / the same time.  "
                    "frac_survey_conv_wgrads_priced_out = algorithmic / (t_encoder - conv-gradient FLOPs at the weight-gradient family's measured rate): derived")
    return out


def quick_roofline(lib, step, batches, B, Lv, Lt, halo, k=3):
    """Dominant-GEMM and encoder-section rooflines of `step` on `batches` (k instrumented steps each; single rank): the same event-pair
    measurements the headline's `roofline` / `roofline_encoder` objects come from, for the companion variant."""
    from univtg_amd import _lib
    lib.uvtg_profile_start()
    for i in range(k):
        step.step(*batches[i % len(batches)])
    ms, fl, n = (C.c_double * 8)(), (C.c_double * 8)(), (C.c_longlong * 8)()
    _lib.check(lib.uvtg_profile_stop(ms, fl, n), "uvtg_profile_stop")
    floor = lib.uvtg_profile_event_floor_ms()
    raw = ms[3] + floor * n[3]
    ach = fl[3] / (raw * 1e-3) / 1e12 if raw > 0 else 0.0
    lib.uvtg_profile_sections_start()
    for i in range(k):
        step.step(*batches[i % len(batches)])
    sm, sn = (C.c_double * 4)(), (C.c_longlong * 4)()
    _lib.check(lib.uvtg_profile_sections_stop(sm, sn), "uvtg_profile_sections_stop")
    t_enc = max((sm[0] + sm[1]) / k * 1e-3, 1e-9)
    lens = [bt[0]["_lens_host"] for bt in batches]
    alg, exe = encoder_flops(lens, B, Lv, Lt, halo)
    return dict(roofline=dict(kernel="gemm_nt256_kernel", achieved=round(ach, 2), peak=2500.0, unit="TFLOP/s", frac=round(ach / 2500.0, 4),
                              launches_per_step=int(n[3] // k), avg_launch_us=round(raw * 1e3 / max(1, n[3]), 2)),
                t_encoder_ms=round(t_enc * 1e3, 3),
                roofline_encoder=dict(achieved=round(exe / t_enc / 1e12, 1), peak=2500.0, unit="TFLOP/s", frac=round(exe / t_enc / 2.5e15, 4),
                                      frac_executed=round(exe / t_enc / 2.5e15, 4), frac_survey=round(alg / t_enc / 2.5e15, 4),
                                      frac_survey_note="SURVEY 8d algorithmic FLOPs of the PADDED shape / t_encoder: on a packed (ragged) stream the padded "
                                                       "positions are counted but not executed"))


def companion_config(cid, dev, precise, lib, k_prof=2, steps=10):
    """One other BASELINE config on this GPU, inside the driver-run line: a fresh model + TrainStep at the config's
    shape, variant B (ragged valid lengths on the packed loss-only stream, `value` = VALID clips), 3 warm-up + `steps` timed steps + the
    instrumented steps of quick_roofline.  Same step, same arithmetic as the headline."""
    from univtg_amd.model import build_model
    from univtg_amd.trainer import TrainStep
    wl = CONFIGS[cid]
    B, Lv, Lt = wl["B"], wl["L_v"], wl["L_t"]
    torch.manual_seed(2018)
    model, crit = build_model(model_args(max_v_l=Lv, proj_precise=precise))
    model.to(dev).train()
    crit.to(dev).train()
    model.set_seed(2018)
    step = TrainStep(model, crit, lr=1e-4, weight_decay=1e-4, grad_clip=0.1, packed="auto")
    lens_fn = (lambda s_: mixed_length_lens(B, seed=s_)) if cid == 5 else (lambda s_: None)
    bt = [synth_batch(B, Lv, Lt, MODEL["D_v"], MODEL["D_t"], 7000 + i, dev, lens_fn(7000 + i), full=False) for i in range(2)]
    for i in range(3):
        step.step(*bt[i % 2])
    el, per = timed_steps(step, bt, steps)
    lens = [b[0]["_lens_host"] for b in bt]
    valid = sum(sum(a) for a, _ in lens) / len(lens)
    out = dict(what=wl["what"] + "; variant B: " + wl["lens"], per_gpu_batch=B, L_v=Lv, L_t=Lt, steps=steps,
               ms_per_step=round(el / steps * 1e3, 3), ms_per_step_event_median=round(per[len(per) // 2], 3),
               valid_clips_per_sec=round(valid * steps / el, 1), clip_positions_per_sec_incl_padded=round(B * Lv * steps / el, 1),
               losses=[round(x, 5) for x in step.losses[:5].tolist()])
    out.update(quick_roofline(lib, step, bt, B, Lv, Lt, True, k_prof))
    del step, model, crit, bt
    torch.cuda.empty_cache()
    return out


def companion_drop_in(dev, batches, precise, Lv, steps=30):
    """What INTEGRATION.md section 1 gives a maintainer who keeps the reference's loop: the body of
    main/train_vlp_ddp.py:56-68 -- model(**model_inputs) -> criterion -> weighted sum -> optimizer.zero_grad() -> losses.backward() ->
    clip_grad_norm_(model.parameters(), 0.1) -> optimizer.step() -- on the drop-in model (autograd path: one C call forward, one backward,
    dense criterion gradients), on the headline batches.  Twice: with the reference's own torch.optim.AdamW (main/config.py:349-350) and with
    univtg_amd.optim.FusedAdamWClip swapped in for that one constructor (clip fused into the step, the clip line dropped)."""
    from univtg_amd.model import build_model
    from univtg_amd.optim import FusedAdamWClip
    res = {}
    for name in ("torch_adamw_and_clip_grad_norm", "fused_adamw_clip"):
        torch.manual_seed(2018)
        model, crit = build_model(model_args(max_v_l=Lv, proj_precise=precise))
        model.to(dev).train()
        crit.to(dev).train()
        model.set_seed(2018)
        group = [{"params": [p for n, p in model.named_parameters() if p.requires_grad]}]
        fused = name == "fused_adamw_clip"
        opt = FusedAdamWClip(group, lr=1e-4, weight_decay=1e-4, max_grad_norm=0.1, model=model) if fused else torch.optim.AdamW(group, lr=1e-4, weight_decay=1e-4)

        def one(i):
            inputs, targets = batches[i % len(batches)]
            model_inputs = {k: v for k, v in inputs.items() if not k.startswith("_")}
            outputs = model(**model_inputs)
            loss_dict = crit(outputs, targets)
            weight_dict = crit.weight_dict
            losses = sum(loss_dict[k] * weight_dict[k] for k in loss_dict.keys() if k in weight_dict)
            opt.zero_grad()
            losses.backward()
            if not fused:
                torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1)
            opt.step()
            return losses
        for i in range(4):
            one(i)
        torch.cuda.synchronize()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        evs[0].record()
        for i in range(steps):
            last = one(i)
            evs[i + 1].record()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
        res[name] = dict(ms_per_step=round(el / steps * 1e3, 3), ms_per_step_event_median=round(per[steps // 2], 3), loss=round(float(last), 5),
                         grads_read_in_place=(opt.in_place_steps == steps + 4) if fused else None)
        del model, crit, opt
        torch.cuda.empty_cache()
    res["what"] = ("the loop body of main/train_vlp_ddp.py:56-68 on the drop-in model (univtg_amd.model.build_model, autograd path), the headline batches, "
                   "host wall clock per step incl. Python; torch_adamw_and_clip_grad_norm = the reference's optimizer lines unchanged; fused_adamw_clip = "
                   "univtg_amd.optim.FusedAdamWClip(param_dicts, lr, weight_decay, max_grad_norm=opt.grad_clip, model=model) in place of "
                   "torch.optim.AdamW(...) and the clip_grad_norm_ line dropped")
    return res


def companion_input_pipeline(dev, step, batches, steps=30):
    """One measured step that STARTS ON THE HOST (main/dataset.py:1037-1100, utils/tensor_utils.py:5-53): the
    headline batches as `PackedHostBatch`es in pinned host memory (what DataLoader workers with collate_fn=pack_batch_host hand over), uploaded by
    `DevicePrefetcher` (depth 2: batch n + 1's H2D copies + uvtg_ragged_to_padded on a side stream under batch n's step), fp32 and bf16 wire, the
    SAME TrainStep as the headline.  `resident` = the same loop over the batches already in HBM, for the in-run ratio."""
    import itertools
    from univtg_amd.pipeline import DevicePrefetcher, pack_batch_host

    def samples_of(batch):
        inputs, tg = batch
        lv, lt = inputs["_lens_host"]
        cpu = {k: v.cpu() for k, v in {**{k: v for k, v in inputs.items() if torch.is_tensor(v)}, **{k: v for k, v in tg.items() if torch.is_tensor(v)}}.items()}
        return [dict(meta=dict(qid=b), model_inputs=dict(
            query_feat=cpu["src_txt"][b, :lt[b]], video_feat=cpu["src_vid"][b, :lv[b]], timestamp=cpu["timestamp"][b, :lv[b]],
            timestamp_window=cpu["timestamp_window"][b, :lv[b]], span_labels_nn=cpu["span_labels_nn"][b, :lv[b]],
            saliency_scores=cpu["saliency_scores"][b, :lv[b]], saliency_pos_labels=[int(cpu["saliency_pos_labels"][b, 0])])) for b in range(len(lv))]

    def run(feed, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for inputs, targets in feed(n):
            step.step(inputs, targets)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    res = {}
    for i in range(3):
        step.step(*batches[i % 2])
    res["resident"] = dict(ms_per_step=round(run(lambda n: (batches[i % 2] for i in range(n)), steps), 3))
    sam = [samples_of(b) for b in batches]
    for name, wire in (("fp32_wire", torch.float32), ("bf16_wire", torch.bfloat16)):
        t0 = time.perf_counter()
        pbs = [pack_batch_host(x, feature_dtype=wire) for x in sam]
        t_pack = (time.perf_counter() - t0) / len(pbs) * 1e3
        mb = sum(blk.numel() * blk.element_size() for pb in pbs[:1] for blk, _, _, _ in pb.padded.values()) / 1e6
        pf = DevicePrefetcher(itertools.islice(itertools.cycle(pbs), steps + 3), dev, depth=2, timing=True)

        it = iter(pf)
        for _ in range(3):                                  # warm-up batches through the same prefetcher
            _, mi, tg = next(it)
            step.step(mi, tg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            _, mi, tg = next(it)
            step.step(mi, tg)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        next(it, None)                                      # exhausted: the prefetcher folds its side-stream event pairs into stats
        res[name] = dict(ms_per_step=round(ms, 3), over_resident=round(ms / res["resident"]["ms_per_step"], 4), wire_mb_per_batch=round(mb, 1),
                         upload_ms_per_batch_side_stream=round(pf.stats["upload_ms"] / max(1, pf.stats["batches"]), 3),
                         host_enqueue_ms_per_batch=round(pf.stats["host_collate_s"] / max(1, pf.stats["batches"]) * 1e3, 3),
                         pack_ms_per_batch_one_host_process=round(t_pack, 1))
    res["what"] = ("the headline TrainStep fed from PINNED HOST memory: PackedHostBatch (univtg_amd.pipeline.pack_batch_host, the collate_fn of the loader's "
                   "workers; its one-process cost is pack_ms_per_batch_one_host_process, outside the timed loop) -> DevicePrefetcher depth 2 (H2D copies + "
                   "uvtg_ragged_to_padded on a side stream) -> step; ms_per_step = host wall clock per step over the timed steps; resident = the same loop "
                   "over batches already in HBM; upload_ms_per_batch_side_stream = device time of one batch's copies + padding kernels (hidden when < the step)")
    return res


def companion_infer(dev, n=20):
    """The inference call (model(...) under no_grad at the default precision + uvtg_postprocess_mr; main/inference_mr.py:88-193) at the
    reference's eval batch 32 and at batch 1 with CLIP-only features (BASELINE config 1's shape), inside the driver-run line."""
    from univtg_amd import ops
    from univtg_amd.model import build_model
    res = {}
    for name, B, Lv, Lt, Dv in (("config2_eval_B32", 32, 75, 32, MODEL["D_v"]), ("config1_B1", 1, 75, 32, 514)):
        batches = [infer_batch(B, Lv, Lt, Dv, MODEL["D_t"], 50 + i, dev) for i in range(2)]
        torch.manual_seed(2018)
        model, _ = build_model(model_args(max_v_l=Lv, v_feat_dim=Dv, precision="auto", packed=False))
        model.to(dev).eval()

        def call(i):
            inp, ts, tm, dur = batches[i % 2]
            with torch.no_grad():
                out = model(**inp)
                return ops.postprocess_mr(out["pred_logits"], out["pred_spans"], out["saliency_scores"], ts, tm, dur, clip_length=2.0, eval_mode="add")
        for i in range(4):
            call(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            call(i)
        torch.cuda.synchronize()
        w = (time.perf_counter() - t0) / n * 1e3
        res[name] = dict(ms_per_batch=round(w, 3), clips_per_sec=round(B * Lv / w * 1e3, 1), precision="fp32x3 (default under no_grad)", batches=n)
        del model
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json config (2 = the headline metric)")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=3, help="extra instrumented steps for the roofline line")
    ap.add_argument("--no-padded-compare", action="store_true", help="variant B: skip the extra timing of the padded / all-clip-rows executions")
    ap.add_argument("--no-companions", action="store_true", help="headline only: skip the ragged-batch (variant B) and bf16-projection companion timings")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the config 3 / 4 / 5 and inference companions of the default line")
    ap.add_argument("--packed", default="auto", choices=["auto", "off"], help="encoder row stream (auto = exact packed stream)")
    ap.add_argument("--variant", default=None, choices=["A", "B"],
                    help="SURVEY 8d: A = all-ones masks -- every clip position is executed (default for config 2: the shape 'L=75, d=1024, batch=256' "
                         "literally names, `value` counts executed clips only); B = ragged valid lengths on the packed stream (default for configs 3-5)")
    ap.add_argument("--proj", default="precise", choices=["precise", "bf16"],
                    help="input projections of the timed train step: precise = fp32-class split operands (saliency_scores within 1e-4 of the fp32 "
                         "reference, the north_star tolerance; default), bf16 = plain bf16 operands (3e-2)")
    ap.add_argument("--mode", default="train", choices=["train", "infer"], help="train = the headline metric; infer = forward + post-processing")
    ap.add_argument("--grad-comm-dtype", default=os.environ.get("UVTG_GRAD_COMM_DTYPE", "fp32"), choices=["fp32", "bf16"], help="wire dtype of the gradient buckets (N > 1)")
    ap.add_argument("--overlap", default="auto", choices=["auto", "force"],
                    help="force = run the N > 1 step (per-layer readiness events, bucketed side-stream exchange, clipping-norm pass) on ONE rank: what the "
                         "data-parallel step costs in compute against the timed single-rank step (dev)")
    ap.add_argument("--comm-cus", type=int, default=int(os.environ.get("UVTG_COMM_CUS", "0")), help="CUs kept out of the persistent GEMM grids for RCCL's kernels (N > 1)")
    args = ap.parse_args()
    if args.variant is None:
        args.variant = "A" if args.config == 2 else "B"

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        print("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)", file=sys.stderr)
        sys.exit(2)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # plumbing test on a 1-GPU box: UVTG_BENCH_BACKEND=gloo UVTG_BENCH_SAME_DEVICE=1 runs the N>1 control flow with every rank on cuda:0
    backend = os.environ.get("UVTG_BENCH_BACKEND", "nccl")
    if os.environ.get("UVTG_BENCH_SAME_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm_size = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.distributed.init_process_group(backend="nccl", device_id=dev)     # "nccl" == RCCL on ROCm
        else:
            torch.distributed.init_process_group(backend=backend)
        comm_size = torch.distributed.get_world_size()
    elif args.overlap == "force":                # the N > 1 step on one rank: a one-rank RCCL group carries the (no-op) range all-reduces
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
        torch.distributed.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    if comm_size != max(world, 1) or (args.gpus > 1 and comm_size != args.gpus):
        print(f"bench.py: --gpus {args.gpus} but the process group has {comm_size} ranks (WORLD_SIZE={world})", file=sys.stderr)
        sys.exit(2)

    if args.mode == "infer":
        if world > 1:
            print("bench.py --mode infer is a single-GPU line", file=sys.stderr)
            sys.exit(2)
        run_infer(args, dev)
        return
    from univtg_amd import _lib
    from univtg_amd.model import build_model
    from univtg_amd.trainer import TrainStep
    wl = CONFIGS[args.config]
    B, Lv, Lt = args.batch or wl["B"], wl["L_v"], wl["L_t"]
    torch.manual_seed(2018)
    precise = args.proj == "precise"
    model, crit = build_model(model_args(max_v_l=Lv, proj_precise=precise))
    model.to(dev).train()
    crit.to(dev).train()
    model.set_seed(2018 + rank)
    packed = False if args.packed == "off" else "auto"
    step = TrainStep(model, crit, lr=1e-4, weight_decay=1e-4, grad_clip=0.1, packed=packed, grad_comm_dtype=args.grad_comm_dtype,
                     comm_cus=args.comm_cus, time_comm=world > 1, overlap_comm="force" if args.overlap == "force" else True)
    full = args.variant == "A"

    def make_batches(full_, n=2):
        lens_fn = (lambda s_: mixed_length_lens(B, seed=s_)) if (args.config == 5 and not full_) else (lambda s_: None)
        return [synth_batch(B, Lv, Lt, MODEL["D_v"], MODEL["D_t"], 1000 * rank + i, dev, lens_fn(1000 * rank + i), full=full_) for i in range(n)]
    batches = make_batches(full)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # ---- the headline: W untimed steps, then EXACTLY K timed steps between barrier + synchronize ----
    for i in range(args.warmup):
        step.step(*batches[i % 2])
    elapsed, per_step = timed_steps(step, batches, args.steps, barrier)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)
    losses = step.losses[:5].tolist()
    exposed = sorted(step.exposed_comm_ms()) if world > 1 else []       # (outside the timed region; one sample per timed / warm-up step)
    # a stable median needs >= 50 samples whatever --steps was: extra steps, outside `value`
    n_med = max(50, args.steps)
    if n_med > args.steps:
        _, per_med = timed_steps(step, batches, n_med, barrier)
    else:
        per_med = per_step
    # data-parallel replicas must hold bit-identical parameters after the same number of steps
    replicas = None
    if world > 1:
        cs = torch.tensor(step.flat_checksum(), device=dev, dtype=torch.float64)
        allcs = [torch.zeros_like(cs) for _ in range(world)]
        torch.distributed.all_gather(allcs, cs)
        same = all(bool((c == allcs[0]).all()) for c in allcs)
        replicas = dict(flat_parameter_checksums_equal=same, checksum_rank0=[float(x) for x in allcs[0].tolist()], ranks=world)

    # ---- companions (single GPU, rank 0): the other variant's step, the bf16-projection step, the padded executions ----
    comp = {}
    if rank == 0 and world == 1 and not args.no_companions:
        other_full = not full
        ob = make_batches(other_full)
        for i in range(3):
            step.step(*ob[i % 2])
        el, per = timed_steps(step, ob, 30)
        lens_o = [bt[0]["_lens_host"] for bt in ob]
        valid = sum(sum(a) for a, _ in lens_o) / len(lens_o)
        alg_o, exe_o = encoder_flops(lens_o, B, Lv, Lt, bool(packed) and not other_full)
        comp["variant_" + ("A" if other_full else "B")] = dict(
            what=("all-ones masks: every clip position executed" if other_full else
                  f"ragged valid lengths ({wl['lens']}) on the packed loss-only stream: valid clips + 3-clip conv halo + valid text tokens; losses and all "
                  "parameter gradients exactly the padded execution's"),
            ms_per_step=round(el / 30 * 1e3, 3), ms_per_step_event_median=round(per[len(per) // 2], 3),
            executed_or_valid_clips_per_sec=round((B * Lv if other_full else valid) * 30 / el, 1),
            clip_positions_per_sec_incl_padded=round(B * Lv * 30 / el, 1),
            encoder_rows_fraction=round(sum(sum(min(Lv, x + 3) for x in a) + sum(b) for a, b in lens_o) / (len(lens_o) * B * (Lv + Lt)), 4) if (packed and not other_full) else 1.0,
            executed_encoder_tflop_per_step=round(exe_o / 1e12, 3))
        comp["variant_" + ("A" if other_full else "B")].update(quick_roofline(_lib.load(), step, ob, B, Lv, Lt, bool(packed) and not other_full,
                                                                               max(1, args.profile_steps)))
        if not other_full and packed and not args.no_padded_compare:        # the same ragged batches through the padded executions
            for kind in ("padded", "allrows"):
                step_p = TrainStep(model, crit, lr=1e-4, weight_decay=1e-4, grad_clip=0.1, packed=False if kind == "padded" else "auto", loss_only=False)
                for i in range(3):
                    step_p.step(*ob[i % 2])
                elp, _ = timed_steps(step_p, ob, 10)
                comp["variant_B"]["padded_execution_ms_per_step" if kind == "padded" else "all_clip_rows_ms_per_step"] = round(elp / 10 * 1e3, 3)
                del step_p
        # the other projection arithmetic on the headline batches
        model.proj_precise = not precise
        for i in range(3):
            step.step(*batches[i % 2])
        el, per = timed_steps(step, batches, 30)
        comp["projections_" + ("bf16" if precise else "precise")] = dict(
            what=("plain bf16 input projections in the forward (saliency_scores within 3e-2 of fp32)" if precise else
                  "fp32-class split-operand input projections in the forward (saliency_scores within 1e-4 of fp32)"),
            ms_per_step=round(el / 30 * 1e3, 3), ms_per_step_event_median=round(per[len(per) // 2], 3))
        model.proj_precise = precise
        for i in range(2):
            step.step(*batches[i % 2])

    # ---- roofline of the dominant kernel: HIP events around every GEMM launch, on the launch stream ----
    lib = _lib.load()
    roof, sect, roof_attn, roof_hbm = None, None, None, None
    lens_a = [bt[0]["_lens_host"] for bt in batches]
    halo = bool(packed) and not full
    if rank == 0:
        lib.uvtg_profile_start()
    for i in range(args.profile_steps):          # EVERY rank runs these steps (they contain the gradient all-reduce); only rank 0 instruments them
        step.step(*batches[i % 2])
    if rank == 0:
        ms, fl, n = (C.c_double * 8)(), (C.c_double * 8)(), (C.c_longlong * 8)()
        _lib.check(lib.uvtg_profile_stop(ms, fl, n), "uvtg_profile_stop")
        floor = lib.uvtg_profile_event_floor_ms()
        pby = (C.c_double * 8)()
        lib.uvtg_profile_bytes(pby)
        fam = ["gemm_nt_kernel<bf16>", "gemm_nt_kernel<split>", "gemm_tn_kernel", "gemm_nt256_kernel"]
        dom = max(range(4), key=lambda i: ms[i])
        raw_ms = ms[dom] + floor * n[dom]                     # durations as the event pairs saw them (no floor correction)
        ach = fl[dom] / (raw_ms * 1e-3) / 1e12 if raw_ms > 0 else 0.0
        ach_corr = fl[dom] / (ms[dom] * 1e-3) / 1e12 if ms[dom] > 0 else 0.0
        # HBM traffic of the dominant kernel: separate rocprofv3 --pmc passes of this same command (tools/pmc_nt256.sh), valid only when taken
        # from THESE kernels (the file carries the hash of the GEMM sources; the GPU box has no .git to compare a HEAD with)
        traffic, traffic_note, traffic_ratio = None, None, None
        alg_bytes = int(pby[dom] / max(1, n[dom])) if pby[dom] > 0 else None      # counted by the library on this run's own launches
        pmc = next((q for q in (os.path.join(ROOT, "profiles", f) for f in ("r06_pmc_nt256.json", "r05_pmc_nt256.json", "r04_pmc_nt256.json")) if os.path.exists(q)), None)
        if fam[dom] == "gemm_nt256_kernel" and pmc:
            with open(pmc) as f:
                pj = json.load(f)
            if (pj.get("variant", "A"), pj.get("config", 2)) != (args.variant, args.config):
                traffic_note = (f"{os.path.relpath(pmc, ROOT)} was taken on config {pj.get('config', 2)} variant {pj.get('variant', 'A')} (the default workload), "
                                f"not on this one: not quoted")
            elif pj.get("kernel_src_sha") == kernel_src_sha():
                traffic = pj["traffic_bytes_per_launch"]
                traffic_ratio = round(traffic / alg_bytes, 3) if alg_bytes else None
                traffic_note = (f"measured by separate rocprofv3 --pmc passes of this command on these kernels ({os.path.relpath(pmc, ROOT)}, kernel source "
                                f"{pj['kernel_src_sha']}, git {pj.get('git_head', '?')}): FETCH_SIZE x {pj.get('fetch_factor', 2.0)} ({pj.get('fetch_factor_source', 'guide')}) "
                                f"+ WRITE_SIZE per launch; MFMA-busy fraction {pj['mfma_busy_frac']:.3f}; not re-measured inside this run (PMC passes cannot share a run with timing)")
            else:
                traffic_note = (f"stale: {os.path.relpath(pmc, ROOT)} was taken from kernel source {pj.get('kernel_src_sha')} (git {pj.get('git_head', '?')}), this build is "
                                f"{kernel_src_sha()} -- re-run tools/pmc_nt256.sh")
        roof = dict(bound="mfma", kernel=fam[dom], achieved=round(ach, 2), peak=2500.0, unit="TFLOP/s", frac=round(ach / 2500.0, 4),
                    traffic=traffic, traffic_note=traffic_note, algorithmic_bytes_per_launch=alg_bytes, traffic_over_algorithmic=traffic_ratio, launches_per_step=int(n[dom] // max(1, args.profile_steps)),
                    avg_launch_us=round(raw_ms * 1e3 / max(1, n[dom]), 2),
                    event_pair_floor_us=round(floor * 1e3, 2), achieved_floor_corrected=round(ach_corr, 2),
                    note="achieved = sum(2MNK of the rows each launch executes) / sum(event-pair duration) over the launches, durations NOT floor-corrected",
                    clock_note=("peak is the guide's dense bf16 figure at the 2.4 GHz nominal clock; the chip does not hold that clock under this load: a resident "
                                "probe wave sampling s_memtime against the 100 MHz real-time counter (tools/clock_probe.py, profiles/r06_shader_clock_under_load.txt; "
                                "not re-measured in this run) reads 2.40-2.42 GHz idle, 1.94 GHz mean (p5 1.48, median 1.90) while this kernel runs back to back and "
                                "2.0-2.1 GHz over the training step -- the matrix cores' ceiling at the sustained clock is ~2.0 PFLOP/s; inside the full-grid K loops "
                                "(profiles/r04_nt_ktile_cycles.txt) 1.6-1.7 GHz, where the K loop holds 81 % of the matrix rate that clock allows"),
                    algorithmic_gflop_per_launch=round(fl[dom] / max(1, n[dom]) / 1e9, 2),
                    all_gemm_kernels={fam[i]: dict(ms_per_step=round(ms[i] / max(1, args.profile_steps), 3),
                                                   tflops=round(fl[i] / (ms[i] * 1e-3) / 1e12, 1) if ms[i] > 0 else 0.0,
                                                   launches_per_step=int(n[i] // max(1, args.profile_steps))) for i in range(4)})
        # attention kernels (families 4 / 5): FLOPs of the rows the stream runs (== the padded S in variant A)
        s2_exe = sum(sum((min(Lv, x + 3) + y) ** 2 for x, y in zip(a, b)) for a, b in lens_a) / len(lens_a) if halo else B * (Lv + Lt) ** 2
        roof_attn = {}
        for name, i, mult in (("forward", 4, 4.0), ("backward", 5, 10.0)):
            t_ms = (ms[i] + floor * n[i]) / max(1, args.profile_steps)
            exe = mult * MODEL["E"] * s2_exe * MODEL["d"]
            roof_attn[name] = dict(ms_per_step=round(t_ms, 3), launches_per_step=int(n[i] // max(1, args.profile_steps)),
                                   achieved=round(exe / max(t_ms * 1e-3, 1e-12) / 1e12, 1), frac=round(exe / max(t_ms * 1e-3, 1e-12) / 2.5e15, 4))
        roof_attn["note"] = ("attention kernels only (HIP event pairs around launch_attn_fwd / launch_attn_bwd incl. the delta pass): achieved = "
                             "4 (fwd) / 10 (bwd) * E * sum_b S_b^2 * d FLOPs of the rows the stream EXECUTES over the measured time, peak 2.5 PFLOP/s")
        # HBM-bound kernels: LayerNorm launches (bytes counted by the library) and the attention kernels (bytes of the rows the stream runs)
        rows_exe = (sum(sum(min(Lv, x + 3) + y for x, y in zip(a, b)) for a, b in lens_a) / len(lens_a)) if halo else B * (Lv + Lt)
        roof_hbm = {}
        for name, i in (("layernorm_forward", 6), ("layernorm_backward", 7)):
            t_ms = (ms[i] + floor * n[i]) / max(1, args.profile_steps)
            gb = fl[i] / max(1, args.profile_steps) / 1e9
            roof_hbm[name] = dict(ms_per_step=round(t_ms, 3), launches_per_step=int(n[i] // max(1, args.profile_steps)), gbytes_per_step=round(gb, 3),
                                  achieved=round(gb / max(t_ms * 1e-3, 1e-12), 0), peak=8000.0, unit="GB/s", frac=round(gb / max(t_ms * 1e-3, 1e-12) / 8000.0, 3))
        for name, i, streams in (("attention_forward", 4, 4), ("attention_backward", 5, 8)):
            t_ms = (ms[i] + floor * n[i]) / max(1, args.profile_steps)
            gb = MODEL["E"] * streams * rows_exe * MODEL["d"] * 2 / 1e9
            roof_hbm[name] = dict(ms_per_step=round(t_ms, 3), gbytes_per_step=round(gb, 3), achieved=round(gb / max(t_ms * 1e-3, 1e-12), 0), peak=8000.0,
                                  unit="GB/s", frac=round(gb / max(t_ms * 1e-3, 1e-12) / 8000.0, 3))
        roof_hbm["note"] = ("HBM-bound kernels against the 8 TB/s peak: algorithmic bytes per step (LayerNorm: input + every output row stream of every "
                            "launch, incl. the feature LayerNorms and their parameter-gradient reduce passes in the time; attention: q,k,v + o forward, "
                            "q,k,v,dO,O + dq,dk,dv backward, bf16, executed rows) over the event-pair time of the launches")
    # ---- section timing: encoder forward / backward (roofline.achieved = encoder fwd+bwd FLOPs / t_encoder / peak) ----
    if rank == 0:
        lib.uvtg_profile_sections_start()
    for i in range(args.profile_steps):
        step.step(*batches[i % 2])
    if rank == 0:
        sm, sn = (C.c_double * 4)(), (C.c_longlong * 4)()
        _lib.check(lib.uvtg_profile_sections_stop(sm, sn), "uvtg_profile_sections_stop")
        k = max(1, args.profile_steps)
        sect = dict(encoder_fwd_ms=round(sm[0] / k, 3), encoder_bwd_ms=round(sm[1] / k, 3), forward_ms=round(sm[2] / k, 3),
                    backward_ms=round(sm[3] / k, 3))
    if world > 1:
        torch.distributed.barrier()

    # every other BASELINE config that fits one GPU, and the inference call, in the SAME driver-run line.  LAST on the
    # device: each builds (and frees) its own model + multi-GB workspace, and the headline's instrumented passes above must not run behind
    # that allocator traffic (visit r5a: the LayerNorm-forward event pairs of the pass that followed it read 10 ms per step)
    if rank == 0 and world == 1 and not args.no_companions and args.config == 2:
        comp["with_input_pipeline"] = companion_input_pipeline(dev, step, batches)
        comp["drop_in_autograd"] = companion_drop_in(dev, batches, precise, Lv)
    if rank == 0 and world == 1 and not args.no_companions and args.config == 2 and not args.no_other_configs:
        for cid in (3, 4, 5):
            comp[f"config{cid}"] = companion_config(cid, dev, precise, lib)
        comp["inference"] = companion_infer(dev)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config == 2:
        cpu = cpu_baseline(batches[0], dev)

    if rank == 0:
        valid_clips = sum(sum(a) for a, _ in lens_a) / len(lens_a)
        # `value` counts clips the step EXECUTES: every position in variant A (nothing is padded); the VALID clips in variant B (the packed stream
        # does not run the padded positions, and counting them was round 3's non-creditable headline)
        counted = (B * Lv) if full else valid_clips
        alg_flops, exe_flops = encoder_flops(lens_a, B, Lv, Lt, halo)
        t_enc = max((sect["encoder_fwd_ms"] + sect["encoder_bwd_ms"]) * 1e-3, 1e-9)
        desc = (f"{wl['what']}: L_v={Lv} L_t={Lt} D_v=2818 D_t=512 d=1024 F=1024 H=8 E=4, full train step (fwd+criterion+bwd+clip+AdamW), dropout 0.5/0/0.1, "
                + ("all-ones masks (SURVEY 8d variant A): every clip position and every text token is executed" if full else
                   f"ragged valid lengths ({wl['lens']}, SURVEY 8d variant B) on the packed loss-only stream: valid clips + the 3 padded clips per sample inside "
                   "the conv heads' receptive field + valid text tokens (losses and all gradients exactly the padded execution's)" if packed else
                   f"ragged valid lengths ({wl['lens']}), padded execution")
                + ("; input projections of the forward on fp32-class split operands (saliency_scores within 1e-4 of the fp32 reference)" if precise else
                   "; input projections on plain bf16 operands (saliency_scores within 3e-2)"))
        out = dict(metric="clips/sec (L=75,d=1024) fwd+bwd" if args.config == 2 else f"clips/sec fwd+bwd (BASELINE config {args.config})",
                   value=round(counted * world * args.steps / elapsed, 1), unit="clips/s", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=round(elapsed / args.steps * 1e3, 3), higher_is_better=True,
                   scaling="weak", vs_baseline=None, dtype="bf16", data="synthetic",
                   config=dict(workload=desc, baseline_config=args.config, variant=args.variant, per_gpu_batch=B, global_batch=B * world, parallelism=f"dp{world}",
                               projections=args.proj),
                   value_counts=("B*L_v clip positions per step, ALL executed (all-ones masks)" if full else
                                 "VALID clips per step (padded positions are neither executed nor counted); clip_positions_per_sec_incl_padded is the "
                                 "reference's own count"),
                   clip_positions_per_sec_incl_padded=round(B * Lv * world * args.steps / elapsed, 1),
                   samples_per_sec=round(B * world * args.steps / elapsed, 1),
                   ms_per_step_event_median=round(per_med[len(per_med) // 2], 3), ms_per_step_event_min=round(per_med[0], 3), median_over_steps=len(per_med),
                   grad_comm_dtype=(args.grad_comm_dtype if world > 1 else None), gemm_cus=getattr(step, "gemm_cus", None), comm_cus=(args.comm_cus if world > 1 else None),
                   exposed_comm_ms_per_step=(round(exposed[len(exposed) // 2], 3) if exposed else None), exposed_comm_samples=len(exposed),
                   world_size=world, comm_backend=(backend if world > 1 else None), comm_world_size=comm_size, replicas=replicas,
                   t_encoder_ms=round(t_enc * 1e3, 3), sections=sect,
                   roofline_encoder=encoder_roofline(alg_flops, exe_flops, t_enc, B, Lv, halo, lens_a, roof),
                   encoder_rows_fraction=round(sum(sum(min(Lv, x + 3) for x in a) + sum(b) for a, b in lens_a) / (len(lens_a) * B * (Lv + Lt)), 4) if halo else 1.0,
                   companions=comp or None,
                   numerics=("forward input projections on fp32-class split operands: saliency_scores of the timed train-mode step within 1e-4 of the fp32 "
                             "oracle for the same dropout masks (tests/test_gpu_parity_full.py::test_bench_path_trainstep_dropout_replayed_through_oracle); "
                             "everything else bf16 operands / fp32 accumulation" if precise else
                             "train-mode calls run the input projections on plain bf16 operands (saliency_scores within 3e-2 of fp32)"),
                   losses=[round(x, 5) for x in losses], roofline=roof, roofline_attention=roof_attn, roofline_hbm=roof_hbm, cpu_baseline=cpu)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
