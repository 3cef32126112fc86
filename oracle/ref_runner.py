"""TEST / MEASUREMENT INFRASTRUCTURE ONLY -- runs THE REFERENCE (showlab/UniVTG, imported unmodified from oracle/_ref/uvtg_reference_model.zip,
see oracle/build_ref.py) in a CHILD process, on the host cores.  Only tests/ and bench.py's `cpu_baseline` leg start it; nothing under
univtg_amd/ knows it exists.  A child process because the archive provides the generic top-level packages `model` / `utils` / `eval`:
here nothing else can shadow them, and `import_ref` refuses modules that did not come out of the archive.

    python oracle/ref_runner.py job.json          # prints ONE JSON line (the result) as its last stdout line

job = {"task": "model" | "postproc", "threads": n, ...}

task "model"   -- reference `build_model(args)` (model/univtg.py:409-450) with the oracle-seeded weights (`param_seed`, loaded through
                  `load_state_dict(strict=True)`), on a batch given either by seeds ({"batch": {B, Lv, Lt, seed, ragged}}: the oracle's
                  deterministic input generator, the one the GPU tests use) or as a file ({"batch_npz": path} with in/<key>, tg/<key>).
                  * "train_steps": k > 0 -> 1 warm-up + k timed TRAIN-mode steps, Model.forward + SetCriterion.forward + backward
                    (model/univtg.py:105-155,195-351; the loop body of main/train_vlp_ddp.py:56-62), fp32
                  * "eval_out": path  -> one EVAL-mode forward under no_grad, outputs written as .npz (pred_logits, pred_spans, saliency_scores)
task "postproc"-- the reference's inference tail on model outputs given as .npz (pred_logits, pred_spans, timestamp, timestamp_mask, durations):
                  compose_predictions() restates the ten glue lines of compute_mr_results (main/inference_mr.py:111-163 -- that file itself
                  imports nncore / h5py, absent from the image), then the REFERENCE's own PostProcessorDETR(..., ["round_multiple"])
                  (eval/postprocessing.py:9-51, as constructed at main/inference_mr.py:184-192) and the REFERENCE's own temporal_nms
                  (utils/temporal_nms.py:25-74, as called by post_processing_mr_nms, main/inference_mr.py:31-40).
"""
from __future__ import annotations

import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _load_batch(job, O, cfg):
    import numpy as np
    import torch
    if job.get("batch_npz"):
        z = np.load(job["batch_npz"])
        grab = lambda pre: {k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)}
        return grab("in/"), grab("tg/")
    b = job["batch"]
    return O.make_batch(cfg, b["B"], b["Lv"], b["Lt"], seed=b["seed"], ragged=b.get("ragged", True))


def task_model(job):
    import numpy as np
    import torch
    from oracle import univtg_oracle as O
    from oracle.build_ref import import_ref
    from oracle.make_golden import ref_args
    (ref,), manifest = import_ref(("model.univtg",))
    cfg = O.make_cfg(**job.get("cfg", {}))
    params = O.init_params(cfg, seed=job.get("param_seed", 0))
    inputs, tg = _load_batch(job, O, cfg)
    model, crit = ref.build_model(ref_args(cfg))
    model.load_state_dict({k: v.clone() for k, v in params.items()}, strict=True)
    B, Lv = inputs["src_vid"].shape[:2]
    res = dict(manifest=manifest["members"], module_file=ref.__file__, B=B, L_v=Lv, threads=torch.get_num_threads(), torch=torch.__version__,
               host_cpus=os.cpu_count())
    k = int(job.get("train_steps", 0))
    if k > 0:
        model.train(); crit.train()
        steps, fwd = [], []
        for i in range(k + 1):                                   # step 0 = warm-up
            model.zero_grad(set_to_none=True)
            t0 = time.perf_counter()
            out = model(**inputs)
            t1 = time.perf_counter()
            ld = crit(out, tg)
            sum(ld[n] * crit.weight_dict[n] for n in ld if n in crit.weight_dict).backward()
            t2 = time.perf_counter()
            if i:
                steps.append(t2 - t0); fwd.append(t1 - t0)
        res.update(train_step_s=steps, train_forward_s=fwd)
    if job.get("eval_out"):
        model.eval(); crit.eval()
        t0 = time.perf_counter()
        with torch.no_grad():
            out = model(**inputs)
        res["eval_forward_s"] = time.perf_counter() - t0
        np.savez(job["eval_out"], **{n: out[n].detach().numpy() for n in ("pred_logits", "pred_spans", "saliency_scores")})
    return res


def compose_predictions(pred_logits, pred_spans, timestamp, timestamp_mask, durations):
    """main/inference_mr.py:111-163 for span_loss_type 'l1', a dense-regression model_id, sorted results: per sample the ranked rows
    [st, ed, score], every number through float(f"{e:.4f}").  Also returns the clip index behind every ranked row (the same stable sort
    applied to the row numbers) -- the "span indices" of north_star's index clause."""
    import torch
    prob = pred_logits.clone()
    scores = prob[..., 0]
    spans_all = timestamp + pred_spans
    scores[~timestamp_mask.bool()] = 0
    res, orders = [], []
    for spans, score, dur in zip(spans_all, scores, durations):
        spans = torch.clamp(spans * dur, 0, dur)
        rows = torch.cat([spans, score[:, None]], dim=1).tolist()
        orders.append(sorted(range(len(rows)), key=lambda i: rows[i][2], reverse=True))
        rows = sorted(rows, key=lambda x: x[2], reverse=True)
        res.append([[float(f"{e:.4f}") for e in row] for row in rows])
    return res, orders


def _rank_positions(pre, nms):
    """post-NMS rows -> their positions in the ranked (pre-NMS) list (first unused match; identical rows are interchangeable)"""
    used, out = set(), []
    for r in nms:
        idx = next(i for i, rr in enumerate(pre) if rr == r and i not in used)
        used.add(idx)
        out.append(idx)
    return out


def task_postproc(job):
    import numpy as np
    import torch
    from oracle.build_ref import import_ref
    (nms_mod, pp_mod), manifest = import_ref(("utils.temporal_nms", "eval.postprocessing"))
    z = np.load(job["outputs_npz"])
    t = lambda n: torch.from_numpy(z[n])
    durations = [float(x) for x in z["durations"]]
    out = {}
    ranked, orders = compose_predictions(t("pred_logits"), t("pred_spans"), t("timestamp"), t("timestamp_mask"), durations)
    for clip_length in job.get("clip_lengths", [0.0, 2.0]):
        lines = [dict(pred_relevant_windows=[list(r) for r in rows]) for rows in ranked]
        if clip_length > 0:                                      # opt.round_multiple > 0 (main/inference_mr.py:190-192)
            pp = pp_mod.PostProcessorDETR(clip_length=clip_length, min_ts_val=0, max_ts_val=150, min_w_l=2, max_w_l=150, move_window_method="left",
                                          process_func_names=["round_multiple"])
            lines = pp(lines)
        pre = [e["pred_relevant_windows"] for e in lines]
        nms = [nms_mod.temporal_nms(e[:job.get("max_before_nms", 1000)], nms_thd=job.get("nms_thd", 0.7), max_after_nms=job.get("max_after_nms", 10))
               for e in pre]
        out[str(clip_length)] = dict(pre=pre, nms=nms, order=orders, keep=[_rank_positions(p, n) for p, n in zip(pre, nms)])
    with open(job["result_json"], "w") as f:
        json.dump(out, f)
    return dict(manifest=manifest["members"], module_files=[nms_mod.__file__, pp_mod.__file__], samples=len(ranked))


def main():
    with open(sys.argv[1]) as f:
        job = json.load(f)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import torch
    if job.get("threads"):
        torch.set_num_threads(int(job["threads"]))
    res = dict(task_model=task_model, task_postproc=task_postproc)["task_" + job["task"]](job)
    print(json.dumps(res))


def run_job(job, timeout=1800):
    """Parent-side helper: start the child, return its result dict (raises RuntimeError with the child's stderr tail on failure)."""
    import subprocess
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump(job, f)
        path = f.name
    try:
        env = dict(os.environ, PYTHONPATH="")                    # nothing but the child's own sys.path entries
        r = subprocess.run([sys.executable, os.path.abspath(__file__), path], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
        if r.returncode != 0:
            raise RuntimeError("oracle/ref_runner.py failed:\n" + r.stderr[-3000:])
        return json.loads(r.stdout.strip().splitlines()[-1])
    finally:
        os.unlink(path)


if __name__ == "__main__":
    main()
