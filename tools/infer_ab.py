"""eager inference call timing (default precision and bf16) at batch 1 / 32 / 64 (INFER_AB_BATCHES, INFER_AB_PRECS), for in-box A/Bs of launch heuristics (env switches)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from univtg_amd import ops
from univtg_amd.model import build_model
dev = torch.device("cuda:0")
res = []
BATCHES = [int(b) for b in os.environ.get("INFER_AB_BATCHES", "1,32,64").split(",")]
PRECS = os.environ.get("INFER_AB_PRECS", "auto,bf16").split(",")
for B, Dv in [(b, 514 if b == 1 else 2818) for b in BATCHES]:
    for prec in PRECS:
        torch.manual_seed(2018)
        model, _ = build_model(bench.model_args(max_v_l=75, v_feat_dim=Dv, precision=prec))
        model.to(dev).eval()
        batches = [bench.infer_batch(B, 75, 32, Dv, 512, 50 + i, dev) for i in range(2)]
        def call(i):
            inp, ts, tm, dur = batches[i % 2]
            with torch.no_grad():
                out = model(**inp)
                ops.postprocess_mr(out["pred_logits"], out["pred_spans"], out["saliency_scores"], ts, tm, dur, clip_length=2.0, eval_mode="add")
        for i in range(5):
            call(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(40):
            call(i)
        torch.cuda.synchronize()
        res.append(f"B={B} {prec}: {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms")
print(" | ".join(res))
