"""profile target: the default (precise) inference call at the reference's eval batch 32 -- run under tools/prof.sh"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from univtg_amd import ops
from univtg_amd.model import build_model
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(2018)
model, _ = build_model(bench.model_args(max_v_l=75))
model.to(dev).eval()
batches = [bench.infer_batch(B, 75, 32, 2818, 512, 50 + i, dev) for i in range(2)]
for i in range(40):
    inp, ts, tm, dur = batches[i % 2]
    with torch.no_grad():
        out = model(**inp)
        ops.postprocess_mr(out["pred_logits"], out["pred_spans"], out["saliency_scores"], ts, tm, dur, clip_length=2.0, eval_mode="add")
torch.cuda.synchronize()
