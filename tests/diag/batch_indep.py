"""eval-mode batch independence at BASELINE config 2: rows of a 32-sample slice vs the same rows of the 256-sample call (split-K of the small launches
sums K in another order than the unsplit large ones)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from oracle import univtg_oracle as O
from tests.test_gpu_model import build, to_dev
dev = torch.device("cuda:0")
cfg = O.make_cfg(input_dropout=0.0, droppath=0.0, dropout=0.0)
params = O.init_params(cfg, seed=1)
inputs, tg = O.make_batch(cfg, 256, 75, 32, seed=2, ragged=True)
ind = to_dev(inputs, dev)
for prec in ("bf16", "fp32x3"):
    model, crit = build(cfg, params, dev, prec)
    model.eval()
    with torch.no_grad():
        out = model(**ind)
        for n in (32, 1):
            sub = model(**{k: v[:n] for k, v in ind.items()})
            print(prec, n, {k: f"{float((out[k][:n] - sub[k]).abs().max()):.2e}" for k in ("pred_logits", "pred_spans", "saliency_scores")})
