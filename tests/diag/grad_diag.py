"""Diagnostic (test infrastructure, uses the oracle; not collected by pytest): per-parameter gradient agreement (HIP bf16 path vs CPU oracle autograd): max-rel error, cosine, norm ratio."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import univtg_oracle as O
from tests.test_gpu_model import args_from_cfg, to_dev
from univtg_amd.model import build_model

def run(cfg, B, Lv, Lt, seed, tag):
    dev = torch.device("cuda:0")
    params = O.init_params(cfg, seed=seed)
    inputs, tg = O.make_batch(cfg, B, Lv, Lt, seed=seed + 1, ragged=True)
    p2 = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = O.forward(p2, cfg, **inputs)
    lref = O.criterion(ref, tg, cfg)
    O.total_loss(lref, cfg).backward()
    model, crit = build_model(args_from_cfg(cfg, precision="bf16"))
    model.load_state_dict(params); model.to(dev).eval(); crit.to(dev)
    out = model(**to_dev(inputs, dev))
    ld = crit(out, to_dev(tg, dev))
    sum(ld[k] * crit.weight_dict[k] for k in ld).backward()
    print(f"=== {tag}: losses " + " ".join(f"{k}={float(ld[k]):.5f}/{float(lref[k]):.5f}" for k in ld))
    for k, p in model.named_parameters():
        if p.grad is None: continue
        g, r = p.grad.cpu().double().flatten(), p2[k].grad.double().flatten()
        cos = float((g @ r) / (g.norm() * r.norm() + 1e-30))
        print(f"{k:58s} maxrel {float((g-r).abs().max()/(r.abs().max()+1e-30)):.3e} cos {cos:.5f} norm {float(g.norm()/(r.norm()+1e-30)):.4f} |r| {float(r.norm()):.2e}")

tiny = O.make_cfg(hidden_dim=64, nheads=2, dim_feedforward=96, enc_layers=2, v_feat_dim=34, t_feat_dim=24, max_q_l=16, input_dropout=0.0, dropout=0.0, droppath=0.0)
run(tiny, 5, 13, 7, 11, "tiny")
mid = O.make_cfg(hidden_dim=256, nheads=4, dim_feedforward=256, enc_layers=2, v_feat_dim=514, t_feat_dim=512, input_dropout=0.0, dropout=0.0, droppath=0.0)
run(mid, 8, 40, 12, 3, "mid")
big = O.make_cfg(enc_layers=2, input_dropout=0.0, dropout=0.0, droppath=0.0)
run(big, 4, 75, 32, 5, "big d=1024")
