"""diagnosis: per-parameter gradient agreement of the tiny fixtures (bf16 training arithmetic vs the reference's fp32 gradients)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import test_gpu_model as T
dev = torch.device("cuda:0")
gd = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "golden")
for name in ("tiny_eval_ragged", "tiny_hl", "tiny_zero_saliency"):
    meta, cfg, params, inputs, tg, out_ref, eval_ref, grads_ref, losses_ref = T.load_case(gd, name)
    for pp in (True,):
        model, crit = T.build(cfg, params, dev, "bf16", proj_precise=pp)
        model.eval()
        out = model(**T.to_dev(inputs, dev))
        losses = crit(out, T.to_dev(tg, dev))
        sum(losses[k] * crit.weight_dict[k] for k in losses).backward()
        named = dict(model.named_parameters())
        print(name, {k: (round(float(losses[k]), 5), round(losses_ref[k], 5)) for k in losses})
        for k in ("pred_logits", "pred_spans", "saliency_scores"):
            print("   out", k, float((out[k].detach().cpu() - out_ref[k]).abs().max()))
        for k, g in grads_ref.items():
            a, r = named[k].grad.cpu().double().flatten(), g.double().flatten()
            cos = float((a @ r) / (a.norm() * r.norm() + 1e-30)); ratio = float(a.norm() / (r.norm() + 1e-30))
            flag = " <<<" if (cos < 0.985 or abs(ratio - 1) > 0.05) else ""
            print(f"   {k:60s} cos {cos:.4f} ratio {ratio:.4f} |ref| {float(r.norm()):.3e}{flag}")

# the same question at production width against the ORACLE (fp32 autograd): is the hl-subset shrink toy-width noise?
from oracle import univtg_oracle as O
for losses_sel, tag in ((("labels", "saliency"), "hl"), (("spans", "labels", "saliency"), "vlp")):
    cfg = O.make_cfg(input_dropout=0.0, droppath=0.0, dropout=0.0, losses=losses_sel)
    params = O.init_params(cfg, seed=51)
    inputs, tg = O.make_batch(cfg, 16, 75, 32, seed=52, ragged=True, curve=True)
    p2 = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = O.forward(p2, cfg, **inputs)
    O.total_loss(O.criterion(ref, tg, cfg), cfg).backward()
    model, crit = T.build(cfg, params, dev, "bf16", proj_precise=True)
    model.eval()
    out = model(**T.to_dev(inputs, dev))
    losses = crit(out, T.to_dev(tg, dev))
    sum(losses[k] * crit.weight_dict[k] for k in losses).backward()
    named = dict(model.named_parameters())
    worst = []
    for k, p in p2.items():
        if p.grad is None or named[k].grad is None:
            continue
        a, r = named[k].grad.cpu().double().flatten(), p.grad.double().flatten()
        if float(r.norm()) == 0:
            continue
        worst.append((float(a.norm() / r.norm()), float((a @ r) / (a.norm() * r.norm() + 1e-30)), k))
    worst.sort()
    print(f"[production width, {tag}] norm ratios: min {worst[0]}, max {worst[-1]}, median {worst[len(worst) // 2][0]:.4f}; min cosine {min(w[1] for w in worst):.5f}")
