import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import univtg_oracle as O
from tests.test_gpu_model import build, to_dev
from univtg_amd.trainer import TrainStep
dev = torch.device("cuda:0")
B, Lv, Lt = 64, 75, 32
cfg = O.make_cfg(input_dropout=0.0, dropout=0.0, droppath=0.0)
params = O.init_params(cfg, seed=41)
inputs, tg = O.make_batch(cfg, B, Lv, Lt, seed=42, ragged=True)
ind, tgd = to_dev(inputs, dev), to_dev(tg, dev)
lens = (inputs["src_vid_mask"].sum(1).int().tolist(), inputs["src_txt_mask"].sum(1).int().tolist())
m32, _ = build(cfg, params, dev, "fp32x3"); m32.eval()
with torch.no_grad():
    ref = m32(**ind)
valid = inputs["src_vid_mask"].bool().to(dev)
res = {}
for mode in (False, True):
    model, crit = build(cfg, params, dev, "bf16"); model.eval()
    step = TrainStep(model, crit, packed=mode)
    batch = dict(ind)
    if mode: batch["_lens_host"] = lens
    step.step(batch, tgd, optimize=False); torch.cuda.synchronize()
    res[mode] = (step.pred_logits.clone()[..., 0], step.pred_spans.clone(), step.grads.clone())
for name, i in (("logits", 0), ("spans", 1)):
    a, b = res[False][i], res[True][i]
    r = ref["pred_logits"][..., 0] if i == 0 else ref["pred_spans"]
    v = valid if i == 0 else valid[..., None].expand_as(a)
    print(f"{name}: |padded-packed| valid {float((a-b)[v].abs().max()):.2e} padded-pos {float((a-b)[~v].abs().max()):.2e} | err vs fp32x3: padded {float((a-r).abs().max()):.2e} packed {float((b-r).abs().max()):.2e}")
g0, g1 = res[False][2].double(), res[True][2].double()
print("grad cos", float((g0@g1)/(g0.norm()*g1.norm())), "ratio", float(g1.norm()/g0.norm()))
