"""Dev check: BASELINE config 4 shape (Ego4D-NLQ long video: B=32, L_v=1200, L_t=32 -> S=1232) runs fwd+bwd+step on one GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from univtg_amd.model import build_model
from univtg_amd.trainer import TrainStep
dev = torch.device("cuda:0")
for (B, Lv, Lt, name) in [(32, 1200, 32, "config 4"), (256, 128, 32, "config 3 per-GPU")]:
    torch.manual_seed(0)
    model, crit = build_model(bench.model_args(max_v_l=Lv))
    model.to(dev).train(); crit.to(dev).train()
    step = TrainStep(model, crit)
    batch = bench.synth_batch(B, Lv, Lt, 2818, 512, 7, dev)
    for _ in range(2): step.step(*batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): step.step(*batch)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
    print(f"{name}: B={B} L_v={Lv}: {t*1e3:.2f} ms/step, {B*Lv/t/1e6:.3f} M clips/s, losses {[round(x,4) for x in step.losses[:5].tolist()]}, ws {step.ws.numel()/2**30:.2f} GiB")
    del step, model, crit, batch
    torch.cuda.empty_cache()
