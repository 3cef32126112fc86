"""Dev tool: socket power and clocks (rocm-smi samples from a side thread) while the config-2 train step loops for ~12 s; then the same for an idle chip."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from univtg_amd.model import build_model
from univtg_amd.trainer import TrainStep

def sample(tag):
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showuse"], capture_output=True, text=True).stdout
    keep = [l.split(":", 1)[1].strip() if ":" in l else l for l in out.splitlines() if any(k in l for k in ("Power (W)", "sclk", "mclk", "fclk", "GPU use"))]
    print(f"{tag}: " + " | ".join(keep), flush=True)

wl = bench.CONFIGS[2]
dev = torch.device("cuda:0")
torch.manual_seed(2018)
model, crit = build_model(bench.model_args(max_v_l=wl["L_v"]))
model.to(dev).train(); crit.to(dev).train(); model.set_seed(2018)
step = TrainStep(model, crit, lr=1e-4, weight_decay=1e-4, grad_clip=0.1, packed="auto")
batch = bench.synth_batch(wl["B"], wl["L_v"], wl["L_t"], bench.MODEL["D_v"], bench.MODEL["D_t"], 0, dev, None, full=True)
for _ in range(10):
    step.step(*batch)
torch.cuda.synchronize()
sample("idle before")
stop = False
def sampler():
    i = 0
    while not stop:
        time.sleep(1.0); sample(f"train step looping, t = {i + 1:2d} s"); i += 1
th = threading.Thread(target=sampler); th.start()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 12.0:
    for _ in range(20):
        step.step(*batch)
    torch.cuda.synchronize(); n += 20
el = time.perf_counter() - t0
stop = True; th.join()
print(f"{n} steps in {el:.2f} s = {el / n * 1e3:.3f} ms per step")
time.sleep(2.0); sample("idle after")
