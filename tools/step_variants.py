"""Dev tool: whole-step time (bench workload, packed stream) with the NT GEMM structure forced, for calibrating the launch
cost model in gemm.hip against the real epilogues rather than the plain kernel-level entry point."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from univtg_amd import _lib
from univtg_amd.model import build_model
from univtg_amd.trainer import TrainStep

dev = torch.device("cuda:0")
torch.manual_seed(2018)
model, crit = build_model(bench.model_args())
model.to(dev).train(); crit.to(dev).train(); model.set_seed(2018)
step = TrainStep(model, crit, lr=1e-4, weight_decay=1e-4, grad_clip=0.1)
W = bench.WORKLOAD
batches = [bench.synth_batch(W["B"], W["L_v"], W["L_t"], W["D_v"], W["D_t"], i, dev) for i in range(2)]
lib = _lib.load()

def run(n=20):
    for i in range(3): step.step(*batches[i % 2])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): step.step(*batches[i % 2])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

settings = [("auto", 0, 0), ("wn4 auto-bm", 4, 0), ("wn4 bm256", 4, 256), ("wn4 bm192", 4, 192), ("wn4 bm128", 4, 128),
            ("wn2 auto-bm", 2, 0), ("wn2 bm192", 2, 192), ("wn2 bm128", 2, 128), ("auto", 0, 0)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    settings = [("auto", 0, 0)] * 3
for name, wn, bm in settings:
    _lib.check(lib.uvtg_debug_force_nt_wn(wn)); _lib.check(lib.uvtg_debug_force_nt_bm(bm))
    print(f"{name:14s}: {run():7.3f} ms/step", flush=True)
lib.uvtg_debug_force_nt_wn(0); lib.uvtg_debug_force_nt_bm(0)
