"""Dev probe (round 6): what would running the per-step weight re-layout (uvtg_prepare_weights: transposes / splits / conv re-layouts, ~100 us, HBM-bound)
BESIDE the head of the forward (seq_prep + the VALU-bound wide feature LayerNorm, ~145 us) buy?  The headline TrainStep with uvtg_prepare_weights
(a) as shipped (on the step's stream), (b) skipped after the first step (lower bound: the work removed; the weights stay the first step's -- timing only),
(c) issued on a SIDE stream into a second operand cache right when the step starts and joined at the end of the step (what an in-forward fork / join
could reach at best, incl. the two cross-stream hand-overs).  ms per step, host wall clock, 40 steps, three alternating rounds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from univtg_amd import _lib
from univtg_amd.model import build_model, _ptr
from univtg_amd.trainer import TrainStep
lib = _lib.load()
dev = torch.device("cuda:0")
B, Lv, Lt = 256, 75, 32
torch.manual_seed(2018)
model, crit = build_model(bench.model_args(max_v_l=Lv, proj_precise=True))
model.to(dev).train(); crit.to(dev).train(); model.set_seed(2018)
step = TrainStep(model, crit, lr=1e-4, weight_decay=1e-4, grad_clip=0.1, packed="auto")
batches = [bench.synth_batch(B, Lv, Lt, 2818, 512, i, dev, None, full=True) for i in range(2)]
for i in range(3): step.step(*batches[i % 2])
torch.cuda.synchronize()
real = lib.uvtg_prepare_weights
side = torch.cuda.Stream()
wc2 = torch.empty_like(step.wcache)
mode = {"m": "a"}
def patched(dims, ptrs, wcache, st):
    if mode["m"] == "a":
        return real(dims, ptrs, wcache, st)
    if mode["m"] == "b":
        return 0
    side.wait_stream(torch.cuda.current_stream())
    return real(dims, ptrs, _ptr(wc2), side.cuda_stream)
lib.uvtg_prepare_weights = patched
def run(m, n=40):
    mode["m"] = m
    for i in range(3):
        step.step(*batches[i % 2]); torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        step.step(*batches[i % 2])
        if m == "c": torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
names = {"a": "(a) as shipped", "b": "(b) re-layout skipped (lower bound)", "c": "(c) re-layout on a side stream beside the step's head"}
for rnd in range(3):
    for m in "abc":
        print(f"{names[m]:60s} {run(m):7.3f} ms per step", flush=True)
