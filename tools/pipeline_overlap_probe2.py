"""Dev probe (round 6): why does the fp32-wire input pipeline cost the step 13 % when its 5.4 ms upload fits under the 8.6 ms step?
The headline TrainStep loops on resident batches while a SIDE stream carries, per step, (a) nothing, (b) the H2D copy of one batch's feature
block out of pinned memory (234 MB fp32 / 117 MB bf16), (c) only a device-side pass of the padding kernel's size (uvtg_ragged_to_padded over
a resident packed block: 234 MB in, 234 MB out), (d) both.  ms per step, host wall clock, 30 steps."""
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
side = torch.cuda.Stream()
def run(name, wire, do_copy, do_kernel, n=30):
    host = torch.randn(B * Lv, 2818).to(wire).pin_memory()
    dpacked = host.to(dev)
    offs = torch.arange(0, B * Lv + 1, Lv, dtype=torch.int32, device=dev)
    out = torch.empty(B, Lv, 2818, device=dev); mask = torch.empty(B, Lv, device=dev)
    for i in range(3): step.step(*batches[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        with torch.cuda.stream(side):
            src = host.to(dev, non_blocking=True) if do_copy else dpacked
            if do_kernel:
                _lib.check(lib.uvtg_ragged_to_padded(_ptr(src), int(wire == torch.bfloat16), _ptr(offs), B, Lv, 2818, _ptr(out), _ptr(mask),
                                                     torch.cuda.current_stream().cuda_stream), "r2p")
        step.step(*batches[i % 2])
    torch.cuda.synchronize()
    print(f"{name:56s} {(time.perf_counter() - t0) / n * 1e3:7.3f} ms per step")
for rnd in range(2):
    run("resident (nothing on the side stream)", torch.float32, False, False)
    for wire, tag in ((torch.float32, "fp32 234 MB"), (torch.bfloat16, "bf16 117 MB")):
        run(f"H2D copy only ({tag})", wire, True, False)
        run(f"padding kernel only ({tag} in)", wire, False, True)
        run(f"H2D copy + padding kernel ({tag})", wire, True, True)
