"""Dev experiment: does a persistent-GEMM step slow down when some CUs are held by another stream, and does the dynamic tile
hand-out (uvtg_set_dynamic_tiles) fix it?  A side stream keeps `blocks` CUs busy for `usec` in every step (as RCCL would)."""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from univtg_amd import _lib
from univtg_amd.model import build_model
from univtg_amd.trainer import TrainStep
dev = torch.device("cuda:0")
hog = C.CDLL(os.path.join(ROOT, "tools", "libcuhog.so"))
hog.cu_hog.argtypes = [C.c_int, C.c_int, C.c_void_p]
torch.manual_seed(2018)
model, crit = build_model(bench.model_args()); model.to(dev).train(); crit.to(dev).train(); model.set_seed(1)
step = TrainStep(model, crit)
W = bench.WORKLOAD
batches = [bench.synth_batch(W["B"], W["L_v"], W["L_t"], W["D_v"], W["D_t"], i, dev) for i in range(2)]
lib = _lib.load()
side = torch.cuda.Stream()
def run(blocks, usec, n=15):
    for i in range(3): step.step(*batches[i % 2])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        if blocks: hog.cu_hog(blocks, usec, C.c_void_p(side.cuda_stream))
        step.step(*batches[i % 2])
        if blocks: torch.cuda.current_stream().wait_stream(side)     # like the optimizer waiting for the exchange
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for dyn in (0, 1, 0, 1):
    lib.uvtg_set_dynamic_tiles(dyn)
    print(f"dynamic={dyn}: free {run(0, 0):.3f} | 16 CUs x 4 ms {run(16, 4000):.3f} | 32 CUs x 4 ms {run(32, 4000):.3f} | 32 CUs x 8 ms {run(32, 8000):.3f} ms/step", flush=True)
lib.uvtg_set_dynamic_tiles(0)
