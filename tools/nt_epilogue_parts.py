"""Dev tool (round 6): where does the exposed epilogue time of the persistent NT GEMM go?  Runs the headline training step on
tools/libuvtg_nostore.so (tools/build_nostore.py: a patched copy of gemm.hip with a run-time epilogue mode) and reads the NT family's
event-pair time per step with (0) the full epilogue, (1) no output stores, (2) no epilogue-operand loads, (3) neither -- the arithmetic, the LDS
transpose and the instruction stream of the epilogue run in every mode.  Results of modes 1-3 are garbage (timing only).
    UVTG_LIB_PATH=tools/libuvtg_nostore.so python tools/nt_epilogue_parts.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("UVTG_LIB_PATH", os.path.join(ROOT, "tools", "libuvtg_nostore.so"))
import torch
import bench
from univtg_amd import _lib
from univtg_amd.model import build_model
from univtg_amd.trainer import TrainStep
lib = _lib.load()
lib.uvtg_debug_nt_epilogue_mode.argtypes = [C.c_int]
dev = torch.device("cuda:0")
wl = bench.CONFIGS[2]
torch.manual_seed(2018)
model, crit = build_model(bench.model_args(max_v_l=wl["L_v"], proj_precise=True))
model.to(dev).train(); crit.to(dev).train(); model.set_seed(2018)
step = TrainStep(model, crit, lr=0.0, weight_decay=0.0, grad_clip=0.1, packed="auto")      # lr 0: garbage gradients of modes 1-3 never reach the weights
batches = [bench.synth_batch(wl["B"], wl["L_v"], wl["L_t"], bench.MODEL["D_v"], bench.MODEL["D_t"], i, dev, None, full=True) for i in range(2)]
for i in range(10): step.step(*batches[i % 2])
torch.cuda.synchronize()
names = {0: "full epilogue", 1: "no output stores", 2: "no epilogue-operand loads", 3: "neither stores nor operand loads"}
K = 5
print("persistent NT GEMM inside the headline training step (48 launches per step); event-pair time of the family per step, floor-corrected")
for rnd in range(2):
    for mode in (0, 1, 2, 3):
        lib.uvtg_debug_nt_epilogue_mode(mode)
        for i in range(3): step.step(*batches[i % 2])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(10): step.step(*batches[i % 2])
        e1.record(); torch.cuda.synchronize()
        ms_step = e0.elapsed_time(e1) / 10
        lib.uvtg_profile_start()
        for i in range(K): step.step(*batches[i % 2])
        ms, fl, n = (C.c_double * 8)(), (C.c_double * 8)(), (C.c_longlong * 8)()
        _lib.check(lib.uvtg_profile_stop(ms, fl, n), "uvtg_profile_stop")
        print(f"mode {mode} ({names[mode]:34s}): NT family {ms[3] / K:6.3f} ms per step ({n[3] // K} launches, {fl[3] / (ms[3] * 1e-3) / 1e12:5.0f} TF/s), "
              f"weight gradients {ms[2] / K:6.3f} ms, step {ms_step:6.3f} ms")
lib.uvtg_debug_nt_epilogue_mode(0)
