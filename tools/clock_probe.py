"""Dev probe (round 6): the shader clock the chip SUSTAINS under the headline's kernels -- one resident wave (tools/lab/clock_probe.hip) samples
(shader cycle counter, 100 MHz real-time counter) pairs every few microseconds on a side stream while the main stream runs (a) nothing, (b) the
persistent NT GEMM at the encoder's N = K = 1024 shape back to back, (c) the training step.  dense bf16 peak = 2.5 PFLOP/s at the nominal
2.4 GHz; at the sustained clock the matrix cores' own ceiling is 2.5 x clock / 2.4."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from univtg_amd import _lib, ops
from univtg_amd.model import build_model, _ptr
from univtg_amd.trainer import TrainStep
lib = _lib.load()
cp = C.CDLL(os.path.join(ROOT, "tools", "lab", "libclock_probe.so"))
cp.clock_probe.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
side = torch.cuda.Stream()
N = 3000

def probe(work, label):
    buf = torch.zeros(2 * N, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    cp.clock_probe(_ptr(buf), N, C.c_void_p(side.cuda_stream))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); work(); e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    v = buf.cpu().view(N, 2).double()
    dc, dr = v[1:, 0] - v[:-1, 0], v[1:, 1] - v[:-1, 1]
    ghz = dc / dr * 0.1                              # cycles per 10 ns tick -> GHz
    span_ms = float(v[-1, 1] - v[0, 1]) / 1e5
    # samples taken while the work ran: the first ms / span_ms of the probe's span (both start together)
    k = max(10, min(N - 1, int((N - 1) * min(1.0, ms / span_ms))))
    g = ghz[:k]
    q = torch.quantile(g, torch.tensor([0.05, 0.5, 0.95], dtype=torch.double))
    print(f"{label:58s} work {ms:7.2f} ms | probe span {span_ms:6.2f} ms, {k} samples under load | shader clock mean {float(g.mean()):.3f} GHz "
          f"(p5 {float(q[0]):.3f} / median {float(q[1]):.3f} / p95 {float(q[2]):.3f}) -> bf16 MFMA ceiling at that clock {2500 * float(g.mean()) / 2.4:6.0f} TFLOP/s")
    return float(g.mean())

wl = bench.CONFIGS[2]
torch.manual_seed(2018)
model, crit = build_model(bench.model_args(max_v_l=wl["L_v"], proj_precise=True))
model.to(dev).train(); crit.to(dev).train(); model.set_seed(2018)
step = TrainStep(model, crit, lr=1e-4, weight_decay=1e-4, grad_clip=0.1, packed="auto")
batches = [bench.synth_batch(wl["B"], wl["L_v"], wl["L_t"], bench.MODEL["D_v"], bench.MODEL["D_t"], i, dev, None, full=True) for i in range(2)]
for i in range(10): step.step(*batches[i % 2])
M, Nn, K = 27392, 1024, 1024
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = (torch.randn(Nn, K, device=dev) / K ** 0.5).to(torch.bfloat16)
for _ in range(5): ops.linear_bf16(a, w, None, 0)
for rnd in range(2):
    probe(lambda: torch.cuda._sleep(int(2.4e9 * 0.010)), "idle chip (a 10 ms device-side sleep)")
    c_nt = probe(lambda: [ops.linear_bf16(a, w, None, 0) for _ in range(150)], "persistent NT GEMM 27392 x 1024 x 1024, 150 launches back to back")
    c_st = probe(lambda: [step.step(*batches[i % 2]) for i in range(1)], "one training step (config 2, variant A)")
    c_s4 = probe(lambda: [step.step(*batches[i % 2]) for i in range(4)], "four training steps")
