"""Dev probe: does a half-batch software pipeline on two streams overlap the MFMA-bound GEMMs of one half with the HBM-bound kernels (LayerNorm,
epilogue write bursts) of the other half?  Chain per iteration: NT GEMM (N = K = 1024, fp32 out) -> LayerNorm forward (fp32 rows) -> NT GEMM ->
LayerNorm backward-like second LN, on (a) ONE stream at the full M = 20160 rows, (b) TWO streams with M / 2 rows each, started together,
(c) the same with the second stream delayed by one kernel.  Prints us per iteration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from univtg_amd import _lib
from univtg_amd.model import _ptr
lib = _lib.load()
dev = torch.device("cuda:0")
M, d = 20160, 1024
def mk(rows):
    return dict(a=torch.randn(rows, d, device=dev).to(torch.bfloat16), w=(torch.randn(d, d, device=dev) / 32).to(torch.bfloat16),
                c=torch.empty(rows, d, device=dev), y=torch.empty(rows, d, device=dev), gam=torch.ones(d, device=dev), bet=torch.zeros(d, device=dev),
                mean=torch.empty(rows, device=dev), rstd=torch.empty(rows, device=dev), rows=rows)
def chain(b, stream, reps):
    s = C.c_void_p(stream.cuda_stream)
    for _ in range(reps):
        for _ in range(2):
            lib.uvtg_linear_bf16(_ptr(b["a"]), _ptr(b["w"]), None, _ptr(b["c"]), b["rows"], d, d, 0, s)
            lib.uvtg_layernorm_fwd(_ptr(b["c"]), _ptr(b["gam"]), _ptr(b["bet"]), _ptr(b["y"]), _ptr(b["mean"]), _ptr(b["rstd"]), b["rows"], d, s)
full, ha, hb = mk(M), mk(M // 2), mk(M // 2)
s0, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
REPS = 10
def seq():
    cur = torch.cuda.current_stream(); s0.wait_stream(cur); chain(full, s0, REPS); cur.wait_stream(s0)
def par(delay):
    cur = torch.cuda.current_stream(); s1.wait_stream(cur); s2.wait_stream(cur)
    if delay:
        sh = C.c_void_p(s1.cuda_stream)
        lib.uvtg_linear_bf16(_ptr(ha["a"]), _ptr(ha["w"]), None, _ptr(ha["c"]), ha["rows"], d, d, 0, sh)     # stream 1 gets a head start of one GEMM
        ev = torch.cuda.Event(); ev.record(s1); s2.wait_event(ev)
    chain(ha, s1, REPS); chain(hb, s2, REPS)
    cur.wait_stream(s1); cur.wait_stream(s2)
for name, fn in (("one stream, full M", seq), ("two streams, M/2 each, in phase", lambda: par(False)), ("two streams, M/2 each, one kernel apart", lambda: par(True)), ("one stream, full M", seq)):
    print(f"{name:45s}: {timed(fn) / REPS:8.1f} us per (2 GEMM + 2 LN) iteration")
# reference points: the pieces alone
def only(kind, b):
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(20):
        if kind == "g": lib.uvtg_linear_bf16(_ptr(b["a"]), _ptr(b["w"]), None, _ptr(b["c"]), b["rows"], d, d, 0, s)
        else: lib.uvtg_layernorm_fwd(_ptr(b["c"]), _ptr(b["gam"]), _ptr(b["bet"]), _ptr(b["y"]), _ptr(b["mean"]), _ptr(b["rstd"]), b["rows"], d, s)
for kind, b, nm in (("g", full, "GEMM full M"), ("g", ha, "GEMM M/2"), ("l", full, "LN full M"), ("l", ha, "LN M/2")):
    print(f"{nm:45s}: {timed(lambda: only(kind, b)) / 20:8.1f} us per launch")
