"""Dev probe: is the persistent NT GEMM slowed by L2-channel camping on power-of-two row strides?  uvtg_linear_bf16 takes lda = ldb = K, so
K = 1024 (2 KB rows: every 8-row piece of a wave-instruction starts 2 KB apart) is compared with K = 960 / 1088 / 1152 (rows 1920 / 2176 /
2304 bytes apart) at the step's launch shape M = 20158, N = 1024: TF/s should be flat in K (within the 15 / 17 / 18 K-tile amortisation of
the prologue and epilogue) unless the stride matters.  act = 100: main loop only (no epilogue), act = 0: fp32 output epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from univtg_amd import _lib
from univtg_amd.model import _ptr
lib = _lib.load()
dev = torch.device("cuda:0")
def run(M, N, K, act, n=20):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        lib.uvtg_linear_bf16(_ptr(a), _ptr(w), None, _ptr(c), M, N, K, act, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        lib.uvtg_linear_bf16(_ptr(a), _ptr(w), None, _ptr(c), M, N, K, act, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    return us, 2.0 * M * N * K / us / 1e6
for M, N in ((20158, 1024), (20158, 3072), (27392, 1024)):
    for act in (100, 0):
        row = []
        for K in (960, 1024, 1088, 1152, 2048, 2112, 3072, 3136):
            us, tf = run(M, N, K, act)
            row.append(f"K={K}: {us:6.1f} us {tf:5.0f} TF")
        print(f"M={M} N={N} act={act}: " + " | ".join(row))
