"""Dev tool: does the persistent NT main loop slow down per CU when more CUs run it?  Same per-CU work (2 tiles of 192 x 256 x K) at
64 .. 256 active CUs, loop only (act=100) and with the fp32-output epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from univtg_amd import _lib, ops
lib = _lib.load()
dev = torch.device("cuda:0")
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
lib.uvtg_debug_force_nt_tile(256); lib.uvtg_debug_force_nt_bm(192)
for K in (1024, 3072):
    for N in (1024,):
        for cus in (64, 128, 192, 224, 256):
            lib.uvtg_debug_gemm_cus(cus)
            M = cus // (N // 256) * 192 * 2
            a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = torch.randn(N, K, device=dev).to(torch.bfloat16)
            tl = min(timeit(lambda: ops.linear_bf16(a, w, None, 100)) for _ in range(2))
            tf = min(timeit(lambda: ops.linear_bf16(a, w, None, 0)) for _ in range(2))
            fl = 2.0 * M * N * K
            print(f"K={K} cus={cus:3d} M={M:6d}: loop {tl:6.1f} us ({fl / tl / 1e6 / cus:5.2f} TF/CU)  +f32 epilogue {tf:6.1f} us (incl. torch.empty)")
lib.uvtg_debug_gemm_cus(0)
