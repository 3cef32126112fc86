"""Dev probe (measurement build, tools/libuvtg_trace.so): which operand's miss latency limits the persistent NT main loop?
act 100 = main loop only; 101 = A pieces re-read K tile 0 (cache-hot); 102 = B pieces do; 103 = both."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("UVTG_LIB_PATH", os.path.join(ROOT, "tools", "libuvtg_trace.so"))
sys.path.insert(0, ROOT)
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
lib.uvtg_debug_force_nt_tile(256)
for bm in (192, 256):
    lib.uvtg_debug_force_nt_bm(bm)
    for (M, N, K) in ((20158, 1024, 1024), (20158, 3072, 1024), (20158, 1024, 3072)):
        a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = torch.randn(N, K, device=dev).to(torch.bfloat16)
        t = {act: min(timeit(lambda: ops.linear_bf16(a, w, None, act)) for _ in range(2)) for act in (100, 101, 102, 103)}
        print(f"bm={bm} {M}x{N}x{K}: loop {t[100]:6.1f} us | A hot {t[101]:6.1f} | B hot {t[102]:6.1f} | both hot {t[103]:6.1f}   (incl. ~6 us torch.empty + launch)")
