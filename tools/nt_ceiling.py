"""Dev tool (round 5, VERDICT r4 item 2: "so that the ceiling of this tile shape is a measurement, not prose"): the persistent NT kernel at the
headline's launch shapes, MAIN LOOP ONLY (act = 100: no epilogue -- what the K loop + the CU-round structure alone allow) against the same
launch with its epilogue (kernel-level entry: bias-free fp32 output), per tile height.  Same box, same process, best of 3 x 20 launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from univtg_amd import _lib, ops
lib = _lib.load()
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
lib.uvtg_debug_force_nt_tile(256)
for (M, N, K) in [(27392, 1024, 1024), (27392, 3072, 1024), (27392, 1024, 3072), (20480, 1024, 1024), (16384, 1024, 1024)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    out = []
    for bm in (256, 320, 192):
        lib.uvtg_debug_force_nt_bm(bm)
        tiles = -(-M // bm) * (N // 256)
        for act in (100, 0):
            t = min(timeit(lambda: ops.linear_bf16(a, w, None, act)) for _ in range(3))
            out.append(f"{bm} rows ({tiles / 256:.2f} rounds) {'loop only' if act == 100 else 'with epilogue'} {t:6.1f} us {2 * M * N * K / t / 1e6:5.0f} TF/s")
    print(f"{M} x {N} x {K}:\n   " + "\n   ".join(out))
lib.uvtg_debug_force_nt_bm(0); lib.uvtg_debug_force_nt_tile(0)
