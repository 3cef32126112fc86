"""Dev tool: A/B the persistent NT GEMM structures (tile height x one/two workgroups per CU) through the C ABI."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from univtg_amd import ops, _lib

dev = torch.device("cuda:0")
lib = _lib.load()
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us

shapes = [(27392, 1024, 1024), (22016, 1024, 1024), (21000, 1024, 1024), (22016, 2048, 1024), (22016, 1024, 2048), (19200, 1024, 2880), (19200, 1024, 3072), (8192, 1024, 1024)]
for (M, N, K) in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    ref = None
    row = []
    for wn, bm in ((4, 256), (4, 192), (4, 128), (2, 192), (2, 128)):
        _lib.check(lib.uvtg_debug_force_nt_tile(256)); _lib.check(lib.uvtg_debug_force_nt_wn(wn)); _lib.check(lib.uvtg_debug_force_nt_bm(bm))
        y = ops.linear_bf16(a, w, None, 0)
        if ref is None: ref = y
        err = float((y - ref).abs().max())
        t = timeit(lambda: ops.linear_bf16(a, w, None, 0))
        t0 = timeit(lambda: ops.linear_bf16(a, w, None, 100))
        row.append(f"wn{wn}/bm{bm}: {t:6.1f} ({t0:6.1f}) us err {err:.1e}")
    lib.uvtg_debug_force_nt_wn(0); lib.uvtg_debug_force_nt_bm(0); lib.uvtg_debug_force_nt_tile(0)
    tauto = timeit(lambda: ops.linear_bf16(a, w, None, 0))
    print(f"{M}x{N}x{K}: " + " | ".join(row) + f" | auto {tauto:6.1f}")
