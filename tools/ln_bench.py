import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from univtg_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rows, D in [(27392, 1024), (19200, 2818)]:
    x = torch.randn(rows, D, device=dev); g = torch.randn(rows, D, device=dev); gam = torch.randn(D, device=dev)
    y, mean, rstd = ops.layernorm_fwd(x, gam, gam)
    print(f"ln_bwd fp32 {rows}x{D} dbg={os.environ.get('UVTG_LN_DEBUG','0')}: {timeit(lambda: ops.layernorm_bwd(g, x, mean, rstd, gam)):.1f} us (incl. 3 small allocs/zeroing)")
