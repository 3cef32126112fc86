"""Dev tool: time the GEMM kernels through the C ABI at the hot-path shapes."""
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
    return e0.elapsed_time(e1) / n * 1e3  # us

for (M, N, K) in [(27392, 1024, 1024), (27392, 2048, 1024), (27392, 1024, 3072), (19200, 2048, 3072), (4096, 4096, 4096), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    t = timeit(lambda: ops.linear_bf16(a, w, None, 0))
    print(f"NT bf16 {M}x{N}x{K}: {t:8.1f} us  {2*M*N*K/t/1e6:8.1f} TFLOP/s (fp32 out)")
    t = timeit(lambda: ops.linear_bf16(a, w, None, 100))
    print(f"NT bf16 {M}x{N}x{K}: {t:8.1f} us  {2*M*N*K/t/1e6:8.1f} TFLOP/s (NO epilogue)")
    af, wf = a.float(), w.float()
    t = timeit(lambda: ops.linear_f32x3(af, wf, None, 0), 5)
    print(f"NT x3   {M}x{N}x{K}: {t:8.1f} us  {2*M*N*K/t/1e6:8.1f} TFLOP/s-equiv")
for (M, N, K) in [(27392, 1024, 1024), (27392, 2048, 1024), (19200, 1024, 2944)]:
    dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    t = timeit(lambda: ops.wgrad_bf16(dy, x, 8))
    print(f"TN bf16 {M}x{N}x{K}: {t:8.1f} us  {2*M*N*K/t/1e6:8.1f} TFLOP/s (incl. zero-init of dW)")
    t = timeit(lambda: ops.wgrad_bf16_ws(dy, x))
    print(f"TN256   {M}x{N}x{K}: {t:8.1f} us  {2*M*N*K/t/1e6:8.1f} TFLOP/s (incl. zero-init of dW, scratch alloc)")
