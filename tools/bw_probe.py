"""Dev tool: what this box's HBM path delivers to plain streaming kernels (torch copy / add / sum over tensors far larger than the caches)."""
import torch
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (64, 256, 1024, 4096):
    n = mb * 1024 * 1024 // 2
    a = torch.empty(n, dtype=torch.bfloat16, device=dev).normal_()
    b = torch.empty_like(a)
    c = torch.empty_like(a)
    t = timeit(lambda: b.copy_(a)); print(f"{mb:5d} MB tensors: copy      {2 * mb / 1024 / t / 1e3 * 1.048576:6.2f} TB/s ({t * 1e6:7.1f} us)")
    t = timeit(lambda: torch.add(a, b, out=c)); print(f"{mb:5d} MB tensors: add       {3 * mb / 1024 / t / 1e3 * 1.048576:6.2f} TB/s ({t * 1e6:7.1f} us)")
    t = timeit(lambda: a.float().sum()); print(f"{mb:5d} MB tensors: cast+sum  (read-mostly) {t * 1e6:7.1f} us")
    t = timeit(lambda: a.zero_()); print(f"{mb:5d} MB tensors: fill      {mb / 1024 / t / 1e3 * 1.048576:6.2f} TB/s ({t * 1e6:7.1f} us)")
