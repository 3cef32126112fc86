"""Dev tool: time the attention kernels through the C ABI at the long-sequence shape (BASELINE config 4) and the config-2 shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from univtg_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, S, H, hd) in [(32, 1232, 8, 128), (64, 632, 8, 128), (256, 107, 8, 128)]:
    d = H * hd
    qkv = (torch.randn(B * S, 3 * d, device=dev) * 0.5).to(torch.bfloat16)
    kv = torch.ones(B, S, dtype=torch.uint8, device=dev)
    o, lse = ops.attention_fwd(qkv, kv, B, S, H, hd, False)
    do = torch.randn(B * S, d, device=dev).to(torch.bfloat16)
    tf = timeit(lambda: ops.attention_fwd(qkv, kv, B, S, H, hd, False))
    tb = timeit(lambda: ops.attention_bwd(qkv, kv, o, lse, do, 1.0, B, S, H, hd))
    fl = 4.0 * B * S * S * d
    print(f"B={B} S={S} H={H} hd={hd}: fwd {tf:8.1f} us {fl / tf / 1e6:6.0f} TF | bwd (delta + dK/dV + dQ, incl. zero-init of dqkv) {tb:8.1f} us {2.5 * fl / tb / 1e6:6.0f} TF")
