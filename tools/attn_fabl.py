"""Dev tool: time the fused (S <= 128) attention backward under the ablation modes of a -DUVTG_ATTN_ABLATE build (UVTG_LIB_PATH, UVTG_ATTN_FABL)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from univtg_amd import ops
dev = torch.device("cuda:0")
B, S, H, hd = 256, 107, 8, 128
d = H * hd
qkv = (torch.randn(B * S, 3 * d, device=dev) * 0.5).to(torch.bfloat16)
g = torch.Generator().manual_seed(1)
lens = torch.randint(70, S + 1, (B,), generator=g)
kv = (torch.arange(S)[None, :] < lens[:, None]).to(torch.uint8).to(dev)
o, lse = ops.attention_fwd(qkv, kv, B, S, H, hd, False)
do = torch.randn(B * S, d, device=dev).to(torch.bfloat16)
fn = lambda: ops.attention_bwd(qkv, kv, o, lse, do, 1.0, B, S, H, hd)
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): fn()
e1.record(); torch.cuda.synchronize()
print(f"UVTG_ATTN_FABL={os.environ.get('UVTG_ATTN_FABL', '0')}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us (delta + zero-init + kernel)")
