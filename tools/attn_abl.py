"""Dev tool: time the long-sequence dK/dV kernel alone under the ablation modes of a -DUVTG_ATTN_ABLATE build (UVTG_LIB_PATH, UVTG_ATTN_ABL)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from univtg_amd import ops
dev = torch.device("cuda:0")
B, S, H, hd = 32, 1232, 8, 128
d = H * hd
qkv = (torch.randn(B * S, 3 * d, device=dev) * 0.5).to(torch.bfloat16)
kv = torch.ones(B, S, dtype=torch.uint8, device=dev)
o, lse = ops.attention_fwd(qkv, kv, B, S, H, hd, False)
do = torch.randn(B * S, d, device=dev).to(torch.bfloat16)
fn = lambda: ops.attention_bwd(qkv, kv, o, lse, do, 1.0, B, S, H, hd)
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): fn()
e1.record(); torch.cuda.synchronize()
print(f"UVTG_ATTN_ABL={os.environ.get('UVTG_ATTN_ABL', '0')}: {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us (delta + zero-init + kernel)")
