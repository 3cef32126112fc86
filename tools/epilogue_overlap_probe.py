"""Dev probe (round 6, VERDICT r5 item 2: "a committed probe showing the non-MFMA worker cannot overlap"): could the epilogue traffic of the
N = K = 1024 launches hide under the main loop if SOMEBODY ELSE on the same CU moved it?  The persistent NT kernel at 256-row tiles (<= 240
registers, 2 waves per SIMD, 128 KB of LDS) leaves 32 registers per lane and 32 KB of LDS per CU: exactly one register-light 4-wave workgroup
per CU fits beside it (tools/lab/stream_probe.hip: 20 VGPRs, no LDS).  Arms, one process, alternating:
  loop      the GEMM, main loop only (act = 100)                                  -> t_loop
  full      the GEMM with its epilogue (bf16 residual in, bf16 out: EPI 1)        -> t_full
  stream    the streamer alone: reads 2 x 56 MB, writes 56 MB (one launch's epilogue bytes), one 4-wave block per CU -> t_stream
  both      streamer launched first on stream 2, the main-loop-only GEMM on stream 1, end = both done -> t_both
t_both ~ max(t_loop, t_stream): the bytes can hide (what a worker design could win = t_full - t_both);
t_both ~ t_loop + t_stream (or more): the memory system is what both wait for, no worker design helps."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from univtg_amd import _lib, ops
from univtg_amd.model import _ptr
lib = _lib.load()
sp = C.CDLL(os.path.join(ROOT, "tools", "lab", "libstream_probe.so"))
sp.stream_copy_add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
M, N, K = 27392, 1024, 1024
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
res = torch.randn(M, N, device=dev).to(torch.bfloat16)
sa, sb, so = (torch.randn(M, N, device=dev).to(torch.bfloat16) for _ in range(3))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
lib.uvtg_debug_force_nt_tile(256); lib.uvtg_debug_force_nt_bm(256)
nbytes = M * N * 2

def gemm(act, st):
    with torch.cuda.stream(st):
        ops.linear_bf16(a, w, None, act)

def streamer(st, blocks=256):
    sp.stream_copy_add(_ptr(sa), _ptr(sb), _ptr(so), nbytes, 1, blocks, C.c_void_p(st.cuda_stream))

def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def arm_loop(): gemm(100, torch.cuda.current_stream())
def arm_full(): gemm(0, torch.cuda.current_stream())
def arm_stream(): streamer(torch.cuda.current_stream())
def arm_both(blocks=256):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    streamer(s2, blocks)
    gemm(100, s1)
    cur.wait_stream(s1); cur.wait_stream(s2)
def arm_fork_only():          # the fork / join alone around the main-loop-only GEMM (what the two-stream plumbing costs)
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    gemm(100, s1)
    cur.wait_stream(s1); cur.wait_stream(s2)

print(f"{M} x {N} x {K}, 256-row tiles (428 tiles = 1.67 rounds); streamer = {3 * nbytes / 1e6:.0f} MB of traffic per launch")
for rnd in range(3):
    r = dict(loop=timed(arm_loop), full=timed(arm_full), stream=timed(arm_stream), fork_only=timed(arm_fork_only), both=timed(arm_both),
             both_128blocks=timed(lambda: arm_both(128)), both_512blocks=timed(lambda: arm_both(512)))
    print("  ".join(f"{k} {v:6.1f} us" for k, v in r.items()), f"| loop + stream = {r['loop'] + r['stream']:.1f}, max = {max(r['loop'], r['stream']):.1f}, "
          f"both - fork_only = {r['both'] - r['fork_only']:+.1f}")
lib.uvtg_debug_force_nt_bm(0); lib.uvtg_debug_force_nt_tile(0)
