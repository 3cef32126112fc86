"""Dev probe (round 6, VERDICT r5 item 2: "a committed probe showing the non-MFMA worker cannot overlap"): could the epilogue traffic of the
N = K = 1024 launches hide under the main loop if SOMEBODY ELSE on the same CU moved it?  The persistent NT kernel at 256-row tiles (<= 240
registers, 2 waves per SIMD, 128 KB of LDS) leaves 32 registers per lane and 32 KB of LDS per CU: exactly one register-light 4-wave workgroup
per CU fits beside it (tools/lab/stream_probe.hip: 20 VGPRs, no LDS).  Arms, one process:
  alone         the GEMM main loop only (act = 100), the GEMM with its epilogue (kernel-level entry: fp32 out, 112 MB), one streamer pass (168 MB)
  side by side  20 GEMM launches on stream 1 WHILE one long streamer launch runs on stream 2 (started first, one block per CU, resident
                throughout); each stream timed by its own event pair
If the main loop keeps its time beside the streamer and the streamer keeps most of its rate, epilogue-sized traffic CAN hide under the MFMAs of
a co-resident workgroup (the upper bound of what a worker design could win is then t_full - t_loop per launch); if the main loop slows by
what the streamer moves, the memory system is what both wait for and no worker design helps."""
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

def side_by_side(act, n_gemm=20, blocks=256):
    """n_gemm GEMM launches on stream 1 WHILE one long streamer launch (enough passes to outlast them) runs on stream 2, started first.
    Returns (us per GEMM launch, streamer passes completed per GEMM launch's time as MB/us is derived by the caller).  Event pairs on each
    stream: the fork / join latency of the two-stream plumbing is outside both measurements."""
    passes = max(8, int(n_gemm * 90 / 26) + 8)
    cur = torch.cuda.current_stream()
    torch.cuda.synchronize()
    s1.wait_stream(cur); s2.wait_stream(cur)
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s2):
        b0.record()
        sp.stream_copy_add(_ptr(sa), _ptr(sb), _ptr(so), nbytes, passes, blocks, C.c_void_p(s2.cuda_stream))
        b1.record()
    with torch.cuda.stream(s1):
        for _ in range(3): ops.linear_bf16(a, w, None, act)        # (the streamer is resident by the time the timed launches start)
        a0.record()
        for _ in range(n_gemm): ops.linear_bf16(a, w, None, act)
        a1.record()
    torch.cuda.synchronize()
    return a0.elapsed_time(a1) / n_gemm * 1e3, b0.elapsed_time(b1) / passes * 1e3

print(f"{M} x {N} x {K}, 256-row tiles (428 tiles = 1.67 rounds); one streamer pass = {3 * nbytes / 1e6:.0f} MB of traffic (2 x 56 MB in, 56 MB out), "
      f"one 4-wave block per CU (20 VGPRs: co-resident with the GEMM's 240-register workgroups)")
for rnd in range(3):
    t_loop, t_full, t_stream = timed(arm_loop), timed(arm_full), timed(arm_stream)
    g_side, s_side = side_by_side(100)
    f_side, s_side_f = side_by_side(0)
    # in the time of one main-loop-only launch beside the streamer, how many streamer MB moved; a launch's own epilogue traffic is 112 MB (fp32 out here)
    mb_per_launch = 3 * nbytes / 1e6 * g_side / s_side
    print(f"alone: loop {t_loop:6.1f} us  full {t_full:6.1f} us  streamer pass {t_stream:6.1f} us | side by side: loop {g_side:6.1f} us "
          f"({g_side / t_loop:4.2f}x) while the streamer runs a pass in {s_side:6.1f} us ({s_side / t_stream:4.2f}x) = {mb_per_launch:5.0f} MB moved per "
          f"GEMM launch | full beside the streamer {f_side:6.1f} us ({f_side / t_full:4.2f}x), streamer pass {s_side_f:6.1f} us")
lib.uvtg_debug_force_nt_bm(0); lib.uvtg_debug_force_nt_tile(0)
