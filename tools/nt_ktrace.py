"""Dev tool (needs tools/libuvtg_ktrace.so, tools/build_ktrace.py): inside ONE training step, per persistent NT launch, the shader cycles a wave spends per
K tile (a) from the end of its K-tile instruction stream to behind the next barrier (`wait`: own LDS-DMA landing, then the slowest wave) and (b) in the
K-tile body (`body`: fragment reads, staging issues, MFMA issue) -- medians over workgroups and local tiles, loader waves (0..3) and the others (4..7)
apart.  usage: nt_ktrace.py [--full] [lw-mask]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("UVTG_LIB_PATH", os.path.join(ROOT, "tools", "libuvtg_ktrace.so"))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from univtg_amd import _lib
from univtg_amd.model import build_model
from univtg_amd.trainer import TrainStep

full = "--full" in sys.argv
wl = bench.CONFIGS[2]
dev = torch.device("cuda:0")
torch.manual_seed(2018)
model, crit = build_model(bench.model_args(max_v_l=wl["L_v"]))
model.to(dev).train(); crit.to(dev).train(); model.set_seed(2018)
step = TrainStep(model, crit, lr=1e-4, weight_decay=1e-4, grad_clip=0.1, packed="auto")
batch = bench.synth_batch(wl["B"], wl["L_v"], wl["L_t"], bench.MODEL["D_v"], bench.MODEL["D_t"], 0, dev, None, full=full)
lib = _lib.load()
for a in sys.argv[1:]:
    if a.isdigit():
        _lib.check(lib.uvtg_debug_nt_loader_waves(int(a)))
for _ in range(6):
    step.step(*batch)
torch.cuda.synchronize()
lib.uvtg_debug_nt_trace.argtypes = [C.c_void_p, C.c_int]
lib.uvtg_debug_nt_trace_info.argtypes = [C.c_int, C.c_void_p]
lib.uvtg_debug_nt_ktrace.argtypes = [C.c_void_p]
NL = 96
buf = torch.zeros(NL, 256, 16, 4, dtype=torch.int64, device=dev)
kbuf = torch.zeros(NL, 256, 16, 8, 2, dtype=torch.int64, device=dev)
lib.uvtg_debug_nt_ktrace(kbuf.data_ptr())
lib.uvtg_debug_nt_trace(buf.data_ptr(), NL)
step.step(*batch)
torch.cuda.synchronize()
t = buf.cpu().numpy().astype(np.float64) * 0.01
k = kbuf.cpu().numpy().astype(np.float64)
info = (C.c_int * 8)()
print(f"{'#':>3} {'M':>6} {'N':>5} {'K':>5} tm e g grid | tile main loop us | per K tile, shader cycles: wait / body  waves 0-3 | waves 4-7 | implied clock GHz")
for i in range(NL):
    if lib.uvtg_debug_nt_trace_info(i, info):
        break
    M, N, K, tm, eop, gather, grid, groups = list(info)
    if M < 0:
        continue
    a = t[i, :grid]; used = a[..., 3] > 0
    kk = k[i, :grid]
    nk = K // 64
    main_us = (a[..., 2] - a[..., 0])[used]
    w = kk[..., 0][used] / nk; b = kk[..., 1][used] / nk            # [n, 8 waves]
    if not (b > 0).any():
        continue
    cyc = (kk[..., 0] + kk[..., 1])[used].mean(axis=1)                 # cycles per tile main loop (mean over waves)
    ghz = np.median(cyc / np.maximum(main_us, 1e-9)) / 1e3
    f = lambda x: f"{np.median(x):6.0f}"
    print(f"{i:3d} {M:6d} {N:5d} {K:5d} {tm:2d} {eop} {gather} {grid:4d} | {np.median(main_us):6.1f} | {f(w[:, :4])} /{f(b[:, :4])} | {f(w[:, 4:])} /{f(b[:, 4:])} | {ghz:.2f}")
