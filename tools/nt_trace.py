"""Dev tool: where the time of each persistent NT GEMM of ONE training step goes (needs tools/libuvtg_trace.so = gemm.hip built with
-DUVTG_NT_TRACE, see tools/build_trace.sh).  Per launch: shape, kernel span, and the median over workgroups, per local tile index, of
   wait  = (stamp after K tile 0) - (tile start) - one average K tile      (what the first barrier waits for: operand landing + the stores
                                                                          of the previous epilogue)
   main  = main-loop time, epi = epilogue time."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("UVTG_LIB_PATH", os.path.join(ROOT, "tools", "libuvtg_trace.so"))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from univtg_amd import _lib
from univtg_amd.model import build_model
from univtg_amd.trainer import TrainStep

cfg = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2
full = "--full" in sys.argv          # SURVEY 8d variant A (all-ones masks: the round-4 headline workload)
wl = bench.CONFIGS[cfg]
dev = torch.device("cuda:0")
torch.manual_seed(2018)
model, crit = build_model(bench.model_args(max_v_l=wl["L_v"]))
model.to(dev).train(); crit.to(dev).train(); model.set_seed(2018)
step = TrainStep(model, crit, lr=1e-4, weight_decay=1e-4, grad_clip=0.1, packed="auto")
lens = bench.mixed_length_lens(wl["B"], seed=0) if cfg == 5 else None
batch = bench.synth_batch(wl["B"], wl["L_v"], wl["L_t"], bench.MODEL["D_v"], bench.MODEL["D_t"], 0, dev, lens, full=full)
for _ in range(8):
    step.step(*batch)
torch.cuda.synchronize()
lib = _lib.load()
lib.uvtg_debug_nt_trace.argtypes = [C.c_void_p, C.c_int]
lib.uvtg_debug_nt_trace_info.argtypes = [C.c_int, C.c_void_p]
NL = 96
buf = torch.zeros(NL, 256, 16, 4, dtype=torch.int64, device=dev)
lib.uvtg_debug_nt_trace(buf.data_ptr(), NL)
step.step(*batch)
torch.cuda.synchronize()
t = buf.cpu().numpy().astype(np.float64) * 0.01          # us (100 MHz)
print(f"{'#':>3} {'M':>6} {'N':>5} {'K':>5} tm e g {'grid':>4} {'span':>7} {'TF/s':>6} | per local tile: wait / main / epi (us, median over workgroups)")
info = (C.c_int * 8)()
tot = {}
for i in range(NL):
    if lib.uvtg_debug_nt_trace_info(i, info):
        break
    M, N, K, tm, eop, gather, grid, groups = list(info)
    if M < 0:      # weight-gradient (TN) launch: -rows, tiles, splits, steps per unit, units
        a = t[i].reshape(-1, 4)[:grid]
        span = a[:, 2].max() - a[:, 0].min()
        main, slab, steps = a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 3] * 100.0
        print(f"{i:3d} TN rows={-M} tiles={N} splits={K} groups={groups} steps/unit<={tm} units={grid} span {span:7.1f} us | main median {np.median(main):6.1f} "
              f"(max {main.max():6.1f}; {np.median(main / np.maximum(steps, 1)):.3f} us/step)  slab write median {np.median(slab):5.1f} (max {slab.max():5.1f})")
        if "--xcd" in sys.argv:
            print("      per-XCD median main:", " ".join(f"{np.median(main[x::8]):6.1f}" for x in range(8)), "| start skew (max - min):", f"{a[:, 0].max() - a[:, 0].min():.1f} us",
                  "| slowest units:", np.argsort(-main)[:8].tolist())
        continue
    a = t[i, :grid]
    used = a[..., 3] > 0
    span = a[..., 3][used].max() - a[..., 0][used].min()
    nk = K // 64
    cols = []
    for lt in range(16):
        u = used[:, lt]
        if not u.any():
            break
        s0, s1, s2, s3 = (a[u, lt, k] for k in range(4))
        main = s2 - s0
        ktile = (s2 - s1) / max(nk - 1, 1)
        wait = (s1 - s0) - ktile
        cols.append(f"{np.median(wait):5.1f}/{np.median(main):5.1f}/{np.median(s3 - s2):5.1f}[{int(u.sum())}]")
    fl = 2.0 * M * N * K * groups
    print(f"{i:3d} {M:6d} {N:5d} {K:5d} {tm:2d} {eop} {gather} {grid:4d} {span:7.1f} {fl / span / 1e6:6.0f} | " + "  ".join(cols))
