"""Dev tool: the persistent NT GEMM launches of ONE inference call (default precision) at batch B, phase by phase (needs tools/libuvtg_trace.so,
tools/build_trace.sh).  Split-K launches: the workgroups whose part arrives last carry fold + epilogue in `tail`, the others only publish + ticket.
usage: nt_trace_infer.py [B] [bf16]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("UVTG_LIB_PATH", os.path.join(ROOT, "tools", "libuvtg_trace.so"))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from univtg_amd import _lib
from univtg_amd.model import build_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
prec = "bf16" if "bf16" in sys.argv else "auto"
dev = torch.device("cuda:0")
torch.manual_seed(2018)
model, _ = build_model(bench.model_args(max_v_l=75, precision=prec))
model.to(dev).eval()
inp, ts, tm_, dur = bench.infer_batch(B, 75, 32, 2818, 512, 50, dev)
with torch.no_grad():
    for _ in range(5):
        model(**inp)
torch.cuda.synchronize()
lib = _lib.load()
lib.uvtg_debug_nt_trace.argtypes = [C.c_void_p, C.c_int]
lib.uvtg_debug_nt_trace_info.argtypes = [C.c_int, C.c_void_p]
NL = 40
buf = torch.zeros(NL, 256, 16, 4, dtype=torch.int64, device=dev)
lib.uvtg_debug_nt_trace(buf.data_ptr(), NL)
with torch.no_grad():
    model(**inp)
torch.cuda.synchronize()
t = buf.cpu().numpy().astype(np.float64) * 0.01
info = (C.c_int * 8)()
print(f"batch {B} {prec}:  #      M     N     K tm g grid   span | first K tile / rest of main loop / tail (us): median [min .. max]; tail split into its two populations when bimodal")
tot = 0.0
for i in range(NL):
    if lib.uvtg_debug_nt_trace_info(i, info):
        break
    M, N, K, tm, eop, gather, grid, groups = list(info)
    a = t[i, :grid, 0]
    used = a[:, 3] > 0
    a = a[used]
    span = a[:, 3].max() - a[:, 0].min(); tot += span
    first, rest, tail = a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2]
    f = lambda x: f"{np.median(x):5.1f} [{x.min():5.1f} ..{x.max():5.1f}]"
    ts_ = np.sort(tail); gap = np.diff(ts_); extra = ""
    if len(ts_) > 3 and gap.max() > 2.0:
        c = int(np.argmax(gap)) + 1
        extra = f"  | tail populations: {c} x {np.median(ts_[:c]):.1f}, {len(ts_) - c} x {np.median(ts_[c:]):.1f}"
    print(f"{i:3d} {M:6d} {N:5d} {K:5d} {tm:2d} {gather} {grid:4d} {span:6.1f} | {f(first)} / {f(rest)} / {f(tail)}{extra}  start skew {a[:, 0].max() - a[:, 0].min():.1f}")
print(f"sum of spans {tot:.1f} us")
