"""Dev probe: do an NT GEMM chain (bf16-ish epilogue traffic) and a TN weight-gradient GEMM run faster side by side on two streams
with split CU budgets than back to back on the whole chip?  (decides whether uvtg_backward should fork the weight gradients)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from univtg_amd import _lib, ops
lib = _lib.load()
dev = torch.device("cuda:0")
M, d = 24300, 1024
a = torch.randn(M, d, device=dev).to(torch.bfloat16)
w = torch.randn(d, d, device=dev).to(torch.bfloat16)
w3 = torch.randn(3 * d, d, device=dev).to(torch.bfloat16)
a3 = torch.randn(M, 3 * d, device=dev).to(torch.bfloat16)
wk3 = torch.randn(d, 3 * d, device=dev).to(torch.bfloat16)
dy = torch.randn(M, d, device=dev).to(torch.bfloat16)
x = torch.randn(M, d, device=dev).to(torch.bfloat16)
dy2 = torch.randn(M, 2 * d, device=dev).to(torch.bfloat16)
outs = [torch.empty(M, d, device=dev) for _ in range(3)] + [torch.empty(M, d, device=dev)]
from univtg_amd.model import _ptr
import ctypes as C
nf = lib.uvtg_wgrad_scratch_floats(M, 2 * d, d)
scr = [torch.empty(nf, device=dev) for _ in range(2)]
dw = [torch.zeros(d, d, device=dev), torch.zeros(2 * d, d, device=dev)]
def nt_chain(st):
    s = C.c_void_p(st.cuda_stream)
    for i in range(3):
        lib.uvtg_linear_bf16(_ptr(a), _ptr(w), None, _ptr(outs[i]), M, d, d, 0, s)
    lib.uvtg_linear_bf16(_ptr(a3), _ptr(wk3), None, _ptr(outs[3]), M, d, 3 * d, 0, s)
def tn_pair(st):
    s = C.c_void_p(st.cuda_stream)
    lib.uvtg_wgrad_bf16_ws(_ptr(dy), _ptr(x), _ptr(dw[0]), None, M, d, d, _ptr(scr[0]), nf, s)
    lib.uvtg_wgrad_bf16_ws(_ptr(dy2), _ptr(x), _ptr(dw[1]), None, M, 2 * d, d, _ptr(scr[1]), nf, s)
    lib.uvtg_wgrad_bf16_ws(_ptr(dy2), _ptr(x), _ptr(dw[1]), None, M, 2 * d, d, _ptr(scr[1]), nf, s)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(mode, c1, c2, n=10):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(2):
        e0.record()
        for _ in range(n):
            if mode == "seq":
                lib.uvtg_debug_gemm_cus(0); nt_chain(torch.cuda.current_stream()); tn_pair(torch.cuda.current_stream())
            else:
                s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
                lib.uvtg_debug_gemm_cus(c1); nt_chain(s1)
                lib.uvtg_debug_gemm_cus(c2); tn_pair(s2)
                torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
        e1.record(); torch.cuda.synchronize()
    lib.uvtg_debug_gemm_cus(0)
    return e0.elapsed_time(e1) / n * 1e3
print("NT chain (3 x d*d + 1 x K=3d, fp32 out) alone: %.1f us" % run("seq", 0, 0) if False else "", end="")
torch.cuda.synchronize()
for mode, c1, c2 in (("seq", 0, 0), ("par", 128, 128), ("par", 160, 96), ("par", 96, 160), ("par", 256, 256), ("par", 192, 64), ("seq", 0, 0)):
    print(f"{mode} NT cus {c1:3d} / TN cus {c2:3d}: {run(mode, c1, c2):8.1f} us per (4 NT + 3 TN)")
