"""Dev tool: tools/libuvtg_ktrace.so = the library with a PATCHED COPY of gemm.hip whose persistent NT kernel also accumulates, per wave and tile, the
shader cycles it spends (a) between finishing a K tile's instruction stream and passing the next K tile's barrier (own LDS-DMA landing + the
other waves) and (b) inside the K tile body -- tools/nt_ktrace.py reads them.  The patch is applied to a copy so that the shipped source (and
the kernel-source hash the PMC summaries are pinned to) stays untouched.  Usage: python tools/build_ktrace.py"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(R, "univtg_amd/csrc/gemm.hip")).read()
def rep(old, new, cnt=1):
    global src
    assert src.count(old) == cnt, (src.count(old), old[:70])
    src = src.replace(old, new)
# device pointer of the per-wave accumulators: [launch][grid 256][16 local tiles][8 waves][2] u64, set together with the phase-stamp pointer
rep("__global__ void nt_trace_set_kernel(unsigned long long* ptr) { g_nt_trace_dev = ptr; }",
    "__device__ unsigned long long* g_nt_ktrace_dev = nullptr;\n"
    "__global__ void nt_trace_set_kernel(unsigned long long* ptr, unsigned long long* kptr) { g_nt_trace_dev = ptr; g_nt_ktrace_dev = kptr; }")
rep("  int it = 0;\n  [[maybe_unused]] int lt = 0;",
    "  int it = 0;\n  [[maybe_unused]] int lt = 0;\n  unsigned long long k_wait = 0, k_body = 0, k_t0 = 0, k_t1 = 0; int k_n = 0;")
rep("        const int cur = it & 1;\n        __syncthreads();",
    "        const int cur = it & 1;\n"
    "        __builtin_amdgcn_sched_barrier(0); k_t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);\n"
    "        __syncthreads();\n"
    "        __builtin_amdgcn_sched_barrier(0); k_t1 = __builtin_amdgcn_s_memtime(); k_wait += k_t1 - k_t0; __builtin_amdgcn_sched_barrier(0);")
rep("      it++;\n      if constexpr (SMALL) st = st == 2 ? 0 : st + 1;",
    "      if constexpr (!SMALL) { __builtin_amdgcn_sched_barrier(0); k_body += __builtin_amdgcn_s_memtime() - k_t1; k_n++; __builtin_amdgcn_sched_barrier(0); }\n"
    "      it++;\n      if constexpr (SMALL) st = st == 2 ? 0 : st + 1;")
rep("    NT_STAMP(3);\n    lt++;",
    "    NT_STAMP(3);\n"
    "    if (g_nt_ktrace_dev && lane == 0 && lt < 16) { unsigned long long* o = g_nt_ktrace_dev + (((size_t)blockIdx.x * 16 + lt) * 8 + wave) * 2; o[0] = k_wait; o[1] = k_body; }\n"
    "    k_wait = 0; k_body = 0; k_n = 0;\n"
    "    lt++;")
rep("static unsigned long long* g_trace_buf = nullptr;",
    "static unsigned long long* g_trace_buf = nullptr;\nstatic unsigned long long* g_ktrace_buf = nullptr;\n"
    "extern \"C\" int uvtg_debug_nt_ktrace(void* buf) { g_ktrace_buf = (unsigned long long*)buf; return 0; }")
rep("  unsigned long long* ptr = nullptr;\n  if (g_trace_buf && g_trace_next < g_trace_max) {\n    ptr = g_trace_buf + (size_t)g_trace_next * 256 * 16 * 4;",
    "  unsigned long long* ptr = nullptr; unsigned long long* kptr = nullptr;\n  if (g_trace_buf && g_trace_next < g_trace_max) {\n"
    "    ptr = g_trace_buf + (size_t)g_trace_next * 256 * 16 * 4;\n    if (g_ktrace_buf) kptr = g_ktrace_buf + (size_t)g_trace_next * 256 * 16 * 8 * 2;")
rep("  hipLaunchKernelGGL(nt_trace_set_kernel, dim3(1), dim3(1), 0, s, ptr);", "  hipLaunchKernelGGL(nt_trace_set_kernel, dim3(1), dim3(1), 0, s, ptr, kptr);")
os.makedirs("/tmp/uvtg_ktrace", exist_ok=True)
open("/tmp/uvtg_ktrace/gemm_k.hip", "w").write(src)
sys.path.insert(0, R)
from univtg_amd import build
build.build()
hipcc = "/opt/rocm/bin/hipcc"
subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed", "-DUVTG_NT_TRACE",
                "-I", os.path.join(R, "univtg_amd/csrc"), "-c", "/tmp/uvtg_ktrace/gemm_k.hip", "-o", "/tmp/uvtg_ktrace/gemm.o"], check=True)
objs = [os.path.join(R, "univtg_amd/csrc/build", f) for f in sorted(os.listdir(os.path.join(R, "univtg_amd/csrc/build"))) if f.endswith(".o") and f != "gemm.o"]
subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(R, "tools/libuvtg_ktrace.so"), "/tmp/uvtg_ktrace/gemm.o"] + objs, check=True)
print("built tools/libuvtg_ktrace.so")
