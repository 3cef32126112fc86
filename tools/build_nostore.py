"""Dev tool (round 6): tools/libuvtg_nostore.so = the library with a PATCHED COPY of gemm.hip whose persistent NT kernel can skip parts of its
EPILOGUE at run time (a device-side mode word, set by uvtg_debug_nt_epilogue_mode): bit 0 = the output stores are not issued (everything else of
the epilogue -- LDS transpose, bias, activation, packing -- still runs), bit 1 = the bf16 epilogue operand (residual / pre-activation) is not
loaded (zeros instead).  Results are garbage; what it answers is WHERE the exposed epilogue time of the N = K = 1024 launches goes: the store
burst, the operand round trips, or the instruction stream (tools/nt_epilogue_parts.py).  The patch is applied to a copy: the shipped source
(and the kernel-source hash of the PMC summaries) stays untouched.  Usage: python tools/build_nostore.py"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(R, "univtg_amd/csrc/gemm.hip")).read()
def rep(old, new, cnt=1):
    global src
    assert src.count(old) == cnt, (src.count(old), old[:80])
    src = src.replace(old, new)
rep("typedef __attribute__((address_space(3))) void lds_void_t;",
    "__device__ int g_nt_epi_mode_dev = 0;\n__global__ void nt_epi_mode_set_kernel(int m) { g_nt_epi_mode_dev = m; }\n"
    "typedef __attribute__((address_space(3))) void lds_void_t;")
# mode word: one scalar load per workgroup, ahead of the tile loop
rep("  int it = 0;\n  [[maybe_unused]] int lt = 0;",
    "  int it = 0;\n  [[maybe_unused]] int lt = 0;\n  const int epi_mode = __builtin_amdgcn_readfirstlane(g_nt_epi_mode_dev);")
rep("            *(u32x4*)(p.outPre + go + orow * p.ldpre_out + n) = t;", "            if (!(epi_mode & 1)) *(u32x4*)(p.outPre + go + orow * p.ldpre_out + n) = t;")
rep("            *(u32x4*)(p.outB + go + orow * p.ldoB + n) = t;", "            if (!(epi_mode & 1)) *(u32x4*)(p.outB + go + orow * p.ldoB + n) = t;")
rep("            *(f32x4*)op = (f32x4){v[0], v[1], v[2], v[3]};\n            *(f32x4*)(op + 4) = (f32x4){v[4], v[5], v[6], v[7]};\n          }\n          if ((SIMPLE || p.outB) && okB) {",
    "            if (!(epi_mode & 1)) { *(f32x4*)op = (f32x4){v[0], v[1], v[2], v[3]};\n            *(f32x4*)(op + 4) = (f32x4){v[4], v[5], v[6], v[7]}; }\n          }\n          if ((SIMPLE || p.outB) && okB) {")
# operand loads: the three fetch forms
rep("      eg[EOP ? i : 0][q] = *(const u32x4*)(esrc + orow * eld + (ncol ? n : 0));",
    "      eg[EOP ? i : 0][q] = (epi_mode & 2) ? (u32x4){0, 0, 0, 0} : *(const u32x4*)(esrc + orow * eld + (ncol ? n : 0));")
rep("        return *(const u32x4*)(esrc + orow * eld + (ncol ? n : 0));", "        return (epi_mode & 2) ? (u32x4){0, 0, 0, 0} : *(const u32x4*)(esrc + orow * eld + (ncol ? n : 0));")
rep("          else if constexpr (EOP) eop = *(const u32x4*)(esrc + orow * eld + n);", "          else if constexpr (EOP) eop = (epi_mode & 2) ? (u32x4){0, 0, 0, 0} : *(const u32x4*)(esrc + orow * eld + n);")
rep("static int check_nt(const GemmArgs& a, int elem) {",
    "extern \"C\" int uvtg_debug_nt_epilogue_mode(int m) { hipLaunchKernelGGL(nt_epi_mode_set_kernel, dim3(1), dim3(1), 0, 0, m); return (int)hipDeviceSynchronize(); }\n"
    "static int check_nt(const GemmArgs& a, int elem) {")
os.makedirs("/tmp/uvtg_nostore", exist_ok=True)
open("/tmp/uvtg_nostore/gemm_n.hip", "w").write(src)
sys.path.insert(0, R)
from univtg_amd import build
build.build()
hipcc = "/opt/rocm/bin/hipcc"
subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed",
                "-I", os.path.join(R, "univtg_amd/csrc"), "-c", "/tmp/uvtg_nostore/gemm_n.hip", "-o", "/tmp/uvtg_nostore/gemm.o"], check=True)
objs = [os.path.join(R, "univtg_amd/csrc/build", f) for f in sorted(os.listdir(os.path.join(R, "univtg_amd/csrc/build"))) if f.endswith(".o") and f != "gemm.o"]
subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(R, "tools/libuvtg_nostore.so"), "/tmp/uvtg_nostore/gemm.o"] + objs, check=True)
print("built tools/libuvtg_nostore.so")
