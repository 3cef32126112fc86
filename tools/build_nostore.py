"""Dev tool (round 6): tools/libuvtg_nostore.so = the library with a PATCHED COPY of gemm.hip in which the kernel-level activation code 104 means
"run the whole epilogue but do not issue the fp32 output stores" (next to the existing 100 = main loop only).  It answers WHERE the exposed
epilogue time of the persistent NT GEMM goes -- the store burst or the instruction stream (LDS transpose, bias, packing) -- at kernel level
(tools/nt_epilogue_parts.py: uvtg_linear_bf16, general epilogue, fp32 output).  The mode rides in GemmArgs::act, which is already a kernel
argument: the NT instantiations sit at the 106-SGPR limit, and a first version that kept a mode word live across the tile loop spilled and
returned NaNs (and, with garbage data, ran 10 % "faster": DVFS).  The patch is applied to a copy: the shipped source (and the kernel-source
hash of the PMC summaries) stays untouched; the script refuses a build whose NT kernels use scratch.  Usage: python tools/build_nostore.py"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(R, "univtg_amd/csrc/gemm.hip")).read()
def rep(old, new, cnt=1):
    global src
    assert src.count(old) == cnt, (src.count(old), old[:80])
    src = src.replace(old, new)
rep("            *(f32x4*)op = (f32x4){v[0], v[1], v[2], v[3]};\n            *(f32x4*)(op + 4) = (f32x4){v[4], v[5], v[6], v[7]};\n          }\n          if ((SIMPLE || p.outB) && okB) {",
    "            if (p.act != 104) { *(f32x4*)op = (f32x4){v[0], v[1], v[2], v[3]};\n            *(f32x4*)(op + 4) = (f32x4){v[4], v[5], v[6], v[7]}; }\n          }\n          if ((SIMPLE || p.outB) && okB) {")
# act 105: the epilogue WITHOUT its LDS transpose round trip (the wave-private [32][64] fp32 slab: 32 ds_write_b32 + 8 ds_read_b128 per lane and
# 32-row group): the values stored are garbage (two accumulator registers + the loop counter), the loads / arithmetic / stores are the epilogue's
rep("          for (int r = 0; r < 16; r++)\n            wbuf[((r & 3) + 8 * (r >> 2) + 4 * g) * 64 + j * 32 + l31] = acc[i][j][r];\n        if (GROUPS && i + 1 < TM && i + 1 > NPF) {      // one group ahead",
    "          for (int r = 0; r < 16; r++)\n            if (p.act != 105) wbuf[((r & 3) + 8 * (r >> 2) + 4 * g) * 64 + j * 32 + l31] = acc[i][j][r];\n        if (GROUPS && i + 1 < TM && i + 1 > NPF) {      // one group ahead")
rep("          const f32x4 v0 = *(const f32x4*)(wbuf + row * 64 + c8), v1 = *(const f32x4*)(wbuf + row * 64 + c8 + 4);\n          if (m >= p.M || !ncol) continue;",
    "          f32x4 v0, v1;\n          if (p.act != 105) { v0 = *(const f32x4*)(wbuf + row * 64 + c8); v1 = *(const f32x4*)(wbuf + row * 64 + c8 + 4); }\n"
    "          else { const float t0 = acc[i][0][0] + (float)q, t1 = acc[i][1][5]; v0 = (f32x4){t0, t1, t0, t1}; v1 = (f32x4){t1, t0, t1, t0}; }\n          if (m >= p.M || !ncol) continue;")
os.makedirs("/tmp/uvtg_nostore", exist_ok=True)
open("/tmp/uvtg_nostore/gemm_n.hip", "w").write(src)
sys.path.insert(0, R)
from univtg_amd import build
build.build()
hipcc = "/opt/rocm/bin/hipcc"
subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed",
                "-I", os.path.join(R, "univtg_amd/csrc"), "-c", "/tmp/uvtg_nostore/gemm_n.hip", "-o", "/tmp/uvtg_nostore/gemm.o"], check=True)
objs = [os.path.join(R, "univtg_amd/csrc/build", f) for f in sorted(os.listdir(os.path.join(R, "univtg_amd/csrc/build"))) if f.endswith(".o") and f != "gemm.o"]
out = os.path.join(R, "tools/libuvtg_nostore.so")
subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, "/tmp/uvtg_nostore/gemm.o"] + objs, check=True)
bad = [k["name"] for k in build.kernel_resources(out) if "gemm_nt256" in k["name"] and k["scratch"] > 0]
# the instantiations the kernel-level entry launches without an epilogue operand (EOP = false, third template argument Lb0) must be spill-free;
# tools/nt_epilogue_parts.py also checks act 0 of this build against a float64 product before it times anything
bad_measured = [n for n in bad if "ELb0ELi" in n.split("gemm_nt256_kernelILb")[1][6:14]]
print("built tools/libuvtg_nostore.so; NT instantiations with scratch:", [n[24:70] for n in bad] or "none")
assert not bad_measured, bad_measured
