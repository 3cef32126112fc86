"""Dev tool (round 6): tools/libuvtg_srdbg.so = the library with a PATCHED COPY of misc.hip whose saliency_rows_kernel takes a mode word from UVTG_SR_DBG
(UVTG_DEV_ENV=1): 1 = clip chunks only, 2 = text chunks only, 3 = clip rows without the x0 / pooled reads (gs treated as 0), 4 = no stores,
5 = no per-row scalar reads (g_sal / vnorm / cosv / vout_map replaced by constants).  Garbage results in every mode but 0; only the kernel's
duration in a kernel trace of the bench step is read (profiles/r06_saliency_rows_phases.txt).  Usage: python tools/build_srdbg.py"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(R, "univtg_amd/csrc/misc.hip")).read()
def rep(old, new, cnt=1):
    global src
    assert src.count(old) == cnt, (src.count(old), old[:80])
    src = src.replace(old, new)
rep("__global__ __launch_bounds__(256) void saliency_rows_kernel(const SaliencyArgs a) {", "__global__ __launch_bounds__(256) void saliency_rows_kernel(const SaliencyArgs a, const int dbg) {")
rep("  const int s_end = chunk < nvc ? min(a.Lv, chunk * SAL_CHUNK + SAL_CHUNK) : (SAL_TXT ? min(a.S, s_begin + SAL_TXT) : a.S);\n",
    "  const int s_end = chunk < nvc ? min(a.Lv, chunk * SAL_CHUNK + SAL_CHUNK) : (SAL_TXT ? min(a.S, s_begin + SAL_TXT) : a.S);\n  if ((dbg == 1 && chunk >= nvc) || (dbg == 2 && chunk < nvc)) return;\n")
rep("      const float gs = a.g_sal ? a.g_sal[b * a.Lv + t] : 0.f;\n      const float vn = fmaxf(a.vnorm[b * a.Lv + t], 1e-8f), cs = a.cosv[b * a.Lv + t];\n      const int vo = a.vout_map ? a.vout_map[b * a.Lv + t] : b * a.Lv + t;",
    "      const float gs = dbg == 3 ? 0.f : (dbg == 5 ? 0.25f : (a.g_sal ? a.g_sal[b * a.Lv + t] : 0.f));\n      const float vn = dbg == 5 ? 1.5f : fmaxf(a.vnorm[b * a.Lv + t], 1e-8f), cs = dbg == 5 ? 0.3f : a.cosv[b * a.Lv + t];\n      const int vo = (a.vout_map && dbg != 5) ? a.vout_map[b * a.Lv + t] : b * a.Lv + t;")
rep("        u32x2 o; o[0] = pack_bf2(g[0], g[1]); o[1] = pack_bf2(g[2], g[3]);\n        *(u32x2*)(out + c) = o;",
    "        u32x2 o; o[0] = pack_bf2(g[0], g[1]); o[1] = pack_bf2(g[2], g[3]);\n        if (dbg != 4 || o[0] == 0x12345678u) *(u32x2*)(out + c) = o;", cnt=2)
rep("  if (a.d == 1024) hipLaunchKernelGGL(saliency_rows_kernel<4>, grid, dim3(256), 0, s, a);\n  else if (a.d == 512) hipLaunchKernelGGL(saliency_rows_kernel<2>, grid, dim3(256), 0, s, a);",
    "  static const int dbg = uvtg_dev_env(\"UVTG_SR_DBG\") ? atoi(uvtg_dev_env(\"UVTG_SR_DBG\")) : 0;\n  if (a.d == 1024) hipLaunchKernelGGL(saliency_rows_kernel<4>, grid, dim3(256), 0, s, a, dbg);\n  else if (a.d == 512) hipLaunchKernelGGL(saliency_rows_kernel<2>, grid, dim3(256), 0, s, a, dbg);")
src = src.replace("hipLaunchKernelGGL(saliency_rows_kernel<1>, grid, dim3(256), 0, s, a);", "hipLaunchKernelGGL(saliency_rows_kernel<1>, grid, dim3(256), 0, s, a, 0);")
os.makedirs("/tmp/uvtg_srdbg", exist_ok=True)
open("/tmp/uvtg_srdbg/misc_d.hip", "w").write(src)
sys.path.insert(0, R)
from univtg_amd import build
build.build()
hipcc = "/opt/rocm/bin/hipcc"
subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed",
                "-I", os.path.join(R, "univtg_amd/csrc"), "-I", os.path.join(R, "include"), "-c", "/tmp/uvtg_srdbg/misc_d.hip", "-o", "/tmp/uvtg_srdbg/misc.o"], check=True)
objs = [os.path.join(R, "univtg_amd/csrc/build", f) for f in sorted(os.listdir(os.path.join(R, "univtg_amd/csrc/build"))) if f.endswith(".o") and f != "misc.o"]
out = os.path.join(R, "tools/libuvtg_srdbg.so")
subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, "/tmp/uvtg_srdbg/misc.o"] + objs, check=True)
k = [k for k in build.kernel_resources(out) if "saliency_rows_kernel" in k["name"]]
print("built tools/libuvtg_srdbg.so;", [(x["name"][-30:], x["vgpr"], x["scratch"]) for x in k])
