"""Dev tool (round 6): tools/libuvtg_hsdbg.so = the library with a PATCHED COPY of misc.hip whose fused head pass (heads_saliency_fwd_kernel) takes a
mode word from UVTG_HS_DBG (UVTG_DEV_ENV=1): 1 = return after the text pooling, 2 = clip walk without the cosine's x0 row reads, 3 = clip walk
without the hidden-row reads, 4 = no pooling (zeros) -- where do the kernel's 83 us go?  Results are garbage in every mode but 0; only the
kernel's duration in a rocprofv3 kernel trace of the bench step is read.  The shipped source stays untouched.  Usage: python tools/build_hsdbg.py"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(R, "univtg_amd/csrc/misc.hip")).read()
def rep(old, new, cnt=1):
    global src
    assert src.count(old) == cnt, (src.count(old), old[:80])
    src = src.replace(old, new)
rep("__global__ __launch_bounds__(512) void heads_saliency_fwd_kernel(const HeadsFinalArgs h, const SaliencyArgs a) {",
    "__global__ __launch_bounds__(512) void heads_saliency_fwd_kernel(const HeadsFinalArgs h, const SaliencyArgs a, const int dbg) {")
rep("  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, d = a.d;\n  const float* xt = a.x0 + ((size_t)b * a.S + a.Lv) * d;      // text rows\n  for (int t = wave; t < a.Lt; t += 8) {",
    "  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, d = a.d;\n  const float* xt = a.x0 + ((size_t)b * a.S + a.Lv) * d;      // text rows\n  for (int t = wave; t < (dbg == 4 ? 0 : a.Lt); t += 8) {")
rep("  for (int c = tid; c < d; c += 512) {\n    float acc = 0.f;\n#pragma unroll 8\n    for (int t = 0; t < a.Lt; t++) acc += s_alpha[t] * xt[(size_t)t * d + c];",
    "  for (int c = tid; c < d; c += 512) {\n    float acc = 0.f;\n#pragma unroll 8\n    for (int t = 0; t < (dbg == 4 ? 0 : a.Lt); t++) acc += s_alpha[t] * xt[(size_t)t * d + c];")
rep("  if (tid == 0 && a.qnorm) a.qnorm[b] = qn;\n  // ---- the clips of this sample",
    "  if (tid == 0 && a.qnorm) a.qnorm[b] = qn;\n  if (dbg == 1) return;\n  // ---- the clips of this sample")
rep("    for (int c = lane * 4; c < d; c += 256) {\n      const f32x4 x = *(const f32x4*)(v + c), y = *(const f32x4*)(s_pool + c);\n      dot += x[0] * y[0]",
    "    for (int c = lane * 4; c < (dbg == 2 ? 0 : d); c += 256) {\n      const f32x4 x = *(const f32x4*)(v + c), y = *(const f32x4*)(s_pool + c);\n      dot += x[0] * y[0]", cnt=1)
rep("    if (framed) load_row(fs + t + 2, rs[2], rc[2]);", "    if (framed && dbg != 3) load_row(fs + t + 2, rs[2], rc[2]);")
rep("  if (a.d == 1024) hipLaunchKernelGGL(heads_saliency_fwd_kernel<2>, dim3(a.B, ny), dim3(512), sh, s, h, a);\n  else hipLaunchKernelGGL(heads_saliency_fwd_kernel<1>, dim3(a.B, ny), dim3(512), sh, s, h, a);",
    "  static const int dbg = uvtg_dev_env(\"UVTG_HS_DBG\") ? atoi(uvtg_dev_env(\"UVTG_HS_DBG\")) : 0;\n  if (a.d == 1024) hipLaunchKernelGGL(heads_saliency_fwd_kernel<2>, dim3(a.B, ny), dim3(512), sh, s, h, a, dbg);\n  else hipLaunchKernelGGL(heads_saliency_fwd_kernel<1>, dim3(a.B, ny), dim3(512), sh, s, h, a, dbg);")
os.makedirs("/tmp/uvtg_hsdbg", exist_ok=True)
open("/tmp/uvtg_hsdbg/misc_d.hip", "w").write(src)
sys.path.insert(0, R)
from univtg_amd import build
build.build()
hipcc = "/opt/rocm/bin/hipcc"
subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed",
                "-I", os.path.join(R, "univtg_amd/csrc"), "-I", os.path.join(R, "include"), "-c", "/tmp/uvtg_hsdbg/misc_d.hip", "-o", "/tmp/uvtg_hsdbg/misc.o"], check=True)
objs = [os.path.join(R, "univtg_amd/csrc/build", f) for f in sorted(os.listdir(os.path.join(R, "univtg_amd/csrc/build"))) if f.endswith(".o") and f != "misc.o"]
out = os.path.join(R, "tools/libuvtg_hsdbg.so")
subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, "/tmp/uvtg_hsdbg/misc.o"] + objs, check=True)
k = [k for k in build.kernel_resources(out) if "heads_saliency_fwd" in k["name"]]
print("built tools/libuvtg_hsdbg.so;", [(x["name"][-40:], x["vgpr"], x["scratch"]) for x in k])
