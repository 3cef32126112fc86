// Dev probe (round 6): ONE resident wave samples (s_memtime, s_memrealtime) pairs every ~4 us while other kernels run: the shader clock the
// chip actually sustains under that load = d(shader cycles) / d(100 MHz real time).  20 VGPRs, no LDS: it stays resident beside the
// persistent GEMM workgroups once it is on a CU (launched first).  tools/clock_probe.py drives it.
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* out, int n) {
  if (threadIdx.x != 0) return;
  for (int i = 0; i < n; i++) {
    const unsigned long long c = __builtin_amdgcn_s_memtime();
    const unsigned long long r = wall_clock64();
    out[2 * i] = c; out[2 * i + 1] = r;
#pragma unroll
    for (int k = 0; k < 4; k++) __builtin_amdgcn_s_sleep(127);
  }
}
extern "C" int clock_probe(void* out, int n, void* stream) {
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out, n);
  return (int)hipGetLastError();
}
