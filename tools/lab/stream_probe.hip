// Dev probe (round 6): a register-light streaming kernel (<= 32 VGPRs, no LDS) that CAN co-reside with the persistent NT GEMM's 256-row
// workgroups (2 waves per SIMD at <= 240 registers leave 32 per lane) and moves what that launch's epilogues move (bf16 residual in, bf16 out).
// tools/epilogue_overlap_probe.py runs it beside the GEMM's main loop on a second stream: if the pair takes max(t_loop, t_stream) the epilogue
// traffic COULD hide under somebody's MFMAs; if it takes the sum, the memory system is what both wait for.
#include <hip/hip_runtime.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_copy_add_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ out, long long n16, int iters) {
  for (int it = 0; it < iters; it++)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) {
      u32x4 x = a[i], y = b[i];
      out[i] = (u32x4){x[0] ^ y[0], x[1] + y[1], x[2] ^ y[2], x[3] + y[3]};
    }
}
extern "C" int stream_copy_add(const void* a, const void* b, void* out, long long bytes, int iters, int blocks, void* stream) {
  hipLaunchKernelGGL(stream_copy_add_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)a, (const u32x4*)b, (u32x4*)out, bytes / 16, iters);
  return (int)hipGetLastError();
}
