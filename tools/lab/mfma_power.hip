// Dev probe: does the MFMA rate per CU depend on how many CUs run MFMAs (clock / power management)?  Pure register MFMA loop, no memory.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_power tools/mfma_power.hip && tools/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512, 2) void mfma_loop(float* out, int iters) {
  s16x8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int it = 0; it < iters; it++) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
  }
  float t = 0.f;
  for (int r = 0; r < 16; r++) t += c0[r] + c1[r] + c2[r] + c3[r];
  if (t == 123.456f) out[0] = t;
}
int main() {
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;     // 4 MFMAs x 32 cycles x 2 waves per SIMD = 256 cycles per iteration per SIMD -> ~2.1 ms at 2.4 GHz
  for (int rep = 0; rep < 2; rep++)
    for (int cus : {32, 64, 128, 192, 256}) {
      hipLaunchKernelGGL(mfma_loop, dim3(cus), dim3(512), 0, 0, out, 100);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_loop, dim3(cus), dim3(512), 0, 0, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flop = (double)cus * 8 * iters * 4 * 32768.0;
      printf("cus=%3d: %.3f ms  %.2f TF/CU  (%.0f TF/s total; implied clock %.2f GHz)\n", cus, ms, flop / ms / 1e9 / cus, flop / ms / 1e9,
             (double)iters * 256.0 / (ms * 1e-3) / 1e9);
    }
  return 0;
}
