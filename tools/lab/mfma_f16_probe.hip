// Does v_mfma_f32_32x32x16_f16 on gfx950 keep fp16 SUBNORMAL inputs (needed by the fp16 hi/lo operand split of the precise GEMM mode:
// the lo image of an O(0.1) value is ~1e-5, below fp16's smallest normal 6.1e-5)?  Prints the products of subnormal x normal operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void probe(float a_val, float b_val, float* out) {
  half8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
  f32x16 c;
  for (int i = 0; i < 16; i++) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
  float* d; hipMalloc(&d, 4);
  const float cases[][2] = {{ldexpf(1.f, -20), 1024.f}, {ldexpf(1.f, -24), 1024.f}, {ldexpf(3.f, -24), 4096.f}, {1024.f, ldexpf(1.f, -20)},
                            {ldexpf(1.f, -14), 1.f}, {ldexpf(1.f, -15), 1.f}, {ldexpf(1.f, -20), ldexpf(1.f, -4)}};
  for (auto& c : cases) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, c[0], c[1], d);
    float h = -1.f; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    const double want = 16.0 * (double)c[0] * (double)c[1];
    printf("a=%.6g b=%.6g  mfma16 sum=%.9g  exact=%.9g  %s\n", c[0], c[1], h, want, fabs(h - want) <= 1e-6 * fabs(want) ? "KEPT" : "FLUSHED/LOST");
  }
  return 0;
}
