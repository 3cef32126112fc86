// Dev tool: occupy `blocks` CUs for `usec` microseconds on a stream (emulates resident RCCL workgroups next to the training step).
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/libcuhog.so tools/cu_hog.hip
#include <hip/hip_runtime.h>
__global__ void hog_kernel(long long ticks) {
  extern __shared__ char lds[];
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(32); }
  if (ticks < 0) lds[threadIdx.x] = 0;
}
extern "C" int cu_hog(int blocks, int usec, void* stream) {
  static bool once = false;
  if (!once) { hipFuncSetAttribute((const void*)hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024); once = true; }
  hipLaunchKernelGGL(hog_kernel, dim3(blocks), dim3(256), 100 * 1024, (hipStream_t)stream, (long long)usec * 100);   // wall_clock64: 100 MHz
  return (int)hipGetLastError();
}
