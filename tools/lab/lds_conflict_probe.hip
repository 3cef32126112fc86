// Dev probe (not shipped): cycles per ds_read_b64_tr_b16 / ds_read_b128 for the attention kernels' LDS layouts of a [rows][128] bf16 tile.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/lds_conflict_probe tools/lds_conflict_probe.hip && tools/lds_conflict_probe
// Layouts: 0 = rows padded to 136 elements (68 dwords: what the backward kernels use), 1 = rows padded to 160 elements (80 dwords: the
// forward kernel's V tile), 2 = unpadded rows with 16-byte chunk c stored at c ^ (4 (r & 3) + ((r >> 2) & 3)).
// Access patterns (the kernels' own): "tr" = the A fragment of dV^T += dO^T P (32-lane group: 4 rows x 64 B), "row" = the A fragment of
// S = Q K^T (lane = row, one 16-byte chunk).  One wave and four waves (one per SIMD) are timed: the second shows the LDS pipe shared.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

template <int LAYOUT> __device__ __forceinline__ int off(int r, int col) {
  if constexpr (LAYOUT == 0) return r * 136 + col;
  else if constexpr (LAYOUT == 1) return r * 160 + col;
  else return r * 128 + ((((col >> 3) ^ (((r & 3) << 2) | ((r >> 2) & 3)))) << 3) + (col & 7);
}

template <int LAYOUT, bool TR>
__global__ __launch_bounds__(256) void probe(int iters, long long* cycles, int* sink) {
  __shared__ __attribute__((aligned(16))) short tile[64 * 160];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 5, l31 = lane & 31, i16 = lane & 15, qd = (lane >> 4) & 1;
  for (int i = tid; i < 64 * 160; i += blockDim.x) tile[i] = (short)i;
  __syncthreads();
  int acc = 0;
  const long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
    int zero = 0;
    asm volatile("" : "+v"(zero));       // opaque per-iteration offset: the reads stay inside the loop
    const short* tl = tile + zero;
    s16x4 ta[16]; s16x8 ra[16];
#pragma unroll
    for (int hf = 0; hf < 2; hf++)
#pragma unroll
      for (int blk = 0; blk < 4; blk++) {
        const int n = (hf * 4 + blk) * 2;
        if constexpr (TR) {
          const int qr = 16 * hf + 4 * g + (i16 >> 2), col = blk * 32 + 16 * qd + 4 * (i16 & 3);
          ta[n] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tl + off<LAYOUT>(qr, col)));
          ta[n + 1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tl + off<LAYOUT>(qr + 8, col)));
        } else {
          ra[n] = *(const s16x8*)(tl + off<LAYOUT>(l31, 16 * (2 * blk + hf) + 8 * g));
          ra[n + 1] = *(const s16x8*)(tl + off<LAYOUT>(32 + l31, 16 * (2 * blk + hf) + 8 * g));
        }
      }
#pragma unroll
    for (int n = 0; n < 16; n++) {       // all 16 reads are in flight before the first is consumed
      if constexpr (TR) { asm volatile("" :: "v"(ta[n])); acc += ta[n][0]; }
      else { asm volatile("" :: "v"(ra[n])); acc += ra[n][0]; }
    }
  }
  const long long t1 = clock64();
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
  if (acc == 0x7fffffff) sink[0] = acc;
}

template <int LAYOUT, bool TR> static void run(const char* name, long long* dcyc, int* sink) {
  for (int threads : {64, 256}) {
    const int iters = 2000;
    hipLaunchKernelGGL((probe<LAYOUT, TR>), dim3(1), dim3(threads), 0, 0, 10, dcyc, sink);
    hipLaunchKernelGGL((probe<LAYOUT, TR>), dim3(1), dim3(threads), 0, 0, iters, dcyc, sink);
    CK(hipDeviceSynchronize());
    long long c; CK(hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost));
    printf("%-28s layout %d  %d wave(s): %7.2f clock64 ticks per wave-instruction (16 per iteration and wave)\n", name, LAYOUT, threads / 64, (double)c / iters / 16);
  }
}

int main() {
  long long* dcyc; int* sink; CK(hipMalloc(&dcyc, 64)); CK(hipMalloc(&sink, 64));
  run<0, true>("ds_read_b64_tr_b16 (4 x 64 B)", dcyc, sink);
  run<1, true>("ds_read_b64_tr_b16 (4 x 64 B)", dcyc, sink);
  run<2, true>("ds_read_b64_tr_b16 (4 x 64 B)", dcyc, sink);
  run<0, false>("ds_read_b128 (lane = row)", dcyc, sink);
  run<1, false>("ds_read_b128 (lane = row)", dcyc, sink);
  run<2, false>("ds_read_b128 (lane = row)", dcyc, sink);
  return 0;
}
