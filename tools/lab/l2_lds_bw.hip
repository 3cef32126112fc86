// Dev probe (not shipped): how many bytes per second can ONE CU pull out of L2 / HBM with the staging pattern of the persistent NT GEMM?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/l2_lds_bw tools/l2_lds_bw.hip && tools/l2_lds_bw
// Every workgroup (512 threads, one per CU) streams "K tiles" of its own 256-row operand panel: per K tile each wave issues NPW loads of 1 KB
// (8 rows x 128 B, 16 B per lane, rows `ld` bytes apart) -- exactly the address pattern of gemm_nt256's pieces.  Variants:
//   lds : global_load_lds (LDS-DMA) into a ring of DEPTH K-tile slots, `s_waitcnt vmcnt(NPW * (DEPTH - 1))` per K tile, no barrier
//   vgpr: global_load_dwordx4 into registers (consumed by an opaque asm), same counts
// The panel is re-read `reps` times, so that the second and later passes come from L2 / MALL (panel = 256 rows x 2 KB = 512 KB per workgroup,
// 128 MB for 256 workgroups; with `share` workgroups per panel the footprint shrinks accordingly, like the N tiles of one A row panel).
// The DESIGN.md section 6 reading this checks: the K loop's operand stream alone runs at ~50 GB/s per CU (24 B/clk) whatever the prefetch depth.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int NPW, int DEPTH, bool TO_LDS>
__global__ __launch_bounds__(512) void stream_kernel(const char* base, int ld, int ktiles, int reps, int share, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sr = lane >> 3, sc = lane & 7;
  const char* panel = base + (size_t)(blockIdx.x / share) * 256 * ld;
  unsigned rowoff[NPW];
#pragma unroll
  for (int i = 0; i < NPW; i++) {
    const int r = ((wave * NPW + i) * 8 + sr) & 255;
    rowoff[i] = (unsigned)r * ld + ((sc ^ ((r >> 1) & 7)) << 4);
  }
  unsigned acc = 0;
  const int total = ktiles * reps;
  auto issue = [&](int t) {
    const int kt = t % ktiles;
    unsigned char* slot = smem + (t % DEPTH) * (NPW * 8 * 1024) + wave * NPW * 1024;
#pragma unroll
    for (int i = 0; i < NPW; i++) {
      const char* src = panel + rowoff[i] + (size_t)kt * 128;
      if constexpr (TO_LDS) __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(slot + i * 1024), 16, 0, 0);
      else { u32x4 v = *(const u32x4*)src; asm volatile("" :: "v"(v)); acc += 1; }
    }
  };
  for (int t = 0; t < DEPTH - 1 && t < total; t++) issue(t);
  for (int t = 0; t < total; t++) {
    if (t + DEPTH - 1 < total) issue(t + DEPTH - 1);
    if constexpr (TO_LDS) {
      if constexpr (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if constexpr (NPW * (DEPTH - 1) <= 63) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPW * (DEPTH - 1)) : "memory");
      else asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0xffffffffu) sink[0] = acc;
}

template <int NPW, int DEPTH, bool TO_LDS>
static void run(const char* buf, int ld, int ktiles, int reps, int share, int grid, unsigned* sink, const char* name) {
  const size_t smem = TO_LDS ? (size_t)DEPTH * NPW * 8 * 1024 : 0;
  if (smem > 160 * 1024) return;
  CK(hipFuncSetAttribute((const void*)stream_kernel<NPW, DEPTH, TO_LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL((stream_kernel<NPW, DEPTH, TO_LDS>), dim3(grid), dim3(512), smem, 0, buf, ld, ktiles, reps, share, sink);
  CK(hipEventRecord(e0));
  const int n = 5;
  for (int i = 0; i < n; i++) hipLaunchKernelGGL((stream_kernel<NPW, DEPTH, TO_LDS>), dim3(grid), dim3(512), smem, 0, buf, ld, ktiles, reps, share, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= n;
  const double bytes = (double)grid * ktiles * reps * NPW * 8 * 1024;
  printf("%-5s pieces/wave %d depth %d grid %3d share %d: %8.1f us  %7.2f TB/s  = %6.1f GB/s per CU\n", name, NPW, DEPTH, grid, share, ms * 1e3, bytes / ms / 1e9,
         bytes / ms / 1e6 / grid);
}

// FETCH_SIZE calibration (tools/fetch_calib.sh): ONE launch per pattern over a known byte count, every byte read exactly once from a
// buffer far larger than the Infinity Cache is hot with: the ratio known bytes / (FETCH_SIZE x 1024) is the correction factor of that pattern
__global__ __launch_bounds__(256) void calib_linear_kernel(const u32x4* src, size_t n16, unsigned* sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { u32x4 v = src[i]; asm volatile("" :: "v"(v)); acc += 1; }
  if (acc == 0xffffffffu) sink[0] = acc;
}

int main(int argc, char** argv) {
  const int ld = 2048, ktiles = 16, reps = 8;          // 1024 bf16 columns per row, K tiles of 64 columns
  char* buf; unsigned* sink;
  CK(hipMalloc(&buf, (size_t)256 * 256 * ld + 4096)); CK(hipMemset(buf, 1, (size_t)256 * 256 * ld + 4096)); CK(hipMalloc(&sink, 64));
  if (argc > 1 && !strcmp(argv[1], "calib")) {
    // three kernels, one launch each, 128 MiB each (the whole buffer, every byte once): (1) the NT GEMM's staging pattern via LDS-DMA (8 rows x
    // 128 B per wave-instruction, rows 2 KB apart; 4 pieces per wave = 256 distinct rows per K tile), (2) the same addresses into registers,
    // (3) a plain linear 16 B / lane stream (the guide's calibrated case: factor 2)
    const size_t bytes = (size_t)256 * ktiles * 4 * 8 * 1024;
    CK(hipFuncSetAttribute((const void*)stream_kernel<4, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 4 * 8 * 1024));
    hipLaunchKernelGGL((stream_kernel<4, 2, true>), dim3(256), dim3(512), 2 * 4 * 8 * 1024, 0, buf, ld, ktiles, 1, 1, sink);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((stream_kernel<4, 2, false>), dim3(256), dim3(512), 0, 0, buf, ld, ktiles, 1, 1, sink);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(calib_linear_kernel, dim3(2048), dim3(256), 0, 0, (const u32x4*)buf, bytes / 16, sink);
    CK(hipDeviceSynchronize());
    printf("calib: %zu bytes per kernel (stream_kernel<4,2,true>, stream_kernel<4,2,false>, calib_linear_kernel)\n", bytes);
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "ld")) {
    // row-stride sweep (round 4): does the 2 KB stride of a K = 1024 bf16 operand (every row of a tile on the same few L2 channels?) cost
    // delivery bandwidth?  Same pattern, rows `ld` bytes apart; buffer sized for the largest stride
    char* big; CK(hipMalloc(&big, (size_t)256 * 256 * 8192 + 4096)); CK(hipMemset(big, 1, (size_t)256 * 256 * 8192 + 4096));
    for (int l : {2048, 2048 + 64, 2048 + 128, 2048 + 256, 2048 + 512, 4096, 4096 + 128, 6144, 6144 + 128, 8192 - 128}) {
      printf("row stride %d B\n", l);
      for (int share : {1, 4}) {
        run<8, 1, true>(big, l, ktiles, reps, share, 256, sink, "lds");
        run<8, 2, true>(big, l, ktiles, reps, share, 256, sink, "lds");
      }
    }
    return 0;
  }
  for (int grid : {256, 64, 8}) {
    for (int share : {1, 4}) {
      run<8, 1, true>(buf, ld, ktiles, reps, share, grid, sink, "lds");
      run<8, 2, true>(buf, ld, ktiles, reps, share, grid, sink, "lds");
      run<4, 4, true>(buf, ld, ktiles * 2, reps, share, grid, sink, "lds");
      run<8, 2, false>(buf, ld, ktiles, reps, share, grid, sink, "vgpr");
      run<8, 4, false>(buf, ld, ktiles, reps, share, grid, sink, "vgpr");
    }
  }
  return 0;
}
