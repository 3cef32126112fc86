// Dev lab (not shipped): standalone A/B bench of NT GEMM structures at the hot-path shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/gemm_lab tools/gemm_lab.hip && tools/gemm_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstring>
#include "../univtg_amd/csrc/uvtg_common.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

struct LabArgs {
  const bf16_t* A; const bf16_t* B; int M, N, K, lda, ldb;
  const float* bias;
  float* outF; bf16_t* outB; int ldo;
  int mode;   // 0: no epilogue (sink), 1: bf16 out, 2: fp32 out, 3: both
  int* counter;   // dynamic tile counter (mode bit 64)
};

// ------------------------------------------------------------------------------------------------
// V1: 256x256x64 tile, 8 waves (2 x 4, each 128 x 64), global_load_lds (16 B) double buffer, one
// barrier per K tile.  LDS image of a stage: A [256][64] bf16 then B [256][64] bf16, rows 128 B,
// 16-byte chunk c of row r stored at chunk position c ^ ((r >> 1) & 7) (source-side permutation).
// ------------------------------------------------------------------------------------------------
template <int WM, int WN>   // wave grid (WM x WN = 8 or 16), tile 256 x 256
__global__ __launch_bounds__(64 * WM * WN) void gemm256_kernel(const LabArgs p) {
  constexpr int BM = 256, BN = 256, BK = 64, NW = WM * WN, PPW = 32 / NW;   // staging pieces (1 KB) per wave per operand
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;   // 32x32 tiles per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 * 64 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, g = lane >> 5, l31 = lane & 31;
  // XCD-aware remap: consecutive logical tiles (sharing the A panel) on one XCD
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tiles_n = (p.N + BN - 1) / BN;
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;

  // staging: each wave issues PPW A pieces + PPW B pieces of 1 KB (8 rows x 128 B) per K tile
  const int sr = lane >> 3, sc = lane & 7;
  const bf16_t* ag[PPW]; const bf16_t* bg[PPW];
#pragma unroll
  for (int i = 0; i < PPW; i++) {
    const int r = (wave * PPW + i) * 8 + sr;
    const int c = (sc ^ ((r >> 1) & 7)) * 8;
    ag[i] = p.A + (size_t)min(m0 + r, p.M - 1) * p.lda + c;
    bg[i] = p.B + (size_t)min(n0 + r, p.N - 1) * p.ldb + c;
  }
  auto stage = [&](int s, int kt) {
    unsigned char* base = smem + s * 65536 + wave * (PPW * 1024);
#pragma unroll
    for (int i = 0; i < PPW; i++) {
      __builtin_amdgcn_global_load_lds((gbl_void*)(ag[i] + kt * BK), (lds_void*)(base + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void*)(bg[i] + kt * BK), (lds_void*)(base + 32768 + i * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // fragment byte offsets (within a stage) for ks = 0; other ks: XOR of the chunk index is applied below
  int aoff[TM], boff[TN], aswz[TM], bswz[TN];
#pragma unroll
  for (int i = 0; i < TM; i++) { const int r = wm * (BM / WM) + i * 32 + l31; aoff[i] = r * 128; aswz[i] = (r >> 1) & 7; }
#pragma unroll
  for (int j = 0; j < TN; j++) { const int r = wn * (BN / WN) + j * 32 + l31; boff[j] = 32768 + r * 128; bswz[j] = (r >> 1) & 7; }

  const int nk = p.K / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; kt++) {
    const int cur = kt & 1;
    __syncthreads();                       // vmcnt(0) + barrier: tile kt landed, buffer cur^1 free
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const unsigned char* base = smem + cur * 65536;
    s16x8 fa[2][TM], fb[2][TN];
#pragma unroll
    for (int i = 0; i < TM; i++) fa[0][i] = *(const s16x8*)(base + aoff[i] + (((g) ^ aswz[i]) << 4));
#pragma unroll
    for (int j = 0; j < TN; j++) fb[0][j] = *(const s16x8*)(base + boff[j] + (((g) ^ bswz[j]) << 4));
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      if (ks < 3) {
#pragma unroll
        for (int i = 0; i < TM; i++) fa[(ks + 1) & 1][i] = *(const s16x8*)(base + aoff[i] + (((2 * ks + 2 + g) ^ aswz[i]) << 4));
#pragma unroll
        for (int j = 0; j < TN; j++) fb[(ks + 1) & 1][j] = *(const s16x8*)(base + boff[j] + (((2 * ks + 2 + g) ^ bswz[j]) << 4));
      }
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = mfma32(fa[ks & 1][i], fb[ks & 1][j], acc[i][j]);
    }
#ifdef SGB
    __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
    for (int ks = 0; ks < 3; ks++) {
#pragma unroll
      for (int n = 0; n < (TM + TN < TM * TN ? TM + TN : TM * TN); n++) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
      if (TM * TN > TM + TN) __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN), 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
#endif
  }
  if (p.mode == 0) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) t += acc[i][j][r];
    if (t == 123.456f) p.outF[0] = t;
    return;
  }
  // epilogue through LDS: per wave a [64][64] fp32 slab (16 KB), TM*TN/4 passes
  __syncthreads();
  float* wbuf = (float*)smem + wave * 4096;
  const int c4 = (lane & 15) * 4;
  constexpr int PI = TM / 2, PJ = TN / 2;    // passes over 64x64 sub-slabs
#pragma unroll
  for (int pi = 0; pi < PI; pi++)
#pragma unroll
    for (int pj = 0; pj < PJ; pj++) {
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int r = 0; r < 16; r++)
            wbuf[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g) * 64 + j * 32 + l31] = acc[pi * 2 + i][pj * 2 + j][r];
      const int n = n0 + wn * (BN / WN) + pj * 64 + c4;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (p.bias && n < p.N) bv = *(const f32x4*)(p.bias + n);
#pragma unroll 4
      for (int it = 0; it < 16; it++) {
        const int row = it * 4 + (lane >> 4);
        const int m = m0 + wm * (BM / WM) + pi * 64 + row;
        f32x4 v = *(const f32x4*)(wbuf + row * 64 + c4);
        if (m >= p.M || n >= p.N) continue;
        v += bv;
        if (p.mode & 2) *(f32x4*)(p.outF + (size_t)m * p.ldo + n) = v;
        if (p.mode & 1) { u32x2 t; t[0] = pack_bf2(v[0], v[1]); t[1] = pack_bf2(v[2], v[3]); *(u32x2*)(p.outB + (size_t)m * p.ldo + n) = t; }
      }
    }
}


// ------------------------------------------------------------------------------------------------
// V2: persistent flat pipeline.  grid = min(tiles, #CU); every block walks tiles bid, bid+G, ...; the K-tile
// stream never drains at tile boundaries: the last K step of a tile prefetches K tile 0 of the next tile
// into the other stage while the epilogue of the finished tile runs out of the stage just consumed
// (8 KB fp32 slab per wave) and its global stores stay in flight behind the next tile's main loop.
// ------------------------------------------------------------------------------------------------
template <int WM, int WN>
__global__ __launch_bounds__(512) void gemm256p_kernel(const LabArgs p) {
  constexpr int BM = 256, BN = 256, BK = 64;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, g = lane >> 5, l31 = lane & 31;
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int ntiles = tiles_m * tiles_n;
  const int nk = p.K / BK;
  const int sr = lane >> 3, sc = lane & 7;

  auto tile_origin = [&](int t, int& m0, int& n0) {
    const int q = ntiles / 8, r = ntiles % 8, xcd = t % 8, idx = t / 8;
    const int l = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    m0 = (l / tiles_n) * BM; n0 = (l % tiles_n) * BN;
  };
  unsigned aofs[4], bofs[4];     // byte offsets of this lane's staging pieces for the tile being loaded
  auto set_offsets = [&](int m0, int n0) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int r = (wave * 4 + i) * 8 + sr;
      const int c = (sc ^ ((r >> 1) & 7)) * 8;
      aofs[i] = ((unsigned)min(m0 + r, p.M - 1) * p.lda + c) * 2;
      bofs[i] = ((unsigned)min(n0 + r, p.N - 1) * p.ldb + c) * 2;
    }
  };
  auto stage = [&](int s, int kt) {
    unsigned char* base = smem + s * 65536 + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      __builtin_amdgcn_global_load_lds((gbl_void*)((const char*)p.A + aofs[i] + kt * (BK * 2)), (lds_void*)(base + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void*)((const char*)p.B + bofs[i] + kt * (BK * 2)), (lds_void*)(base + 32768 + i * 1024), 16, 0, 0);
    }
  };
  int aoff[TM], boff[TN];
  const int swz = (l31 >> 1) & 7;           // same for every 32-row tile of the wave
#pragma unroll
  for (int i = 0; i < TM; i++) aoff[i] = (wm * (BM / WM) + i * 32 + l31) * 128;
#pragma unroll
  for (int j = 0; j < TN; j++) boff[j] = 32768 + (wn * (BN / WN) + j * 32 + l31) * 128;

  __shared__ int s_next;
  const bool dyn = (p.mode & 64) != 0;
  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  int m0, n0;
  tile_origin(tile, m0, n0);
  set_offsets(m0, n0);
  stage(0, 0);
  int it = 0;
  while (true) {
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    int next = tile + gridDim.x;
    int nm0 = 0, nn0 = 0;
    for (int kt = 0; kt < nk; kt++, it++) {
      const int cur = it & 1;
      if (!(p.mode & 32)) __syncthreads();
      if (dyn) {
        if (kt == nk - 2 && tid == 0) s_next = gridDim.x + atomicAdd(p.counter, 1);
        if (kt == nk - 1) next = s_next;
      }
      if (!(p.mode & 16)) {
      if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
      else if (next < ntiles) { tile_origin(next, nm0, nn0); set_offsets(nm0, nn0); stage(cur ^ 1, 0); }
      }
      const unsigned char* base = smem + cur * 65536;
      s16x8 fa[2][TM], fb[2][TN];
#pragma unroll
      for (int i = 0; i < TM; i++) fa[0][i] = *(const s16x8*)(base + aoff[i] + ((g ^ swz) << 4));
#pragma unroll
      for (int j = 0; j < TN; j++) fb[0][j] = *(const s16x8*)(base + boff[j] + ((g ^ swz) << 4));
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        if (ks < 3) {
#pragma unroll
          for (int i = 0; i < TM; i++) fa[(ks + 1) & 1][i] = *(const s16x8*)(base + aoff[i] + (((2 * ks + 2 + g) ^ swz) << 4));
#pragma unroll
          for (int j = 0; j < TN; j++) fb[(ks + 1) & 1][j] = *(const s16x8*)(base + boff[j] + (((2 * ks + 2 + g) ^ swz) << 4));
        }
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++) acc[i][j] = mfma32(fa[ks & 1][i], fb[ks & 1][j], acc[i][j]);
      }
#ifdef SGB
      __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
      for (int ks = 0; ks < 3; ks++) {
#pragma unroll
        for (int n = 0; n < TM + TN; n++) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN), 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
#endif
    }
    // ---- epilogue of `tile` out of the stage consumed last ----
    if ((p.mode & 7) == 0) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) t += acc[i][j][r];
      if (t == 123.456f) p.outF[0] = t;
    } else {
      __builtin_amdgcn_s_barrier();          // every wave is done reading that stage (no vmcnt drain)
      float* wbuf = (float*)(smem + ((it - 1) & 1) * 65536) + wave * 2048;    // [32][64] fp32
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int jp = 0; jp < TN / 2; jp++) {
#pragma unroll
          for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++)
              wbuf[((r & 3) + 8 * (r >> 2) + 4 * g) * 64 + j * 32 + l31] = acc[i][jp * 2 + j][r];
          const int mb = m0 + wm * (BM / WM) + i * 32, nb = n0 + wn * (BN / WN) + jp * 64;
          if (p.mode & 2) {
            const int c4 = (lane & 15) * 4, n = nb + c4;
#pragma unroll
            for (int q = 0; q < 8; q++) {
              const int row = q * 4 + (lane >> 4), m = mb + row;
              f32x4 v = *(const f32x4*)(wbuf + row * 64 + c4);
              if (p.bias && n < p.N) v += *(const f32x4*)(p.bias + n);
              if (m < p.M && n < p.N) { if (p.mode & 8) __builtin_nontemporal_store(v, (f32x4*)(p.outF + (size_t)m * p.ldo + n)); else *(f32x4*)(p.outF + (size_t)m * p.ldo + n) = v; }
            }
          }
          if (p.mode & 1) {
            const int c8 = (lane & 7) * 8, n = nb + c8;
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const int row = q * 8 + (lane >> 3), m = mb + row;
              f32x4 v0 = *(const f32x4*)(wbuf + row * 64 + c8), v1 = *(const f32x4*)(wbuf + row * 64 + c8 + 4);
              if (p.bias && n < p.N) { v0 += *(const f32x4*)(p.bias + n); v1 += *(const f32x4*)(p.bias + n + 4); }
              u32x4 t; t[0] = pack_bf2(v0[0], v0[1]); t[1] = pack_bf2(v0[2], v0[3]); t[2] = pack_bf2(v1[0], v1[1]); t[3] = pack_bf2(v1[2], v1[3]);
              if (m < p.M && n < p.N) { if (p.mode & 8) __builtin_nontemporal_store(t, (u32x4*)(p.outB + (size_t)m * p.ldo + n)); else *(u32x4*)(p.outB + (size_t)m * p.ldo + n) = t; }
            }
          }
          if (p.mode == 4) {
            const int c4 = (lane & 15) * 4;
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 8; q++) s += *(const f32x4*)(wbuf + (q * 4 + (lane >> 4)) * 64 + c4);
            if (s[0] == 123.456f) p.outF[0] = s[1];
          }
        }
    }
    if (next >= ntiles) break;
    tile = next; m0 = nm0; n0 = nn0;
  }
}
template <int WM, int WN> void launch256p(const LabArgs& a, int grid_cap = 256) {
  static bool once = false;
  if (!once) { CK(hipFuncSetAttribute((const void*)gemm256p_kernel<WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); once = true; }
  int grid = ((a.M + 255) / 256) * ((a.N + 255) / 256);
  if (grid > grid_cap) grid = grid_cap;
  hipLaunchKernelGGL((gemm256p_kernel<WM, WN>), dim3(grid), dim3(512), 131072, 0, a);
}

// naive reference
__global__ void ref_kernel(const bf16_t* A, const bf16_t* B, const float* bias, float* C, int M, int N, int K) {
  const int n = blockIdx.x * 16 + threadIdx.x, m = blockIdx.y * 16 + threadIdx.y;
  if (m >= M || n >= N) return;
  float s = 0.f;
  for (int k = 0; k < K; k++) s += bf2f(A[(size_t)m * K + k]) * bf2f(B[(size_t)n * K + k]);
  C[(size_t)m * N + n] = s + (bias ? bias[n] : 0.f);
}
__global__ void fill_kernel(bf16_t* p, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  p[i] = f2bf(((x >> 8) * (1.0f / 8388608.0f)) - 1.0f);
}

template <typename F> float time_us(F f, int n = 20) {
  for (int i = 0; i < 3; i++) f();
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < n; i++) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / n;
}

template <int WM, int WN> void launch256(const LabArgs& a) {
  static bool once = false;
  if (WM * WN > 8 && a.mode != 0) return;
  if (!once) { CK(hipFuncSetAttribute((const void*)gemm256_kernel<WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); once = true; }
  const int grid = ((a.M + 255) / 256) * ((a.N + 255) / 256);
  hipLaunchKernelGGL((gemm256_kernel<WM, WN>), dim3(grid), dim3(64 * WM * WN), 131072, 0, a);
}

int main(int argc, char** argv) {
  if (argc >= 6 && !strcmp(argv[1], "one")) {
    const int M = atoi(argv[2]), N = atoi(argv[3]), K = atoi(argv[4]), mode = atoi(argv[5]);
    bf16_t *A, *B, *Cb; float* C;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 4)); CK(hipMalloc(&Cb, (size_t)M * N * 2));
    fill_kernel<<<(unsigned)(((size_t)M * K + 255) / 256), 256>>>(A, (size_t)M * K, 1);
    fill_kernel<<<(unsigned)(((size_t)N * K + 255) / 256), 256>>>(B, (size_t)N * K, 2);
    LabArgs a{A, B, M, N, K, K, K, nullptr, C, Cb, N, mode, nullptr};
    for (int i = 0; i < 5; i++) launch256p<2, 4>(a);
    CK(hipDeviceSynchronize());
    return 0;
  }
  if (argc >= 6 && !strcmp(argv[1], "time")) {      // time M N K mode [mode ...] : persistent 2x4 kernel only
    const int M = atoi(argv[2]), N = atoi(argv[3]), K = atoi(argv[4]);
    bf16_t *A, *B, *Cb; float* C;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 4)); CK(hipMalloc(&Cb, (size_t)M * N * 2));
    fill_kernel<<<(unsigned)(((size_t)M * K + 255) / 256), 256>>>(A, (size_t)M * K, 1);
    fill_kernel<<<(unsigned)(((size_t)N * K + 255) / 256), 256>>>(B, (size_t)N * K, 2);
    const double fl = 2.0 * M * N * K;
    for (int i = 5; i < argc; i++) {
      LabArgs a{A, B, M, N, K, K, K, nullptr, C, Cb, N, atoi(argv[i]), nullptr};
      const float t = time_us([&] { launch256p<2, 4>(a); });
      printf("%dx%dx%d mode %3d: %8.1f us %7.1f TF\n", M, N, K, a.mode, t, fl / t / 1e6);
    }
    return 0;
  }
  // ---- correctness at an awkward shape ----
  {
    const int M = 700, N = 520, K = 192;
    bf16_t *A, *B; float *C, *R, *bias; bf16_t* Cb;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 4));
    CK(hipMalloc(&R, (size_t)M * N * 4)); CK(hipMalloc(&Cb, (size_t)M * N * 2)); CK(hipMalloc(&bias, N * 4));
    fill_kernel<<<(M * K + 255) / 256, 256>>>(A, (size_t)M * K, 1); fill_kernel<<<(N * K + 255) / 256, 256>>>(B, (size_t)N * K, 2);
    std::vector<float> hb(N); for (int i = 0; i < N; i++) hb[i] = 0.01f * i;
    CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
    ref_kernel<<<dim3((N + 15) / 16, (M + 15) / 16), dim3(16, 16)>>>(A, B, bias, R, M, N, K);
    LabArgs a{A, B, M, N, K, K, K, bias, C, Cb, N, 3, nullptr};
    CK(hipMemset(C, 0, (size_t)M * N * 4));
    launch256<2, 4>(a);
    CK(hipDeviceSynchronize());
    std::vector<float> hc((size_t)M * N), hr((size_t)M * N);
    CK(hipMemcpy(hc.data(), C, hc.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hr.data(), R, hr.size() * 4, hipMemcpyDeviceToHost));
    double mx = 0; for (size_t i = 0; i < hc.size(); i++) mx = fmax(mx, fabs(hc[i] - hr[i]));
    printf("check 2x4 M=%d N=%d K=%d: max abs err %.3e\n", M, N, K, mx);
    CK(hipMemset(C, 0, (size_t)M * N * 4));
    launch256<4, 2>(a);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hc.data(), C, hc.size() * 4, hipMemcpyDeviceToHost));
    mx = 0; for (size_t i = 0; i < hc.size(); i++) mx = fmax(mx, fabs(hc[i] - hr[i]));
    printf("check 4x2: max abs err %.3e\n", mx);
    for (int cap : {256, 3}) {
      CK(hipMemset(C, 0, (size_t)M * N * 4)); CK(hipMemset(Cb, 0, (size_t)M * N * 2));
      launch256p<2, 4>(a, cap);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(hc.data(), C, hc.size() * 4, hipMemcpyDeviceToHost));
      std::vector<bf16_t> hb2((size_t)M * N);
      CK(hipMemcpy(hb2.data(), Cb, hb2.size() * 2, hipMemcpyDeviceToHost));
      mx = 0; double mxb = 0;
      for (size_t i = 0; i < hc.size(); i++) { mx = fmax(mx, fabs(hc[i] - hr[i])); unsigned u = (unsigned)hb2[i] << 16; float f; memcpy(&f, &u, 4); mxb = fmax(mxb, fabs(f - hr[i])); }
      printf("check persistent 2x4 cap %d: max abs err fp32 %.3e bf16 %.3e\n", cap, mx, mxb);
    }
    hipFree(A); hipFree(B); hipFree(C); hipFree(R); hipFree(Cb); hipFree(bias);
  }
  // ---- timing ----
  const int shapes[][3] = {{27392, 1024, 1024}, {27392, 2048, 1024}, {27392, 3072, 1024}, {27392, 1024, 3072}, {19200, 2048, 3072}, {4096, 4096, 4096}, {8192, 8192, 8192}};
  for (auto& s : shapes) {
    const int M = s[0], N = s[1], K = s[2];
    bf16_t *A, *B, *Cb; float* C;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 4)); CK(hipMalloc(&Cb, (size_t)M * N * 2));
    fill_kernel<<<(unsigned)(((size_t)M * K + 255) / 256), 256>>>(A, (size_t)M * K, 1);
    fill_kernel<<<(unsigned)(((size_t)N * K + 255) / 256), 256>>>(B, (size_t)N * K, 2);
    const double fl = 2.0 * M * N * K;
    int* counter; CK(hipMalloc(&counter, 4));
    for (int mode : {0, 64, 1, 65, 2, 66}) {
      LabArgs a{A, B, M, N, K, K, K, nullptr, C, Cb, N, mode, counter};
      float t1 = mode >= 3 ? 0.f : time_us([&] { launch256<2, 4>(a); });
      if (mode == 0) { float t16 = time_us([&] { launch256<4, 4>(a); }); printf("%5dx%4dx%4d mode 0: V1 4x4 (16 waves) %8.1f us %7.1f TF\n", M, N, K, t16, fl / t16 / 1e6); }
      float t3 = time_us([&] { if (a.mode & 64) hipMemsetAsync(counter, 0, 4, 0); launch256p<2, 4>(a); });
      float t4 = time_us([&] { if (a.mode & 64) hipMemsetAsync(counter, 0, 4, 0); launch256p<4, 2>(a); });
      printf("%5dx%4dx%4d mode %d: V1 2x4 %8.1f us %7.1f TF | P 2x4 %8.1f us %7.1f TF | P 4x2 %8.1f us %7.1f TF\n", M, N, K, mode, t1, fl / t1 / 1e6, t3, fl / t3 / 1e6, t4, fl / t4 / 1e6);
    }
    hipFree(A); hipFree(B); hipFree(C); hipFree(Cb);
  }
  return 0;
}
