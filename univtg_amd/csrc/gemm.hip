// MFMA GEMMs for gfx950 (CDNA4): 64-wide waves, v_mfma_f32_32x32x16_bf16, LDS-staged swizzled tiles.
//
//  gemm_nt : C[M,N] = A[M,K] * B[N,K]^T  + fused epilogue.  Every Linear / Conv1d(k=3) forward and
//            every dgrad of the hot path (model/univtg.py:399-406,375-382;
//            model/transformer_encoder_droppath.py:117-125) runs through it.
//            bf16: operands bf16, persistent (64 TM) x 256 x 64 tiles with LDS-DMA staging (large shapes) or 128x128x64
//            register-staged tiles (small shapes), ds_read_b128 fragments.
//            split-operand ("fp32x3") variant of both: operand rows hold fp16 hi / lo images interleaved in 32-column blocks
//            (uvtg_common.h), one staged 64-element K tile yields hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16
//            (~2^-22 relative: fp32-class accuracy at 3 MFMAs per product).
//  gemm_tn : C[N,K] += P[M,N]^T * Q[M,K] (weight gradients): both operands are row-major in the
//            reduction dimension, fragments come from ds_read_b64_tr_b16 transposing LDS reads.
#include "uvtg_kernels.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace {

constexpr int BM = 128, BN = 128;

template <int BK> struct Swz {
  static constexpr int CPR = BK * 2 / 16;                      // 16-byte chunks per tile row
  static constexpr int RPB = (256 / (BK * 2)) > 0 ? (256 / (BK * 2)) : 1;   // rows per 256-B bank row
  __device__ static __forceinline__ int f(int r) { return (r / RPB) % CPR; }
};

__device__ __forceinline__ int map_row(int m, int seg, int stride, int off) {
  return seg ? (m / seg) * stride + (m % seg) + off : m + off;
}

// HALF: split-operand precise mode -- rows hold fp16 hi / lo images interleaved in 32-column blocks (uvtg_common.h, split_col): a 64-element
// K tile is 32 real columns, k-steps 0, 1 = hi halves, 2, 3 = lo halves; products hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16.
template <bool HALF>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmArgs p) {
#pragma clang fp contract(off)            // (epilogue rounding identical to the persistent kernel's)
  constexpr int BK = 64;
  constexpr int TILE = BM * BK;
  constexpr int NT = 2;                     // operand tiles per stage (A, B)
  using SW = Swz<BK>;
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * NT * TILE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, g = lane >> 5, l31 = lane & 31;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int gz = blockIdx.z;

  const char* Ab = (const char*)((p.A2 && n0 >= p.a2_n0) ? p.A2 : p.A) + (size_t)gz * p.gA * 2;
  const char* Bb = (const char*)p.B + (size_t)gz * p.gB * 2;

  // per-thread staging coordinates: 4 x 16-byte pieces of A and of B per K tile
  int srow[4], sch[4];
  size_t aoff[4], boff[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int q = tid + 256 * i;
    srow[i] = q >> 3; sch[i] = q & 7;
    int m = min(m0 + srow[i], p.M - 1);
    int n = min(n0 + srow[i], p.N - 1);
    aoff[i] = (size_t)max(map_row(m, p.a_seg, p.a_seg_stride, p.a_off), 0) * p.lda;
    boff[i] = (size_t)n * p.ldb;
  }
  const int nk = (p.K + BK - 1) / BK;

  u32x4 ra[4], rb[4];
  auto gload = [&](int kt) {
    const int k0 = kt * BK;
    const int tap = k0 / p.ktap, kk = k0 - tap * p.ktap;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int kc = sch[i] * 8;
      const bool ok = (k0 + kc) < p.K;
      u32x4 z = {0, 0, 0, 0};
      ra[i] = ok ? *(const u32x4*)(Ab + ((aoff[i] + (size_t)tap * p.lda + kk + kc) * 2)) : z;
      rb[i] = ok ? *(const u32x4*)(Bb + ((boff[i] + k0 + kc) * 2)) : z;
    }
  };
  auto sstore = [&](int stage) {
    bf16_t* base = smem + stage * NT * TILE;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int r = srow[i];
      const int off = r * BK + ((sch[i] ^ SW::f(r)) * 8);
      *(u32x4*)(base + off) = ra[i];
      *(u32x4*)(base + TILE + off) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  gload(0);
  sstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; kt++) {
    if (kt + 1 < nk) gload(kt + 1);
    const bf16_t* base = smem + (kt & 1) * NT * TILE;
    if constexpr (HALF) {
      s16x8 a[4][2], b[4][2];                 // all four k-steps of the tile: hi halves (0, 1), lo halves (2, 3)
#pragma unroll
      for (int ks = 0; ks < 4; ks++)
#pragma unroll
        for (int i = 0; i < 2; i++) {
          const int ra_ = wm * 64 + i * 32 + l31, rb_ = wn * 64 + i * 32 + l31;
          a[ks][i] = *(const s16x8*)(base + ra_ * BK + (((2 * ks + g) ^ SW::f(ra_)) * 8));
          b[ks][i] = *(const s16x8*)(base + TILE + rb_ * BK + (((2 * ks + g) ^ SW::f(rb_)) * 8));
        }
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 2; j++) {
            acc[i][j] = mfma32h(a[h][i], b[h][j], acc[i][j]);          // hi . hi   (the persistent kernel's order: both tile paths give the same bits)
            acc[i][j] = mfma32h(a[h][i], b[2 + h][j], acc[i][j]);      // hi . lo
            acc[i][j] = mfma32h(a[2 + h][i], b[h][j], acc[i][j]);      // lo . hi
          }
    } else {
#pragma unroll
    for (int ks = 0; ks < BK / 16; ks++) {
      s16x8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int ra_ = wm * 64 + i * 32 + l31, rb_ = wn * 64 + i * 32 + l31;
        const int oa = ra_ * BK + (((2 * ks + g) ^ SW::f(ra_)) * 8);
        const int ob = rb_ * BK + (((2 * ks + g) ^ SW::f(rb_)) * 8);
        a[i] = *(const s16x8*)(base + oa);
        b[i] = *(const s16x8*)(base + TILE + ob);
      }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
    }
    }
    if (kt + 1 < nk) sstore((kt + 1) & 1);
    __syncthreads();
  }

  // ---------------- epilogue ----------------
  // Each wave parks its 64x64 fp32 accumulator tile in its own 16 KB slice of the (now idle) staging LDS, then
  // re-reads it row-wise so that every lane owns 4 consecutive columns: bias / residual / positional loads and all
  // stores become 8-16 byte vector accesses on full 128-256 byte row segments (the MFMA register layout would
  // give 2-4 byte stores at a row stride).  Accumulator indices stay compile-time constants (no scratch).
  if (p.act == 100) {   // measurement aid: main loop only (keeps the accumulators live, stores nothing)
    float t = 0.f;
#pragma clang loop unroll(full)
    for (int i = 0; i < 2; i++)
#pragma clang loop unroll(full)
      for (int j = 0; j < 2; j++)
#pragma clang loop unroll(full)
        for (int r = 0; r < 16; r++) t += acc[i][j][r];
    if (t == 123.456f) p.outF[0] = t;
    return;
  }
  float* wbuf = (float*)smem + wave * 4096;
#pragma clang loop unroll(full)
  for (int i = 0; i < 2; i++)
#pragma clang loop unroll(full)
    for (int j = 0; j < 2; j++)
#pragma clang loop unroll(full)
      for (int r = 0; r < 16; r++)
        wbuf[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g) * 64 + j * 32 + l31] = acc[i][j][r];
  __syncthreads();
  const float* bias = p.bias ? p.bias + (size_t)gz * p.gBias : nullptr;
  const size_t go = (size_t)gz * p.gOut, gp = (size_t)gz * p.gPre;
  const int c4 = (lane & 15) * 4;
  const int n = n0 + wn * 64 + c4;
  if (n >= p.N) return;
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias) bv = *(const f32x4*)(bias + n);
  if (p.bias2) { const f32x4 t = *(const f32x4*)(p.bias2 + n); bv += t; }
#pragma unroll 4
  for (int it = 0; it < 16; it++) {
    const int row = it * 4 + (lane >> 4);
    const int m = m0 + wm * 64 + row;
    if (m >= p.M) continue;
    f32x4 v = *(const f32x4*)(wbuf + row * 64 + c4);
    size_t orow = (size_t)map_row(m, p.o_seg, p.o_seg_stride, p.o_off), frow = orow;
    bool okB = true;                                // the non-outF outputs of this row are written (row tables: < 0 drops them)
    if (p.o_rows) { const int t = p.o_rows[m]; okB = t >= 0; if (p.f_rows) frow = (size_t)p.f_rows[m]; orow = okB ? (size_t)t : 0; }
    if constexpr (HALF) v *= p.accscale;            // operand scales of the split images folded back
    v += bv;
    if (n < p.colscale_n) v *= p.colscale;
    if (p.outPre && okB) {
      u32x2 t; t[0] = pack_bf2(v[0], v[1]); t[1] = pack_bf2(v[2], v[3]);
      *(u32x2*)(p.outPre + go + orow * p.ldpre_out + n) = t;
    }
    if (p.act == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
    else if (p.act == 2) {      // (the bf16 mode shares ONE GELU with the persistent kernel: both tile paths must give the same bits)
      if constexpr (HALF) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
      else { v[0] = gelu_fast(v[0]); v[1] = gelu_fast(v[1]); v[2] = gelu_fast(v[2]); v[3] = gelu_fast(v[3]); }
    }
    if (p.actgrad) {
      const u32x2 t = *(const u32x2*)(p.gradPre + gp + orow * p.ldgp + n);
      const float q0 = __uint_as_float(t[0] << 16), q1 = __uint_as_float(t[0] & 0xffff0000u);
      const float q2 = __uint_as_float(t[1] << 16), q3 = __uint_as_float(t[1] & 0xffff0000u);
      if (p.actgrad == 1) { v[0] = q0 > 0.f ? v[0] : 0.f; v[1] = q1 > 0.f ? v[1] : 0.f; v[2] = q2 > 0.f ? v[2] : 0.f; v[3] = q3 > 0.f ? v[3] : 0.f; }
      else { v[0] *= gelu_fast_grad(q0); v[1] *= gelu_fast_grad(q1); v[2] *= gelu_fast_grad(q2); v[3] *= gelu_fast_grad(q3); }
    }
    if (p.rowscale) { const float rs = p.rowscale[p.row_sample ? p.row_sample[m] : m / p.rs_seg]; v = rs == 0.f ? (f32x4){0.f, 0.f, 0.f, 0.f} : v * rs; }
    if (p.resid) { const f32x4 t = *(const f32x4*)(p.resid + orow * p.ldr + n); v += t; }
    if (p.residB) {
      const u32x2 t = *(const u32x2*)(p.residB + orow * p.ldrB + n);
      v[0] += __uint_as_float(t[0] << 16); v[1] += __uint_as_float(t[0] & 0xffff0000u);
      v[2] += __uint_as_float(t[1] << 16); v[3] += __uint_as_float(t[1] & 0xffff0000u);
    }
    if (p.outF) *(f32x4*)(p.outF + go + frow * p.ldoF + n) = v;
    if (p.outB && okB) {
      u32x2 t; t[0] = pack_bf2(v[0], v[1]); t[1] = pack_bf2(v[2], v[3]);
      *(u32x2*)(p.outB + go + orow * p.ldoB + n) = t;
    }
    if constexpr (HALF) {
      if (p.outS && okB) {
        const float vv[4] = {v[0], v[1], v[2], v[3]};
        u32x2 hi, lo; split4_f16(vv, p.sscale, hi, lo);
        unsigned short* o = p.outS + orow * p.ldoS + split_col((int)go + n);
        *(u32x2*)o = hi; *(u32x2*)(o + 32) = lo;
      }
    }
    if ((p.outU || p.outUF || (HALF && p.outUS)) && okB) {
      f32x4 u = v;
      if (p.pos && m < p.pos_rows) { const f32x4 t = *(const f32x4*)(p.pos + (size_t)(p.pos_map ? p.pos_map[m] : m) * p.ldpos + n); u += t; }
      if (p.outU) { u32x2 t; t[0] = pack_bf2(u[0], u[1]); t[1] = pack_bf2(u[2], u[3]); *(u32x2*)(p.outU + orow * p.ldoU + n) = t; }
      if (p.outUF) *(f32x4*)(p.outUF + orow * p.ldoU + n) = u;
      if constexpr (HALF) {
        if (p.outUS) {
          const float uu[4] = {u[0], u[1], u[2], u[3]};
          u32x2 hi, lo; split4_f16(uu, p.sscale, hi, lo);
          unsigned short* o = p.outUS + orow * p.ldoS + split_col(n);
          *(u32x2*)o = hi; *(u32x2*)(o + 32) = lo;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// gemm_nt256: the large-shape bf16 path.  (64 TM) x 256 x 64 tiles, 8 waves (2 x 4, (32 TM) x 64 each), operands staged with
// global_load_lds (16 B per lane, no VGPR round trip) into two 64 KB stages, ONE barrier per K tile.  Persistent: grid =
// min(tiles, #CU); every workgroup walks tiles bid, bid + G, ... and the K-tile stream never drains at a tile boundary -- the last
// K tile of an output tile prefetches K tile 0 of the next one into the other stage while the epilogue of the finished tile runs
// out of the stage just consumed (an 8 KB fp32 slab per wave, row-contiguous 16-byte global accesses).
// LDS image of a stage: A [64 TM][64] bf16, then (at 32 KB) B [256][64] bf16; rows are 128 B; the 16-byte chunk c of row r sits
// at chunk position c ^ ((r >> 1) & 7): global_load_lds writes lane-linearly, so the permutation is applied to the per-lane
// SOURCE address and again on the ds_read_b128 fragment reads (conflict-free for the 32-row fragments).
// K-tile body (round 2, measured in tools/gemm_pp_lab.hip): the staging pieces of the NEXT K tile are not issued in one burst at
// the head of the K tile (8 waves x 8 pieces hit the CU's one address path together: ~1000 cycles with the matrix cores idle,
// 941 TF/s at 27392 x 1024 x 1024) but AFTER the fragment reads of k-step 0 and interleaved with the MFMAs of k-steps 0 and 1
// (1075 TF/s; a ping-pong structure with the two wave halves one barrier out of phase measured slower than both: 8 barriers per
// K tile).  The body is straight-line code (pieces are issued unconditionally: without a next tile they re-load the current one
// into the idle stage), because sched_group_barrier pins only work inside one basic block.
// Epilogue operand (EOP: the bf16 residual, or the pre-activation of an activation gradient): for tiles of <= 192 rows the
// wave's whole (32 TM) x 64 operand block is fetched into registers during k-steps 2 and 3 of the LAST K tile, under the MFMAs --
// in the epilogue every wave of every workgroup would wait for these loads at the same time (bf16 out + residual: 853 vs
// 1075 TF/s main loop only at N = K = 1024).
// Tiles are enumerated XCD-aware: the 8 workgroups that run concurrently on one XCD take neighbouring tiles, which share the A
// row panel through that XCD's L2.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

// sched_group_barrier sequences (the builtin wants literal arguments): NV VMEM issues two at a time with one MFMA between the
// groups while MFMAs are left (M), then the remaining MFMAs
template <int NV, int M> __device__ __forceinline__ void sgb_vmem_mfma() {
  if constexpr (NV >= 1) {
    __builtin_amdgcn_sched_group_barrier(0x020, NV >= 2 ? 2 : 1, 0);
    if constexpr (M > 0) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); sgb_vmem_mfma<(NV >= 2 ? NV - 2 : 0), M - 1>(); }
    else sgb_vmem_mfma<(NV >= 2 ? NV - 2 : 0), 0>();
  } else if constexpr (M > 0) {
    __builtin_amdgcn_sched_group_barrier(0x008, M, 0);
  }
}
template <int N> __device__ __forceinline__ void sgb_pairs() {      // N x (one MFMA, one fragment read)
  if constexpr (N > 0) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); sgb_pairs<N - 1>(); }
}
// one k-step: Mf MFMAs, (READS) the R fragment reads of the next k-step, NV VMEM issues
template <int Mf, int R, bool READS, int NV> __device__ __forceinline__ void sgb_kstep() {
  constexpr int PAIRS = READS ? (Mf < R ? Mf : R) : 0;
  sgb_pairs<PAIRS>();
  if constexpr (READS && R > PAIRS) __builtin_amdgcn_sched_group_barrier(0x100, R - PAIRS, 0);
  sgb_vmem_mfma<NV, Mf - PAIRS>();
}

// 320-row tiles (single-buffered A fragments): per A fragment i its TN MFMAs, then the read that refills it for the next k-step (the
// first TN groups also take one B fragment of the next k-step), then one of the NV (<= TM) VMEM issues
template <int I, int TM_, int TN_, bool READS, int NV> __device__ __forceinline__ void sgb_sb() {
  if constexpr (I < TM_) {
    __builtin_amdgcn_sched_group_barrier(0x008, TN_, 0);
    if constexpr (READS) __builtin_amdgcn_sched_group_barrier(0x100, I < TN_ ? 2 : 1, 0);
    if constexpr (I < NV) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    sgb_sb<I + 1, TM_, TN_, READS, NV>();
  }
}

// GATHER: row gather / scatter / conv taps / groups present (integer divisions per row); the plain variant has none.
// TM: 32-row fragments per wave along M -> tile height BM = 64 TM (320 / 256 / 192 / 128; 320 without GATHER only): the launcher picks the height that
// wastes the fewest CU-rounds for the launch's tile count (ragged batches give awkward row counts).
// EOP: the launch has a bf16 epilogue operand (residB / gradPre).
// ORD: 0 = all staging pieces of the next K tile right behind the barrier (round-1 order), 1 = behind the fragment reads of k-step 0,
// interleaved with the MFMAs of k-steps 0 and 1
#ifdef UVTG_NT_TRACE
// measurement build only (tools/nt_trace.py): per-tile phase timestamps (100 MHz wall clock) of every workgroup
__device__ unsigned long long* g_nt_trace_dev = nullptr;     // [grid][16 tiles][4 stamps] of the launch being traced
__global__ void nt_trace_set_kernel(unsigned long long* ptr) { g_nt_trace_dev = ptr; }
#define NT_STAMP(k) do { if (g_nt_trace_dev && tid == 0 && lt < 16) g_nt_trace_dev[((size_t)blockIdx.x * 16 + lt) * 4 + (k)] = wall_clock64(); } while (0)
#else
#define NT_STAMP(k) do { } while (0)
#endif
// EPI (plain row mapping only): the epilogue is VALU-issue-bound (8 waves x 4 TM iterations x ~80 instructions, most of them the
// run-time feature tests and 64-bit address arithmetic of paths the launch does not take), so the step's recurring feature sets get
// bodies without the rest: 0 = general; 1 = bias, column scale, row factor, bf16 residual (EOP), bf16 out (q,k,v / out-proj / FFN2 /
// dgrads: 24 of the 40 launches); 2 = bias, pre-activation copy, GELU, bf16 out (FFN1); 3 = GELU' of the bf16 pre-activation (EOP),
// bf16 out (the activation-gradient GEMM); 4 = bf16 out + attention backward's delta: the epilogue operand is O, every row adds the dot product
// of its rounded outputs with it into delta[sample][head][row] (the out-projection dgrad; GemmArgs::delta).
// HALF: split-operand precise mode -- rows of fp16 hi / lo images interleaved in 32-column blocks (uvtg_common.h): staging, LDS image and
// fragment reads are the bf16 kernel's; the K-tile body issues hi.hi, hi.lo, lo.hi from the four fragment sets (6 MFMA groups instead of 4 per
// K tile: 2/3 of the LDS and staging traffic per MFMA) on v_mfma_f32_32x32x16_f16; general epilogue only, fp32 residual, exact erf GELU,
// re-split outputs outS / outUS for the next GEMM.
#ifndef UVTG_NT_GROUPS_MAX_TM
#define UVTG_NT_GROUPS_MAX_TM 4      // experiment switches of the epilogue-operand prefetch (3 / 0 = the round-3 behaviour)
#define UVTG_NT_EOP_RING 1
#endif
// SMALL (general epilogue; TM = 2: 128 x 256 tiles, TM = 1: 128 x 128): the launches with at most one tile per CU (inference batches: M = 32 x
// 107 rows are 108 tiles of 128 x 256, batch 1 is 4).  Such a launch leaves CUs idle while every busy one walks its whole K loop at the rate
// operands reach ONE CU (0.9 - 1.1 us per 48 KB K tile, tools/nt_trace_infer.py), so this variant (a) has no next tile to prefetch and spends
// its LDS on a THREE-stage ring instead: K tile t + 2 is in flight while t is multiplied (counted vmcnt: the wait at the head of a K tile
// leaves the newest tile's pieces outstanding); (b) comes in a NARROW shape that doubles the workgroups and halves the epilogues -- same
// products, same K order, same epilogue as every other instantiation: bit-identical results; and (c) when the launcher asks for it (p.sk > 1:
// at most half as many tiles as CUs) runs a grid of `ntiles x p.sk` workgroups, one K range of one tile each.  The parts of a tile meet like the split parts of the hybrid weight-gradient kernel below: publish the fp32 partial (write-through,
// the accumulator registers in lane order), take a ticket, and the part that draws the last ticket reads the OTHER parts behind one
// agent-scope acquire, sums all of them in part order (the result does not depend on the arrival order) and runs the epilogue; nothing spins.
// LW ("loader waves", round 4; plain row mapping, tile heights 128 - 256): the staging pieces of a K tile are issued by waves 0..3 only -- one
// per SIMD, each for itself and for the wave it shares the SIMD with (w + 4), all at the head of the K tile.  An LDS-DMA issue holds its
// wave's instruction stream for ~60-180 cycles (MI355X_MICROARCH.md); with every wave issuing its pieces right behind the barrier both waves
// of a SIMD are held at the same time, with one loader per SIMD the other wave multiplies meanwhile.  Measured (profiles/r04_ab_nt_loader_waves.txt):
// the kernel +2 % (872 -> 889 TFLOP/s), the step -0.8 % -- against the interleaved order (ORD = 1) at 256 rows and the head order elsewhere;
// nothing at 320 rows.  (The model behind it -- ~1000 idle pipe cycles per K tile from simultaneous DMA issue -- predicted ten times that:
// the issue stall is NOT what the K loop waits for.)  Same products, same K order: results are bit-identical.
template <bool GATHER, int TM, bool EOP, int ORD, int EPI, bool HALF = false, bool SMALL = false, bool LW = false>
__global__ __launch_bounds__(512, 2) void gemm_nt256_kernel(const GemmArgs p) {
#pragma clang fp contract(off)            // every instantiation must round the epilogue alike (the tile paths are compared bit for bit)
  static_assert(!(EPI != 0 && GATHER), "the specialised epilogues have the plain row mapping");
  static_assert(!HALF || (EPI == 0 && !EOP), "the split-operand mode uses the general epilogue without a bf16 operand");
  auto MF = [](s16x8 a, s16x8 b, f32x16 c) -> f32x16 { if constexpr (HALF) return mfma32h(a, b, c); else return mfma32(a, b, c); };
  static_assert(!SMALL || (TM <= 2 && EPI == 0 && ORD == 0), "single-tile variant: 128 x 128 / 128 x 256 tiles, general epilogue, pieces issued at the head of a K tile");
  static_assert(TM >= 2 || SMALL, "TM = 1 exists in the single-tile variant only");
  static_assert(!LW || (ORD == 0 && !SMALL && !HALF), "loader waves: pieces at the head of the K tile, persistent bf16 kernel");
  static_assert(EPI != 2 || !EOP, "FFN1 has no epilogue operand");
  static_assert(EPI != 3 || EOP, "the activation gradient reads its pre-activation");
  static_assert(EPI != 4 || (EOP && !HALF && !SMALL), "the delta epilogue reads O as its operand");
  static_assert(TM < 5 || ((TM + 4 + 1) / 2 <= TM), "320-row tiles: at most one staging piece per A-fragment group of a k-step");
  constexpr bool SIMPLE = EPI != 0;
  // NARROW (the single-tile variant at TM = 1): 128 x 128 tiles, the eight waves as 4 x 2 (32 x 64 outputs each) -- as many tiles as 64 x 256
  // ones would give, at 32 instead of 40 KB of staging per K tile (these launches are bound by operand delivery, not by the MFMAs)
  constexpr bool NARROW = SMALL && TM == 1;
  constexpr int TB = NARROW ? 128 : 256, KB = 64, TN = 2, WR = NARROW ? 4 : 2, BM = WR * 32 * TM, PA = BM / 64, PB = TB / 64, NP = PA + PB;
  // (320-row tiles: 40 KB of A rows per stage; SMALL: 16 KB of A rows + 32 KB of B rows, three stages; the epilogue's 64 KB of fp32 staging
  // then start at the head of the ring -- nothing is in flight by then)
  constexpr int BOFF = SMALL ? BM * 128 : (BM * 128 > 32768 ? BM * 128 : 32768), SSTR = BOFF + TB * 128;
  static_assert(!SMALL || ((NP == 6 || NP == 4) && 3 * SSTR >= 65536), "the counted wait below is written for four / six pieces per K tile");
  using Yes = std::true_type; using No = std::false_type;
  constexpr int RT = BM > 256 ? 512 : 256;       // tile rows covered by the per-row staging tables (one thread per row)
  // epilogue-operand groups (32 rows x 64 columns = 4 x 16 B per lane each) fetched during the last K tile; the remaining ones are
  // fetched inside the epilogue one group ahead, into the registers the fragments no longer need
  constexpr int NPF = !EOP ? 0 : (TM >= 4 ? 0 : (TM == 3 || TM == 1 ? 1 : 2));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem256[];
  __shared__ float s_rs[RT];               // per-row factors (DropPath / frame mask) of the tile in its epilogue
  // attention-backward delta (GemmArgs::delta): the launches that can carry it are the plain-row bf16 ones with an epilogue operand
  constexpr bool DELTA = EPI == 4;
  __shared__ int s_dl[DELTA ? RT : 1];     // per tile row: index of delta[b][0][s]
  __shared__ int s_tab[GATHER ? 3 : 1][RT]; // GATHER: the tile rows' entries of the output-row tables (o_rows, f_rows, pos_map)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = NARROW ? wave >> 1 : wave >> 2, wn = NARROW ? wave & 1 : wave & 3, g = lane >> 5, l31 = lane & 31;
  const int tiles_n = (p.N + TB - 1) / TB, tiles_m = (p.M - p.m_begin + BM - 1) / BM;
  const int per_group = tiles_m * tiles_n;
  const int ntiles = per_group * p.groups;
  const int nk = p.K / KB;
  const int sr_ = lane >> 3, sc_ = lane & 7;

  auto tile_origin = [&](int t, int& gz, int& m0, int& n0) {
    const int q = ntiles / 8, r = ntiles % 8, xcd = t % 8, idx = t / 8;
    int l = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    gz = l / per_group; l -= gz * per_group;
    if (p.cgw > 0 && p.cgw < tiles_n) {      // column groups of cgw tiles, row-block-major inside a group (the last group may be narrower)
      const int gsz = tiles_m * p.cgw, cg = l / gsz, rem = l - cg * gsz, w = min(p.cgw, tiles_n - cg * p.cgw);
      m0 = p.m_begin + (rem / w) * BM; n0 = (cg * p.cgw + rem % w) * TB;
    } else {
      m0 = p.m_begin + (l / tiles_n) * BM; n0 = (l % tiles_n) * TB;
    }
  };
  constexpr int LWF = LW ? 2 : 1;          // staging rows per issuing wave: its own, (LW) and those of wave + 4
  unsigned aofs[LWF * PA], bofs[LWF * PB];   // byte offsets of this lane's staging pieces (1 KB = 8 rows x 128 B each) for the tile being loaded
  const char* Abase = (const char*)p.A;      // A operand of the tile being loaded (A2 for the column tiles from a2_n0 on)
  auto set_offsets = [&](int gz, int m0, int n0) {
    Abase = (const char*)((p.A2 && n0 >= p.a2_n0) ? p.A2 : p.A);
    // 320-row tiles: the per-piece swizzled chunk columns are lane constants the compiler would keep in NP registers for the whole
    // kernel; behind an opaque move they are recomputed here, once per tile
    int sr = sr_, sc = sc_;
    if constexpr (TM >= 5) asm volatile("" : "+v"(sr), "+v"(sc));
#pragma unroll
    for (int h = 0; h < LWF; h++) {
      const int vw = (wave + 4 * h) & 7;        // (LW: h = 1 are the rows of the wave on the same SIMD; meaningless, and unused, on waves 4..7)
#pragma unroll
      for (int i = 0; i < PA; i++) {
        const int r = (vw * PA + i) * 8 + sr;
        const int c = (sc ^ ((r >> 1) & 7)) * 8;
        const int am = GATHER ? max(map_row(min(m0 + r, p.M - 1), p.a_seg, p.a_seg_stride, p.a_off), 0) : min(m0 + r, p.M - 1);
        aofs[h * PA + i] = (unsigned)(((size_t)am * p.lda + c + (size_t)gz * p.gA) * 2);
      }
#pragma unroll
      for (int i = 0; i < PB; i++) {
        const int r = (vw * PB + i) * 8 + sr;
        const int c = (sc ^ ((r >> 1) & 7)) * 8;
        bofs[h * PB + i] = (unsigned)(((size_t)min(n0 + r, p.N - 1) * p.ldb + c + (size_t)gz * p.gB) * 2);
      }
    }
  };
  // byte offsets (A, B) of K tile kt inside the operand rows; the conv taps switch the A row every ktap columns
  auto k_offsets = [&](int kt, unsigned& ka, unsigned& kb) {
    kb = (unsigned)kt * (KB * 2u);
    ka = kb;
#ifdef UVTG_NT_TRACE
    // latency probes (results are garbage): 101 = the A pieces always re-read K tile 0 (cache-hot after the first touch), 102 = the B
    // pieces do, 103 = both -- against 100 (main loop only) they tell which operand's miss latency the two-stage ring fails to cover
    if (p.act == 101 || p.act == 103) ka = 0;
    if (p.act == 102 || p.act == 103) kb = 0;
#endif
    if constexpr (GATHER) {
      const int k0 = kt * KB, tap = k0 / p.ktap;
      ka = (unsigned)(tap * p.lda + (k0 - tap * p.ktap)) * 2u;
    }
  };
  auto piece = [&](unsigned char* sbase, unsigned ka, unsigned kb, int idx) {      // staging piece idx of NP (LW: of 2 NP, the second NP for wave + 4): A pieces first
    const int h = idx / NP, i = idx % NP, vw = (wave + 4 * h) & 7;
    if (i < PA) __builtin_amdgcn_global_load_lds((gbl_void_t*)(Abase + (aofs[h * PA + (i < PA ? i : 0)] + ka)), (lds_void_t*)(sbase + (vw * PA + i) * 1024), 16, 0, 0);
    else __builtin_amdgcn_global_load_lds((gbl_void_t*)((const char*)p.B + (bofs[h * PB + (i >= PA ? i - PA : 0)] + kb)), (lds_void_t*)(sbase + BOFF + (vw * PB + (i - PA)) * 1024), 16, 0, 0);
  };
  // every staging piece of one K tile this wave is responsible for, at the head of the K tile (ORD == 0)
  auto pieces_head = [&](unsigned char* sbase, unsigned ka, unsigned kb) {
    if constexpr (LW) {
      if (wave < 4) {
#pragma unroll
        for (int i = 0; i < 2 * NP; i++) piece(sbase, ka, kb, i);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NP; i++) piece(sbase, ka, kb, i);
    }
  };
  int aoff[TM], boff[TN];
  const int swz = (l31 >> 1) & 7;           // identical for every 32-row fragment of the wave
#pragma unroll
  for (int i = 0; i < TM; i++) aoff[i] = (wm * (32 * TM) + i * 32 + l31) * 128;
#pragma unroll
  for (int j = 0; j < TN; j++) boff[j] = BOFF + (wn * 64 + j * 32 + l31) * 128;

  int tile = blockIdx.x;
  [[maybe_unused]] int sk_part = 0;
  int kt_begin = 0, kt_end = nk;           // this workgroup's K tiles (SMALL with p.sk > 1: one part of the tile's K range)
  [[maybe_unused]] const int sk_parts = SMALL ? (p.sk > 1 ? p.sk : 1) : 1;
  if constexpr (SMALL) {
    sk_part = tile / ntiles; tile -= sk_part * ntiles;
    const int per = (nk + sk_parts - 1) / sk_parts;
    kt_begin = sk_part * per; kt_end = min(kt_begin + per, nk);
    if (sk_part >= sk_parts) return;
  }
  if (tile >= ntiles) return;
  int gz, m0, n0;
  tile_origin(tile, gz, m0, n0);
  set_offsets(gz, m0, n0);
  {
    unsigned ka, kb;
    k_offsets(kt_begin, ka, kb);
    pieces_head(smem256, ka, kb);
    if constexpr (SMALL) {
      if (kt_begin + 1 < kt_end) {
        k_offsets(kt_begin + 1, ka, kb);
        pieces_head(smem256 + SSTR, ka, kb);
      }
    }
  }
  [[maybe_unused]] int st = 0;             // SMALL: ring stage of the K tile being multiplied
  // this lane's 8 output columns
  const int c8 = (lane & 7) * 8;
  int it = 0;
  [[maybe_unused]] int lt = 0;
  while (true) {
    NT_STAMP(0);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    const int next = SMALL ? ntiles : tile + gridDim.x;      // (SMALL: one unit per workgroup)
    int ngz = gz, nm0 = m0, nn0 = n0;
    const int n = n0 + wn * 64 + c8;
    const bool ncol = n < p.N;
    // bias of the lane's columns, fetched at the head of the tile (in the epilogue the load would sit behind the barrier with
    // every wave of the block waiting on it); consumed here on every path (see the note on pending loads in the epilogue)
    float rs_reg = 1.f;       // (set in the last K tile)
    [[maybe_unused]] int tab_reg[3] = {0, 0, 0};
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; e++) bv[e] = 0.f;
    if (ncol && (p.act < 100 || p.act > 103)) {
      if (p.bias) {
        const float* bias = p.bias + (size_t)gz * p.gBias + n;
        const f32x4 b0 = *(const f32x4*)bias, b1 = *(const f32x4*)(bias + 4);
#pragma unroll
        for (int e = 0; e < 4; e++) { bv[e] += b0[e]; bv[4 + e] += b1[e]; }
      }
      if (p.bias2) {
        const f32x4 b0 = *(const f32x4*)(p.bias2 + n), b1 = *(const f32x4*)(p.bias2 + n + 4);
#pragma unroll
        for (int e = 0; e < 4; e++) { bv[e] += b0[e]; bv[4 + e] += b1[e]; }
      }
    }
    // a load some path never waits for reaches the K loop as "maybe pending", and hipcc then drains vmcnt(0) before the first
    // fragment read that re-uses its register: make every path wait here
    asm volatile("" :: "v"(bv[0]), "v"(bv[1]), "v"(bv[2]), "v"(bv[3]), "v"(bv[4]), "v"(bv[5]), "v"(bv[6]), "v"(bv[7]));
    // epilogue operand: residual, else the pre-activation of the activation gradient
    const size_t go = (size_t)gz * p.gOut, gp = (size_t)gz * p.gPre;
    const bf16_t* esrc = !EOP ? nullptr : (DELTA ? p.deltaO : (p.residB ? p.residB : p.gradPre + gp));
    const int eld = !EOP ? 0 : (DELTA ? p.ldDO : (p.residB ? p.ldrB : p.ldgp));
    u32x4 eg[EOP ? TM : 1][4];
    auto fetch_group = [&](int i, int q) {     // one 16-byte piece: rows q * 8 + lane / 8 of 32-row group i
      const int m = min(m0 + wm * (32 * TM) + i * 32 + q * 8 + (lane >> 3), p.M - 1);     // clamped: loaded, not used
      const size_t orow = GATHER ? (size_t)map_row(m, p.o_seg, p.o_seg_stride, p.o_off) : (size_t)m;
      eg[EOP ? i : 0][q] = *(const u32x4*)(esrc + orow * eld + (ncol ? n : 0));
    };

    // ---- one K tile: barrier, fragment reads of k-step 0, then the MFMAs of every k-step cover the reads of the next one, the
    // staging pieces of the next K tile (k-steps 0, 1) and -- LAST only -- the epilogue-operand block (k-steps 2, 3) ----
    // PREF: the K tile issues staging pieces (always, except the last two K tiles of the SMALL ring: nothing left to fetch);
    // DRAIN (SMALL): the wait at its head is vmcnt(0) (the last K tile), else it leaves the newest six pieces (K tile kt + 1) in flight
    auto ktile = [&](int kt, auto last_tag, auto pref_tag, auto drain_tag) {
      constexpr bool LAST = decltype(last_tag)::value;
      constexpr bool PREF = decltype(pref_tag)::value;
      [[maybe_unused]] constexpr bool DRAIN = decltype(drain_tag)::value;
      static_assert(SMALL || PREF, "only the ring has K tiles without staging");
      unsigned ka = 0, kb = 0;
      unsigned char* sbase; const unsigned char* base;
      if constexpr (SMALL) {
        if constexpr (DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if constexpr (NP == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // K tile kt landed (this wave's pieces; the barrier covers the others')
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();        // ... and every wave is done reading the stage K tile kt + 2 goes to
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PREF) k_offsets(kt + 2, ka, kb);
        sbase = smem256 + (st >= 1 ? st - 1 : 2) * SSTR;
        base = smem256 + st * SSTR;
      } else {
        const int cur = it & 1;
        __syncthreads();                       // vmcnt(0) + barrier: K tile `it` landed, the other stage is free
        k_offsets(LAST ? 0 : kt + 1, ka, kb);
        sbase = smem256 + (cur ^ 1) * SSTR;
        base = smem256 + cur * SSTR;
      }
      if constexpr (LAST) {
        // row factor of tile row (tid & 255), fetched under the last K tile: in the epilogue the lookup -- two DEPENDENT loads per
        // q iteration, rowscale[row_sample[m]] -- was a latency chain of its own (8.5-10 us epilogues with DropPath vs 5 without).
        // Branch-free (this block is scheduled as one): without a table the loads hit a valid dummy address and are not used.
        const int mr = min(m0 + (tid & (RT - 1)), p.M - 1);
        const int* rsm = p.row_sample ? p.row_sample : (const int*)p.B;
        const int i1 = rsm[p.row_sample ? mr : 0];
        const float* rsp = p.rowscale ? p.rowscale : (const float*)p.B;
        rs_reg = rsp[p.rowscale ? (p.row_sample ? i1 : mr / p.rs_seg) : 0];
        if constexpr (GATHER) {             // the row tables likewise (same dummy-address trick)
          const int* t0 = p.o_rows ? p.o_rows : (const int*)p.B;
          const int* t1 = p.f_rows ? p.f_rows : (const int*)p.B;
          const int* t2 = p.pos_map ? p.pos_map : (const int*)p.B;
          tab_reg[0] = t0[p.o_rows ? mr : 0]; tab_reg[1] = t1[p.f_rows ? mr : 0]; tab_reg[2] = t2[p.pos_map ? mr : 0];
        }
      }
      constexpr int G0 = ORD == 0 ? 0 : (NP + 1) / 2, G1 = ORD == 0 ? 0 : NP - (NP + 1) / 2;          // pieces issued under k-step 0 / 1
      constexpr int E2 = (NPF * 4 + 1) / 2, E3 = NPF * 4 - E2;  // epilogue-operand loads under k-step 2 / 3
      constexpr int Mf = TM * TN, R = TM + TN;
      if constexpr (HALF) {
        // Split-operand K tile: 32 real columns, hi halves in k-steps 0 / 1, lo halves in k-steps 2 / 3 of the staged 64 elements.
        // Three MFMA groups per half h: hi.hi, hi.lo, lo.hi.  Fragment registers are refilled in place right behind their last use (A hi
        // after the hi.lo group, B lo after it too, A lo after the lo.hi group); only B hi -- needed by the first group of the next half
        // while the last group of this one still reads it -- has two buffers: 2 TM + 3 TN fragment sets.
        auto rdA = [&](int i, int q) { return *(const s16x8*)(base + aoff[i] + (((2 * q + g) ^ swz) << 4)); };
        auto rdB = [&](int j, int q) { return *(const s16x8*)(base + boff[j] + (((2 * q + g) ^ swz) << 4)); };
        s16x8 ah[TM], al[TM], bh[2][TN], bl[TN];
        if (PREF && ORD == 0) pieces_head(sbase, ka, kb);
#pragma unroll
        for (int i = 0; i < TM; i++) ah[i] = rdA(i, 0);
#pragma unroll
        for (int j = 0; j < TN; j++) bh[0][j] = rdB(j, 0);
#pragma unroll
        for (int j = 0; j < TN; j++) bl[j] = rdB(j, 2);
#pragma unroll
        for (int i = 0; i < TM; i++) al[i] = rdA(i, 2);
        if (ORD == 1) {
#pragma unroll
          for (int i = 0; i < G0; i++) piece(sbase, ka, kb, i);
        }
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++) acc[i][j] = MF(ah[i], bh[0][j], acc[i][j]);          // hi . hi   (columns 0..15)
#pragma unroll
        for (int j = 0; j < TN; j++) bh[1][j] = rdB(j, 1);
#pragma unroll
        for (int i = 0; i < TM; i++) {
#pragma unroll
          for (int j = 0; j < TN; j++) acc[i][j] = MF(ah[i], bl[j], acc[i][j]);             // hi . lo
          ah[i] = rdA(i, 1);
        }
#pragma unroll
        for (int j = 0; j < TN; j++) bl[j] = rdB(j, 3);
        if (ORD == 1) {
#pragma unroll
          for (int i = G0; i < NP; i++) piece(sbase, ka, kb, i);
        }
#pragma unroll
        for (int i = 0; i < TM; i++) {
#pragma unroll
          for (int j = 0; j < TN; j++) acc[i][j] = MF(al[i], bh[0][j], acc[i][j]);          // lo . hi
          al[i] = rdA(i, 3);
        }
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++) acc[i][j] = MF(ah[i], bh[1][j], acc[i][j]);          // hi . hi   (columns 16..31)
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++) acc[i][j] = MF(ah[i], bl[j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++) acc[i][j] = MF(al[i], bh[1][j], acc[i][j]);
      } else if constexpr (TM >= 5) {
        // 320-row tiles: 160 accumulator registers leave no room for two sets of A fragments.  Each A fragment is refilled for the
        // next k-step right behind its own TN MFMAs (the refill then has the other (TM - 1) TN MFMAs of this k-step and i TN of the
        // next one to land: >= 8 MFMAs); only the B fragments, used by every MFMA of a k-step, stay double-buffered.
        s16x8 fa[TM], fb[2][TN];
        if (ORD == 0) pieces_head(sbase, ka, kb);
#pragma unroll
        for (int i = 0; i < TM; i++) fa[i] = *(const s16x8*)(base + aoff[i] + ((g ^ swz) << 4));
#pragma unroll
        for (int j = 0; j < TN; j++) fb[0][j] = *(const s16x8*)(base + boff[j] + ((g ^ swz) << 4));
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
#pragma unroll
          for (int i = 0; i < TM; i++) {
#pragma unroll
            for (int j = 0; j < TN; j++) acc[i][j] = MF(fa[i], fb[ks & 1][j], acc[i][j]);
            if (ks < 3) {
              fa[i] = *(const s16x8*)(base + aoff[i] + (((2 * ks + 2 + g) ^ swz) << 4));
              if (i < TN) fb[(ks + 1) & 1][i] = *(const s16x8*)(base + boff[i] + (((2 * ks + 2 + g) ^ swz) << 4));
            }
            // (LDS-DMA and fragment reads keep their source order -- both touch LDS as far as the compiler knows -- so a piece goes
            // where the pinned sequence wants it: one behind the reads of each group of k-steps 0 and 1)
            if (ORD == 1 && ks == 0 && i < G0) piece(sbase, ka, kb, i);
            if (ORD == 1 && ks == 1 && G0 + i < NP) piece(sbase, ka, kb, G0 + i);
          }
        }
        if (ORD == 0 && !LW) __builtin_amdgcn_sched_group_barrier(0x020, NP, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, R, 0);
        sgb_sb<0, TM, TN, true, G0>();
        sgb_sb<0, TM, TN, true, G1>();
        sgb_sb<0, TM, TN, true, 0>();
        sgb_sb<0, TM, TN, false, 0>();
      } else {
      s16x8 fa[2][TM], fb[2][TN];
      if (PREF && ORD == 0) pieces_head(sbase, ka, kb);
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
        if (ORD == 1 && ks == 0) {
#pragma unroll
          for (int i = 0; i < G0; i++) piece(sbase, ka, kb, i);
        }
        if (ORD == 1 && ks == 1) {
#pragma unroll
          for (int i = G0; i < NP; i++) piece(sbase, ka, kb, i);
        }
        if constexpr (LAST && NPF > 0) {
          if (ks >= 2) {
#pragma unroll
            for (int e = (ks == 2 ? 0 : E2); e < (ks == 2 ? E2 : NPF * 4); e++) fetch_group(e >> 2, e & 3);
          }
        }
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++) acc[i][j] = MF(fa[ks & 1][i], fb[ks & 1][j], acc[i][j]);
      }
      // pin the software pipeline the source expresses (hipcc otherwise sinks every fragment read next to its MFMAs and moves the
      // pieces to the head): R reads up front; per k-step (MFMA, read) pairs, then the VMEM issues two at a time between MFMAs
      if (PREF && ORD == 0 && !LW) __builtin_amdgcn_sched_group_barrier(0x020, NP, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, R, 0);
      sgb_kstep<Mf, R, true, G0>();
      sgb_kstep<Mf, R, true, G1>();
      sgb_kstep<Mf, R, true, LAST ? E2 : 0>();
      sgb_kstep<Mf, R, false, LAST ? E3 : 0>();
      }
      it++;
      if constexpr (SMALL) st = st == 2 ? 0 : st + 1;
    };
    if constexpr (SMALL) {
      int kt = kt_begin;
      for (; kt + 2 < kt_end; kt++) {
        ktile(kt, No{}, Yes{}, No{});
#ifdef UVTG_NT_TRACE
        if (kt == kt_begin) NT_STAMP(1);
#endif
      }
      if (kt + 1 < kt_end) { ktile(kt, No{}, No{}, No{}); kt++; }
      ktile(kt, Yes{}, No{}, Yes{});
    } else {
#ifdef UVTG_NT_TRACE
    for (int kt = kt_begin; kt + 1 < kt_end; kt++) { ktile(kt, No{}, Yes{}, Yes{}); if (kt == kt_begin) NT_STAMP(1); }
#else
    for (int kt = kt_begin; kt + 1 < kt_end; kt++) ktile(kt, No{}, Yes{}, Yes{});
#endif
    // last K tile of this output tile: the pieces now belong to K tile 0 of the next tile (or, without one, re-load this tile's)
    if (next < ntiles) tile_origin(next, ngz, nm0, nn0);
    set_offsets(ngz, nm0, nn0);
    ktile(kt_end - 1, Yes{}, Yes{}, Yes{});
    }
    NT_STAMP(2);
    [[maybe_unused]] bool sk_last = true;
    if constexpr (SMALL) if (sk_parts > 1) {
      // lane-order image of the accumulators: float4 group (i, j, r4) of wave w at floats ((w * TM * TN * 4 + (i * TN + j) * 4 + r4) * 64 + lane) * 4
      constexpr int TILE_F = BM * TB;
      const unsigned vo = (unsigned)lane * 16u;
      const int wbase = wave * (TM * TN * 4);
      {
        float* slab = p.sk_slab + ((size_t)tile * sk_parts + sk_part) * TILE_F;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slab, 0, TILE_F * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) {
              const f32x4 v = {acc[i][j][4 * r4], acc[i][j][4 * r4 + 1], acc[i][j][4 * r4 + 2], acc[i][j][4 * r4 + 3]};
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, vo, (wbase + (i * TN + j) * 4 + r4) * 1024, 16);      // aux 16 = sc1: write-through
            }
      }
      __shared__ unsigned s_ticket;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains its write-through stores (and the dead prefetch) ...
      __syncthreads();                                        // ... before ONE lane takes the ticket
      if (tid == 0) s_ticket = __hip_atomic_fetch_add(p.sk_tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      sk_last = s_ticket == (unsigned)(sk_parts - 1);
      if (sk_last) {
        if (tid == 0) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // drop this CU's stale lines of the other parts' slabs
          __hip_atomic_store(p.sk_tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // the tickets are left zero for the next launch
        }
        __syncthreads();
        // Sum of ALL parts in part order, this part's term from its registers: the other (<= 3) slabs of each 32 x 32 accumulator block are
        // fetched branch-free -- an absent part (o >= sk) and the own one read through an EMPTY buffer resource (out-of-range loads return
        // zero without touching memory) -- so the loads of all blocks are in flight together (one round trip, not one per part).
        const int S = sk_parts;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++) {
            u32x4 buf[4][4];
#pragma unroll
            for (int o = 0; o < 4; o++) {
              const bool fetch = o < S && o != sk_part;
              const float* os = p.sk_slab + ((size_t)tile * S + (fetch ? o : 0)) * TILE_F;
              const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)os, 0, fetch ? TILE_F * 4 : 0, 0x00020000);
#pragma unroll
              for (int r4 = 0; r4 < 4; r4++)         // (aux 17 = sc0 | sc1: system-scope loads -- served beyond the L2, whatever an earlier launch left there)
                buf[o][r4] = __builtin_amdgcn_raw_buffer_load_b128(ro, vo, (wbase + (i * TN + j) * 4 + r4) * 1024, 17);
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
              float sum = 0.f;
#pragma unroll
              for (int o = 0; o < 4; o++) {
                const float x = o == sk_part ? acc[i][j][r] : __uint_as_float(buf[o][r >> 2][r & 3]);
                sum = o == 0 ? x : (o < S ? sum + x : sum);
              }
              acc[i][j][r] = sum;
            }
          }
      }
    }
    // ---------------- epilogue of `tile`, out of the stage consumed last ----------------
    if (SMALL && !sk_last) {
      // (the tile's sums are some other part's to finish)
    } else if (p.act >= 100 && p.act <= 103) {   // measurement aid: main loop only (101-103: latency probes of the measurement build)
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) t += acc[i][j][r];
      if (t == 123.456f) p.outF[0] = t;
    } else {
      // 320-row tiles: the lane's column index is re-derived here behind an opaque move, so that the 64-bit output / operand addresses
      // built from it are computed after the K loop instead of living through it (they were the values the register allocator spilled)
      int c8e = c8;
      if constexpr (TM >= 5) asm volatile("" : "+v"(c8e));
      const int n = n0 + wn * 64 + c8e;
      const bool ncol = n < p.N;
      if (tid < RT) s_rs[tid] = rs_reg;
      if constexpr (DELTA) {
        if (tid < RT) {
          const int mr = min(m0 + tid, p.M - 1);
          int bb, ss;
          if (p.delta_row_sample) { bb = p.delta_row_sample[mr]; ss = mr - p.delta_seq_start[bb]; }      // (packed stream: two dependent look-ups per tile row, once per tile)
          else { bb = mr / p.delta_S; ss = mr - bb * p.delta_S; }
          s_dl[tid] = bb * p.delta_H * p.delta_S + ss;
        }
      }
      if constexpr (GATHER) { if (tid < RT) { s_tab[0][tid] = tab_reg[0]; s_tab[1][tid] = tab_reg[1]; s_tab[2][tid] = tab_reg[2]; } }
      __builtin_amdgcn_s_barrier();          // every wave is done reading that stage (the prefetch is NOT drained)
      float* wbuf = (float*)(smem256 + (SMALL ? 0 : ((it - 1) & 1) * SSTR)) + wave * 2048;    // [32][64] fp32, wave-private
      const float cs = (n < p.colscale_n) ? p.colscale : 1.0f;
      // Epilogue operand (bf16 residual / pre-activation), one 16-byte piece per lane and q iteration.  Round 4 (tools/nt_trace.py,
      // profiles/r04_nt_tile_phases_variantA.txt): with the piece loaded AT USE inside the rolled q loop, the 256- and 320-row tiles' epilogues
      // were chains of 16-20 exposed load round trips -- 13-16 us per tile against 4-5 us for the same stores without an operand.  Now:
      //   GROUPS: a whole 32-row group (4 pieces) is fetched one group ahead into the dead fragment registers (<= 192 rows since round 2;
      //           256-row tiles with the plain row mapping since round 4: no scratch);
      //   RING  : the heights that have no 16 registers to spare (320 rows, 256-row gather) keep TWO pieces in flight instead.  (Whole groups
      //           behind a ring for the first one were tried for them too: 8-140 B of scratch whatever pinned the loads -- not kept.)
      constexpr bool GROUPS = EOP && (TM < 4 || (TM == 4 && !GATHER && UVTG_NT_GROUPS_MAX_TM >= 4));
      constexpr bool RING = EOP && !GROUPS && UVTG_NT_EOP_RING && !(TM >= 5 && EPI == 0);      // (320 rows + the general epilogue: 28 B of scratch with the ring -- stays at direct loads)
      auto eop_piece = [&](int pc) {             // piece pc = 4 i + q of the wave's (32 TM) x 64 operand block (clamped: loaded, not used)
        pc = min(pc, 4 * TM - 1);
        const int m = min(m0 + wm * (32 * TM) + (pc >> 2) * 32 + (pc & 3) * 8 + (lane >> 3), p.M - 1);
        const size_t orow = GATHER ? (size_t)map_row(m, p.o_seg, p.o_seg_stride, p.o_off) : (size_t)m;
        return *(const u32x4*)(esrc + orow * eld + (ncol ? n : 0));
      };
      [[maybe_unused]] u32x4 ring0 = {0, 0, 0, 0}, ring1 = {0, 0, 0, 0};
      if constexpr (RING) { ring0 = eop_piece(0); ring1 = eop_piece(1); }
      if (GROUPS && NPF < TM) {
#pragma unroll
        for (int q = 0; q < 4; q++) fetch_group(NPF, q);
      }
#pragma unroll
      for (int i = 0; i < TM; i++) {
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int r = 0; r < 16; r++)
            wbuf[((r & 3) + 8 * (r >> 2) + 4 * g) * 64 + j * 32 + l31] = acc[i][j][r];
        if (GROUPS && i + 1 < TM && i + 1 > NPF) {      // one group ahead
#pragma unroll
          for (int q = 0; q < 4; q++) fetch_group(i + 1, q);
        }
        // the q loop stays ROLLED: TM x 4 unrolled copies of this body are ~100 KB of code per kernel -- more than the instruction
        // cache, and the epilogue then runs at the cache-miss rate (measured: the whole gain of the new main loop was lost again);
        // the prefetched operand pieces rotate through e0
        u32x4 e0 = eg[EOP ? i : 0][0], e1 = eg[EOP ? i : 0][1], e2 = eg[EOP ? i : 0][2], e3 = eg[EOP ? i : 0][3];
#pragma unroll 1
        for (int q = 0; q < 4; q++) {
          const u32x4 ecur = e0;
          e0 = e1; e1 = e2; e2 = e3;
          [[maybe_unused]] u32x4 ring_cur = ring0;
          if constexpr (RING) { ring0 = ring1; ring1 = eop_piece(4 * i + q + 2); }      // (issued on every path, before the row guard below)
          const int row = q * 8 + (lane >> 3);
          const int m = m0 + wm * (32 * TM) + i * 32 + row;
          const f32x4 v0 = *(const f32x4*)(wbuf + row * 64 + c8), v1 = *(const f32x4*)(wbuf + row * 64 + c8 + 4);
          if (m >= p.M || !ncol) continue;
          float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          size_t orow = GATHER ? (size_t)map_row(m, p.o_seg, p.o_seg_stride, p.o_off) : (size_t)m, frow = orow;
          bool okB = true;                          // the non-outF outputs of this row are written (row tables: < 0 drops them)
          if constexpr (GATHER) {
            if (p.o_rows) {
              const int rl = wm * (32 * TM) + i * 32 + row;
              const int t = s_tab[0][rl]; okB = t >= 0; if (p.f_rows) frow = (size_t)s_tab[1][rl]; orow = okB ? (size_t)t : 0;
            }
          }
          if constexpr (HALF) {          // operand scales of the split images folded back first
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] *= p.accscale;
          }
#pragma unroll
          for (int e = 0; e < 8; e++) v[e] = (v[e] + bv[e]) * cs;
          if ((EPI == 2 || (EPI == 0 && p.outPre)) && okB) {
            u32x4 t; t[0] = pack_bf2(v[0], v[1]); t[1] = pack_bf2(v[2], v[3]); t[2] = pack_bf2(v[4], v[5]); t[3] = pack_bf2(v[6], v[7]);
            *(u32x4*)(p.outPre + go + orow * p.ldpre_out + n) = t;
          }
          if (EPI == 0 && p.act == 1) {
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = fmaxf(v[e], 0.f);
          } else if (EPI == 2 || (EPI == 0 && p.act == 2)) {
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = HALF ? gelu_erf(v[e]) : gelu_fast(v[e]);
          }
          u32x4 eop = {0, 0, 0, 0};
          if constexpr (GROUPS) eop = ecur;
          else if constexpr (RING) eop = ring_cur;
          else if constexpr (EOP) eop = *(const u32x4*)(esrc + orow * eld + n);
          if (EOP && (EPI == 3 || (EPI == 0 && p.actgrad && !p.residB))) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
              const float q0 = __uint_as_float(eop[e] << 16), q1 = __uint_as_float(eop[e] & 0xffff0000u);
              if (EPI == 0 && p.actgrad == 1) { v[2 * e] = q0 > 0.f ? v[2 * e] : 0.f; v[2 * e + 1] = q1 > 0.f ? v[2 * e + 1] : 0.f; }
              else { v[2 * e] *= gelu_fast_grad(q0); v[2 * e + 1] *= gelu_fast_grad(q1); }
            }
          }
          if (EPI <= 1 && p.rowscale) {      // (a zero factor SELECTS zero: the masked frame rows of the conv heads may have accumulated garbage)
            const float rs = s_rs[wm * (32 * TM) + i * 32 + row];
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = rs == 0.f ? 0.f : v[e] * rs;
          }
          if (!SIMPLE && p.resid) {
            const float* rp = p.resid + orow * p.ldr + n;
            const f32x4 r0 = *(const f32x4*)rp, r1 = *(const f32x4*)(rp + 4);
#pragma unroll
            for (int e = 0; e < 4; e++) { v[e] += r0[e]; v[4 + e] += r1[e]; }
          }
          if constexpr (DELTA) {
            {      // delta[b][head][s] += sum over this lane's 8 columns of bf16(dO) * O, reduced over the lanes that share (row, head)
              float dot = 0.f;
#pragma unroll
              for (int e = 0; e < 4; e++) {
                const unsigned t = pack_bf2(v[2 * e], v[2 * e + 1]);
                dot += __uint_as_float(t << 16) * __uint_as_float(eop[e] << 16);
                dot += __uint_as_float(t & 0xffff0000u) * __uint_as_float(eop[e] & 0xffff0000u);
              }
              dot += dpp_mov<0xB1>(dot);          // quad_perm [1,0,3,2]
              dot += dpp_mov<0x4E>(dot);          // quad_perm [2,3,0,1]
              const bool wide = p.delta_hd >= 64;
              const float other = dpp_mov<0x141>(dot);      // row_half_mirror: the other quad of this row's 8 lanes
              if (wide) dot += other;
              if ((lane & (wide ? 7 : 3)) == 0)
                atomicAdd(p.delta + (size_t)s_dl[wm * (32 * TM) + i * 32 + row] + (size_t)(n / p.delta_hd) * p.delta_S, dot);
            }
          }
          if (EOP && (EPI == 1 || (EPI == 0 && p.residB))) {
#pragma unroll
            for (int e = 0; e < 4; e++) { v[2 * e] += __uint_as_float(eop[e] << 16); v[2 * e + 1] += __uint_as_float(eop[e] & 0xffff0000u); }
          }
          if (!SIMPLE && p.outF) {
            float* op = p.outF + go + frow * p.ldoF + n;
            *(f32x4*)op = (f32x4){v[0], v[1], v[2], v[3]};
            *(f32x4*)(op + 4) = (f32x4){v[4], v[5], v[6], v[7]};
          }
          if ((SIMPLE || p.outB) && okB) {
            u32x4 t; t[0] = pack_bf2(v[0], v[1]); t[1] = pack_bf2(v[2], v[3]); t[2] = pack_bf2(v[4], v[5]); t[3] = pack_bf2(v[6], v[7]);
            *(u32x4*)(p.outB + go + orow * p.ldoB + n) = t;
          }
          if constexpr (HALF) {
            if (p.outS && okB) {
              const float va[4] = {v[0], v[1], v[2], v[3]}, vb[4] = {v[4], v[5], v[6], v[7]};
              u32x2 h0, l0, h1, l1; split4_f16(va, p.sscale, h0, l0); split4_f16(vb, p.sscale, h1, l1);
              unsigned short* o = p.outS + orow * p.ldoS + split_col((int)go + n);
              *(u32x4*)o = (u32x4){h0[0], h0[1], h1[0], h1[1]};
              *(u32x4*)(o + 32) = (u32x4){l0[0], l0[1], l1[0], l1[1]};
            }
          }
          if (!SIMPLE && (p.outU || p.outUF || (HALF && p.outUS)) && okB) {
            if (p.pos && m < p.pos_rows) {
              const float* pp = p.pos + (size_t)((GATHER && p.pos_map) ? s_tab[GATHER ? 2 : 0][wm * (32 * TM) + i * 32 + row] : m) * p.ldpos + n;
              const f32x4 r0 = *(const f32x4*)pp, r1 = *(const f32x4*)(pp + 4);
#pragma unroll
              for (int e = 0; e < 4; e++) { v[e] += r0[e]; v[4 + e] += r1[e]; }
            }
            if (p.outU) {
              u32x4 t; t[0] = pack_bf2(v[0], v[1]); t[1] = pack_bf2(v[2], v[3]); t[2] = pack_bf2(v[4], v[5]); t[3] = pack_bf2(v[6], v[7]);
              *(u32x4*)(p.outU + orow * p.ldoU + n) = t;
            }
            if (p.outUF) {
              float* op = p.outUF + orow * p.ldoU + n;
              *(f32x4*)op = (f32x4){v[0], v[1], v[2], v[3]};
              *(f32x4*)(op + 4) = (f32x4){v[4], v[5], v[6], v[7]};
            }
            if constexpr (HALF) {
              if (p.outUS) {
                const float va[4] = {v[0], v[1], v[2], v[3]}, vb[4] = {v[4], v[5], v[6], v[7]};
                u32x2 h0, l0, h1, l1; split4_f16(va, p.sscale, h0, l0); split4_f16(vb, p.sscale, h1, l1);
                unsigned short* o = p.outUS + orow * p.ldoS + split_col(n);
                *(u32x4*)o = (u32x4){h0[0], h0[1], h1[0], h1[1]};
                *(u32x4*)(o + 32) = (u32x4){l0[0], l0[1], l1[0], l1[1]};
              }
            }
          }
        }
      }
    }
    NT_STAMP(3);
    lt++;
    if (next >= ntiles) break;
    tile = next; gz = ngz; m0 = nm0; n0 = nn0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA may outlive the workgroup's LDS allocation
}

// ------------------------------------------------------------------------------------------------
// TN: out[n][k] += sum_m P[m][n] * Q[m + q_row_off][k]
// ------------------------------------------------------------------------------------------------
constexpr int TBM = 64;                 // reduction rows per step
constexpr int TLD = 128 + 32;           // LDS row stride (elements): 320 B keeps tr-reads conflict-free

__global__ __launch_bounds__(256) void gemm_tn_kernel(const GemmTNArgs p) {
  __shared__ __attribute__((aligned(16))) bf16_t sP[2][TBM * TLD];
  __shared__ __attribute__((aligned(16))) bf16_t sQ[2][TBM * TLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1, g = lane >> 5, l31 = lane & 31;
  const int i16 = lane & 15, qd = (lane >> 4) & 1;
  const int tiles_k = (p.K + 127) / 128;
  const int tile_n = blockIdx.x / tiles_k, tile_k = blockIdx.x % tiles_k;
  const int n0 = tile_n * 128, k0 = tile_k * 128;
  // rows of this split, in multiples of TBM
  const int steps_total = (p.M + TBM - 1) / TBM;
  const int steps_per = (steps_total + p.splits - 1) / p.splits;
  const int st0 = blockIdx.y * steps_per, st1 = min(steps_total, st0 + steps_per);
  if (st0 >= st1) return;

  int srow[4], sch[4];
#pragma unroll
  for (int i = 0; i < 4; i++) { int q = tid + 256 * i; srow[i] = q >> 4; sch[i] = q & 15; }
  u32x4 rp[4], rq[4];
  auto gload = [&](int st) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int m = st * TBM + srow[i];
      const int mq = m + p.q_row_off;
      u32x4 z = {0, 0, 0, 0};
      const int nc = n0 + sch[i] * 8, kc = k0 + sch[i] * 8;
      rp[i] = (m < p.M && nc < p.N) ? *(const u32x4*)(p.P + (size_t)m * p.ldp + nc) : z;
      rq[i] = (m < p.M && mq >= 0 && mq < p.Mq && kc < p.ldq) ? *(const u32x4*)(p.Q + (size_t)mq * p.ldq + kc) : z;
    }
  };
  auto sstore = [&](int stage) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      *(u32x4*)(&sP[stage][srow[i] * TLD + sch[i] * 8]) = rp[i];
      *(u32x4*)(&sQ[stage][srow[i] * TLD + sch[i] * 8]) = rq[i];
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  float bsum = 0.f;   // column sum of P for the bias gradient (threads 0..127 own one column each)

  gload(st0);
  sstore(0);
  __syncthreads();
  for (int st = st0; st < st1; st++) {
    const int cur = (st - st0) & 1;
    if (st + 1 < st1) gload(st + 1);
    const bf16_t* bp = sP[cur];
    const bf16_t* bq = sQ[cur];
#pragma unroll
    for (int ks = 0; ks < TBM / 16; ks++) {
      s16x8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int mrow = ks * 16 + 8 * g + (i16 >> 2);
        const int ca = wn * 64 + i * 32 + 16 * qd + 4 * (i16 & 3);
        const int cb = wk * 64 + i * 32 + 16 * qd + 4 * (i16 & 3);
        s16x4 a0 = lds_tr16(bp + mrow * TLD + ca), a1 = lds_tr16(bp + (mrow + 4) * TLD + ca);
        s16x4 b0 = lds_tr16(bq + mrow * TLD + cb), b1 = lds_tr16(bq + (mrow + 4) * TLD + cb);
        a[i] = (s16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        b[i] = (s16x8){b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
      }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
    }
    if (p.dbias && tile_k == 0 && tid < 128) {
#pragma unroll 8
      for (int r = 0; r < TBM; r++) bsum += bf2f(bp[r * TLD + tid]);
    }
    if (st + 1 < st1) sstore(cur ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
      if (n >= p.N) continue;
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int k = k0 + wk * 64 + j * 32 + l31;
        if (k < p.K) atomicAdd(p.out + (size_t)n * p.ldo + (size_t)k * p.col_stride, acc[i][j][r]);
      }
    }
  if (p.dbias && tile_k == 0 && tid < 128 && n0 + tid < p.N) atomicAdd(p.dbias + n0 + tid, bsum);
}

// ------------------------------------------------------------------------------------------------
// gemm_tn256: weight gradients at the large shapes.  out[N,K] += P[M,N]^T Q[M,K] as 256 x 256 output tiles, the M
// reduction cut into `splits` ranges so that tiles x splits ~ #CU; 8 waves (2 x 4, 128 n x 64 k each).  Both operands
// are row-major in the reduction dimension: they are staged as [64 m][256] bf16 tiles with buffer_load ... lds
// (out-of-range rows -- the M tail and the conv taps' row -1 / row Mq -- come back as zeros from the buffer bounds
// check) and the MFMA fragments are built with ds_read_b64_tr_b16.  The 16-byte chunk c of tile row m is stored at
// chunk c ^ ((m & 3) << 2), which spreads the four rows one transposing read touches over all 64 banks.
// fp32 atomics top out at ~0.3 T adds/s on this chip (a 16-way split of a 1024^2 gradient would spend as long in
// atomics as in MFMAs), so every (tile, split) unit writes its partial tile as a plain fp32 slab and
// gemm_tn_reduce_kernel folds the slabs (and the bias-gradient partials) into the gradient buffer.
// ------------------------------------------------------------------------------------------------
struct TNPlan {              // per-launch plan of the grouped 256-tile weight-gradient kernel (host-computed, passed by value)
  GemmTNArgs g[UVTG_TN_MAX_GROUPS];
  int tiles_k[UVTG_TN_MAX_GROUPS], tile_base[UVTG_TN_MAX_GROUPS + 1], n_pad[UVTG_TN_MAX_GROUPS];
  unsigned bytes_p[UVTG_TN_MAX_GROUPS], bytes_q[UVTG_TN_MAX_GROUPS];
  long long bias_base[UVTG_TN_MAX_GROUPS];      // float offset of the group's bias partials in scratch
  int count, splits, steps_per, total_tiles;
  float* scratch;
};
typedef int __attribute__((ext_vector_type(4))) i32x4;
// raw buffer descriptor (base, stride 0, num_records bytes, the flags __builtin_amdgcn_make_buffer_rsrc callers here use), wave-uniform
__device__ __forceinline__ i32x4 tn_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)(uintptr_t)base;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
// 64 lanes x 16 B from buffer offset voff (out-of-range lanes read zeros) to LDS bytes [lds_dst, lds_dst + 1024), lane-linear.
// M0 is the compiler's: saved and restored inside the statement that uses it.
__device__ __forceinline__ void tn_dma16(i32x4 rsrc, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}
__global__ __launch_bounds__(512) void gemm_tn256_kernel(const TNPlan plan) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem256[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3, g = lane >> 5, l31 = lane & 31, i16 = lane & 15, qd = (lane >> 4) & 1;
  const int tiles = plan.total_tiles, units = gridDim.x, splits = plan.splits, steps_per = plan.steps_per;
  int l;
  {
    const int q = units / 8, r = units % 8, xcd = blockIdx.x % 8, idx = blockIdx.x / 8;
    l = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int split = l / tiles, tile_g = l % tiles;
  int gi = 0;
  while (gi + 1 < plan.count && tile_g >= plan.tile_base[gi + 1]) gi++;
  const GemmTNArgs p = plan.g[gi];
  const int tile = tile_g - plan.tile_base[gi], tiles_k = plan.tiles_k[gi], n_pad = plan.n_pad[gi];
  const unsigned bytes_p = plan.bytes_p[gi], bytes_q = plan.bytes_q[gi];
  const int tile_n = tile / tiles_k, tile_k = tile % tiles_k;
  const int n0 = tile_n * 256, k0 = tile_k * 256;
  const int steps_total = (p.M + 63) / 64;
  const int st0 = split * steps_per, st1 = min(steps_total, st0 + steps_per);

  const int q_tap = p.ktap > 0 ? k0 / p.ktap : 0, kq0 = k0 - q_tap * (p.ktap > 0 ? p.ktap : 0);   // conv tap of this k tile
  // The LDS-DMA is issued from inline asm: hipcc answers every ds_read_b64_tr_b16 INTRINSIC that follows an LDS-DMA it knows of with
  // s_waitcnt vmcnt(0) (it cannot prove the transposing read does not alias the DMA's LDS target; plain ds_read_b128 loads do not get
  // this), which serialised round 1's kernel completely -- stage, wait for it, compute.  A DMA the compiler does not see costs no wait;
  // its completion is awaited by the explicit vmcnt(0) in front of every barrier below.
  const i32x4 rp = tn_rsrc(p.P, bytes_p), rq = tn_rsrc(p.Q, bytes_q);
  const unsigned lds0 = (unsigned)(uintptr_t)smem256;      // LDS byte address of the dynamic segment (the only LDS object of the kernel)
  unsigned vp[4], vq[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int rr = (wave * 4 + i) * 2 + g;                 // tile row 0..63
    const int cs = (l31 ^ ((rr & 3) << 2)) * 8;            // source column chunk of this lane (LDS stays lane-linear)
    vp[i] = (unsigned)(((st0 * 64 + rr) * p.ldp + n0 + cs) * 2);
    vq[i] = (unsigned)(((st0 * 64 + rr + p.q_row_off + q_tap) * p.ldq + kq0 + cs) * 2);   // negative rows wrap to out-of-range
  }
  const unsigned dp = (unsigned)(64 * p.ldp * 2), dq = (unsigned)(64 * p.ldq * 2);
  auto stage = [&](int s) {
    unsigned char* base = smem256 + s * 65536 + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      tn_dma16(rp, vp[i], lds0 + (unsigned)(s * 65536 + wave * 4096 + i * 1024));
      tn_dma16(rq, vq[i], lds0 + (unsigned)(s * 65536 + wave * 4096 + 32768 + i * 1024));
      vp[i] += dp; vq[i] += dq;
    }
  };
  // transposing-read offsets: lane (i16, qd) addresses row m = mb + (i16 >> 2), 4 columns at 16 qd + 4 (i16 & 3)
  const int mr = i16 >> 2;
  int aoff[4], boff[2];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int n = wm * 128 + i * 32 + 16 * qd + 4 * (i16 & 3);
    aoff[i] = (8 * g + mr) * 512 + ((((n >> 3) ^ (mr << 2))) << 4) + (n & 7) * 2;
  }
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int k = wn * 64 + j * 32 + 16 * qd + 4 * (i16 & 3);
    boff[j] = 32768 + (8 * g + mr) * 512 + ((((k >> 3) ^ (mr << 2))) << 4) + (k & 7) * 2;
  }
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool do_bias = p.dbias && tile_k == 0 && wn == 0;

#ifdef UVTG_NT_TRACE
  if (g_nt_trace_dev && tid == 0 && blockIdx.x < 1024) g_nt_trace_dev[(size_t)blockIdx.x * 4 + 0] = wall_clock64();
#endif
  if (st0 < st1) stage(0);
  // K-step body: straight-line (the staging pieces of the next step are issued unconditionally -- past the last step they fetch rows
  // beyond this split into the idle stage, or zeros through the buffer bounds check -- so that the scheduling pins below see ONE basic
  // block): the 12 transposing reads of k-step 0 go first, the 8 buffer_load ... lds pieces follow between the MFMAs of k-steps 0 and 1
  // instead of bursting at the head of the step (same measurement as for the NT kernel, tools/gemm_pp_lab.hip)
  auto main_loop = [&](auto with_bias) {
  constexpr bool BIAS = decltype(with_bias)::value;
  for (int st = st0; st < st1; st++) {
    const int cur = (st - st0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the stage about to be read have landed
    __syncthreads();
    const unsigned char* base = smem256 + cur * 65536;
    const unsigned sb = lds0 + (unsigned)((cur ^ 1) * 65536 + wave * 4096);
    s16x4 ta[2][4][2], tb[2][2][2];       // [buffer][tile][row half]: 4 m-rows each, two halves make one MFMA operand
#pragma unroll
    for (int i = 0; i < 4; i++) { ta[0][i][0] = lds_tr16((const bf16_t*)(base + aoff[i])); ta[0][i][1] = lds_tr16((const bf16_t*)(base + aoff[i] + 2048)); }
#pragma unroll
    for (int j = 0; j < 2; j++) { tb[0][j][0] = lds_tr16((const bf16_t*)(base + boff[j])); tb[0][j][1] = lds_tr16((const bf16_t*)(base + boff[j] + 2048)); }
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      if (ks < 3) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
          ta[(ks + 1) & 1][i][0] = lds_tr16((const bf16_t*)(base + aoff[i] + (ks + 1) * 8192));
          ta[(ks + 1) & 1][i][1] = lds_tr16((const bf16_t*)(base + aoff[i] + (ks + 1) * 8192 + 2048));
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
          tb[(ks + 1) & 1][j][0] = lds_tr16((const bf16_t*)(base + boff[j] + (ks + 1) * 8192));
          tb[(ks + 1) & 1][j][1] = lds_tr16((const bf16_t*)(base + boff[j] + (ks + 1) * 8192 + 2048));
        }
      }
      if (ks < 2) {
#pragma unroll
        for (int i = 2 * ks; i < 2 * ks + 2; i++) {
          tn_dma16(rp, vp[i], sb + i * 1024);
          tn_dma16(rq, vq[i], sb + 32768 + i * 1024);
          vp[i] += dp; vq[i] += dq;
        }
      }
      s16x8 a[4], b[2];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const s16x4 a0 = ta[ks & 1][i][0], a1 = ta[ks & 1][i][1];
        a[i] = (s16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      }
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const s16x4 b0 = tb[ks & 1][j][0], b1 = tb[ks & 1][j][1];
        b[j] = (s16x8){b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
      }
      // column sums of P (bias gradient): v_dot2c_f32_bf16 against (1, 1) adds both halves of a dword in ONE op; the four ops of
      // fragment i sit behind the MFMAs that consumed it (no fragment is waited for earlier than the matrix cores need it) -- as 64
      // convert + add pairs scheduled freely they made the bias units the launch's stragglers (+18 % on a quarter of the units)
#pragma unroll
      for (int i = 0; i < 4; i++) {
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
        if constexpr (BIAS) {
          const u32x2 h0 = __builtin_bit_cast(u32x2, ta[ks & 1][i][0]), h1 = __builtin_bit_cast(u32x2, ta[ks & 1][i][1]);
          // (inline asm: the v2bf16 builtin form read dword 0 of each pair twice under hipcc 7.2)
          asm("v_dot2c_f32_bf16 %0, %1, %2\n\tv_dot2c_f32_bf16 %0, %1, %3\n\tv_dot2c_f32_bf16 %0, %1, %4\n\tv_dot2c_f32_bf16 %0, %1, %5"
              : "+v"(bsum[i]) : "s"(0x3f803f80u), "v"(h0[0]), "v"(h0[1]), "v"(h1[0]), "v"(h1[1]));
        }
      }
    }
    {   // pin the fragment pipeline: 12 transposing reads up front; per k-step 6 x (MFMA, 2 reads) + 2 MFMAs (the asm DMA statements keep
        // their program order among the reads: 4 after the reads issued under k-step 0, 4 after those under k-step 1); BIAS: the 4 dot
        // products of a fragment follow its two MFMAs
      __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
      for (int ks = 0; ks < 3; ks++) {
#pragma unroll
        for (int n = 0; n < 3; n++) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    }
  }
  };
  if (do_bias) main_loop(std::true_type{}); else main_loop(std::false_type{});
#ifdef UVTG_NT_TRACE
  if (g_nt_trace_dev && tid == 0 && blockIdx.x < 1024) { g_nt_trace_dev[(size_t)blockIdx.x * 4 + 1] = wall_clock64(); g_nt_trace_dev[(size_t)blockIdx.x * 4 + 3] = (unsigned long long)(st1 - st0); }
#endif
  // ---- partial tile -> fp32 slab [256 n][256 k] of this (split, tile) unit, row-contiguous 16-byte stores ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the pieces issued past the last step must not land in the slabs below
  __syncthreads();
  float* slab = plan.scratch + ((size_t)split * tiles + tile_g) * 65536;
  float* wbuf = (float*)smem256 + wave * 2048;        // [32][64] fp32, wave-private
  const int c8 = (lane & 7) * 8;
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++)
        wbuf[((r & 3) + 8 * (r >> 2) + 4 * g) * 64 + j * 32 + l31] = acc[i][j][r];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int row = q * 8 + (lane >> 3);
      const f32x4 v0 = *(const f32x4*)(wbuf + row * 64 + c8), v1 = *(const f32x4*)(wbuf + row * 64 + c8 + 4);
      float* op = slab + (size_t)(wm * 128 + i * 32 + row) * 256 + wn * 64 + c8;
      *(f32x4*)op = v0;
      *(f32x4*)(op + 4) = v1;
    }
  }
#ifdef UVTG_NT_TRACE
  if (g_nt_trace_dev && tid == 0 && blockIdx.x < 1024) g_nt_trace_dev[(size_t)blockIdx.x * 4 + 2] = wall_clock64();
#endif
  if (do_bias) {
    float* bpart = plan.scratch + plan.bias_base[gi] + (size_t)split * n_pad;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float t = bsum[i] + __shfl_xor(bsum[i], 32, 64);
      if (g == 0) bpart[n0 + wm * 128 + i * 32 + l31] = t;
    }
  }
}

// out[n * ldo + k * col_stride] += sum over splits of the unit slabs; dbias[n] += sum of the bias partials
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(const TNPlan plan) {
  const int gi = blockIdx.y;
  const GemmTNArgs p = plan.g[gi];
  const int tiles = plan.total_tiles, tiles_k = plan.tiles_k[gi], splits = plan.splits, n_pad = plan.n_pad[gi];
  const int kq = (p.K + 3) / 4;
  const long long total = (long long)p.N * kq;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int taps = p.ktap > 0 ? p.K / p.ktap : 0;
  float sq = 0.f;          // sum of squares of the values this thread assigns (p.sqsum)
  if (taps >= 2 && taps <= 4 && p.col_stride == taps && p.ktap % 4 == 0 && taps * p.ktap == p.K && p.ldo % 4 == 0 &&
      (((uintptr_t)p.out) & 15) == 0) {
    // conv taps: a thread folds the `taps` tiles that hold columns kk .. kk + 3 of every tap and writes the 4 x taps floats of
    // out[n][kk .. kk + 3][0 .. taps) as ONE contiguous run (one tap at a time is a stride-`taps` read-modify-write scatter:
    // 38 us instead of 16 for the 61 MB of slabs of a conv gradient)
    const int kq2 = p.ktap / 4;
    if (idx < (long long)p.N * kq2) {
      const int n = (int)(idx / kq2), kk = (int)(idx % kq2) * 4;
      float v[16];
#pragma unroll
      for (int t = 0; t < 4; t++) {
        if (t < taps) {
          const int k = t * p.ktap + kk;
          const int tile = plan.tile_base[gi] + (n >> 8) * tiles_k + (k >> 8);
          const float* sp = plan.scratch + (size_t)tile * 65536 + (size_t)(n & 255) * 256 + (k & 255);
          f32x4 s = {0.f, 0.f, 0.f, 0.f};
          for (int sidx = 0; sidx < splits; sidx++) s += *(const f32x4*)(sp + (size_t)sidx * tiles * 65536);
#pragma unroll
          for (int e = 0; e < 4; e++) v[e * taps + t] = s[e];      // (taps is uniform: the indices resolve per branch below)
        }
      }
      float* op = p.out + (size_t)n * p.ldo + (size_t)kk * taps;
      for (int j = 0; j < taps; j++) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (!p.assign) o = *(f32x4*)(op + 4 * j);
        o[0] += v[4 * j]; o[1] += v[4 * j + 1]; o[2] += v[4 * j + 2]; o[3] += v[4 * j + 3];
        *(f32x4*)(op + 4 * j) = o;
        sq += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
      }
    }
  } else if (idx < total) {
    const int n = (int)(idx / kq), k = (int)(idx % kq) * 4;
    const int tile = plan.tile_base[gi] + (n >> 8) * tiles_k + (k >> 8);
    const float* sp = plan.scratch + (size_t)tile * 65536 + (size_t)(n & 255) * 256 + (k & 255);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int sidx = 0; sidx < splits; sidx++) s += *(const f32x4*)(sp + (size_t)sidx * tiles * 65536);
    const int tap = p.ktap > 0 ? k / p.ktap : 0;
    float* op = p.out + (size_t)n * p.ldo + (size_t)(k - tap * (p.ktap > 0 ? p.ktap : 0)) * p.col_stride + tap;
    if (p.col_stride == 1 && p.ktap == 0 && k + 3 < p.K && ((((uintptr_t)op) & 15) == 0)) {
      if (p.assign) *(f32x4*)op = s;
      else { f32x4 o = *(f32x4*)op; o += s; *(f32x4*)op = o; }
      sq += (s[0] * s[0] + s[1] * s[1]) + (s[2] * s[2] + s[3] * s[3]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; e++) if (k + e < p.K) { if (p.assign) op[(size_t)e * p.col_stride] = s[e]; else op[(size_t)e * p.col_stride] += s[e]; sq += s[e] * s[e]; }
    }
  }
  if (p.sqsum && p.assign) {      // (block-uniform condition; with assign the values above ARE the final gradient)
    __shared__ float red[4];
    sq = wave_sum(sq);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
    __syncthreads();
    // (64 slots, 128 bytes apart: thousands of blocks adding into ONE address serialise in L2 -- 15 -> 37 us per launch, measured)
    if (threadIdx.x == 0) { const float t = (red[0] + red[1]) + (red[2] + red[3]); if (t != 0.f) atomicAdd(p.sqsum + 32 + (blockIdx.x & 63) * 32, t); }
  }
  if (p.dbias && idx < p.N) {
    const float* bp = plan.scratch + plan.bias_base[gi] + idx;
    float s = 0.f;
    for (int sidx = 0; sidx < splits; sidx++) s += bp[(size_t)sidx * n_pad];
    p.dbias[idx] += s;
  }
}


// ------------------------------------------------------------------------------------------------
// gemm_tn256h ("hybrid"): SEVERAL weight gradients over the same reduction rows in ONE launch with NO reduce pass.
// The split-M kernel above pays for parallelism with partial slabs: a 1024 x 1024 gradient is 16 tiles, so its M rows are cut 8-16 ways to
// fill 256 CUs, every unit writes a 256 KB fp32 slab and gemm_tn_reduce_kernel folds them (0.25 ms of the 7.5 ms training step, plus the
// slab writes inside the units).  With the weight gradients of ALL encoder layers deferred to the end of the encoder backward there are
// 384 tiles over the same rows: each workgroup takes WHOLE tiles (tile ids [0, full_tiles): results go straight to the gradient buffer)
// and the `total_tiles - full_tiles` remaining tiles are cut into `nsplit` row ranges (2 at config 2) so that the last round fills the
// chip too.  The parts of a split tile meet without a second kernel: every part stores its fp32 slab write-through (sc1), takes a ticket
// (one device-scope atomic per part), and the part that draws the last ticket re-reads the other slabs behind ONE agent-scope acquire, adds
// its own registers and writes the gradient (cdna_hip_programming.md Guideline 16, "splitk-seam": publish write-through, combine by the last
// arriver; nothing spins, so no residency assumption).  A workgroup runs its split part FIRST: the parts of a tile then finish together.
// Restrictions (the launcher falls back to the slab + reduce path otherwise): N, K multiples of 256, contiguous outputs, no conv taps.
// ------------------------------------------------------------------------------------------------
constexpr int TNH_SLAB = 65536 + 256;            // floats per part: the 256 x 256 partial tile + 256 bias-gradient partials
struct TNHGroup { const bf16_t* P; const bf16_t* Q; float* out; float* dbias; int ldp, ldq, ldo, tiles_k, tile_base, steps_total, ktap, q_row_off, col_stride; unsigned bytes_p, bytes_q; };
struct TNHPlan {
  TNHGroup g[UVTG_TNH_MAX_GROUPS];
  int count, M, steps_total, total_tiles, full_tiles, nsplit, steps_per;
  float* slabs; unsigned* tickets; float* sqsum;
};
// (Measured and dropped, round 3: an L2 "touch-ahead" -- every thread loads one 128-byte line of the rows the staging DMA will fetch 2 or 3
// steps later, the step's waits relaxed to vmcnt(1) -- made the launch 4 % SLOWER: the one-step-ahead DMA already covers the HBM round trip;
// what the K loop is sensitive to is how many workgroups share a row panel through an XCD's L2 at the same time, see engine.hip.)
__global__ __launch_bounds__(512) void gemm_tn256h_kernel(const TNHPlan plan) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem256[];
  __shared__ unsigned s_ticket;
  __shared__ float s_sq[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3, g = lane >> 5, l31 = lane & 31, i16 = lane & 15, qd = (lane >> 4) & 1;
  int l;
  {
    const int units = gridDim.x, q = units / 8, r = units % 8, xcd = blockIdx.x % 8, idx = blockIdx.x / 8;
    l = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const unsigned lds0 = (unsigned)(uintptr_t)smem256;
  const int mr = i16 >> 2;
  int aoff[4], boff[2];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int n = wm * 128 + i * 32 + 16 * qd + 4 * (i16 & 3);
    aoff[i] = (8 * g + mr) * 512 + ((((n >> 3) ^ (mr << 2))) << 4) + (n & 7) * 2;
  }
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int k = wn * 64 + j * 32 + 16 * qd + 4 * (i16 & 3);
    boff[j] = 32768 + (8 * g + mr) * 512 + ((((k >> 3) ^ (mr << 2))) << 4) + (k & 7) * 2;
  }
  const int n_split_tiles = plan.total_tiles - plan.full_tiles;
  float sq_total = 0.f;
  // unit list of this workgroup: its part of a split tile first, then whole tiles l, l + grid, ...
  int tile_g = (l < n_split_tiles * plan.nsplit) ? plan.full_tiles + l / plan.nsplit : l;
  int part = (l < n_split_tiles * plan.nsplit) ? l % plan.nsplit : -1;
  if (part < 0 && tile_g >= plan.full_tiles) return;
  while (true) {
    int gi = 0;
    while (gi + 1 < plan.count && tile_g >= plan.g[gi + 1].tile_base) gi++;
    const TNHGroup p = plan.g[gi];
    const int tile = tile_g - p.tile_base;
    const int tile_n = tile / p.tiles_k, tile_k = tile % p.tiles_k;
    const int n0 = tile_n * 256, k0 = tile_k * 256;
    const int st0 = part < 0 ? 0 : part * plan.steps_per;
    const int st1 = part < 0 ? p.steps_total : min(p.steps_total, st0 + plan.steps_per);      // (a group may reduce over fewer rows than the launch's longest: its late parts are short or empty)
    const i32x4 rp = tn_rsrc(p.P, p.bytes_p), rq = tn_rsrc(p.Q, p.bytes_q);
    // conv tap of this k tile (Conv1d weight gradients, round 5: K = taps x ktap columns; tap t reads the rows shifted by t + q_row_off)
    const int q_tap = p.ktap > 0 ? k0 / p.ktap : 0, kq0 = k0 - q_tap * p.ktap;
    unsigned vp[4], vq[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int rr = (wave * 4 + i) * 2 + g;
      const int cs = (l31 ^ ((rr & 3) << 2)) * 8;
      vp[i] = (unsigned)(((st0 * 64 + rr) * p.ldp + n0 + cs) * 2);
      vq[i] = (unsigned)(((st0 * 64 + rr + p.q_row_off + q_tap) * p.ldq + kq0 + cs) * 2);      // negative rows wrap to out-of-range: zeros
    }
    const unsigned dp = (unsigned)(64 * p.ldp * 2), dq = (unsigned)(64 * p.ldq * 2);
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    const bool do_bias = p.dbias && tile_k == 0 && wn == 0;
    __syncthreads();                          // the previous unit's epilogue is done with the staging LDS
    if (st0 < st1) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        tn_dma16(rp, vp[i], lds0 + (unsigned)(wave * 4096 + i * 1024));
        tn_dma16(rq, vq[i], lds0 + (unsigned)(wave * 4096 + 32768 + i * 1024));
        vp[i] += dp; vq[i] += dq;
      }
    }
    auto main_loop = [&](auto with_bias) {
      constexpr bool BIAS = decltype(with_bias)::value;
      for (int st = st0; st < st1; st++) {
        const int cur = (st - st0) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const unsigned char* base = smem256 + cur * 65536;
        const unsigned sb = lds0 + (unsigned)((cur ^ 1) * 65536 + wave * 4096);
        s16x4 ta[2][4][2], tb[2][2][2];
#pragma unroll
        for (int i = 0; i < 4; i++) { ta[0][i][0] = lds_tr16((const bf16_t*)(base + aoff[i])); ta[0][i][1] = lds_tr16((const bf16_t*)(base + aoff[i] + 2048)); }
#pragma unroll
        for (int j = 0; j < 2; j++) { tb[0][j][0] = lds_tr16((const bf16_t*)(base + boff[j])); tb[0][j][1] = lds_tr16((const bf16_t*)(base + boff[j] + 2048)); }
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
          if (ks < 3) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
              ta[(ks + 1) & 1][i][0] = lds_tr16((const bf16_t*)(base + aoff[i] + (ks + 1) * 8192));
              ta[(ks + 1) & 1][i][1] = lds_tr16((const bf16_t*)(base + aoff[i] + (ks + 1) * 8192 + 2048));
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
              tb[(ks + 1) & 1][j][0] = lds_tr16((const bf16_t*)(base + boff[j] + (ks + 1) * 8192));
              tb[(ks + 1) & 1][j][1] = lds_tr16((const bf16_t*)(base + boff[j] + (ks + 1) * 8192 + 2048));
            }
          }
          if (ks < 2) {
#pragma unroll
            for (int i = 2 * ks; i < 2 * ks + 2; i++) {
              tn_dma16(rp, vp[i], sb + i * 1024);
              tn_dma16(rq, vq[i], sb + 32768 + i * 1024);
              vp[i] += dp; vq[i] += dq;
            }
          }
          s16x8 a[4], b[2];
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const s16x4 a0 = ta[ks & 1][i][0], a1 = ta[ks & 1][i][1];
            a[i] = (s16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
          }
#pragma unroll
          for (int j = 0; j < 2; j++) {
            const s16x4 b0 = tb[ks & 1][j][0], b1 = tb[ks & 1][j][1];
            b[j] = (s16x8){b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
          }
#pragma unroll
          for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int j = 0; j < 2; j++) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
            if constexpr (BIAS) {
              const u32x2 h0 = __builtin_bit_cast(u32x2, ta[ks & 1][i][0]), h1 = __builtin_bit_cast(u32x2, ta[ks & 1][i][1]);
              asm("v_dot2c_f32_bf16 %0, %1, %2\n\tv_dot2c_f32_bf16 %0, %1, %3\n\tv_dot2c_f32_bf16 %0, %1, %4\n\tv_dot2c_f32_bf16 %0, %1, %5"
                  : "+v"(bsum[i]) : "s"(0x3f803f80u), "v"(h0[0]), "v"(h0[1]), "v"(h1[0]), "v"(h1[1]));
            }
          }
        }
        {
          __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
          for (int ks = 0; ks < 3; ks++) {
#pragma unroll
            for (int n = 0; n < 3; n++) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        }
      }
    };
    if (do_bias) main_loop(std::true_type{}); else main_loop(std::false_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the pieces issued past the last step must not land in the epilogue's LDS slabs
    __syncthreads();
    // ---- epilogue ----  (buffer accesses: ONE per-lane byte offset + wave-uniform row offsets in SGPRs; with 64-bit row addresses
    // the unrolled store sequence kept ~60 address registers alive through the K loop and spilled)
    float* wbuf = (float*)smem256 + wave * 2048;             // [32][64] fp32, wave-private
    const int c8 = (lane & 7) * 8;
    const int tcol = wn * 64 + c8;
    const unsigned vo_slab = (unsigned)(((lane >> 3) * 256 + tcol) * 4);
    bool last = true;                                        // whole tile: this unit holds the final sums
    if (part >= 0) {
      // split part: publish the partial tile write-through, take a ticket; the last part of the tile to arrive folds the others into its own values
      const int slab_id = (tile_g - plan.full_tiles) * plan.nsplit + part;
      float* slab = plan.slabs + (size_t)slab_id * TNH_SLAB;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slab, 0, TNH_SLAB * 4, 0x00020000);
#pragma unroll
      for (int i = 0; i < 4; i++) {
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int r = 0; r < 16; r++)
            wbuf[((r & 3) + 8 * (r >> 2) + 4 * g) * 64 + j * 32 + l31] = acc[i][j][r];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int row = q * 8 + (lane >> 3);
          const u32x4 v0 = *(const u32x4*)(wbuf + row * 64 + c8), v1 = *(const u32x4*)(wbuf + row * 64 + c8 + 4);
          const int so = (wm * 128 + i * 32 + q * 8) * 1024;              // bytes: tile rows are 256 floats
          __builtin_amdgcn_raw_buffer_store_b128(v0, rs, vo_slab, so, 16);         // aux 16 = sc1: write-through
          __builtin_amdgcn_raw_buffer_store_b128(v1, rs, vo_slab + 16, so, 16);
        }
      }
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const float t = bsum[i] + __shfl_xor(bsum[i], 32, 64);
          if (g == 0) __hip_atomic_store(slab + 65536 + wm * 128 + i * 32 + l31, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // EVERY storing wave drains its write-through stores ...
      __syncthreads();                                        // ... before ONE lane takes the ticket
      if (tid == 0) s_ticket = __hip_atomic_fetch_add(plan.tickets + (tile_g - plan.full_tiles), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      last = s_ticket == (unsigned)(plan.nsplit - 1);
      if (last) {
        if (tid == 0) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // drop this CU's stale lines of the other parts' slabs
          // the ticket is left zero for the NEXT launch (round 6: a backward may issue two hybrid launches -- the group behind layer 1 and
          // layer 0's -- between two zero fills of the ticket array)
          __hip_atomic_store(plan.tickets + (tile_g - plan.full_tiles), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
      }
    }
    if (last) {
      // (conv weights: element (n, tap, c) lives at out[n ldo + c col_stride + tap])
      float* out = p.out + (size_t)n0 * p.ldo + (size_t)kq0 * p.col_stride + q_tap;
      const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7ffffff0, 0x00020000);
      const unsigned vo_out = (unsigned)(((lane >> 3) * p.ldo + tcol * p.col_stride) * 4);
      const bool strided = p.col_stride != 1;
      const int row_bytes = p.ldo * 4;
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < 4; i++) {
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int r = 0; r < 16; r++)
            wbuf[((r & 3) + 8 * (r >> 2) + 4 * g) * 64 + j * 32 + l31] = acc[i][j][r];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int row = q * 8 + (lane >> 3);
          f32x4 v0 = *(const f32x4*)(wbuf + row * 64 + c8), v1 = *(const f32x4*)(wbuf + row * 64 + c8 + 4);
          const int trow0 = wm * 128 + i * 32 + q * 8;                     // wave-uniform
          if (part >= 0) {
            // the parts are summed in PART order, this part's registers at its own position: the gradient does not depend on which part
            // arrived last (round 6: with three parts "own + others" gave (a2 + a0) + a1 or (a0 + a1) + a2 from run to run)
            const f32x4 own0 = v0, own1 = v1;
            for (int o = 0; o < plan.nsplit; o++) {
              f32x4 t0 = own0, t1 = own1;
              if (o != part) {
                const float* os = plan.slabs + (size_t)((tile_g - plan.full_tiles) * plan.nsplit + o) * TNH_SLAB;
                const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc((void*)os, 0, TNH_SLAB * 4, 0x00020000);
                t0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rso, vo_slab, trow0 * 1024, 0));
                t1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rso, vo_slab + 16, trow0 * 1024, 0));
              }
              if (o == 0) { v0 = t0; v1 = t1; } else { v0 += t0; v1 += t1; }
            }
          }
          if (!strided) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v0), ro, vo_out, trow0 * row_bytes, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v1), ro, vo_out + 16, trow0 * row_bytes, 0);
          } else {
            const unsigned cb = (unsigned)p.col_stride * 4;
#pragma unroll
            for (int e = 0; e < 4; e++) {
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0[e]), ro, vo_out + e * cb, trow0 * row_bytes, 0);
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1[e]), ro, vo_out + (4 + e) * cb, trow0 * row_bytes, 0);
            }
          }
          sq += (v0[0] * v0[0] + v0[1] * v0[1]) + (v0[2] * v0[2] + v0[3] * v0[3]) + (v1[0] * v1[0] + v1[1] * v1[1]) + (v1[2] * v1[2] + v1[3] * v1[3]);
        }
      }
      sq_total += sq;
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
          float t = bsum[i] + __shfl_xor(bsum[i], 32, 64);
          const int nn = wm * 128 + i * 32 + l31;
          if (g == 0) {
            if (part >= 0) {                                    // (part order, like the tile above)
              const float own = t;
              for (int o = 0; o < plan.nsplit; o++) {
                const float x = o == part ? own : plan.slabs[(size_t)((tile_g - plan.full_tiles) * plan.nsplit + o) * TNH_SLAB + 65536 + nn];
                t = o == 0 ? x : t + x;
              }
            }
            p.dbias[n0 + nn] += t;                              // (one tile column block per bias entry: single writer)
          }
        }
      }
    }
    // next unit
    if (part >= 0) { part = -1; tile_g = l; }
    else tile_g += gridDim.x;
    if (tile_g >= plan.full_tiles) break;
  }
  if (plan.sqsum) {
    sq_total = wave_sum(sq_total);
    if (lane == 0) s_sq[wave] = sq_total;
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 8; i++) t += s_sq[i];
      if (t != 0.f) atomicAdd(plan.sqsum + 32 + (blockIdx.x & 63) * 32, t);
    }
  }
}

}  // namespace

void uvtg_prof_begin_launch(int family, double flops, hipStream_t s);
void uvtg_prof_end_launch(int family, hipStream_t s);
void uvtg_prof_add_bytes(int family, double bytes);

static int check_nt(const GemmArgs& a, int elem) {
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return -1;
  if (a.A2 && ((a.a2_n0 % 256) || ((uintptr_t)a.A2 & 15))) return -3;
  const int al = 16 / elem;   // elements per 16 bytes
  if (a.lda % al || a.ldb % al || a.ktap <= 0) return -2;
  if (a.ktap < a.K && (a.ktap % (elem == 2 ? 64 : 32))) return -2;
  if ((a.o_rows || a.f_rows) && (a.residB || a.resid || a.gradPre || (a.groups > 1))) return -2;   // row tables: no epilogue operand
  if (a.f_rows && !a.o_rows) return -2;
  if (((uintptr_t)a.A & 15) || ((uintptr_t)a.B & 15)) return -3;
  if (a.N % 4 || a.ldoF % 4 || a.ldoB % 4 || a.ldoU % 4 || a.ldr % 4 || a.ldrB % 4 || a.ldgp % 4 || a.ldpre_out % 4 || a.ldpos % 4 ||
      a.colscale_n % 4) return -7;
  return 0;
}

// eligibility of the 256-tile path: whole 64-wide K tiles (also per conv tap), 16-byte rows everywhere the epilogue
// touches 8 columns at a time, 32-bit byte offsets, and enough work that 256 x 256 tiles do not waste the chip
static double nt256_cost(int M, int N, int groups, int tm, int cus, bool eop = true);
static int g_num_cu = 0;
static int g_force_tile = 0;   // 0: automatic, 128 / 256: force that NT tile size where it is legal (parity tests)
static int g_force_bm = 0;     // 0: automatic, 128 / 192 / 256: force the tile height of the 256-wide persistent kernel
extern "C" int uvtg_debug_force_nt_bm(int bm) { if (bm != 0 && bm != 128 && bm != 192 && bm != 256 && bm != 320) return -21; g_force_bm = bm; return 0; }
extern "C" int uvtg_debug_force_nt_tile(int tile) { if (tile != 0 && tile != 128 && tile != 256) return -21; g_force_tile = tile; return 0; }
static int g_cu_cap = 0;       // experiment knob: the persistent GEMM grids use at most this many CUs (0 = all)
extern "C" int uvtg_debug_gemm_cus(int n) { g_cu_cap = n > 0 ? n : 0; return 0; }
static int g_cu_reserved = 0;  // CUs left out of every persistent grid for the communication kernels of a data-parallel run
static int eff_cus() {
  int n = (g_cu_cap > 0 && g_cu_cap < g_num_cu) ? g_cu_cap : g_num_cu;
  if (g_cu_reserved > 0) n = n - g_cu_reserved > 8 ? n - g_cu_reserved : 8;
  return n;
}
static int ensure_num_cu() {
  if (!g_num_cu) {
    int dev = 0; hipDeviceProp_t pr;
    if (hipError_t e = hipGetDevice(&dev)) return (int)e;
    if (hipError_t e = hipGetDeviceProperties(&pr, dev)) return (int)e;
    g_num_cu = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
  }
  return 0;
}
static bool nt256_ok(const GemmArgs& a) {
  if (g_force_tile == 128) return false;
  if (a.K % 64 || a.ktap % 64 || a.lda % 8 || a.ldb % 8 || a.N % 8) return false;
  if (a.ldoF % 8 || a.ldoB % 8 || a.ldoU % 8 || a.ldr % 8 || a.ldrB % 8 || a.ldgp % 8 || a.ldpre_out % 8 || a.ldpos % 8 || a.colscale_n % 8) return false;
  if (a.gA % 8 || a.gB % 8 || a.gBias % 4 || a.gOut % 8 || a.gPre % 8) return false;
  const int groups = a.groups > 0 ? a.groups : 1;
  if (g_force_tile != 256) {  // pick the structure that wastes less of the chip: the persistent 256-wide kernel at ITS best tile height (whole or
    // partly filled CU rounds, nt256_cost) vs rounds of 512 register-staged 128 x 128 tiles (two per CU = the output area of one 128 x 256 tile,
    // at 620 / 950 of the persistent kernel's in-loop rate).  (Round 2 compared 256-row tiles only: the 8192-row text GEMMs -- 128 tiles of
    // 256 rows, half a round -- went to the 128-tile kernel at 443 TF/s although 256 tiles of 128 rows fill exactly one round.)
    const long long t128 = (long long)cdiv(a.M, 128) * cdiv(a.N, 128) * groups;
    // (Round 4, measured and dropped -- profiles/r04_ab_small_m_heuristic.txt: sending single-round small-M launches (inference batches, M = 32 x 107
    // rows) to the 128 x 128 kernel for twice the tile count made every inference case SLOWER -- batch 32 1.63 -> 1.70 ms, batch 1 bf16
    // 0.65 -> 0.74 ms: its register-staged K loop is longer per tile than half a 128 x 256 tile of the LDS-DMA kernel.)
    const double c128 = (double)((t128 + 511) / 512) * 128.0 * (950.0 / 620.0);
    const bool gather_ = a.a_seg || a.o_seg || a.a_off || a.o_off || a.ktap != a.K || groups != 1 || a.o_rows || a.pos_map;
    // the SAME CU count the launch plan will use (reserved communication CUs / the experiment cap subtracted)
    const int cus = ensure_num_cu() ? 256 : eff_cus();
    double c256 = 1e30;
    for (int tm = gather_ ? 4 : 5; tm >= 2; tm--) { const double c = nt256_cost(a.M, a.N, groups, tm, cus); if (c < c256) c256 = c; }
    static const bool old_choice = uvtg_dev_env("UVTG_NT_OLD_128_CHOICE") != nullptr;
    if (old_choice) {
      const long long t256 = (long long)cdiv(a.M, 256) * cdiv(a.N, 256) * groups;
      const double e256 = (double)t256 / (double)((t256 + 255) / 256 * 256) * 950.0, e128 = (double)t128 / (double)((t128 + 511) / 512 * 512) * 620.0;
      if (e256 < e128) return false;
    } else if (c128 < c256) return false;
  }
  const long long amax = ((long long)(a.a_seg ? (a.M / a.a_seg + 1) * a.a_seg_stride : a.M) + a.a_off + 3) * a.lda + (long long)groups * a.gA;
  const long long bmax = (long long)a.N * a.ldb + (long long)groups * a.gB;
  return amax * 2 < (1LL << 32) && bmax * 2 < (1LL << 32);
}
extern "C" int uvtg_set_reserved_cus(int k) {
  g_cu_reserved = k > 0 ? k : 0;
  if (ensure_num_cu()) return -1;
  return eff_cus();
}
// relative time per output element of the persistent NT structure at the four tile heights
#ifndef UVTG_NT_F4
#define UVTG_NT_F5 1.02
#define UVTG_NT_F4 1.00
#define UVTG_NT_F3 1.07
#define UVTG_NT_F2 1.20
#endif
// 320-row tiles WITHOUT a bf16 epilogue operand (QKV projection, FFN1, out-projection dgrad): their epilogue has no two-piece operand ring to
// walk, and the factor fitted on the residual launches overprices them -- round 5, in-box (profiles/r05_ab_nt_plans.txt): the N = 3 d launch of
// the headline (27392 rows) 199 us as 7 rounds of 192-row tiles, 187 as 4.03 rounds of 320-row ones, 177 as 4 rounds + a single-tile tail
#ifndef UVTG_NT_F5_NOEOP
#define UVTG_NT_F5_NOEOP 0.90
#endif
// Tile height (TM = rows / 64) of a persistent NT launch, pure host arithmetic.  Candidates: 256-wide tiles of 320 / 256 / 192 / 128 rows,
// one workgroup per CU; cost = rounds x rows of a tile x the per-height factor.  A partly filled last round costs ~0.84 of a full one up to
// ~70 % fill (fewer active CUs run their K loops faster), then rises to a full round (fitted on the per-height timings of tools/tm5_ab.sh at
// the step's shapes).  320-row tiles: plain row mapping only -- the gather variants have no registers left for them; a forced 320 falls back
// to 256 there.  Returns 0 when the forced height is not a candidate.
static int nt256_pick_tm(int M, int N, int groups, bool gather, int cus, int force, bool eop = true) {
  static const bool tm5_off = uvtg_dev_env("UVTG_NT_TM5_OFF") != nullptr;          // experiment: no 320-row tiles
  int best_tm = 0; double best = 1e30;
  const int force_bm = (force == 320 && gather) ? 256 : force;
  if (cus < 1) cus = 1;
  for (int tm = 5; tm >= 2; tm--) {
    if (force_bm && tm * 64 != force_bm) continue;
    if (tm == 5 && (gather || (tm5_off && force_bm != 320))) continue;
    const double cost = nt256_cost(M, N, groups, tm, cus, eop);
    if (cost < best) { best = cost; best_tm = tm; }
  }
  return best_tm;
}
extern "C" int uvtg_debug_nt_tile_rows(int M, int N, int groups, int gather, int cus) {
  if (M <= 0 || N <= 0 || groups <= 0 || cus <= 0) return -20;
  return 64 * nt256_pick_tm(M, N, groups, gather != 0, cus, 0);
}
// Launch plan of one persistent NT GEMM: either ONE launch at the height nt256_pick_tm chooses, or -- when that leaves a sparsely filled
// last round -- a HEAD launch of whole CU rounds of tall tiles over rows [0, rows1) followed by a TAIL launch of short tiles over
// [rows1, M).  A round costs the same whether 12 or 200 of its tiles exist (every tile runs its full K loop), so a 27392-row launch at
// N = 1024 (all-ones masks: 344 tiles of 320 rows = 1.34 rounds, paid as 1.84) runs as 256 tiles of 320 rows + 216 tiles of 128 rows, and a
// ragged batch whose packed row count lands just above 320 x 64 = 20480 pays a 128-row tail instead of a second 320-row round.
// Same cost model as nt256_pick_tm (units: rows of a tile x per-height factor per round); the tail launch is charged its launch gap.
struct NtPlan { int tm1, rows1, tm2; };      // rows1 == 0: single launch at tm1
static double nt256_cost(int M, int N, int groups, int tm, int cus, bool eop) {
  static const double f[6] = {0, 0, UVTG_NT_F2, UVTG_NT_F3, UVTG_NT_F4, UVTG_NT_F5};
  const long long tiles = (long long)cdiv(M, 64 * tm) * cdiv(N, 256) * groups;
  const long long full = tiles / cus, rem = tiles % cus;
  const double fill = (double)rem / cus;
  const double rounds = (double)full + (rem ? 0.84 + 0.16 * (fill > 0.7 ? (fill - 0.7) / 0.3 : 0.0) : 0.0);
  return rounds * (64.0 * tm) * ((tm == 5 && !eop) ? UVTG_NT_F5_NOEOP : f[tm]);
}
// Experiment knob (include/uvtg_dev.h): the plan of the plain-row launches of one M x N shape, forced (in-box A/B of the cost model's choice).
struct NtPlanOverride { int M, N, tm1, rows1, tm2; };
static NtPlanOverride g_plan_ovr[8]; static int g_plan_novr = 0;
extern "C" int uvtg_debug_nt_plan_override(int M, int N, int tm1_rows, int rows1, int tm2_rows) {
  if (M <= 0) { g_plan_novr = 0; return 0; }
  if (tm1_rows % 64 || tm1_rows < 128 || tm1_rows > 320 || (rows1 && (tm2_rows % 64 || tm2_rows < 128 || tm2_rows > 256 || rows1 % tm1_rows || rows1 >= M))) return -21;
  for (int i = 0; i < g_plan_novr; i++) if (g_plan_ovr[i].M == M && g_plan_ovr[i].N == N) { g_plan_ovr[i] = NtPlanOverride{M, N, tm1_rows / 64, rows1, tm2_rows / 64}; return 0; }
  if (g_plan_novr >= 8) return -17;
  g_plan_ovr[g_plan_novr++] = NtPlanOverride{M, N, tm1_rows / 64, rows1, tm2_rows / 64};
  return 0;
}
struct SmallPlan { int tm, parts; double us; };
static SmallPlan nt256_small_plan(int rows, int N, int groups, int nk, int cus, int cap_units, bool have_ws, int mode);
static int g_nt_small = -1;          // the single-tile (three-stage ring) variant for launches of at most one tile per CU: 1 on (default), 0 off
static NtPlan nt256_plan(int M, int N, int K, int groups, bool gather, int cus, int force, bool eop = true, bool have_ws = false, int cap_units = 0, bool can_small = true) {
  static const bool split_off = uvtg_dev_env("UVTG_NT_SPLIT_OFF") != nullptr;      // experiment: single launches only
  static const bool ovr_env = [] {      // UVTG_NT_PLAN_OVR="M,N,tm1_rows,rows1,tm2_rows;..." = uvtg_debug_nt_plan_override calls (A/B runs of bench.py)
    const char* e = uvtg_dev_env("UVTG_NT_PLAN_OVR");
    while (e && *e) {
      int v[5] = {0, 0, 0, 0, 0}, n = 0;
      if (sscanf(e, "%d,%d,%d,%d,%d%n", &v[0], &v[1], &v[2], &v[3], &v[4], &n) == 5) uvtg_debug_nt_plan_override(v[0], v[1], v[2], v[3], v[4]);
      else break;
      e += n; if (*e == ';') e++;
    }
    return true;
  }();
  (void)ovr_env;
  if (!gather && groups == 1 && !force)
    for (int i = 0; i < g_plan_novr; i++) if (g_plan_ovr[i].M == M && g_plan_ovr[i].N == N) return NtPlan{g_plan_ovr[i].tm1, g_plan_ovr[i].rows1, g_plan_ovr[i].tm2};
  NtPlan pl{nt256_pick_tm(M, N, groups, gather, cus, force, eop), 0, 0};
  if (!pl.tm1 || split_off || force || cus < 1) return pl;
  double best = nt256_cost(M, N, groups, pl.tm1, cus, eop);
  const double gap = 24.0 * 1024.0 / (K > 64 ? K : 64);          // ~3 us launch gap in units of one 320-row round at K = 1024 (~326 units ~ 40 us)
  const double us_per_unit = 0.1227 * (K > 64 ? K : 64) / 1024.0;
  if (g_nt_small < 0) g_nt_small = uvtg_dev_env("UVTG_NT_SMALL_OFF") ? 0 : (uvtg_dev_env("UVTG_NT_SMALL_TM1_OFF") ? 2 : 1);
  const int tn_g = cdiv(N, 256) * groups;
  for (int tm1 = gather ? 4 : 5; tm1 >= 3; tm1--) {
    const long long rt_total = cdiv(M, 64 * tm1);
    const long long full_rounds = rt_total * tn_g / cus;
    if (full_rounds < 1) continue;
    const int rt1 = (int)(full_rounds * cus / tn_g);                // row tiles of the head: the most that fit into whole rounds
    const long long rows1 = (long long)rt1 * 64 * tm1;
    if (rt1 < 1 || rows1 >= M) continue;
    const double head = nt256_cost((int)rows1, N, groups, tm1, cus, eop);
    for (int tm2 = 2; tm2 <= 4; tm2++) {
      double tail = nt256_cost(M - (int)rows1, N, groups, tm2, cus, eop);
      if (tm2 == 2 && g_nt_small && can_small && (long long)cdiv(M - (int)rows1, 128) * cdiv(N, 256) * groups <= cus) {      // (can_small: launch_nt256 keeps delta-carrying launches off the single-tile variant -- ADVICE r5)
        // a tail of at most one 128-row tile per CU runs the single-tile variant (launch_nt256): priced by that variant's own estimate
        const SmallPlan sp = nt256_small_plan(M - (int)rows1, N, groups, K / 64, cus, cap_units, have_ws, g_nt_small);
        if (sp.tm) tail = sp.us / us_per_unit;
      }
      const double c = head + tail + gap;
      if (c < best * 0.97) { best = c; pl = NtPlan{tm1, (int)rows1, tm2}; }      // (3 % margin: do not split for noise)
    }
  }
  return pl;
}
// host arithmetic only (tests): out3 = {tile rows of the head (or only) launch, rows covered by the head (0 = single launch), tile rows of the tail}
extern "C" int uvtg_debug_nt_plan(int M, int N, int K, int groups, int gather, int cus, int* out3) {
  if (M <= 0 || N <= 0 || K <= 0 || groups <= 0 || cus <= 0 || !out3) return -20;
  const NtPlan pl = nt256_plan(M, N, K, groups, gather != 0, cus, 0);
  out3[0] = 64 * pl.tm1; out3[1] = pl.rows1; out3[2] = 64 * pl.tm2;
  return 0;
}
// ... with the launch's epilogue class (eop != 0: it reads a bf16 residual / pre-activation operand) and whether the caller provides the split-K workspace
extern "C" int uvtg_debug_nt_plan2(int M, int N, int K, int groups, int gather, int cus, int eop, int have_ws, int* out3) {
  if (M <= 0 || N <= 0 || K <= 0 || groups <= 0 || cus <= 0 || !out3) return -20;
  const NtPlan pl = nt256_plan(M, N, K, groups, gather != 0, cus, 0, eop != 0, have_ws != 0, UVTG_SK_UNITS);
  out3[0] = 64 * pl.tm1; out3[1] = pl.rows1; out3[2] = 64 * pl.tm2;
  return 0;
}
// split-operand (fp16 images) instantiations: one staging order per tile height (the defaults of nt_order), general epilogue
template <int TM> static int launch_nt256_half(const GemmArgs& b, int grid, bool gather, hipStream_t s) {
  constexpr int smem = TM == 5 ? 147456 : 131072;
  constexpr int ORD = TM >= 4 ? 1 : 0;
  static bool attr = false;
  if (!attr) {
    if (hipError_t e = hipFuncSetAttribute((const void*)gemm_nt256_kernel<false, TM, false, ORD, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem)) return (int)e;
    if constexpr (TM < 5) { if (hipError_t e = hipFuncSetAttribute((const void*)gemm_nt256_kernel<true, TM, false, ORD, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem)) return (int)e; }
    attr = true;
  }
  if (gather) {
    if constexpr (TM < 5) hipLaunchKernelGGL((gemm_nt256_kernel<true, TM, false, ORD, 0, true>), dim3(grid), dim3(512), smem, s, b);
    else return -21;
  } else hipLaunchKernelGGL((gemm_nt256_kernel<false, TM, false, ORD, 0, true>), dim3(grid), dim3(512), smem, s, b);
  return 0;
}
template <int TM, int ORD> static int launch_nt256_tm(const GemmArgs& b, int grid, bool gather, bool eop, int epi, hipStream_t s) {
  constexpr int smem = TM == 5 ? 147456 : 131072;
  static bool attr = false;
#define NT256_ATTR(G, E, P) if (hipError_t e = hipFuncSetAttribute((const void*)gemm_nt256_kernel<G, TM, E, ORD, P>, hipFuncAttributeMaxDynamicSharedMemorySize, smem)) return (int)e;
  if (!attr) {
    NT256_ATTR(false, false, 0) NT256_ATTR(false, true, 0) NT256_ATTR(false, false, 1) NT256_ATTR(false, true, 1)
    NT256_ATTR(false, false, 2) NT256_ATTR(false, true, 3) NT256_ATTR(false, true, 4)
    if constexpr (TM < 5) { NT256_ATTR(true, false, 0) NT256_ATTR(true, true, 0) }
    attr = true;
  }
#undef NT256_ATTR
#define NT256_GO(G, E, P) hipLaunchKernelGGL((gemm_nt256_kernel<G, TM, E, ORD, P>), dim3(grid), dim3(512), smem, s, b)
  if (gather) {
    if constexpr (TM < 5) { if (eop) NT256_GO(true, true, 0); else NT256_GO(true, false, 0); }
    else return -21;
  } else if (epi == 1) { if (eop) NT256_GO(false, true, 1); else NT256_GO(false, false, 1); }
  else if (epi == 2 && !eop) NT256_GO(false, false, 2);
  else if (epi == 3 && eop) NT256_GO(false, true, 3);
  else if (epi == 4 && eop) NT256_GO(false, true, 4);
  else { if (eop) NT256_GO(false, true, 0); else NT256_GO(false, false, 0); }
#undef NT256_GO
  return 0;
}
// loader-wave instantiations: plain row mapping, pieces at the head of the K tile, tile heights 128 / 192 / 256
static int g_nt_lw = -1;
extern "C" int uvtg_debug_nt_loader_waves(int mask) { if (mask < 0 || mask > 7) return -21; g_nt_lw = mask; return 0; }
template <int TM> static int launch_nt256_lw(const GemmArgs& b, int grid, bool eop, int epi, hipStream_t s) {
  constexpr int smem = 131072;
  static_assert(TM <= 4, "320-row tiles keep every wave staging for itself (measured: no gain)");
  static bool attr = false;
#define NTLW_ATTR(E, P) if (hipError_t e = hipFuncSetAttribute((const void*)gemm_nt256_kernel<false, TM, E, 0, P, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem)) return (int)e;
  if (!attr) {
    NTLW_ATTR(false, 0) NTLW_ATTR(false, 1) NTLW_ATTR(true, 1) NTLW_ATTR(false, 2) NTLW_ATTR(true, 3) NTLW_ATTR(true, 4)
    if constexpr (TM < 4) { NTLW_ATTR(true, 0) }       // (general epilogue + operand at 256 / 320 rows: 16-40 B of scratch with the second set of piece offsets -- not built)
    attr = true;
  }
#undef NTLW_ATTR
#define NTLW_GO(E, P) hipLaunchKernelGGL((gemm_nt256_kernel<false, TM, E, 0, P, false, false, true>), dim3(grid), dim3(512), smem, s, b)
  if (epi == 1) { if (eop) NTLW_GO(true, 1); else NTLW_GO(false, 1); }
  else if (epi == 2 && !eop) NTLW_GO(false, 2);
  else if (epi == 3 && eop) NTLW_GO(true, 3);
  else if (epi == 4 && eop) NTLW_GO(true, 4);
  else if (eop) { if constexpr (TM < 4) NTLW_GO(true, 0); else return -100; }
  else NTLW_GO(false, 0);
#undef NTLW_GO
  return 0;
}
// single-tile instantiations (SMALL: 128-row tiles, three-stage ring, optional split-K, general epilogue): bf16 with / without the bf16
// epilogue operand, split operands
template <int TM> static int launch_nt256_small(const GemmArgs& b, int grid, bool gather, bool eop, bool half, hipStream_t s) {
  constexpr int smem = TM == 1 ? 3 * 32768 : 3 * 49152;       // three stages of 128 x 128 / 128 x 256 operand rows
  static bool attr = false;
#define NTSK_ATTR(G, E, H) if (hipError_t e = hipFuncSetAttribute((const void*)gemm_nt256_kernel<G, TM, E, 0, 0, H, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem)) return (int)e;
  if (!attr) {
    NTSK_ATTR(false, false, false) NTSK_ATTR(false, true, false) NTSK_ATTR(true, false, false) NTSK_ATTR(true, true, false)
    NTSK_ATTR(false, false, true) NTSK_ATTR(true, false, true)
    attr = true;
  }
#undef NTSK_ATTR
#define NTSK_GO(G, E, H) hipLaunchKernelGGL((gemm_nt256_kernel<G, TM, E, 0, 0, H, true>), dim3(grid), dim3(512), smem, s, b)
  if (half) { if (eop) return -6; if (gather) NTSK_GO(true, false, true); else NTSK_GO(false, false, true); }
  else if (gather) { if (eop) NTSK_GO(true, true, false); else NTSK_GO(true, false, false); }
  else { if (eop) NTSK_GO(false, true, false); else NTSK_GO(false, false, false); }
#undef NTSK_GO
  return 0;
}
// Parts per tile of a small launch (0 = no split).  The K loop of a 128-row tile is a latency chain of ~1.2-1.5 us per K tile whatever the
// chip does besides, and the last arriver folds `parts` slabs of 128 KB alone (~2 us each from L2 / Infinity Cache): K-loop time / parts +
// fold time x parts has its minimum near 4 at the encoder's K, so: at most g_nt_splitk_max (4) parts, at least 4 K tiles each, never more
// workgroups than CUs (every part must be resident with the others only for SPEED -- nothing waits).
static int g_delta_fuse = -1;        // 0: attention backward's delta by its own kernel (parity tests / A-B); else fused into the dO GEMM's epilogue where legal
extern "C" int uvtg_debug_delta_fuse(int on) { g_delta_fuse = on ? 1 : 0; return 0; }
static int g_nt_cgw = -1;            // column-group width of the tile order of wide plain-row launches (0 = row-block-major over the whole width)
extern "C" int uvtg_debug_nt_cgw(int tiles_per_group) { if (tiles_per_group < 0 || tiles_per_group > 64) return -21; g_nt_cgw = tiles_per_group; return 0; }
static int g_nt_splitk_max = -1;
extern "C" int uvtg_debug_nt_small(int on) { if (on < 0 || on > 2) return -21; g_nt_small = on; return 0; }       // (2: 128-row tiles only)
extern "C" int uvtg_debug_nt_splitk(int max_parts) { if (max_parts < 0 || max_parts > 4) return -21; g_nt_splitk_max = max_parts; return 0; }
static int nt256_splitk_parts(long long tiles, int nk, int cus, int cap_units) {
  if (g_nt_splitk_max < 0) g_nt_splitk_max = uvtg_dev_env("UVTG_NT_SPLITK_MAX") ? atoi(uvtg_dev_env("UVTG_NT_SPLITK_MAX")) : 4;
  static const int max_tiles = uvtg_dev_env("UVTG_NT_SPLITK_MAX_TILES") ? atoi(uvtg_dev_env("UVTG_NT_SPLITK_MAX_TILES")) : 1 << 30;      // experiment: no split above this many tiles
  if (g_nt_splitk_max < 2 || tiles < 1 || tiles * 2 > cus || tiles > cap_units || tiles > max_tiles) return 0;
  long long parts = cus / tiles;
  if (parts > g_nt_splitk_max) parts = g_nt_splitk_max;
  if (parts > 4) parts = 4;                   // (the kernel's fold holds at most 4 parts)
  if (parts > nk / 4) parts = nk / 4;
  if (parts * tiles > cap_units) parts = cap_units / tiles;
  if (parts < 2) return 0;
  const int per = cdiv(nk, (int)parts);
  parts = cdiv(nk, per);                      // no empty part
  return parts >= 2 ? (int)parts : 0;
}
// Tile shape (tm = 1: 128 x 128, tm = 2: 128 x 256) and K parts of a launch that fits one tile per CU; {0, 0}: not such a launch.  Narrow
// tiles put twice the workgroups on a chip the launch cannot fill anyway, halve every epilogue and stage 32 instead of 48 KB per K tile; a K
// split shortens the serial K loop, but its last arriver publishes, waits for a ticket and folds alone.  Estimated microseconds, fitted on
// tools/nt_trace_infer.py at batch 1 / 32 (profiles/r04_nt_small_tile_phases.txt; split operands): K tile 0.68 / 1.10 us, epilogue 4 / 7 us,
// split: + 3 + 0.7 x parts / + 7 + 1.5 x parts for the part that finishes the tile.  The choice needs to be right only where the candidates
// differ by more than noise: batch 32 encoder GEMMs (216 narrow tiles, no split: 29 us against 39 for 108 wide ones x 2 parts), batch 32
// video projection (K = 2880: 76 wide tiles x 3 parts: 52 us against 68 for 152 unsplit narrow ones), batch 1 (8 narrow tiles x 4 parts).
#ifndef UVTG_NT_SMALL_TK1
#define UVTG_NT_SMALL_TK1 0.68
#endif
static long long nt_small_tiles(int rows, int N, int groups, int tm) {
  return (long long)cdiv(rows, 128) * cdiv(N, tm == 1 ? 128 : 256) * groups;
}
static SmallPlan nt256_small_plan(int rows, int N, int groups, int nk, int cus, int cap_units, bool have_ws, int mode) {
  SmallPlan best{0, 0, 0.0}; double bc = 1e30;
  for (int tm = 1; tm <= 2; tm++) {                     // 1: 128 x 128 tiles, 2: 128 x 256
    if (tm == 1 && mode == 2) continue;                 // (experiment switch: 128 x 256 tiles only)
    const long long tiles = nt_small_tiles(rows, N, groups, tm);
    if (tiles > cus) continue;
    int parts = have_ws ? nt256_splitk_parts(tiles, nk, cus, cap_units) : 0;
    if (parts < 2) parts = 1;
    const double cost = cdiv(nk, parts) * (tm == 1 ? UVTG_NT_SMALL_TK1 : 1.10) + (tm == 1 ? 4.0 : 7.0) + (parts > 1 ? (tm == 1 ? 3.0 + 0.7 * parts : 7.0 + 1.5 * parts) : 0.0);
    if (cost < bc) { bc = cost; best = SmallPlan{tm, parts, cost}; }
  }
  return best;
}
extern "C" int uvtg_debug_nt_splitk_parts(int M, int N, int K, int groups, int cus) {       // host arithmetic only (tests): parts a launch of this shape would get
  if (M <= 0 || N <= 0 || K < 64 || groups <= 0 || cus <= 0) return -20;
  const SmallPlan sp = nt256_small_plan(M, N, groups, K / 64, cus, UVTG_SK_UNITS, true, 1);
  return sp.parts > 1 ? sp.parts : 0;
}
extern "C" int uvtg_debug_nt_small_tile(int M, int N, int K, int groups, int cus) {         // ... and its tile width (128: 128 x 128, 256: 128 x 256; 0: not a single-tile launch)
  if (M <= 0 || N <= 0 || K < 64 || groups <= 0 || cus <= 0) return -20;
  return 128 * nt256_small_plan(M, N, groups, K / 64, cus, UVTG_SK_UNITS, true, 1).tm;
}
// Staging order per tile height (TM = 2, 3, 4, 5; the 320-row tiles were only measured interleaved).  Measured on the whole training step (tools/ord_ab.sh, same box, two rounds): the
// interleaved order wins 5-10 % on the bare main loop at every height (tools/nt_ab.py) but only the 256-row tiles keep a gain once the
// real epilogues run (conv / K = 3072 launches -7 %); 192-row tiles LOSE 3 % (their K tile has 24 MFMAs to cover the same pieces), so
// they stay on the round-1 order.  Experiment override: UVTG_NT_ORD="<o2><o3><o4>".
static int nt_order(int tm) {
  static int ord[4] = {-1, -1, -1, -1};
  if (ord[0] < 0) {
    static const int dflt[4] = {0, 0, 1, 1};
    const char* e = uvtg_dev_env("UVTG_NT_ORD");
    for (int i = 0; i < 4; i++) ord[i] = (e && strlen(e) >= 3 && (int)strlen(e) > i && (e[i] == '0' || e[i] == '1')) ? e[i] - '0' : dflt[i];
  }
  return ord[tm - 2];
}
#ifdef UVTG_NT_TRACE
static unsigned long long* g_trace_buf = nullptr;
static int g_trace_next = 0, g_trace_max = 0;
static int g_trace_info[256][8];
// buf: device [max_launches][256][16][4] u64 (zeroed by the caller); the next max_launches persistent-NT launches are traced
extern "C" int uvtg_debug_nt_trace(void* buf, int max_launches) {
  g_trace_buf = (unsigned long long*)buf; g_trace_next = 0; g_trace_max = max_launches < 256 ? max_launches : 256;
  return 0;
}
extern "C" int uvtg_debug_nt_trace_info(int i, int* out8) {
  if (i < 0 || i >= g_trace_next) return -1;
  for (int k = 0; k < 8; k++) out8[k] = g_trace_info[i][k];
  return 0;
}
static void nt_trace_launch(const GemmArgs& b, int tm, int grid, bool gather, bool eop, hipStream_t s) {
  unsigned long long* ptr = nullptr;
  if (g_trace_buf && g_trace_next < g_trace_max) {
    ptr = g_trace_buf + (size_t)g_trace_next * 256 * 16 * 4;
    int* o = g_trace_info[g_trace_next++];
    o[0] = b.M; o[1] = b.N; o[2] = b.K; o[3] = tm; o[4] = eop; o[5] = gather; o[6] = grid; o[7] = b.groups;
  }
  hipLaunchKernelGGL(nt_trace_set_kernel, dim3(1), dim3(1), 0, s, ptr);
}
#endif
static int launch_nt256(const GemmArgs& a, hipStream_t s, bool half = false) {
  if (int e = ensure_num_cu()) return e;
  GemmArgs b = a;
  if (b.groups <= 0) b.groups = 1;
  const bool gather = b.a_seg || b.o_seg || b.a_off || b.o_off || b.ktap != b.K || b.groups != 1 || b.o_rows || b.pos_map;
  if (b.deltaO && (gather || half || b.residB || b.gradPre || !b.delta || b.delta_S <= 0 || b.delta_H <= 0 || (b.delta_hd != 32 && b.delta_hd != 64 && b.delta_hd != 128) ||
                   b.N != b.delta_H * b.delta_hd || b.ldDO % 8)) return -2;
  const bool eop = b.residB || (b.actgrad && b.gradPre) || b.deltaO;
  static const bool epi_off = uvtg_dev_env("UVTG_NT_EPI_OFF") != nullptr;       // experiment: the general epilogue everywhere
  static const int epi_mask = uvtg_dev_env("UVTG_NT_EPI_MASK") ? atoi(uvtg_dev_env("UVTG_NT_EPI_MASK")) : 14;      // bit e: specialisation e allowed
  int epi = 0;
  if (!epi_off && !gather && b.outB && !b.resid && !b.outF && !b.outU && !b.outUF && !b.pos) {
    if (b.deltaO) epi = 4;
    else if (!b.outPre && !b.act && !b.actgrad && (!eop || b.residB)) epi = 1;
    else if (b.outPre && b.act == 2 && !b.actgrad && !eop && !b.rowscale) epi = 2;
    else if (!b.outPre && !b.act && b.actgrad == 2 && eop && !b.residB && !b.rowscale) epi = 3;
  }
  if (epi != 4 && !((epi_mask >> epi) & 1)) epi = 0;
  if (b.deltaO && epi != 4) return -2;       // (gemm_nt_delta_ok() is the contract: plain bf16 out, nothing else in the epilogue)
  const NtPlan plan = nt256_plan(b.M, b.N, b.K, b.groups, gather, eff_cus(), g_force_bm, eop, b.sk_slab && b.sk_tickets, b.sk_cap_units, !b.deltaO);
  if (!plan.tm1) return -21;
  const int M_all = b.M;
  int rc = 0;
  for (int part = 0; part < (plan.rows1 ? 2 : 1) && !rc; part++) {
    const int best_tm = part == 0 ? plan.tm1 : plan.tm2;
    b.m_begin = part == 0 ? 0 : plan.rows1;
    b.M = (part == 0 && plan.rows1) ? plan.rows1 : M_all;
    const int rows = b.M - b.m_begin;
    long long tiles = (long long)cdiv(rows, 64 * best_tm) * cdiv(b.N, 256) * b.groups;
    int grid = (int)(tiles < eff_cus() ? tiles : eff_cus());
    // At most one tile per CU: the single-tile variant (three-stage ring; bit-identical results) -- 128 x 128 tiles while those still fit
    // one per CU, and cut along K too where the caller gave the launch a workspace and the tiles leave half the chip idle and the shorter K
    // loop pays for the fold (nt256_small_plan; b.sk = parts per tile, 0 = the persistent kernel).
    b.sk = 0;
    {   // column-group tile order for wide outputs (plain row mapping): UVTG_NT_CGW = tiles per group (0 = off), default off until measured
      static const int cgw_env = uvtg_dev_env("UVTG_NT_CGW") ? atoi(uvtg_dev_env("UVTG_NT_CGW")) : 0;
      if (g_nt_cgw < 0) g_nt_cgw = cgw_env;
      b.cgw = (!gather && cdiv(b.N, 256) >= 8) ? g_nt_cgw : 0;
    }
    int small_tm = 0;
    if (g_nt_small < 0) g_nt_small = uvtg_dev_env("UVTG_NT_SMALL_OFF") ? 0 : (uvtg_dev_env("UVTG_NT_SMALL_TM1_OFF") ? 2 : 1);
    if (g_nt_small && !b.deltaO && (!plan.rows1 || part == 1) && best_tm == 2 && tiles <= eff_cus() && !g_force_tile && !g_force_bm && !(b.act >= 100 && b.act <= 103)) {
      const SmallPlan sp = nt256_small_plan(rows, b.N, b.groups, b.K / 64, eff_cus(), b.sk_cap_units, b.sk_slab && b.sk_tickets, g_nt_small);
      small_tm = sp.tm; b.sk = sp.parts;
      tiles = nt_small_tiles(rows, b.N, b.groups, small_tm);
      grid = (int)tiles * b.sk;
    }
#ifdef UVTG_NT_TRACE
    nt_trace_launch(b, small_tm ? small_tm : best_tm, grid, gather, eop, s);
#endif

    if (half) {       // split-operand launch: family 1, ALGORITHMIC flops (one product per element; the kernel runs three MFMA segments)
      uvtg_prof_begin_launch(1, 1.0 * rows * b.N * b.K * b.groups, s);       // (K counts both images: 2 M N K / 2)
      rc = small_tm == 1 ? launch_nt256_small<1>(b, grid, gather, eop, true, s) : small_tm == 2 ? launch_nt256_small<2>(b, grid, gather, eop, true, s) : best_tm == 5 ? launch_nt256_half<5>(b, grid, gather, s) : best_tm == 4 ? launch_nt256_half<4>(b, grid, gather, s)
         : (best_tm == 3 ? launch_nt256_half<3>(b, grid, gather, s) : launch_nt256_half<2>(b, grid, gather, s));
      uvtg_prof_end_launch(1, s);
      continue;
    }
    uvtg_prof_begin_launch(3, 2.0 * rows * b.N * b.K * b.groups, s);
    {   // algorithmic bytes of the launch: both operands once, every output once, the epilogue operands once
      const double mn = (double)rows * b.N * b.groups;
      double by = 2.0 * ((double)rows * b.K + (double)b.N * b.K) * b.groups;
      by += mn * ((b.outB ? 2 : 0) + (b.outF ? 4 : 0) + (b.outPre ? 2 : 0) + (b.outU ? 2 : 0) + (b.outUF ? 4 : 0));
      by += mn * ((b.residB ? 2 : 0) + (b.resid ? 4 : 0) + ((b.actgrad && b.gradPre) ? 2 : 0) + (b.pos ? 4 : 0) + (b.deltaO ? 2 : 0));
      uvtg_prof_add_bytes(3, by);
    }
    if (g_nt_lw < 0) g_nt_lw = uvtg_dev_env("UVTG_NT_LW") ? atoi(uvtg_dev_env("UVTG_NT_LW")) & 7 : 7;      // bit (TM - 2) = loader waves at that tile height (128 / 192 / 256 rows)
    const int lw_mask = g_nt_lw;
    if (small_tm) rc = small_tm == 1 ? launch_nt256_small<1>(b, grid, gather, eop, false, s) : launch_nt256_small<2>(b, grid, gather, eop, false, s);
    else if (!gather && best_tm <= 4 && ((lw_mask >> (best_tm - 2)) & 1) && !(eop && epi == 0 && best_tm >= 4))
      rc = best_tm == 4 ? launch_nt256_lw<4>(b, grid, eop, epi, s) : (best_tm == 3 ? launch_nt256_lw<3>(b, grid, eop, epi, s) : launch_nt256_lw<2>(b, grid, eop, epi, s));
    else if (nt_order(best_tm) == 0)
      rc = best_tm == 5 ? launch_nt256_tm<5, 0>(b, grid, gather, eop, epi, s) : best_tm == 4 ? launch_nt256_tm<4, 0>(b, grid, gather, eop, epi, s) : (best_tm == 3 ? launch_nt256_tm<3, 0>(b, grid, gather, eop, epi, s) : launch_nt256_tm<2, 0>(b, grid, gather, eop, epi, s));
    else
      rc = best_tm == 5 ? launch_nt256_tm<5, 1>(b, grid, gather, eop, epi, s) : best_tm == 4 ? launch_nt256_tm<4, 1>(b, grid, gather, eop, epi, s) : (best_tm == 3 ? launch_nt256_tm<3, 1>(b, grid, gather, eop, epi, s) : launch_nt256_tm<2, 1>(b, grid, gather, eop, epi, s));
    uvtg_prof_end_launch(3, s);
  }
  if (rc) return rc;
  UVTG_CHECK_LAUNCH();
  return 0;
}

// can this launch carry the attention-backward delta in its epilogue?  (the persistent 256-wide kernel with the plain row mapping)
bool gemm_nt_delta_ok(const GemmArgs& a) {
  static const bool off = uvtg_dev_env("UVTG_DELTA_FUSE_OFF") != nullptr;       // experiment: attn_delta_kernel's own pass
  if (off || g_delta_fuse == 0) return false;
  if (check_nt(a, 2) || ensure_num_cu() || !nt256_ok(a)) return false;
  const int groups = a.groups > 0 ? a.groups : 1;
  const bool gather = a.a_seg || a.o_seg || a.a_off || a.o_off || a.ktap != a.K || groups != 1 || a.o_rows || a.pos_map;
  static const bool epi_off = uvtg_dev_env("UVTG_NT_EPI_OFF") != nullptr;
  return !epi_off && !gather && !a.residB && !a.gradPre && !a.resid && !a.outF && !a.outU && !a.outUF && !a.pos && a.outB && !a.outPre && !a.act && !a.actgrad && !a.rowscale;
}
int launch_gemm_nt_bf16(const GemmArgs& a, hipStream_t s) {
  if (int e = check_nt(a, 2)) return e;
  if (a.deltaO && !nt256_ok(a)) return -2;
  if (nt256_ok(a)) return launch_nt256(a, s);
  dim3 grid(cdiv(a.M, BM) * cdiv(a.N, BN), 1, a.groups > 0 ? a.groups : 1);
  uvtg_prof_begin_launch(0, 2.0 * a.M * a.N * a.K * grid.z, s);
  hipLaunchKernelGGL(gemm_nt_kernel<false>, grid, dim3(256), 0, s, a);
  uvtg_prof_end_launch(0, s);
  UVTG_CHECK_LAUNCH();
  return 0;
}
// Split-operand GEMM (the precise "fp32x3" mode): a.A / a.B rows hold the fp16 hi / lo images interleaved in 32-column blocks; a.K, a.lda,
// a.ldb, a.ktap, a.gA, a.gB count elements of such rows (two per real column; the producers zero-pad to whole 64-element K tiles).  Both
// tile structures take it: the persistent 256-wide kernel where it pays, the 128 x 128 kernel for small shapes.
int launch_gemm_nt_split(const GemmArgs& a0, hipStream_t s) {
  GemmArgs a = a0;
  if (a.K % 64 || (a.ktap > 0 && a.ktap < a.K && a.ktap % 64)) return -2;       // whole K tiles: 32 real columns with both images
  if ((a.outS || a.outUS) && a.ldoS % 64) return -2;
  if (a.residB || a.gradPre || a.outPre) return -6;       // no bf16 epilogue operand in this mode (fp32 residual; fp32 / bf16 / split outputs)
  if (a.accscale == 0.f) a.accscale = 1.0f / (UVTG_SPLIT_A_SCALE * UVTG_SPLIT_W_SCALE);
  if (a.sscale == 0.f) a.sscale = UVTG_SPLIT_A_SCALE;
  if (int e = check_nt(a, 2)) return e;
  const double alg_flops = (double)a.M * a.N * a.K * (a.groups > 0 ? a.groups : 1);       // 2 x M x N x (K / 2 real columns): ALGORITHMIC flops
  if (nt256_ok(a)) return launch_nt256(a, s, true);
  dim3 grid(cdiv(a.M, BM) * cdiv(a.N, BN), 1, a.groups > 0 ? a.groups : 1);
  uvtg_prof_begin_launch(1, alg_flops, s);
  hipLaunchKernelGGL(gemm_nt_kernel<true>, grid, dim3(256), 0, s, a);
  uvtg_prof_end_launch(1, s);
  UVTG_CHECK_LAUNCH();
  return 0;
}
// ---- 256-tile weight-gradient path -----------------------------------------------------------------
static int g_tn_cus = 256;
static void tn256_splits(int M, int total_tiles, int& splits, int& steps_per) {
  const int steps_total = cdiv(M, 64);
  int cus = (g_cu_cap > 0 && g_cu_cap < g_tn_cus) ? g_cu_cap : g_tn_cus;
  if (g_cu_reserved > 0) cus = cus - g_cu_reserved > 8 ? cus - g_cu_reserved : 8;
  int want = cus / total_tiles;                        // tiles x splits ~ one unit per CU
  if (want < 1) want = 1;
  if (want > steps_total) want = steps_total;
  steps_per = cdiv(steps_total, want);
  splits = cdiv(steps_total, steps_per);               // no empty split
}
static void tn256_plan(int M, int N, int K, int& tiles_n, int& tiles_k, int& splits, int& steps_per) {
  tiles_n = cdiv(N, 256); tiles_k = cdiv(K, 256);
  tn256_splits(M, tiles_n * tiles_k, splits, steps_per);
}
long long gemm_tn_scratch_floats(int M, int N, int K) {
  int tn, tk, sp, per;
  tn256_plan(M, N, K, tn, tk, sp, per);
  return (long long)sp * tn * tk * 65536 + (long long)sp * tn * 256;
}
static bool tn256_group_ok(const GemmTNArgs& a) {
  if (g_force_tile == 128) return false;
  if (a.ktap < 0 || (a.ktap > 0 && (a.ktap % 256 || a.K % a.ktap))) return false;
  if (a.ldp % 8 || a.ldq % 8 || ((uintptr_t)a.P & 15) || ((uintptr_t)a.Q & 15)) return false;
  if (((long long)a.M + 64) * a.ldp * 2 >= (1LL << 31) || ((long long)a.Mq + 64) * a.ldq * 2 >= (1LL << 31)) return false;
  return a.M > 0 && a.N > 0 && a.K > 0;
}
static int tn_fill_plan(const GemmTNBatch& b, float* scratch, TNPlan& pl) {
  pl.count = b.count; pl.scratch = scratch;
  int tiles = 0;
  for (int i = 0; i < b.count; i++) {
    const GemmTNArgs& a = b.g[i];
    pl.g[i] = a;
    const int tn = cdiv(a.N, 256), tk = cdiv(a.K, 256);
    pl.tiles_k[i] = tk; pl.tile_base[i] = tiles; pl.n_pad[i] = tn * 256;
    pl.bytes_p[i] = (unsigned)((((long long)a.M - 1) * a.ldp + a.N) * 2);
    const int kq = a.ktap > 0 ? a.ktap : a.K;
    const int kcols = a.ldq < (kq + 7) / 8 * 8 ? a.ldq : (kq + 7) / 8 * 8;
    pl.bytes_q[i] = (unsigned)((((long long)a.Mq - 1) * a.ldq + kcols) * 2);
    tiles += tn * tk;
  }
  pl.tile_base[b.count] = tiles; pl.total_tiles = tiles;
  tn256_splits(b.g[0].M, tiles, pl.splits, pl.steps_per);
  long long off = (long long)pl.splits * tiles * 65536;
  for (int i = 0; i < b.count; i++) { pl.bias_base[i] = off; off += (long long)pl.splits * pl.n_pad[i]; }
  return 0;
}
long long gemm_tn_batch_scratch_floats(const GemmTNBatch& b) {
  TNPlan pl;
  tn_fill_plan(b, nullptr, pl);
  return pl.bias_base[b.count - 1] + (long long)pl.splits * pl.n_pad[b.count - 1];
}
bool gemm_tn_batch_ok(const GemmTNBatch& b) {
  if (b.count < 1 || b.count > UVTG_TN_MAX_GROUPS) return false;
  for (int i = 0; i < b.count; i++) {
    if (!tn256_group_ok(b.g[i]) || b.g[i].M != b.g[0].M) return false;
  }
  const GemmTNArgs& a0 = b.g[0];
  if (!a0.scratch || ((uintptr_t)a0.scratch & 15)) return false;
  return a0.scratch_floats >= gemm_tn_batch_scratch_floats(b);
}
int launch_gemm_tn_batch(const GemmTNBatch& b, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    if (hipError_t e = hipFuncSetAttribute((const void*)gemm_tn256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)) return (int)e;
    int dev = 0; hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) g_tn_cus = pr.multiProcessorCount;
    attr = true;
  }
  if (!gemm_tn_batch_ok(b)) return -2;
  TNPlan pl;
  tn_fill_plan(b, b.g[0].scratch, pl);
  double flops = 0; long long mx = 0;
  for (int i = 0; i < b.count; i++) {
    flops += 2.0 * b.g[i].M * b.g[i].N * b.g[i].K;
    const long long tot = (long long)b.g[i].N * ((b.g[i].K + 3) / 4);
    mx = tot > mx ? tot : mx;
  }
  uvtg_prof_begin_launch(2, flops, s);
#ifdef UVTG_NT_TRACE
  {
    GemmArgs ti; memset(&ti, 0, sizeof(ti));
    ti.M = -b.g[0].M; ti.N = pl.total_tiles; ti.K = pl.splits; ti.groups = b.count;
    nt_trace_launch(ti, pl.steps_per, pl.total_tiles * pl.splits, false, false, s);
  }
#endif
  hipLaunchKernelGGL(gemm_tn256_kernel, dim3(pl.total_tiles * pl.splits), dim3(512), 131072, s, pl);
  hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)((mx + 255) / 256), b.count), dim3(256), 0, s, pl);
  uvtg_prof_end_launch(2, s);
  UVTG_CHECK_LAUNCH();
  return 0;
}
// ---- hybrid (no reduce pass) launch of several weight gradients over the same rows -----------------------------------------
static int g_tnh_max_split = -1;
static void tnh_plan_counts(int M, int total_tiles, int cus, int& full_tiles, int& nsplit, int& steps_per) {
  const int steps_total = cdiv(M, 64);
  const int rem = total_tiles % cus;
  full_tiles = total_tiles - rem;
  nsplit = rem ? cus / rem : 0;
  if (g_tnh_max_split < 0) g_tnh_max_split = uvtg_dev_env("UVTG_TN_HYBRID_MAXSPLIT") ? atoi(uvtg_dev_env("UVTG_TN_HYBRID_MAXSPLIT")) : 3;
  if (nsplit > g_tnh_max_split && total_tiles > cus) nsplit = g_tnh_max_split;      // (launches of more than one round: fewer, longer parts instead of the fallback)
  if (nsplit > steps_total / 8) nsplit = steps_total / 8;          // (a part needs a real reduction)
  if (nsplit <= 1) { full_tiles = total_tiles; nsplit = 0; steps_per = steps_total; return; }
  steps_per = cdiv(steps_total, nsplit);
  nsplit = cdiv(steps_total, steps_per);                           // no empty part
}
static int tnh_cus() {
  int cus = (g_cu_cap > 0 && g_cu_cap < g_tn_cus) ? g_cu_cap : g_tn_cus;
  if (g_cu_reserved > 0) cus = cus - g_cu_reserved > 8 ? cus - g_cu_reserved : 8;
  return cus;
}
static int tnh_total_tiles(const GemmTNMulti& b) {
  int t = 0;
  for (int i = 0; i < b.count; i++) t += cdiv(b.g[i].N, 256) * cdiv(b.g[i].K, 256);
  return t;
}
long long gemm_tn_multi_slab_floats(int total_tiles, int cus_hint) {
  // the split parts of a launch never outnumber its CUs (one part per workgroup), and a tile has <= 3 parts (gemm_tn_multi_ok)
  const long long by_tiles = (long long)total_tiles * 3, by_cus = cus_hint > 0 ? cus_hint : 320;
  return (by_tiles < by_cus ? by_tiles : by_cus) * TNH_SLAB;
}
// (round 5: the groups of a launch may reduce over different row counts -- the last encoder layer's FFN runs on the clip rows only; the plan is
// laid out for the longest, a shorter group's tiles end early)
static int tnh_max_rows(const GemmTNMulti& b) { int m = 0; for (int i = 0; i < b.count; i++) m = b.g[i].M > m ? b.g[i].M : m; return m; }
static int tnh_min_rows(const GemmTNMulti& b) { int m = b.g[0].M; for (int i = 1; i < b.count; i++) m = b.g[i].M < m ? b.g[i].M : m; return m; }
bool gemm_tn_multi_ok(const GemmTNMulti& b) {
  if (g_force_tile == 128 || b.count < 1 || b.count > UVTG_TNH_MAX_GROUPS || !b.slabs || !b.tickets) return false;
  static const bool off = uvtg_dev_env("UVTG_TN_HYBRID_OFF") != nullptr;       // experiment: always the slab + reduce path
  if (off) return false;
  for (int i = 0; i < b.count; i++) {
    const GemmTNArgs& a = b.g[i];
    if (!tn256_group_ok(a) || !a.assign || a.Mq != a.M || a.col_stride < 1) return false;
    if (a.ktap ? (a.ktap % 256 || a.K % a.ktap) : (a.q_row_off != 0)) return false;                 // conv taps: whole k tiles per tap
    if (a.N % 256 || a.K % 256 || ((uintptr_t)a.out & 15) || (a.col_stride == 1 && a.ldo % 4)) return false;
  }
  if (((uintptr_t)b.slabs & 15)) return false;
  const int tiles = tnh_total_tiles(b);
  int full, nsplit, per;
  tnh_plan_counts(tnh_max_rows(b), tiles, tnh_cus(), full, nsplit, per);
  if (g_tnh_max_split < 0) g_tnh_max_split = uvtg_dev_env("UVTG_TN_HYBRID_MAXSPLIT") ? atoi(uvtg_dev_env("UVTG_TN_HYBRID_MAXSPLIT")) : 3;
  if (nsplit > g_tnh_max_split) return false;                       // the last arriver of a tile folds nsplit - 1 slabs alone: only short folds pay
  if (full == 0 && nsplit == 0) return false;
  if ((long long)(tiles - full) * nsplit * TNH_SLAB > b.slab_floats || tiles - full > b.n_tickets) return false;
  return tnh_min_rows(b) >= 2048;
}
int launch_gemm_tn_multi(const GemmTNMulti& b, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    if (hipError_t e = hipFuncSetAttribute((const void*)gemm_tn256h_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)) return (int)e;
    int dev = 0; hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) g_tn_cus = pr.multiProcessorCount;
    attr = true;
  }
  if (!gemm_tn_multi_ok(b)) return -2;
  TNHPlan pl;
  memset(&pl, 0, sizeof(pl));
  pl.count = b.count; pl.M = tnh_max_rows(b); pl.steps_total = cdiv(pl.M, 64);
  int tiles = 0;
  double flops = 0;
  for (int i = 0; i < b.count; i++) {
    const GemmTNArgs& a = b.g[i];
    TNHGroup& g = pl.g[i];
    g.P = a.P; g.Q = a.Q; g.out = a.out; g.dbias = a.dbias; g.ldp = a.ldp; g.ldq = a.ldq; g.ldo = a.ldo;
    g.tiles_k = a.K / 256; g.tile_base = tiles; g.steps_total = cdiv(a.M, 64);
    g.ktap = a.ktap; g.q_row_off = a.q_row_off; g.col_stride = a.col_stride;
    g.bytes_p = (unsigned)((((long long)a.M - 1) * a.ldp + a.N) * 2);
    g.bytes_q = (unsigned)((((long long)a.Mq - 1) * a.ldq + (a.ktap > 0 ? a.ktap : a.K)) * 2);
    tiles += (a.N / 256) * (a.K / 256);
    flops += 2.0 * a.M * a.N * a.K;
  }
  pl.total_tiles = tiles;
  const int cus = tnh_cus();
  tnh_plan_counts(pl.M, tiles, cus, pl.full_tiles, pl.nsplit, pl.steps_per);
  pl.slabs = b.slabs; pl.tickets = b.tickets; pl.sqsum = b.g[0].sqsum;
  const int split_units = (tiles - pl.full_tiles) * pl.nsplit;
  int grid = pl.full_tiles > split_units ? pl.full_tiles : split_units;
  if (grid > cus) grid = cus;
  uvtg_prof_begin_launch(2, flops, s);
  hipLaunchKernelGGL(gemm_tn256h_kernel, dim3(grid), dim3(512), 131072, s, pl);
  uvtg_prof_end_launch(2, s);
  UVTG_CHECK_LAUNCH();
  return 0;
}
static int launch_tn256(const GemmTNArgs& a, hipStream_t s) {
  GemmTNBatch b; b.count = 1; b.g[0] = a;
  return launch_gemm_tn_batch(b, s);
}
static bool tn256_ok(const GemmTNArgs& a) {
  if (!a.scratch) return false;
  GemmTNBatch b; b.count = 1; b.g[0] = a;
  if (!gemm_tn_batch_ok(b)) return false;
  if (g_force_tile == 256) return true;
  // worth it only when 256-wide tiles do not waste much of the output and there is a real reduction to split
  const double fill = ((double)a.N * a.K) / ((double)cdiv(a.N, 256) * 256 * (double)cdiv(a.K, 256) * 256);
  return fill >= 0.85 && a.M >= 2048;
}
bool gemm_tn_taps_ok(const GemmTNArgs& a) { return a.ktap > 0 && tn256_ok(a); }
int launch_gemm_tn_bf16(const GemmTNArgs& a, hipStream_t s) {
  if (a.M <= 0 || a.N <= 0 || a.K <= 0 || a.splits <= 0) return -1;
  if (tn256_ok(a)) return launch_tn256(a, s);
  if (a.ktap > 0) return -2;      // taps exist only in the 256-tile kernel: callers check gemm_tn_taps_ok() first
  if (a.ldp % 8 || a.ldq % 8 || ((uintptr_t)a.P & 15) || ((uintptr_t)a.Q & 15)) return -2;
  if (a.assign) {                 // the atomic kernel accumulates: give it a zeroed output (contiguous outputs only)
    if (a.col_stride != 1) return -2;
    if (hipError_t e = hipMemsetAsync(a.out, 0, (size_t)a.N * a.ldo * sizeof(float), s)) return (int)e;
  }
  dim3 grid(cdiv(a.N, 128) * cdiv(a.K, 128), a.splits, 1);
  uvtg_prof_begin_launch(2, 2.0 * a.M * a.N * a.K, s);
  hipLaunchKernelGGL(gemm_tn_kernel, grid, dim3(256), 0, s, a);
  uvtg_prof_end_launch(2, s);
  UVTG_CHECK_LAUNCH();
  if (a.assign && a.sqsum) return launch_sqsum(a.out, (long long)a.N * a.ldo, a.sqsum + 32, s);      // (small shapes: a pass over the matrix, into slot 0)
  return 0;
}
