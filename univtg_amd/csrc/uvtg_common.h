// Shared device/host helpers for the UniVTG gfx950 kernels (internal; the public C-ABI is include/uvtg.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;   // raw bfloat16 bits
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define UVTG_LOG_TINY (-103.27892990343184f)   // logf(1e-45f) evaluated in fp32 (denormal 2^-149)

// fp32 -> bf16, round-to-nearest-even, on the gfx950 converter (v_cvt_pk_bf16_f32: one instruction per pair instead of
// ~6 integer ops per element)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

// ---- fp16 hi/lo operand split of the precise ("fp32x3") GEMM mode -------------------------------------------------------------
// x * scale = hi + lo + O(2^-22 |x|): hi = fp16(x s) carries 11 significant bits, lo = fp16(x s - hi) the next 11 (lo may be an fp16
// subnormal: v_mfma_f32_32x32x16_f16 keeps subnormal inputs on gfx950 -- tools/mfma_f16_probe.hip, profiles/r04_mfma_f16_subnormal_probe.txt
// -- so its absolute error is <= 2^-25).  A product a.w is then hi_a hi_w + hi_a lo_w + lo_a hi_w (three fp16 MFMAs, fp32 accumulation)
// with the dropped lo.lo term at 2^-22: fp32-class, where the round-1..3 bf16 hi/lo split (8 + 8 bits) stopped at 2^-16.
// Operand scales (powers of two, folded back by GemmArgs.accscale): activations x16, weights x64 -- both keep |x s| far below fp16's
// 65504 for LayerNorm outputs / GELU / ReLU activations and for any sane weight, and lift the lo parts out of the subnormal range.
// LAYOUT of a split operand row: the two images are interleaved in blocks of 32 columns -- real column c lives at element
// split_col(c) (hi) and split_col(c) + 32 (lo) -- so that ONE 64-element K tile of the GEMM kernels (128 bytes, staged exactly like a bf16
// K tile) carries 32 real columns with both images: k-steps 0, 1 are the hi halves, k-steps 2, 3 the lo halves, and the K-tile body issues
// hi.hi, hi.lo and lo.hi from the same staged bytes and the same fragment reads (6 MFMA groups per 4 fragment sets: 2/3 of the LDS and
// staging traffic per MFMA of three separate passes).  Rows are zero-padded to a multiple of 64 real columns by their producers.
typedef __attribute__((ext_vector_type(8))) _Float16 half8_t;
__host__ __device__ __forceinline__ int split_col(int c) { return ((c >> 5) << 6) | (c & 31); }
#define UVTG_SPLIT_A_SCALE 16.0f
#define UVTG_SPLIT_W_SCALE 64.0f
__device__ __forceinline__ void split_f16(float x, unsigned short& hi, unsigned short& lo) {
  x = fminf(fmaxf(x, -65000.f), 65000.f);
  const _Float16 h = (_Float16)x;
  const _Float16 l = (_Float16)(x - (float)h);
  hi = __builtin_bit_cast(unsigned short, h); lo = __builtin_bit_cast(unsigned short, l);
}
// four consecutive columns -> the 8-byte hi group and the 8-byte lo group
__device__ __forceinline__ void split4_f16(const float (&v)[4], float scale, u32x2& hi, u32x2& lo) {
  unsigned short h[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; e++) split_f16(v[e] * scale, h[e], l[e]);
  hi[0] = h[0] | ((unsigned)h[1] << 16); hi[1] = h[2] | ((unsigned)h[3] << 16);
  lo[0] = l[0] | ((unsigned)l[1] << 16); lo[1] = l[2] | ((unsigned)l[3] << 16);
}
__device__ __forceinline__ f32x16 mfma32h(s16x8 a, s16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(s16x8 a, s16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// ds_read_b64_tr_b16: within each 16-lane group the 16 lanes x 4 elements form a [4][16] matrix in
// lane order (lane i supplies row i>>2, columns 4*(i&3)..+3); lane i receives column i (rows 0..3).
// Verified on gfx950 by tools/probe_tr.hip.
__device__ __forceinline__ s16x4 lds_tr16(const bf16_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// wave64 sum without the LDS crossbar: four DPP steps inside each row of 16 lanes, then the four row sums through SGPRs
// (6 dependent ds_bpermute round trips per __shfl_xor reduction otherwise -- per-row loops are latency chains); every lane gets the sum
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_mov<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);    // row_half_mirror
  v += dpp_mov<0x140>(v);    // row_mirror
  const int b = __float_as_int(v);      // (the builtin is typed int: a float argument would be converted by VALUE)
  return (__int_as_float(__builtin_amdgcn_readlane(b, 0)) + __int_as_float(__builtin_amdgcn_readlane(b, 16))) +
         (__int_as_float(__builtin_amdgcn_readlane(b, 32)) + __int_as_float(__builtin_amdgcn_readlane(b, 48)));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
// GELU(erf) and its derivative for the bf16 GEMM epilogues: erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16
// rounding of the result), one v_rcp + one v_exp shared by value and derivative -- erff() plus a second exponential made the
// activation-gradient epilogue ALU-bound (18 us per 192 x 256 tile against 8.5 us for the same bytes without it).
__device__ __forceinline__ void gelu_fast_parts(float x, float& cdf, float& pdf_x) {
#pragma clang fp contract(off)      // the same bits from every kernel instantiation that inlines this (contraction is a per-site scheduling choice)
  const float ax = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  const float e = __expf(-ax * ax);                         // exp(-x^2 / 2)
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float half_erfc = 0.5f * poly * t * e;              // 0.5 * erfc(|x| / sqrt 2)
  cdf = x >= 0.f ? 1.0f - half_erfc : half_erfc;
  pdf_x = x * 0.3989422804014327f * e;
}
__device__ __forceinline__ float gelu_fast(float x) {
#pragma clang fp contract(off)
  float c, p; gelu_fast_parts(x, c, p); return x * c;
}
__device__ __forceinline__ float gelu_fast_grad(float x) {
#pragma clang fp contract(off)
  float c, p; gelu_fast_parts(x, c, p); return c + p;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// Counter-based RNG (Philox-4x32-10) for dropout / DropPath: stateless, reproducible in backward.
__device__ __forceinline__ void philox4(unsigned long long seed, unsigned long long ctr_lo, unsigned ctr_hi, unsigned out[4]) {
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
  unsigned c0 = (unsigned)ctr_lo, c1 = (unsigned)(ctr_lo >> 32), c2 = ctr_hi, c3 = 0x5eed5eedu;
#pragma unroll
  for (int r = 0; r < 10; r++) {
    unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float u01(unsigned x) { return (x >> 8) * (1.0f / 16777216.0f); }

// RNG stream ids (ctr_hi) so that every stochastic op draws from its own stream.
enum { UVTG_RNG_IN_VID = 0x100, UVTG_RNG_IN_TXT = 0x200, UVTG_RNG_ATTN = 0x300, UVTG_RNG_PATH = 0x400, UVTG_RNG_TXT_POS = 0x500 };

#define UVTG_CHECK_LAUNCH() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
