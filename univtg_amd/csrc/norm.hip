// LayerNorm forward/backward for the UniVTG hot path (HBM-bound; one 64-lane wave per row, the row
// lives in registers, vectorised global accesses).  Forward fuses everything that consumes the
// normalised row: dropout (input projections, model/univtg.py:392-404), the bf16 / zero-padded GEMM
// operand, the (x + pos) copy that feeds the Q/K projection
// (model/transformer_encoder_droppath.py:116) and the zero-framed copy of the video rows that the
// Conv1d heads read (model/univtg.py:127-130).
#include "uvtg_kernels.h"
#include <cstdlib>

namespace {

template <int VEC> struct VecT;   // (VEC == 8: two 16-byte accesses in fp32, one in bf16)
template <> struct VecT<4> { typedef f32x4 T; };
template <> struct VecT<2> { typedef f32x2 T; };
template <> struct VecT<1> { typedef float T; };

template <int VEC> __device__ __forceinline__ void loadv(const float* p, float (&v)[VEC]) {
  if constexpr (VEC == 8) {
    const f32x4 t = *(const f32x4*)p, u = *(const f32x4*)(p + 4);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; v[4] = u[0]; v[5] = u[1]; v[6] = u[2]; v[7] = u[3];
  } else if constexpr (VEC == 4) { f32x4 t = *(const f32x4*)p; v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
  else if constexpr (VEC == 2) { f32x2 t = *(const f32x2*)p; v[0] = t[0]; v[1] = t[1]; }
  else v[0] = *p;
}
template <int VEC> __device__ __forceinline__ void storev(float* p, const float (&v)[VEC]) {
  if constexpr (VEC == 8) {
    *(f32x4*)p = (f32x4){v[0], v[1], v[2], v[3]};
    *(f32x4*)(p + 4) = (f32x4){v[4], v[5], v[6], v[7]};
  } else if constexpr (VEC == 4) { f32x4 t = {v[0], v[1], v[2], v[3]}; *(f32x4*)p = t; }
  else if constexpr (VEC == 2) { f32x2 t = {v[0], v[1]}; *(f32x2*)p = t; }
  else *p = v[0];
}
template <int VEC> __device__ __forceinline__ void loadb(const bf16_t* p, float (&v)[VEC]) {
  if constexpr (VEC == 8) {
    const u32x4 t = *(const u32x4*)p;
#pragma unroll
    for (int e = 0; e < 4; e++) { v[2 * e] = __uint_as_float(t[e] << 16); v[2 * e + 1] = __uint_as_float(t[e] & 0xffff0000u); }
  } else if constexpr (VEC == 4) {
    const u32x2 t = *(const u32x2*)p;
    v[0] = __uint_as_float(t[0] << 16); v[1] = __uint_as_float(t[0] & 0xffff0000u);
    v[2] = __uint_as_float(t[1] << 16); v[3] = __uint_as_float(t[1] & 0xffff0000u);
  } else if constexpr (VEC == 2) {
    const unsigned t = *(const unsigned*)p;
    v[0] = __uint_as_float(t << 16); v[1] = __uint_as_float(t & 0xffff0000u);
  } else v[0] = bf2f(*p);
}
template <int VEC> __device__ __forceinline__ void storeb(bf16_t* p, const float (&v)[VEC]) {
  if constexpr (VEC == 8) {
    u32x4 t; t[0] = pack_bf2(v[0], v[1]); t[1] = pack_bf2(v[2], v[3]); t[2] = pack_bf2(v[4], v[5]); t[3] = pack_bf2(v[6], v[7]);
    *(u32x4*)p = t;
  } else if constexpr (VEC == 4) { u32x2 t; t[0] = pack_bf2(v[0], v[1]); t[1] = pack_bf2(v[2], v[3]); *(u32x2*)p = t; }
  else if constexpr (VEC == 2) { *(unsigned*)p = pack_bf2(v[0], v[1]); }
  else *p = f2bf(v[0]);
}

// fp16 hi / lo images of VEC consecutive columns starting at real column c of a split operand row (interleaved layout, split_col)
template <int VEC> __device__ __forceinline__ void stores(unsigned short* row, int c, const float (&v)[VEC], float scale) {
  unsigned short* p = row + split_col(c);
  if constexpr (VEC == 8) {
    const float a[4] = {v[0], v[1], v[2], v[3]}, b[4] = {v[4], v[5], v[6], v[7]};
    u32x2 h0, l0, h1, l1; split4_f16(a, scale, h0, l0); split4_f16(b, scale, h1, l1);
    *(u32x4*)p = (u32x4){h0[0], h0[1], h1[0], h1[1]}; *(u32x4*)(p + 32) = (u32x4){l0[0], l0[1], l1[0], l1[1]};
  } else if constexpr (VEC == 4) {
    u32x2 h, l; split4_f16(v, scale, h, l);
    *(u32x2*)p = h; *(u32x2*)(p + 32) = l;
  } else {
#pragma unroll
    for (int e = 0; e < VEC; e++) { unsigned short h, l; split_f16(v[e] * scale, h, l); p[e] = h; p[32 + e] = l; }     // (VEC <= 2 at even c: same 32-block)
  }
}

// keep-mask for element column c of row `row` (4 consecutive columns share one Philox call)
__device__ __forceinline__ void drop_mask4(unsigned long long seed, unsigned stream, long long row, int c4, int D4,
                                           float p, float (&keep)[4]) {
  unsigned r[4];
  philox4(seed, (unsigned long long)row * (unsigned long long)D4 + (unsigned long long)c4, stream, r);
#pragma unroll
  for (int e = 0; e < 4; e++) keep[e] = (u01(r[e]) >= p) ? 1.0f / (1.0f - p) : 0.0f;
}
__device__ __forceinline__ float drop_scale(unsigned long long seed, unsigned stream, long long row, int c, int D4, float p) {
  float k[4];
  drop_mask4(seed, stream, row, c >> 2, D4, p, k);
  return k[c & 3];
}

// VEC == 2: the two lanes of an even/odd pair cover ONE Philox counter (4 columns) per iteration.  The even lane draws the counter
// of iteration 2*ip, the odd lane the one of 2*ip+1, and the halves are swapped with one cross-lane exchange: one Philox call per
// lane per TWO iterations instead of one per iteration (the 2818-wide input LayerNorm was ALU-bound on Philox, not HBM-bound).
__device__ __forceinline__ void drop_pair2(unsigned long long seed, unsigned stream, long long row, int ip, int lane, int D4,
                                           float p, float (&ks)[2][2]) {
  const int odd = lane & 1;
  const int c4 = ((2 * ip + odd) * 64 + (lane & ~1)) >> 1;
  unsigned r[4];
  philox4(seed, (unsigned long long)row * (unsigned long long)D4 + (unsigned long long)c4, stream, r);
  const unsigned s0 = odd ? r[0] : r[2], s1 = odd ? r[1] : r[3];
  const unsigned g0 = (unsigned)__shfl_xor((int)s0, 1, 64), g1 = (unsigned)__shfl_xor((int)s1, 1, 64);
  const float inv = 1.0f / (1.0f - p);
  ks[0][0] = (u01(odd ? g0 : r[0]) >= p) ? inv : 0.0f;      // iteration 2*ip   : the even lane's draw (words 0,1 | 2,3)
  ks[0][1] = (u01(odd ? g1 : r[1]) >= p) ? inv : 0.0f;
  ks[1][0] = (u01(odd ? r[2] : g0) >= p) ? inv : 0.0f;      // iteration 2*ip+1 : the odd lane's draw
  ks[1][1] = (u01(odd ? r[3] : g1) >= p) ? inv : 0.0f;
}

template <int VEC, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnFwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const int D = a.D, D4 = (D + 3) >> 2;
  for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < a.rows; row += gridDim.x * wpb) {
    const int lrow = a.src_rows ? a.src_rows[row] : row;          // row of the padded layout: dropout counter key (+ x row if gather_x)
    const int xrow = a.gather_x ? lrow : row;
    const float* xr = a.x ? a.x + (size_t)xrow * a.ldx : nullptr;
    const bf16_t* xbr = a.x ? nullptr : a.xB + (size_t)xrow * a.ldxB;
    float v[NV][VEC];
    float sum = 0.f;
    const float* addr = a.addtab ? a.addtab + (size_t)(row % a.add_L) * D : nullptr;     // text position embedding row (position_encoding.py:33-38)
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int c = (i * 64 + lane) * VEC;
      if (c < D) {
        if (xr) loadv<VEC>(xr + c, v[i]); else loadb<VEC>(xbr + c, v[i]);
        if (addr) {
          float t[VEC];
          loadv<VEC>(addr + c, t);
#pragma unroll
          for (int e = 0; e < VEC; e++) v[i][e] += t[e];
          if (a.xsum) storev<VEC>(a.xsum + (size_t)row * D + c, v[i]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < VEC; e++) v[i][e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < VEC; e++) sum += v[i][e];
    }
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int c = (i * 64 + lane) * VEC;
      if (c < D) {
#pragma unroll
        for (int e = 0; e < VEC; e++) { const float t = v[i][e] - mean; sq += t * t; }
      }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)D + a.eps);
    if (lane == 0) {
      if (a.mean) a.mean[row] = mean;
      if (a.rstd) a.rstd[row] = rstd;
    }
    // row classification for the positional / conv-frame copies
    int b = 0, s = 0;
    bool is_vid = false;
    if (a.S > 0 && !a.pos_row) { b = row / a.S; s = row - b * a.S; is_vid = s < a.Lv; }
    const float* posr = (a.pos && is_vid) ? a.pos + (size_t)(b * a.Lv + s) * D : nullptr;
    if (a.pos_row) { const int pr = a.pos_row[row]; posr = (a.pos && pr >= 0) ? a.pos + (size_t)pr * D : nullptr; }
    const size_t prow = is_vid ? (size_t)(b * (a.Lv + 2) + s + 1) : 0;
    float ks2[2][2] = {{1.f, 1.f}, {1.f, 1.f}};
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int c = (i * 64 + lane) * VEC;
      if constexpr (VEC == 2 && NV % 2 == 0) {            // all lanes take part in the exchange (outside the c < D guard)
        if ((i & 1) == 0 && a.p_drop > 0.f) drop_pair2(a.seed, a.stream_id, lrow, i >> 1, lane, D4, a.p_drop, ks2);
      }
      if (c < D) {
        float y[VEC], gm[VEC], bt[VEC];
        loadv<VEC>(a.gamma + c, gm);
        loadv<VEC>(a.beta + c, bt);
#pragma unroll
        for (int e = 0; e < VEC; e++) y[e] = (v[i][e] - mean) * rstd * gm[e] + bt[e];
        if (a.p_drop > 0.f) {
          if constexpr (VEC == 2 && NV % 2 == 0) {
#pragma unroll
            for (int e = 0; e < VEC; e++) y[e] *= ks2[i & 1][e];
          } else {
#pragma unroll
            for (int e = 0; e < VEC; e++) y[e] *= drop_scale(a.seed, a.stream_id, lrow, c + e, D4, a.p_drop);
          }
        }
        if (a.yF) storev<VEC>(a.yF + (size_t)row * a.ldyF + c, y);
        if (a.yS) stores<VEC>(a.yS + (size_t)row * a.ldyS, c, y, a.sscale);
        if (a.yB) storeb<VEC>(a.yB + (size_t)row * a.ldyB + c, y);
        if ((a.yU || a.yUF || a.yUS) && a.u_from_x) {       // u = x + LN(x + table): the row read above minus the table, at the row's padded position
          float u[VEC], t[VEC];
#pragma unroll
          for (int e = 0; e < VEC; e++) t[e] = 0.f;
          if (addr) loadv<VEC>(addr + c, t);
#pragma unroll
          for (int e = 0; e < VEC; e++) u[e] = y[e] + (v[i][e] - t[e]);
          if (a.yU) storeb<VEC>(a.yU + (size_t)lrow * a.ldyU + c, u);
          if (a.yUF) storev<VEC>(a.yUF + (size_t)lrow * a.ldyU + c, u);
          if (a.yUS) stores<VEC>(a.yUS + (size_t)lrow * a.ldyS, c, u, a.sscale);
        } else if (a.yU || a.yUF || a.yUS) {
          float u[VEC];
          if (posr) {
            float pv[VEC];
            loadv<VEC>(posr + c, pv);
#pragma unroll
            for (int e = 0; e < VEC; e++) u[e] = y[e] + pv[e];
          } else {
#pragma unroll
            for (int e = 0; e < VEC; e++) u[e] = y[e];
          }
          if (a.yU) storeb<VEC>(a.yU + (size_t)row * a.ldyU + c, u);
          if (a.yUF) storev<VEC>(a.yUF + (size_t)row * a.ldyU + c, u);
          if (a.yUS) stores<VEC>(a.yUS + (size_t)row * a.ldyS, c, u, a.sscale);
        }
        if (is_vid) {
          if (a.yP) storeb<VEC>(a.yP + prow * a.ldyP + c, y);
          if (a.yPF) storev<VEC>(a.yPF + prow * a.ldyP + c, y);
          if (a.yPS) stores<VEC>(a.yPS + prow * a.ldyS, c, y, a.sscale);
        }
      }
    }
    // zero padding columns [D, Dpad) of the GEMM operands
    if (a.Dpad > D) {
      for (int c = D + lane; c < a.Dpad; c += 64) {
        if (a.yB) a.yB[(size_t)row * a.ldyB + c] = 0;
        if (a.yS) { a.yS[(size_t)row * a.ldyS + split_col(c)] = 0; a.yS[(size_t)row * a.ldyS + split_col(c) + 32] = 0; }
      }
    }
  }
}

// The encoder's LayerNorm forward in the bf16 mode (round 5; D = 512 / 1024, bf16 rows in, no dropout): same math as ln_fwd_kernel, built
// like ln_bwd_lean_kernel for the HBM rate.  The generic kernel takes ONE row per wave through a chain load -> wave sum -> wave sum ->
// stores with ~20 run-time feature tests in it and re-reads gamma / beta per row (LN1: 112 MB in 31.5 us = 3.6 TB/s).  Here a wave walks
// its rows with the NEXT row's 16-byte pieces already in flight (still packed: 4 registers per piece), gamma / beta live in registers,
// and the position row of a clip token (fp32, the y + pos output of LN2) is requested before the two reductions that precede its use.
// Outputs: y (bf16), y + pos (bf16; clip rows: pos by (S, Lv) arithmetic or by the packed stream's row table), the clip rows' copy into
// the zero-framed conv layout (last layer), mean / rstd.
template <int NV>
__global__ __launch_bounds__(256, 4) void ln_fwd_lean_kernel(const LnFwdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const int D = a.D;
  float gm[NV][8], bt[NV][8];
#pragma unroll
  for (int i = 0; i < NV; i++) {
    const int c = (i * 64 + lane) * 8;
    const f32x4 g0 = *(const f32x4*)(a.gamma + c), g1 = *(const f32x4*)(a.gamma + c + 4);
    const f32x4 b0 = *(const f32x4*)(a.beta + c), b1 = *(const f32x4*)(a.beta + c + 4);
#pragma unroll
    for (int e = 0; e < 4; e++) { gm[i][e] = g0[e]; gm[i][4 + e] = g1[e]; bt[i][e] = b0[e]; bt[i][4 + e] = b1[e]; }
  }
  const int stride = gridDim.x * wpb;
  const float invD = 1.0f / (float)D;
  u32x4 px[NV];
  auto fetch = [&](int r) {
    const size_t rin = a.x_rows ? (size_t)a.x_rows[r] : (a.x_seg ? (size_t)(r / a.x_seg) * a.x_seg_stride + (size_t)(r % a.x_seg) : (size_t)r);      // (last layer: the clip rows of the stream)
#pragma unroll
    for (int i = 0; i < NV; i++) px[i] = *(const u32x4*)(a.xB + rin * a.ldxB + (i * 64 + lane) * 8);
  };
  int row = blockIdx.x * wpb + wave;
  if (row < a.rows) fetch(row);
  for (; row < a.rows; row += stride) {
    u32x4 cx[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) cx[i] = px[i];
    fetch(min(row + stride, a.rows - 1));
    // row classification (wave-uniform): position row of a clip token, its row in the conv frame
    int prow = -1; size_t frow = 0; bool is_vid = false;
    if (a.pos_row) prow = a.pos_row[row];
    else if (a.S > 0) { const int b = row / a.S, sidx = row - b * a.S; is_vid = sidx < a.Lv; if (is_vid) { prow = b * a.Lv + sidx; frow = (size_t)(b * (a.Lv + 2) + sidx + 1); } }
    const bool want_u = a.yU != nullptr;
    const bool have_pos = want_u && a.pos && prow >= 0;
    const size_t yrow = a.yB_rows ? (size_t)a.yB_rows[row] : (size_t)row;
    f32x4 pv[NV][2];
    if (have_pos) {
#pragma unroll
      for (int i = 0; i < NV; i++) {
        const float* pp = a.pos + (size_t)prow * D + (i * 64 + lane) * 8;
        pv[i][0] = *(const f32x4*)pp; pv[i][1] = *(const f32x4*)(pp + 4);
      }
    }
    float v[NV][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        v[i][2 * e] = __uint_as_float(cx[i][e] << 16); v[i][2 * e + 1] = __uint_as_float(cx[i][e] & 0xffff0000u);
        sum += v[i][2 * e]; sum += v[i][2 * e + 1];
      }
    const float mean = wave_sum_dpp(sum) * invD;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++)
#pragma unroll
      for (int e = 0; e < 8; e++) { const float t = v[i][e] - mean; sq += t * t; }
    const float rstd = rsqrtf(wave_sum_dpp(sq) * invD + a.eps);
    if (lane == 0) {
      if (a.mean) a.mean[row] = mean;
      if (a.rstd) a.rstd[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int c = (i * 64 + lane) * 8;
      float y[8];
#pragma unroll
      for (int e = 0; e < 8; e++) y[e] = (v[i][e] - mean) * rstd * gm[i][e] + bt[i][e];
      u32x4 t; t[0] = pack_bf2(y[0], y[1]); t[1] = pack_bf2(y[2], y[3]); t[2] = pack_bf2(y[4], y[5]); t[3] = pack_bf2(y[6], y[7]);
      if (a.yB) *(u32x4*)(a.yB + yrow * a.ldyB + c) = t;
      if (is_vid && a.yP) *(u32x4*)(a.yP + frow * a.ldyP + c) = t;
      if (want_u) {
        u32x4 u = t;
        if (have_pos) {
          u[0] = pack_bf2(y[0] + pv[i][0][0], y[1] + pv[i][0][1]); u[1] = pack_bf2(y[2] + pv[i][0][2], y[3] + pv[i][0][3]);
          u[2] = pack_bf2(y[4] + pv[i][1][0], y[5] + pv[i][1][1]); u[3] = pack_bf2(y[6] + pv[i][1][2], y[7] + pv[i][1][3]);
        }
        *(u32x4*)(a.yU + (size_t)row * a.ldyU + c) = u;
      }
    }
  }
}

// dx = rstd * (gh - mean(gh) - xhat * mean(gh * xhat)),  gh = g * gamma;  dgamma += g * xhat; dbeta += g
// RPW rows per wave are in flight together (the row loop is a chain load -> two wave reductions -> store: one row at a time
// leaves the kernel latency-bound at half the HBM rate).
#ifndef UVTG_LN_RPW
#define UVTG_LN_RPW 1
#define UVTG_LN_WPE 4   // second __launch_bounds__ argument = waves per SIMD (not workgroups per CU)
#endif
template <int VEC, int NV, bool BF>     // BF: x, g, g2 are bf16 (the fast mode's streams); else fp32
__global__ __launch_bounds__(NV * VEC > 32 ? 128 : 512, (NV * VEC <= 16) ? UVTG_LN_WPE : 1) void ln_bwd_kernel(const LnBwdArgs a) {
  constexpr int RPW = (NV * VEC > 32) ? 1 : (NV * VEC > 16 ? 2 : UVTG_LN_RPW);
  extern __shared__ float red[];       // [waves][2][D] per-wave dgamma / dbeta partials
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const int D = a.D, D4 = (D + 3) >> 2;
  float dgam[NV][VEC], dbet[NV][VEC], gm[NV][VEC];
#pragma unroll
  for (int i = 0; i < NV; i++) {
    const int c = (i * 64 + lane) * VEC;
#pragma unroll
    for (int e = 0; e < VEC; e++) { dgam[i][e] = 0.f; dbet[i][e] = 0.f; gm[i][e] = 0.f; }
    if (c < D) loadv<VEC>(a.gamma + c, gm[i]);
  }
  const int stride = gridDim.x * wpb;
  const bool have_g = BF ? (a.gB != nullptr) : (a.g != nullptr);
  const bool have_g2 = BF ? (a.g2B != nullptr) : (a.g2 != nullptr);

  for (int row0 = blockIdx.x * wpb + wave; row0 < a.rows; row0 += stride * RPW) {
    // ---- all loads of the RPW rows first, no control flow between them (so they are all in flight together) ----
    float xv[RPW][NV][VEC], gv[RPW][NV][VEC];
    float mean[RPW], rstd[RPW];
    int rowi[RPW], lrow[RPW];
#pragma unroll
    for (int rr = 0; rr < RPW; rr++) {
      rowi[rr] = min(row0 + rr * stride, a.rows - 1);      // clamped duplicate row: loaded, never stored / accumulated
      const size_t row = (size_t)rowi[rr];
      lrow[rr] = a.src_rows ? a.src_rows[row] : rowi[rr];
      const size_t xrow = a.gather_x ? (size_t)lrow[rr] : row;
      mean[rr] = a.mean[row]; rstd[rr] = a.rstd[row];
#pragma unroll
      for (int i = 0; i < NV; i++) {
        const int c = (i * 64 + lane) * VEC;
        if (c < D) {
          if constexpr (BF) loadb<VEC>(a.xB + xrow * a.ldxB + c, xv[rr][i]);
          else loadv<VEC>(a.x + xrow * a.ldx + c, xv[rr][i]);
        } else {
#pragma unroll
          for (int e = 0; e < VEC; e++) xv[rr][i][e] = 0.f;
        }
      }
    }
    if (have_g) {
#pragma unroll
      for (int rr = 0; rr < RPW; rr++) {
        const size_t row = (size_t)rowi[rr];
#pragma unroll
        for (int i = 0; i < NV; i++) {
          const int c = (i * 64 + lane) * VEC;
          if (c < D) {
            if constexpr (BF) loadb<VEC>(a.gB + row * a.ldgB + c, gv[rr][i]);
            else loadv<VEC>(a.g + row * a.ldg + c, gv[rr][i]);
          } else {
#pragma unroll
            for (int e = 0; e < VEC; e++) gv[rr][i][e] = 0.f;
          }
        }
      }
    } else {
#pragma unroll
      for (int rr = 0; rr < RPW; rr++)
#pragma unroll
        for (int i = 0; i < NV; i++)
#pragma unroll
          for (int e = 0; e < VEC; e++) gv[rr][i][e] = 0.f;
    }
    if (have_g2) {
#pragma unroll
      for (int rr = 0; rr < RPW; rr++) {
        size_t r2 = (size_t)rowi[rr];
        bool on = true;
        if (a.g2_S > 0) {
          const int b = rowi[rr] / a.g2_S, s = rowi[rr] - b * a.g2_S;
          on = s < a.g2_Lv;
          r2 = (size_t)(b * a.g2_Lv + s);
        }
        if (on) {
#pragma unroll
          for (int i = 0; i < NV; i++) {
            const int c = (i * 64 + lane) * VEC;
            if (c < D) {
              float t[VEC];
              if constexpr (BF) loadb<VEC>(a.g2B + r2 * a.ldg2B + c, t);
              else loadv<VEC>(a.g2 + r2 * a.ldg2 + c, t);
#pragma unroll
              for (int e = 0; e < VEC; e++) gv[rr][i][e] += t[e];
            }
          }
        }
      }
    }
    // ---- math ----
#pragma unroll
    for (int rr = 0; rr < RPW; rr++) {
      const int row = row0 + rr * stride;
      const bool live = row < a.rows;
      unsigned pm[NV];                  // bit e set: x > 0 (ReLU mask of the producing layer)
      float s1 = 0.f, s2 = 0.f;
      float ks2[2][2] = {{1.f, 1.f}, {1.f, 1.f}};
#pragma unroll
      for (int i = 0; i < NV; i++) {
        const int c = (i * 64 + lane) * VEC;
        pm[i] = 0;
        if (a.relu_from_x) {
#pragma unroll
          for (int e = 0; e < VEC; e++) pm[i] |= (xv[rr][i][e] > 0.f ? 1u : 0u) << e;
        }
        if constexpr (VEC == 2 && NV % 2 == 0) {
          if ((i & 1) == 0 && a.p_drop > 0.f) drop_pair2(a.seed, a.stream_id, lrow[rr], i >> 1, lane, D4, a.p_drop, ks2);
#pragma unroll
          for (int e = 0; e < VEC; e++) gv[rr][i][e] *= ks2[i & 1][e];
        } else if (a.p_drop > 0.f && c < D) {
#pragma unroll
          for (int e = 0; e < VEC; e++) gv[rr][i][e] *= drop_scale(a.seed, a.stream_id, lrow[rr], c + e, D4, a.p_drop);
        }
#pragma unroll
        for (int e = 0; e < VEC; e++) {
          const float xh = (c < D) ? (xv[rr][i][e] - mean[rr]) * rstd[rr] : 0.f;
          const float g = live ? gv[rr][i][e] : 0.f;
          dgam[i][e] += g * xh;
          dbet[i][e] += g;
          xv[rr][i][e] = xh;                      // reuse: xhat
          gv[rr][i][e] = g * gm[i][e];            // reuse: g * gamma
          s1 += gv[rr][i][e];
          s2 += gv[rr][i][e] * xh;
        }
      }
      const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
      const float rs = (live && a.rowscale) ? a.rowscale[a.row_sample ? a.row_sample[row] : row / a.rs_seg] : 1.0f;
#pragma unroll
      for (int i = 0; i < NV; i++) {
        const int c = (i * 64 + lane) * VEC;
        if (live && c < D) {
          float dx[VEC];
#pragma unroll
          for (int e = 0; e < VEC; e++) dx[e] = rstd[rr] * (gv[rr][i][e] - c1 - xv[rr][i][e] * c2);
          if (a.relu_from_x) {
#pragma unroll
            for (int e = 0; e < VEC; e++) dx[e] = ((pm[i] >> e) & 1u) ? dx[e] : 0.f;
          }
          if (a.dxF) storev<VEC>(a.dxF + (size_t)row * a.lddxF + c, dx);
          if (a.dxB2) storeb<VEC>(a.dxB2 + (size_t)row * a.lddxB2 + c, dx);
          if (a.dxB) {
#pragma unroll
            for (int e = 0; e < VEC; e++) dx[e] *= rs;
            storeb<VEC>(a.dxB + (size_t)row * a.lddxB + c, dx);
          }
        }
      }
    }
  }
  if (a.dgamma) {
    // cross-wave reduction through wave-private LDS slabs (LDS float atomics from 8 waves on the same 2 D addresses were
    // ~55 % of this kernel's time), then one coalesced global atomic pass per block
    float* mine = red + (size_t)wave * 2 * D;
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int c = (i * 64 + lane) * VEC;
      if (c < D) {
#pragma unroll
        for (int e = 0; e < VEC; e++) { mine[c + e] = dgam[i][e]; mine[D + c + e] = dbet[i][e]; }
      }
    }
    __syncthreads();
    // Hundreds of blocks adding into the same 2 D floats serialise in L2 (same-line atomics ~0.1 us each): with caller
    // scratch the block partials go out as plain stores and ln_bwd_reduce_kernel folds them; atomics only as fallback.
    for (int c = threadIdx.x; c < 2 * D; c += blockDim.x) {
      float t = 0.f;
      for (int w = 0; w < wpb; w++) t += red[(size_t)w * 2 * D + c];
      if (a.partial) a.partial[(size_t)blockIdx.x * 2 * D + c] = t;
      else atomicAdd((c < D ? a.dgamma + c : a.dbeta + (c - D)), t);
    }
  }
}
// The encoder's LayerNorm backward (bf16 streams, D <= 1024, no dropout / ReLU mask): same math as ln_bwd_kernel, built for the
// HBM rate -- 164 MB per launch at config 2 (x, g in; dx scaled + unscaled out).
//  * the next row's x / g (/ g2) are fetched, still PACKED (4 registers per 16 bytes), before the current row's math;
//  * dgamma / dbeta accumulate in the wave's LDS slab and gamma is read from LDS: 48 fewer live registers than ln_bwd_kernel,
//    which spilled 20 dwords inside the row loop at its 128-register (4 waves per SIMD) budget.
template <int NV>
__global__ __launch_bounds__(512, 4) void ln_bwd_lean_kernel(const LnBwdArgs a) {
  extern __shared__ float red[];       // [waves][2][D] dgamma / dbeta partials, then gamma [D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const int D = a.D;
  float* mine = red + (size_t)wave * 2 * D;
  float* sgam = red + (size_t)wpb * 2 * D;
  for (int c = threadIdx.x; c < D; c += blockDim.x) sgam[c] = a.gamma[c];
  for (int c = lane; c < 2 * D; c += 64) mine[c] = 0.f;
  __syncthreads();
  const int stride = gridDim.x * wpb;
  const bool have_g2 = a.g2B != nullptr;
  u32x4 px[NV], pg[NV], pg2[NV];
  float pmean = 0.f, prstd = 0.f;
  auto xrow_of = [&](int r) -> size_t { return a.x_rows ? (size_t)a.x_rows[r] : (a.x_seg ? (size_t)(r / a.x_seg) * a.x_seg_stride + (size_t)(r % a.x_seg) : (size_t)r); };
  auto fetch = [&](int r) {            // r is clamped by the caller: always a legal row
    const size_t row = (size_t)r, xrow = xrow_of(r);
    pmean = a.mean[row]; prstd = a.rstd[row];
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int c = (i * 64 + lane) * 8;
      px[i] = (u32x4){0, 0, 0, 0}; pg[i] = (u32x4){0, 0, 0, 0}; pg2[i] = (u32x4){0, 0, 0, 0};
      if (c < D) {
        px[i] = *(const u32x4*)(a.xB + xrow * a.ldxB + c);
        if (a.gB) pg[i] = *(const u32x4*)(a.gB + row * a.ldgB + c);       // (the top layer has only the heads' gradient: g2)
      }
    }
    if (have_g2) {
      size_t r2 = row;
      bool on = true;
      if (a.g2_rows) r2 = (size_t)a.g2_rows[r];
      else if (a.g2_S > 0) {
        const int b = r / a.g2_S, sidx = r - b * a.g2_S;
        on = sidx < a.g2_Lv;
        r2 = (size_t)(b * a.g2_Lv + sidx);
      }
      if (on) {
#pragma unroll
        for (int i = 0; i < NV; i++) {
          const int c = (i * 64 + lane) * 8;
          if (c < D) pg2[i] = *(const u32x4*)(a.g2B + r2 * a.ldg2B + c);
        }
      }
    }
  };
  int row = blockIdx.x * wpb + wave;
  if (row < a.rows) fetch(row);
  for (; row < a.rows; row += stride) {
    u32x4 cx[NV], cg[NV], cg2[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) { cx[i] = px[i]; cg[i] = pg[i]; cg2[i] = pg2[i]; }
    const float mean = pmean, rstd = prstd;
    fetch(min(row + stride, a.rows - 1));
    float xh[NV][8], gh[NV][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int c = (i * 64 + lane) * 8;
      if (c < D) {
        float g[8];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          xh[i][2 * e] = (__uint_as_float(cx[i][e] << 16) - mean) * rstd;
          xh[i][2 * e + 1] = (__uint_as_float(cx[i][e] & 0xffff0000u) - mean) * rstd;
          g[2 * e] = __uint_as_float(cg[i][e] << 16) + __uint_as_float(cg2[i][e] << 16);
          g[2 * e + 1] = __uint_as_float(cg[i][e] & 0xffff0000u) + __uint_as_float(cg2[i][e] & 0xffff0000u);
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
          f32x4 dg = *(f32x4*)(mine + c + 4 * h), db = *(f32x4*)(mine + D + c + 4 * h);
          const f32x4 gm = *(const f32x4*)(sgam + c + 4 * h);
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const float gv = g[4 * h + e], x = xh[i][4 * h + e];
            dg[e] += gv * x; db[e] += gv;
            const float t = gv * gm[e];
            gh[i][4 * h + e] = t;
            s1 += t; s2 += t * x;
          }
          *(f32x4*)(mine + c + 4 * h) = dg; *(f32x4*)(mine + D + c + 4 * h) = db;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; e++) { xh[i][e] = 0.f; gh[i][e] = 0.f; }
      }
    }
    const float c1 = wave_sum_dpp(s1) / (float)D, c2 = wave_sum_dpp(s2) / (float)D;
    const float rs = a.rowscale ? a.rowscale[a.row_sample ? a.row_sample[row] : row / a.rs_seg] : 1.0f;
    const size_t orow = xrow_of(row);
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int c = (i * 64 + lane) * 8;
      if (c < D) {
        float dx[8];
#pragma unroll
        for (int e = 0; e < 8; e++) dx[e] = rstd * (gh[i][e] - c1 - xh[i][e] * c2);
        if (a.dxF) storev<8>(a.dxF + orow * a.lddxF + c, dx);
        if (a.dxB2) storeb<8>(a.dxB2 + orow * a.lddxB2 + c, dx);
        if (a.dxB) {
#pragma unroll
          for (int e = 0; e < 8; e++) dx[e] *= rs;
          storeb<8>(a.dxB + orow * a.lddxB + c, dx);
        }
      }
    }
  }
  // clip-row launch of the last encoder layer: the rows of the gradient stream that have no clip row behind them (text rows) are zero
  for (int i = blockIdx.x * wpb + wave; i < a.zero_n; i += stride) {
    const long long zr = a.zero_tab ? (long long)a.zero_tab[i] : (long long)(i / a.zero_seg) * a.zero_stride + a.zero_off + i % a.zero_seg;
    if (zr < 0) continue;
    const u32x4 z = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < NV; k++) {
      const int c = (k * 64 + lane) * 8;
      if (c < D) {
        if (a.dxB) *(u32x4*)(a.dxB + (size_t)zr * a.lddxB + c) = z;
        if (a.dxB2) *(u32x4*)(a.dxB2 + (size_t)zr * a.lddxB2 + c) = z;
      }
    }
  }
  if (a.dgamma) {
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * D; c += blockDim.x) {
      float t = 0.f;
      for (int w = 0; w < wpb; w++) t += red[(size_t)w * 2 * D + c];
      if (a.partial) a.partial[(size_t)blockIdx.x * 2 * D + c] = t;
      else atomicAdd((c < D ? a.dgamma + c : a.dbeta + (c - D)), t);
    }
  }
}
// Folds the per-block partials [nblocks][2 D] into dgamma / dbeta.  grid = (2 D / 64 column chunks, row splits): a block owns 64
// consecutive columns (256-byte rows of the partial matrix: full lines) and a slice of the partial rows, 4 row lanes per column,
// 4 independent loads in flight per thread; the few row splits meet in fp32 atomics (<= 8 per address).
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const LnBwdArgs a, int nblocks) {
  __shared__ float red2[4][64];
  const int il = threadIdx.x & 63, bl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + il, n = 2 * a.D;
  const int per = (nblocks + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(b0 + per, nblocks);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < n) {
    const float* pp = a.partial + c;
    int b = b0 + bl;
    for (; b + 12 < b1; b += 16) {
      s0 += pp[(size_t)b * n]; s1 += pp[(size_t)(b + 4) * n]; s2 += pp[(size_t)(b + 8) * n]; s3 += pp[(size_t)(b + 12) * n];
    }
    for (; b < b1; b += 4) s0 += pp[(size_t)b * n];
  }
  red2[bl][il] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (bl == 0 && c < n) {
    const float t = (red2[0][il] + red2[1][il]) + (red2[2][il] + red2[3][il]);
    if (gridDim.y == 1) { if (c < a.D) a.dgamma[c] += t; else a.dbeta[c - a.D] += t; }
    else atomicAdd(c < a.D ? a.dgamma + c : a.dbeta + (c - a.D), t);
  }
}

// The same fold for SEVERAL LayerNorm launches at once (blockIdx.z = launch): the encoder's 2 E backward launches keep their partials in
// their own buffers and are folded by one launch behind the encoder loop (single-rank steps: nobody waits for a layer's gamma / beta gradient).
__global__ __launch_bounds__(256) void ln_bwd_reduce_multi_kernel(const LnReduceMulti m) {
  __shared__ float red2[4][64];
  const int z = blockIdx.z, il = threadIdx.x & 63, bl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + il, n = 2 * m.D, nblocks = m.nblocks[z];
  const int per = (nblocks + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(b0 + per, nblocks);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < n) {
    const float* pp = m.partial[z] + c;
    int b = b0 + bl;
    for (; b + 12 < b1; b += 16) {
      s0 += pp[(size_t)b * n]; s1 += pp[(size_t)(b + 4) * n]; s2 += pp[(size_t)(b + 8) * n]; s3 += pp[(size_t)(b + 12) * n];
    }
    for (; b < b1; b += 4) s0 += pp[(size_t)b * n];
  }
  red2[bl][il] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (bl == 0 && c < n) {
    const float t = (red2[0][il] + red2[1][il]) + (red2[2][il] + red2[3][il]);
    atomicAdd(c < m.D ? m.dgamma[z] + c : m.dbeta[z] + (c - m.D), t);
  }
}

// ------------------------------------------------------------------------------------------------
// Wide rows (the 2818-wide feature LayerNorm of the video projection, model/univtg.py:91-100).  One wave per row keeps
// 48 values + gamma/beta in registers per lane (218-450 VGPRs: one or two waves per SIMD); here a whole 256-thread block
// takes a row, 2 columns x NVW per thread, and the two row statistics go through LDS -- 8 waves per SIMD instead.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum256(float v, float* sh) {     // all 256 threads get the sum; sh: 4 floats, reused safely
  v = wave_sum(v);
  __syncthreads();                         // previous use of sh is over
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}
// keep-scales of the 2 columns this thread owns at column-iteration k (columns (k*256 + tid)*2 ..+1) of `row`: the even/odd
// thread pair shares one Philox counter (4 columns); even threads draw for even k, odd threads for odd k, halves are exchanged
__device__ __forceinline__ void drop_pair_wide(unsigned long long seed, unsigned stream, long long row, int kp, int tid, int D4,
                                               float p, float (&ks)[2][2]) {
  const int odd = tid & 1;
  const int c4 = ((2 * kp + odd) * 256 + (tid & ~1)) >> 1;
  unsigned r[4];
  philox4(seed, (unsigned long long)row * (unsigned long long)D4 + (unsigned long long)c4, stream, r);
  const unsigned s0 = odd ? r[0] : r[2], s1 = odd ? r[1] : r[3];
  const unsigned g0 = (unsigned)__shfl_xor((int)s0, 1, 64), g1 = (unsigned)__shfl_xor((int)s1, 1, 64);
  const float inv = 1.0f / (1.0f - p);
  ks[0][0] = (u01(odd ? g0 : r[0]) >= p) ? inv : 0.0f;
  ks[0][1] = (u01(odd ? g1 : r[1]) >= p) ? inv : 0.0f;
  ks[1][0] = (u01(odd ? r[2] : g0) >= p) ? inv : 0.0f;
  ks[1][1] = (u01(odd ? r[3] : g1) >= p) ? inv : 0.0f;
}
template <int NVW>      // NVW even: column iterations of 512 columns each
__global__ __launch_bounds__(256) void ln_fwd_wide_kernel(const LnFwdArgs a) {
  __shared__ float sh[4];
  const int tid = threadIdx.x, D = a.D, D4 = (D + 3) >> 2;
  // (measured: issuing the next row's loads before the reductions of the current one changes nothing -- the kernel is bound by the
  // Philox draws of the input dropout, 20 quarter-rate 32 x 32 -> 64 bit multiplies per 4 columns, not by the load -> reduce chain)
  for (int row = blockIdx.x; row < a.rows; row += gridDim.x) {
    const int lrow = a.src_rows ? a.src_rows[row] : row;
    const float* xr = a.x + (size_t)(a.gather_x ? lrow : row) * a.ldx;
    float v[NVW][2];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NVW; k++) {
      const int c = (k * 256 + tid) * 2;
      v[k][0] = 0.f; v[k][1] = 0.f;
      if (c < D) loadv<2>(xr + c, v[k]);
      sum += v[k][0] + v[k][1];
    }
    const float mean = block_sum256(sum, sh) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < NVW; k++) {
      const int c = (k * 256 + tid) * 2;
      if (c < D) { const float t0 = v[k][0] - mean, t1 = v[k][1] - mean; sq += t0 * t0 + t1 * t1; }
    }
    const float rstd = rsqrtf(block_sum256(sq, sh) / (float)D + a.eps);
    if (tid == 0) {
      if (a.mean) a.mean[row] = mean;
      if (a.rstd) a.rstd[row] = rstd;
    }
    float ks[2][2] = {{1.f, 1.f}, {1.f, 1.f}};
#pragma unroll
    for (int k = 0; k < NVW; k++) {
      const int c = (k * 256 + tid) * 2;
      if ((k & 1) == 0 && a.p_drop > 0.f) drop_pair_wide(a.seed, a.stream_id, lrow, k >> 1, tid, D4, a.p_drop, ks);
      if (c < D) {
        float gm[2], bt[2], y[2];
        loadv<2>(a.gamma + c, gm);
        loadv<2>(a.beta + c, bt);
#pragma unroll
        for (int e = 0; e < 2; e++) y[e] = ((v[k][e] - mean) * rstd * gm[e] + bt[e]) * ks[k & 1][e];
        if (a.yS) stores<2>(a.yS + (size_t)row * a.ldyS, c, y, a.sscale);
        if (a.yB) storeb<2>(a.yB + (size_t)row * a.ldyB + c, y);
      }
    }
    if (a.Dpad > D) {
      for (int c = D + tid; c < a.Dpad; c += 256) {
        if (a.yB) a.yB[(size_t)row * a.ldyB + c] = 0;
        if (a.yS) { a.yS[(size_t)row * a.ldyS + split_col(c)] = 0; a.yS[(size_t)row * a.ldyS + split_col(c) + 32] = 0; }
      }
    }
  }
}
// Round 3: the same with ONE WAVE per row.  The block-per-row kernel above pays two block-wide reductions (four barriers) per row on a
// load -> reduce -> reduce -> draw -> store latency chain: 90 us for the 268 MB of the video features at config 2 (3 TB/s).  Here a lane owns
// whole column quads c4 = lane + 64 k (two 8-byte loads each: the rows are only 8-byte aligned), the two row sums are DPP wave sums (no LDS, no
// barrier), and the dropout draw is one Philox call per owned quad -- the same (row, column) -> mask mapping as everywhere else
// (counter = row * ceil(D / 4) + c / 4, word c % 4).  NQ = quads per lane (12 covers D <= 3072).
// Round 6: FULL = number of leading quad groups k that are complete for EVERY lane (64 (k + 1) * 4 <= D, chosen by the host: 11 at D = 2818) -- those
// groups carry no per-lane column guards at all (the guarded code was 216 exec-mask branches per row for a kernel that is VALU-bound: 12 Philox
// calls + 48 hi / lo splits per lane and row); only the groups k >= FULL keep them.  The fused multiply-adds are spelled out so that both
// instantiations round alike (left to the compiler, the guarded code fused sq += t * t and the unguarded one did not: one sample of the index
// clause's 1024 moved across its fp32 tie).
template <int NQ, int FULL>
__global__ __launch_bounds__(256) void ln_fwd_wide_wave_kernel(const LnFwdArgs a) {
  const int lane = threadIdx.x & 63, D = a.D, D4 = (D + 3) >> 2;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  const int lrow = a.src_rows ? a.src_rows[row] : row;
  const float* xr = a.x + (size_t)(a.gather_x ? lrow : row) * a.ldx;
  float v[NQ][4];
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < NQ; k++) {
    const int c = (lane + 64 * k) * 4;
    v[k][0] = v[k][1] = v[k][2] = v[k][3] = 0.f;
    if (k < FULL || c + 1 < D) { const f32x2 t = *(const f32x2*)(xr + c); v[k][0] = t[0]; v[k][1] = t[1]; }
    if (k < FULL || c + 3 < D) { const f32x2 t = *(const f32x2*)(xr + c + 2); v[k][2] = t[0]; v[k][3] = t[1]; }
    sum += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
  }
  const float mean = wave_sum_dpp(sum) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int k = 0; k < NQ; k++) {
    const int c = (lane + 64 * k) * 4;
#pragma unroll
    for (int e = 0; e < 4; e++) if (k < FULL || c + e < D) { const float t = v[k][e] - mean; sq = __builtin_fmaf(t, t, sq); }
  }
  const float rstd = rsqrtf(wave_sum_dpp(sq) / (float)D + a.eps);
  if (lane == 0) {
    if (a.mean) a.mean[row] = mean;
    if (a.rstd) a.rstd[row] = rstd;
  }
  const float inv = a.p_drop > 0.f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
  const int dmax = a.Dpad > D ? a.Dpad : D;
#pragma unroll
  for (int k = 0; k < NQ; k++) {
    const int c4 = lane + 64 * k, c = c4 * 4;
    if (k >= FULL && c >= dmax) continue;
    float y[4] = {0.f, 0.f, 0.f, 0.f};
    if (k < FULL || c < D) {
      float ks[4] = {1.f, 1.f, 1.f, 1.f};
      if (a.p_drop > 0.f) {
        unsigned r[4];
        philox4(a.seed, (unsigned long long)lrow * (unsigned long long)D4 + (unsigned long long)c4, a.stream_id, r);
#pragma unroll
        for (int e = 0; e < 4; e++) ks[e] = (u01(r[e]) >= a.p_drop) ? inv : 0.0f;
      }
      if (k < FULL || c + 3 < D) {
        const f32x4 gm = *(const f32x4*)(a.gamma + c), bt = *(const f32x4*)(a.beta + c);
#pragma unroll
        for (int e = 0; e < 4; e++) y[e] = __builtin_fmaf((v[k][e] - mean) * rstd, gm[e], bt[e]) * ks[e];
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++)
          if (c + e < D) y[e] = __builtin_fmaf((v[k][e] - mean) * rstd, a.gamma[c + e], a.beta[c + e]) * ks[e];
      }
    }
    // (columns [D, Dpad) are written as zeros: the GEMM operand is zero-padded to whole K tiles)
    const bool whole = k < FULL || c + 3 < dmax;
    if (a.yS) {      // fp16 hi / lo images of the split-operand projection GEMM (y == 0 in the padding columns: both images get zeros)
      unsigned short* o = a.yS + (size_t)row * a.ldyS + split_col(c);       // (c is a multiple of 4: the quad stays inside one 32-column block)
      if (whole) { u32x2 h, l; split4_f16(y, a.sscale, h, l); *(u32x2*)o = h; *(u32x2*)(o + 32) = l; }
      else {
#pragma unroll
        for (int e = 0; e < 4; e++) if (c + e < dmax) { unsigned short h, l; split_f16(y[e] * a.sscale, h, l); o[e] = h; o[32 + e] = l; }
      }
    }
    if (a.yB) {
      bf16_t* o = a.yB + (size_t)row * a.ldyB + c;
      if (whole) { u32x2 t; t[0] = pack_bf2(y[0], y[1]); t[1] = pack_bf2(y[2], y[3]); *(u32x2*)o = t; }
      else {
#pragma unroll
        for (int e = 0; e < 4; e++) if (c + e < dmax) o[e] = f2bf(y[e]);
      }
    }
  }
}
// dgamma / dbeta only (the feature LayerNorm has no upstream: dx is never needed), i.e. a pure column reduction over the rows:
// thread = 2 columns, block = 512 columns x a slice of rows, per-block partials folded by ln_bwd_reduce_kernel.
// GB (round 6): the upstream gradient is a bf16 stream (a.gB) -- the projection dgrad GEMM that produces it then writes, and this pass reads, half
// the bytes; x stays the fp32 feature rows.
template <bool GB>
__global__ __launch_bounds__(256) void ln_dgb_wide_kernel(const LnBwdArgs a, int rows_per_block) {
  const int tid = threadIdx.x, D = a.D, D4 = (D + 3) >> 2;
  const int c = (blockIdx.x * 256 + tid) * 2;
  const bool on = c < D;
  const int cc = on ? c : 0;                       // inactive threads still take part in the pair exchange
  const int r0 = blockIdx.y * rows_per_block, r1 = min(a.rows, r0 + rows_per_block);
  float dg[2] = {0.f, 0.f}, db[2] = {0.f, 0.f};
  for (int r = r0; r < r1; r += 2) {
    const int rb = min(r + 1, r1 - 1);             // second row of the pair (duplicate of the first on an odd tail: weight 0)
    const float wb = (r + 1 < r1) ? 1.f : 0.f;
    float ga[2], xa[2], gb[2], xb[2];
    const int lr = a.src_rows ? a.src_rows[r] : r, lrb = a.src_rows ? a.src_rows[rb] : rb;
    if constexpr (GB) { loadb<2>(a.gB + (size_t)r * a.ldgB + cc, ga); loadb<2>(a.gB + (size_t)rb * a.ldgB + cc, gb); }
    else { loadv<2>(a.g + (size_t)r * a.ldg + cc, ga); loadv<2>(a.g + (size_t)rb * a.ldg + cc, gb); }
    loadv<2>(a.x + (size_t)(a.gather_x ? lr : r) * a.ldx + cc, xa);
    loadv<2>(a.x + (size_t)(a.gather_x ? lrb : rb) * a.ldx + cc, xb);
    const float ma = a.mean[r], sa = a.rstd[r], mb = a.mean[rb], sb = a.rstd[rb];
    float ka[2] = {1.f, 1.f}, kb[2] = {1.f, 1.f};
    if (a.p_drop > 0.f) {
      // even threads draw the pair's counter for row r, odd threads for row rb; halves exchanged (one Philox call per thread per 2 rows)
      const int odd = tid & 1;
      const int c4 = c >> 2;                       // the even/odd pair covers columns 4*c4 .. 4*c4+3 (also when one of them is past D)
      unsigned q[4];
      philox4(a.seed, (unsigned long long)(odd ? lrb : lr) * (unsigned long long)D4 + (unsigned long long)c4, a.stream_id, q);
      const unsigned s0 = odd ? q[0] : q[2], s1 = odd ? q[1] : q[3];
      const unsigned g0 = (unsigned)__shfl_xor((int)s0, 1, 64), g1 = (unsigned)__shfl_xor((int)s1, 1, 64);
      const float inv = 1.0f / (1.0f - a.p_drop);
      ka[0] = (u01(odd ? g0 : q[0]) >= a.p_drop) ? inv : 0.f; ka[1] = (u01(odd ? g1 : q[1]) >= a.p_drop) ? inv : 0.f;
      kb[0] = (u01(odd ? q[2] : g0) >= a.p_drop) ? inv : 0.f; kb[1] = (u01(odd ? q[3] : g1) >= a.p_drop) ? inv : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const float g0 = ga[e] * ka[e], g1 = gb[e] * kb[e] * wb;
      dg[e] += g0 * ((xa[e] - ma) * sa) + g1 * ((xb[e] - mb) * sb);
      db[e] += g0 + g1;
    }
  }
  if (on) {
    float* out = a.partial + (size_t)blockIdx.y * 2 * D;
    out[c] = dg[0]; out[c + 1] = dg[1]; out[D + c] = db[0]; out[D + c + 1] = db[1];
  }
}

template <int VEC, int NV> int run_fwd(const LnFwdArgs& a, hipStream_t s) {
  const int blocks = min(cdiv(a.rows, 4), 8192);
  hipLaunchKernelGGL((ln_fwd_kernel<VEC, NV>), dim3(blocks), dim3(256), 0, s, a);
  UVTG_CHECK_LAUNCH();
  return 0;
}
#ifndef UVTG_LN_BLOCKS
#define UVTG_LN_BLOCKS 512
#endif
static int ln_bwd_wpb(int D) {       // waves per block: as many as fit 64 KB of per-wave partial slabs (8 at D <= 1024, 2 at D = 2818); ~12 rows per wave
  int wpb = (int)(65536 / (2 * (size_t)D * sizeof(float)));
  return wpb >= 8 ? 8 : (wpb >= 4 ? 4 : (wpb >= 2 ? 2 : 1));
}
static int ln_bwd_blocks(int rows, int D) { const int wpb = ln_bwd_wpb(D); return min(cdiv(rows, wpb), max(1, (UVTG_LN_BLOCKS) * 8 / wpb)); }
template <int VEC, int NV> int run_bwd(const LnBwdArgs& a, hipStream_t s) {
  const bool bf = a.x == nullptr;
  if (bf ? ((a.g != nullptr) || (a.g2 != nullptr) || !a.xB) : ((a.gB != nullptr) || (a.g2B != nullptr))) return -4;   // streams are all bf16 or all fp32
  const int wpb = ln_bwd_wpb(a.D);
  const int blocks = ln_bwd_blocks(a.rows, a.D);
  LnBwdArgs b = a;
  if (!b.dgamma || b.partial_floats < (long long)blocks * 2 * b.D) b.partial = nullptr;
  const bool defer = b.defer_blocks && b.partial;      // the caller folds the partials (launch_ln_bwd_reduce_multi)
  if (b.defer_blocks) *b.defer_blocks = defer ? blocks : 0;
  if constexpr (VEC == 8 && NV <= 2) {
    static const bool lean_off = uvtg_dev_env("UVTG_LN_LEAN_OFF") != nullptr;       // experiment: the generic kernel
    if (bf && (a.gB || a.g2B) && a.p_drop == 0.f && !a.relu_from_x && !a.gather_x && wpb == 8 && !lean_off) {
      hipLaunchKernelGGL((ln_bwd_lean_kernel<NV>), dim3(blocks), dim3(512), (size_t)(wpb * 2 + 1) * a.D * sizeof(float), s, b);
      if (b.partial && !defer) hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(cdiv(2 * b.D, 64), blocks >= 128 ? 8 : 1), dim3(256), 0, s, b, blocks);
      UVTG_CHECK_LAUNCH();
      return 0;
    }
  }
  if (a.x_seg || a.x_rows || a.g2_rows || a.zero_n) return -4;      // (the clip-row maps exist in the lean kernel only: the engine asks ln_clip_rows_ok first)
  if (bf) hipLaunchKernelGGL((ln_bwd_kernel<VEC, NV, true>), dim3(blocks), dim3(64 * wpb), (size_t)wpb * 2 * a.D * sizeof(float), s, b);
  else hipLaunchKernelGGL((ln_bwd_kernel<VEC, NV, false>), dim3(blocks), dim3(64 * wpb), (size_t)wpb * 2 * a.D * sizeof(float), s, b);
  if (b.partial && !defer) hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(cdiv(2 * b.D, 64), blocks >= 128 ? 8 : 1), dim3(256), 0, s, b, blocks);
  UVTG_CHECK_LAUNCH();
  return 0;
}

}  // namespace

long long ln_bwd_partial_floats(int rows, int D) { return (long long)ln_bwd_blocks(rows, D) * 2 * D; }
int launch_ln_bwd_reduce_multi(const LnReduceMulti& m, hipStream_t s) {
  if (m.count <= 0) return 0;
  if (m.count > UVTG_LN_MULTI_MAX) return -17;
  int mx = 0;
  for (int i = 0; i < m.count; i++) mx = max(mx, m.nblocks[i]);
  hipLaunchKernelGGL(ln_bwd_reduce_multi_kernel, dim3(cdiv(2 * m.D, 64), mx >= 128 ? 8 : 1, m.count), dim3(256), 0, s, m);
  UVTG_CHECK_LAUNCH();
  return 0;
}

#define LN_DISPATCH(FN, ARGS)                                                          \
  const int D = ARGS.D;                                                                 \
  if (D % 8 == 0 && alignv8 && D >= 512) {                                              \
    if (D <= 512) return FN<8, 1>(ARGS, s);                                             \
    if (D <= 1024) return FN<8, 2>(ARGS, s);                                            \
    if (D <= 2048) return FN<8, 4>(ARGS, s);                                            \
    if (D <= 4096) return FN<8, 8>(ARGS, s);                                            \
  }                                                                                     \
  if (D % 4 == 0 && align16) {                                                          \
    if (D <= 256) return FN<4, 1>(ARGS, s);                                             \
    if (D <= 1024) return FN<4, 4>(ARGS, s);                                            \
    if (D <= 2048) return FN<4, 8>(ARGS, s);                                            \
    if (D <= 4096) return FN<4, 16>(ARGS, s);                                           \
  } else if (D % 2 == 0 && align8) {                                                    \
    if (D <= 512) return FN<2, 4>(ARGS, s);                                             \
    if (D <= 2048) return FN<2, 16>(ARGS, s);                                           \
    if (D <= 3072) return FN<2, 24>(ARGS, s);                                           \
  } else {                                                                              \
    if (D <= 512) return FN<1, 8>(ARGS, s);                                             \
    if (D <= 3072) return FN<1, 48>(ARGS, s);                                           \
  }                                                                                     \
  return -4;

static bool al(const void* p, int ld_elems, int bytes_per, int want) {
  return p == nullptr || ((((uintptr_t)p) % want == 0) && ((size_t)ld_elems * bytes_per) % want == 0);
}

static int g_ln_fwd_lean = -1;       // 1 (default): the encoder's bf16 LayerNorm launches take ln_fwd_lean_kernel; 0: the generic kernel (parity tests / A-B)
extern "C" int uvtg_debug_ln_fwd_lean(int on) { g_ln_fwd_lean = on ? 1 : 0; return 0; }
static int launch_ln_fwd_impl(const LnFwdArgs& a, hipStream_t s) {
  if (a.rows <= 0) return 0;
  if (a.addtab && (a.D > 2048 || a.add_L <= 0)) return -4;      // (only the generic kernel adds the table)
  const bool align16 = al(a.x, a.ldx, 4, 16) && al(a.xB, a.ldxB, 2, 8) && al(a.yF, a.ldyF, 4, 16) && al(a.yS, a.ldyS, 2, 8) && al(a.yUS, a.ldyS, 2, 8) && al(a.yPS, a.ldyS, 2, 8) &&
                       al(a.yB, a.ldyB, 2, 8) && al(a.yU, a.ldyU, 2, 8) && al(a.yUF, a.ldyU, 4, 16) &&
                       al(a.yP, a.ldyP, 2, 8) && al(a.yPF, a.ldyP, 4, 16) && al(a.gamma, 0, 4, 16) && al(a.beta, 0, 4, 16) &&
                       al(a.pos, a.D, 4, 16);
  const bool align8 = al(a.x, a.ldx, 4, 8) && al(a.xB, a.ldxB, 2, 4) && al(a.yF, a.ldyF, 4, 8) && al(a.yS, a.ldyS, 2, 4) && al(a.yUS, a.ldyS, 2, 4) && al(a.yPS, a.ldyS, 2, 4) &&
                      al(a.yB, a.ldyB, 2, 4) && al(a.yU, a.ldyU, 2, 4) && al(a.yUF, a.ldyU, 4, 8) &&
                      al(a.yP, a.ldyP, 2, 4) && al(a.yPF, a.ldyP, 4, 8) && al(a.gamma, 0, 4, 8) && al(a.beta, 0, 4, 8) &&
                      al(a.pos, a.D, 4, 8);
  const bool alignv8 = align16 && al(a.xB, a.ldxB, 2, 16) && al(a.yB, a.ldyB, 2, 16) && al(a.yU, a.ldyU, 2, 16) && al(a.yP, a.ldyP, 2, 16) &&
                       al(a.yS, a.ldyS, 2, 16) && al(a.yUS, a.ldyS, 2, 16) && al(a.yPS, a.ldyS, 2, 16);
  if (a.D > 2048 && a.D <= 3072 && a.D % 2 == 0 && align8 && a.x && !a.yF && !a.pos && !a.pos_row && !a.yU && !a.yUF && !a.yP && !a.yPF && !a.addtab && !a.yUS && !a.yPS) {
    static const bool wave_off = uvtg_dev_env("UVTG_LN_WIDE_WAVE_OFF") != nullptr;     // experiment: the block-per-row kernel
    const int dmax = a.Dpad > a.D ? a.Dpad : a.D;
    if (!wave_off && dmax <= 3072 && (a.ldx % 2 == 0) && (!a.yB || a.ldyB % 4 == 0) && (!a.yS || a.ldyS % 4 == 0) && al(a.gamma, 0, 4, 16) && al(a.beta, 0, 4, 16))
    {                                                                                                     // wave per row
      static const bool full_off = uvtg_dev_env("UVTG_LN_WIDE_FULL_OFF") != nullptr;                     // experiment: every quad group guarded (rounds 3-5)
      if (a.D >= 11 * 256 && !full_off) hipLaunchKernelGGL((ln_fwd_wide_wave_kernel<12, 11>), dim3(cdiv(a.rows, 4)), dim3(256), 0, s, a);
      else hipLaunchKernelGGL((ln_fwd_wide_wave_kernel<12, 0>), dim3(cdiv(a.rows, 4)), dim3(256), 0, s, a);
    }
    else
      hipLaunchKernelGGL((ln_fwd_wide_kernel<6>), dim3(min(a.rows, 4096)), dim3(256), 0, s, a);         // block per row (see the kernel)
    UVTG_CHECK_LAUNCH();
    return 0;
  }
  {   // the encoder's bf16 LayerNorms (round 5): lean kernel, next row in flight
    if (g_ln_fwd_lean < 0) g_ln_fwd_lean = uvtg_dev_env("UVTG_LN_FWD_LEAN_OFF") ? 0 : 1;
    const bool lean_off = g_ln_fwd_lean == 0;
    const bool plain = a.xB && !a.x && a.p_drop == 0.f && !a.yF && !a.yS && !a.yUS && !a.yPS && !a.yUF && !a.yPF && !a.addtab && !a.src_rows && !a.u_from_x &&
                       !a.xsum && (a.Dpad <= a.D) && (a.D == 1024 || a.D == 512) && alignv8 && !(a.pos_row && a.yP);
    if (plain && !lean_off && (a.rows >= 1024 || a.x_seg || a.x_rows || a.yB_rows)) {
      static const int blocks_env = uvtg_dev_env("UVTG_LN_FWD_BLOCKS") ? atoi(uvtg_dev_env("UVTG_LN_FWD_BLOCKS")) : 1024;      // (A/B: 512 / 1024 / 2048 measured in round 5)
      const int blocks = min(cdiv(a.rows, 4), blocks_env > 0 ? blocks_env : 1024);
      if (a.D == 1024) hipLaunchKernelGGL((ln_fwd_lean_kernel<2>), dim3(blocks), dim3(256), 0, s, a);
      else hipLaunchKernelGGL((ln_fwd_lean_kernel<1>), dim3(blocks), dim3(256), 0, s, a);
      UVTG_CHECK_LAUNCH();
      return 0;
    }
  }
  if (a.x_seg || a.x_rows || a.yB_rows) return -4;         // (the clip-row maps exist in the lean kernel only)
  LN_DISPATCH(run_fwd, a)
}
// the clip-row launches of the last encoder layer need BOTH lean kernels (bf16 streams, D = 512 / 1024, neither switched off)
bool ln_clip_rows_ok(int D) {
  if (g_ln_fwd_lean < 0) g_ln_fwd_lean = uvtg_dev_env("UVTG_LN_FWD_LEAN_OFF") ? 0 : 1;
  static const bool bwd_off = uvtg_dev_env("UVTG_LN_LEAN_OFF") != nullptr;
  return (D == 512 || D == 1024) && g_ln_fwd_lean == 1 && !bwd_off;
}

static int launch_ln_bwd_impl(const LnBwdArgs& a, hipStream_t s) {
  if (a.defer_blocks) *a.defer_blocks = 0;      // (set by the row kernels' launcher when it leaves the fold to the caller)
  if (a.rows <= 0) return 0;
  const bool align16 = al(a.x, a.ldx, 4, 16) && al(a.g, a.ldg, 4, 16) && al(a.g2, a.ldg2, 4, 16) && al(a.xB, a.ldxB, 2, 8) &&
                       al(a.gB, a.ldgB, 2, 8) && al(a.g2B, a.ldg2B, 2, 8) && al(a.dxB2, a.lddxB2, 2, 8) &&
                       al(a.dxF, a.lddxF, 4, 16) && al(a.dxB, a.lddxB, 2, 8) && al(a.gamma, 0, 4, 16);
  const bool align8 = al(a.x, a.ldx, 4, 8) && al(a.g, a.ldg, 4, 8) && al(a.g2, a.ldg2, 4, 8) && al(a.xB, a.ldxB, 2, 4) &&
                      al(a.gB, a.ldgB, 2, 4) && al(a.g2B, a.ldg2B, 2, 4) && al(a.dxB2, a.lddxB2, 2, 4) &&
                      al(a.dxF, a.lddxF, 4, 8) && al(a.dxB, a.lddxB, 2, 4) && al(a.gamma, 0, 4, 8);
  const bool alignv8 = align16 && al(a.xB, a.ldxB, 2, 16) && al(a.gB, a.ldgB, 2, 16) && al(a.g2B, a.ldg2B, 2, 16) &&
                       al(a.dxB, a.lddxB, 2, 16) && al(a.dxB2, a.lddxB2, 2, 16);
  if (a.D > 2048 && a.D % 2 == 0 && align8 && a.x && (a.g != nullptr) != (a.gB != nullptr) && !a.g2 && !a.g2B && a.dgamma && a.dbeta && !a.dxF && !a.dxB && !a.dxB2 &&
      a.partial) {
    // parameter gradients only: column reduction (ln_dgb_wide_kernel)
    int rpb = cdiv(a.rows, 384);
    rpb = (rpb < 32 ? 32 : rpb + (rpb & 1));
    const int rb = cdiv(a.rows, rpb);
    if ((long long)rb * 2 * a.D <= a.partial_floats) {
      if (a.gB) hipLaunchKernelGGL(ln_dgb_wide_kernel<true>, dim3(cdiv(a.D, 512), rb), dim3(256), 0, s, a, rpb);
      else hipLaunchKernelGGL(ln_dgb_wide_kernel<false>, dim3(cdiv(a.D, 512), rb), dim3(256), 0, s, a, rpb);
      hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(cdiv(2 * a.D, 64), rb >= 128 ? 8 : 1), dim3(256), 0, s, a, rb);
      UVTG_CHECK_LAUNCH();
      return 0;
    }
  }
  LN_DISPATCH(run_bwd, a)
}

// measurement hooks (uvtg_profile_start / stop, families 6 / 7): event pairs around every LayerNorm launch; the "flops" slot carries the
// algorithmic bytes of the row streams (input + every output + the fp32 position rows the +pos outputs add)
void uvtg_prof_begin_launch(int family, double flops, hipStream_t s);
void uvtg_prof_end_launch(int family, hipStream_t s);

int launch_ln_fwd(const LnFwdArgs& a, hipStream_t s) {
  const double rd = (double)a.rows * a.D;
  double bytes = rd * (a.x ? 4 : 2);
  if (a.yF) bytes += rd * 4;
  if (a.yS) bytes += (double)a.rows * (a.Dpad > a.D ? a.Dpad : a.D) * 4;
  if (a.yUS) bytes += rd * 4;
  if (a.yB) bytes += (double)a.rows * (a.Dpad > a.D ? a.Dpad : a.D) * 2;
  if (a.yU) bytes += rd * 2;
  if (a.yUF) bytes += rd * 4;
  if (a.pos && (a.yU || a.yUF) && a.S > 0) bytes += rd * 4 * a.Lv / a.S;      // fp32 position rows of the clip rows (packed streams: >= this share)
  uvtg_prof_begin_launch(6, bytes, s);
  const int rc = launch_ln_fwd_impl(a, s);
  uvtg_prof_end_launch(6, s);
  return rc;
}
int launch_ln_bwd(const LnBwdArgs& a, hipStream_t s) {
  const double rd = (double)a.rows * a.D;
  double bytes = rd * (a.x ? 4 : 2) + ((a.g || a.gB) ? rd * (a.g ? 4 : 2) : 0.0) + ((a.g2 || a.g2B) ? rd * (a.g2 ? 4 : 2) : 0.0);
  if (a.dxF) bytes += rd * 4;
  if (a.dxB) bytes += rd * 2;
  if (a.dxB2) bytes += rd * 2;
  uvtg_prof_begin_launch(7, bytes, s);
  const int rc = launch_ln_bwd_impl(a, s);
  uvtg_prof_end_launch(7, s);
  return rc;
}
