// LayerNorm forward/backward for the UniVTG hot path (HBM-bound; one 64-lane wave per row, the row
// lives in registers, vectorised global accesses).  Forward fuses everything that consumes the
// normalised row: dropout (input projections, model/univtg.py:392-404), the bf16 / zero-padded GEMM
// operand, the (x + pos) copy that feeds the Q/K projection
// (model/transformer_encoder_droppath.py:116) and the zero-framed copy of the video rows that the
// Conv1d heads read (model/univtg.py:127-130).
#include "uvtg_kernels.h"

namespace {

template <int VEC> struct VecT;
template <> struct VecT<4> { typedef f32x4 T; };
template <> struct VecT<2> { typedef f32x2 T; };
template <> struct VecT<1> { typedef float T; };

template <int VEC> __device__ __forceinline__ void loadv(const float* p, float (&v)[VEC]) {
  if constexpr (VEC == 4) { f32x4 t = *(const f32x4*)p; v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
  else if constexpr (VEC == 2) { f32x2 t = *(const f32x2*)p; v[0] = t[0]; v[1] = t[1]; }
  else v[0] = *p;
}
template <int VEC> __device__ __forceinline__ void storev(float* p, const float (&v)[VEC]) {
  if constexpr (VEC == 4) { f32x4 t = {v[0], v[1], v[2], v[3]}; *(f32x4*)p = t; }
  else if constexpr (VEC == 2) { f32x2 t = {v[0], v[1]}; *(f32x2*)p = t; }
  else *p = v[0];
}
template <int VEC> __device__ __forceinline__ void loadb(const bf16_t* p, float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    const u32x2 t = *(const u32x2*)p;
    v[0] = __uint_as_float(t[0] << 16); v[1] = __uint_as_float(t[0] & 0xffff0000u);
    v[2] = __uint_as_float(t[1] << 16); v[3] = __uint_as_float(t[1] & 0xffff0000u);
  } else if constexpr (VEC == 2) {
    const unsigned t = *(const unsigned*)p;
    v[0] = __uint_as_float(t << 16); v[1] = __uint_as_float(t & 0xffff0000u);
  } else v[0] = bf2f(*p);
}
template <int VEC> __device__ __forceinline__ void storeb(bf16_t* p, const float (&v)[VEC]) {
  if constexpr (VEC == 4) { u32x2 t; t[0] = pack_bf2(v[0], v[1]); t[1] = pack_bf2(v[2], v[3]); *(u32x2*)p = t; }
  else if constexpr (VEC == 2) { *(unsigned*)p = pack_bf2(v[0], v[1]); }
  else *p = f2bf(v[0]);
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

template <int VEC, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnFwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const int D = a.D, D4 = (D + 3) >> 2;
  for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < a.rows; row += gridDim.x * wpb) {
    const float* xr = a.x ? a.x + (size_t)row * a.ldx : nullptr;
    const bf16_t* xbr = a.x ? nullptr : a.xB + (size_t)row * a.ldxB;
    float v[NV][VEC];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int c = (i * 64 + lane) * VEC;
      if (c < D) { if (xr) loadv<VEC>(xr + c, v[i]); else loadb<VEC>(xbr + c, v[i]); }
      else {
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
    if (a.S > 0) { b = row / a.S; s = row - b * a.S; is_vid = s < a.Lv; }
    const float* posr = (a.pos && is_vid) ? a.pos + (size_t)(b * a.Lv + s) * D : nullptr;
    const size_t prow = is_vid ? (size_t)(b * (a.Lv + 2) + s + 1) : 0;
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int c = (i * 64 + lane) * VEC;
      if (c < D) {
        float y[VEC], gm[VEC], bt[VEC];
        loadv<VEC>(a.gamma + c, gm);
        loadv<VEC>(a.beta + c, bt);
#pragma unroll
        for (int e = 0; e < VEC; e++) y[e] = (v[i][e] - mean) * rstd * gm[e] + bt[e];
        if (a.p_drop > 0.f) {
#pragma unroll
          for (int e = 0; e < VEC; e++) y[e] *= drop_scale(a.seed, a.stream_id, row, c + e, D4, a.p_drop);
        }
        if (a.yF) storev<VEC>(a.yF + (size_t)row * a.ldyF + c, y);
        if (a.yF2) storev<VEC>(a.yF2 + (size_t)row * a.ldyF2 + c, y);
        if (a.yB) storeb<VEC>(a.yB + (size_t)row * a.ldyB + c, y);
        if (a.yU || a.yUF) {
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
        }
        if (is_vid) {
          if (a.yP) storeb<VEC>(a.yP + prow * a.ldyP + c, y);
          if (a.yPF) storev<VEC>(a.yPF + prow * a.ldyP + c, y);
        }
      }
    }
    // zero padding columns [D, Dpad) of the GEMM operands
    if (a.Dpad > D) {
      for (int c = D + lane; c < a.Dpad; c += 64) {
        if (a.yB) a.yB[(size_t)row * a.ldyB + c] = 0;
        if (a.yF2) a.yF2[(size_t)row * a.ldyF2 + c] = 0.f;
      }
    }
  }
}

// dx = rstd * (gh - mean(gh) - xhat * mean(gh * xhat)),  gh = g * gamma;  dgamma += g * xhat; dbeta += g
template <int VEC, int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnBwdArgs a) {
  extern __shared__ float red[];       // [2][D] block-level dgamma / dbeta
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const int D = a.D, D4 = (D + 3) >> 2;
  float dgam[NV][VEC], dbet[NV][VEC];
#pragma unroll
  for (int i = 0; i < NV; i++)
#pragma unroll
    for (int e = 0; e < VEC; e++) { dgam[i][e] = 0.f; dbet[i][e] = 0.f; }
  for (int c = threadIdx.x; c < 2 * D; c += blockDim.x) red[c] = 0.f;
  __syncthreads();

  for (int row = blockIdx.x * wpb + wave; row < a.rows; row += gridDim.x * wpb) {
    const float* xr = a.x ? a.x + (size_t)row * a.ldx : nullptr;
    const bf16_t* xbr = a.x ? nullptr : a.xB + (size_t)row * a.ldxB;
    const float* gr = a.g ? a.g + (size_t)row * a.ldg : nullptr;
    const bf16_t* gbr = (!a.g && a.gB) ? a.gB + (size_t)row * a.ldgB : nullptr;
    const float* g2r = nullptr;
    const bf16_t* g2br = nullptr;
    if (a.g2 || a.g2B) {
      size_t r2 = (size_t)row;
      bool on = true;
      if (a.g2_S > 0) {
        const int b = row / a.g2_S, s = row - b * a.g2_S;
        on = s < a.g2_Lv;
        r2 = (size_t)(b * a.g2_Lv + s);
      }
      if (on) { if (a.g2) g2r = a.g2 + r2 * a.ldg2; else g2br = a.g2B + r2 * a.ldg2B; }
    }
    const float mean = a.mean[row], rstd = a.rstd[row];
    float xh[NV][VEC], gh[NV][VEC];
    unsigned pm[NV];                    // bit e set: x > 0 (ReLU mask of the producing layer)
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int c = (i * 64 + lane) * VEC;
      if (c < D) {
        float xv[VEC], gv[VEC], gm[VEC];
        if (xr) loadv<VEC>(xr + c, xv); else loadb<VEC>(xbr + c, xv);
        if (gr) loadv<VEC>(gr + c, gv);
        else if (gbr) loadb<VEC>(gbr + c, gv);
        else {
#pragma unroll
          for (int e = 0; e < VEC; e++) gv[e] = 0.f;
        }
        loadv<VEC>(a.gamma + c, gm);
        pm[i] = 0;
#pragma unroll
        for (int e = 0; e < VEC; e++) pm[i] |= (xv[e] > 0.f ? 1u : 0u) << e;
        if (g2r || g2br) {
          float t[VEC];
          if (g2r) loadv<VEC>(g2r + c, t); else loadb<VEC>(g2br + c, t);
#pragma unroll
          for (int e = 0; e < VEC; e++) gv[e] += t[e];
        }
        if (a.p_drop > 0.f) {
#pragma unroll
          for (int e = 0; e < VEC; e++) gv[e] *= drop_scale(a.seed, a.stream_id, row, c + e, D4, a.p_drop);
        }
#pragma unroll
        for (int e = 0; e < VEC; e++) {
          xh[i][e] = (xv[e] - mean) * rstd;
          dgam[i][e] += gv[e] * xh[i][e];
          dbet[i][e] += gv[e];
          gh[i][e] = gv[e] * gm[e];
          s1 += gh[i][e];
          s2 += gh[i][e] * xh[i][e];
        }
      } else {
        pm[i] = 0;
#pragma unroll
        for (int e = 0; e < VEC; e++) { xh[i][e] = 0.f; gh[i][e] = 0.f; }
      }
    }
    const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
    const float rs = a.rowscale ? a.rowscale[row / a.rs_seg] : 1.0f;
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int c = (i * 64 + lane) * VEC;
      if (c < D) {
        float dx[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e++) dx[e] = rstd * (gh[i][e] - c1 - xh[i][e] * c2);
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
  if (a.dgamma) {
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int c = (i * 64 + lane) * VEC;
      if (c < D) {
#pragma unroll
        for (int e = 0; e < VEC; e++) { atomicAdd(&red[c + e], dgam[i][e]); atomicAdd(&red[D + c + e], dbet[i][e]); }
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
      atomicAdd(a.dgamma + c, red[c]);
      atomicAdd(a.dbeta + c, red[D + c]);
    }
  }
}

template <int VEC, int NV> int run_fwd(const LnFwdArgs& a, hipStream_t s) {
  const int blocks = min(cdiv(a.rows, 4), 4096);
  hipLaunchKernelGGL((ln_fwd_kernel<VEC, NV>), dim3(blocks), dim3(256), 0, s, a);
  UVTG_CHECK_LAUNCH();
  return 0;
}
template <int VEC, int NV> int run_bwd(const LnBwdArgs& a, hipStream_t s) {
  const int blocks = min(cdiv(a.rows, 4), 512);
  hipLaunchKernelGGL((ln_bwd_kernel<VEC, NV>), dim3(blocks), dim3(256), 2 * a.D * sizeof(float), s, a);
  UVTG_CHECK_LAUNCH();
  return 0;
}

}  // namespace

#define LN_DISPATCH(FN, ARGS)                                                          \
  const int D = ARGS.D;                                                                 \
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

int launch_ln_fwd(const LnFwdArgs& a, hipStream_t s) {
  if (a.rows <= 0) return 0;
  const bool align16 = al(a.x, a.ldx, 4, 16) && al(a.xB, a.ldxB, 2, 8) && al(a.yF, a.ldyF, 4, 16) && al(a.yF2, a.ldyF2, 4, 16) &&
                       al(a.yB, a.ldyB, 2, 8) && al(a.yU, a.ldyU, 2, 8) && al(a.yUF, a.ldyU, 4, 16) &&
                       al(a.yP, a.ldyP, 2, 8) && al(a.yPF, a.ldyP, 4, 16) && al(a.gamma, 0, 4, 16) && al(a.beta, 0, 4, 16) &&
                       al(a.pos, a.D, 4, 16);
  const bool align8 = al(a.x, a.ldx, 4, 8) && al(a.xB, a.ldxB, 2, 4) && al(a.yF, a.ldyF, 4, 8) && al(a.yF2, a.ldyF2, 4, 8) &&
                      al(a.yB, a.ldyB, 2, 4) && al(a.yU, a.ldyU, 2, 4) && al(a.yUF, a.ldyU, 4, 8) &&
                      al(a.yP, a.ldyP, 2, 4) && al(a.yPF, a.ldyP, 4, 8) && al(a.gamma, 0, 4, 8) && al(a.beta, 0, 4, 8) &&
                      al(a.pos, a.D, 4, 8);
  LN_DISPATCH(run_fwd, a)
}

int launch_ln_bwd(const LnBwdArgs& a, hipStream_t s) {
  if (a.rows <= 0) return 0;
  const bool align16 = al(a.x, a.ldx, 4, 16) && al(a.g, a.ldg, 4, 16) && al(a.g2, a.ldg2, 4, 16) && al(a.xB, a.ldxB, 2, 8) &&
                       al(a.gB, a.ldgB, 2, 8) && al(a.g2B, a.ldg2B, 2, 8) && al(a.dxB2, a.lddxB2, 2, 8) &&
                       al(a.dxF, a.lddxF, 4, 16) && al(a.dxB, a.lddxB, 2, 8) && al(a.gamma, 0, 4, 16);
  const bool align8 = al(a.x, a.ldx, 4, 8) && al(a.g, a.ldg, 4, 8) && al(a.g2, a.ldg2, 4, 8) && al(a.xB, a.ldxB, 2, 4) &&
                      al(a.gB, a.ldgB, 2, 4) && al(a.g2B, a.ldg2B, 2, 4) && al(a.dxB2, a.lddxB2, 2, 4) &&
                      al(a.dxF, a.lddxF, 4, 8) && al(a.dxB, a.lddxB, 2, 4) && al(a.gamma, 0, 4, 8);
  LN_DISPATCH(run_bwd, a)
}
