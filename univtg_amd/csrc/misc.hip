// Small HBM-bound kernels of the UniVTG hot path: sine position table + key mask
// (model/position_encoding.py:60-83, model/univtg.py:119-124), DropPath factors
// (model/transformer_encoder_droppath.py:154-167), weight re-layouts, the last Conv1d layer of both heads
// with sigmoid / sign (model/univtg.py:129-136,375-382), weighted text pooling + cosine saliency
// (model/univtg.py:36-49,143-147) and their backward passes.
#include "uvtg_kernels.h"
#include <cstdlib>

namespace {

// ---------------- position table + key validity ----------------
// skip (optional, [B*Lv]): rows with skip[row] < 0 have no packed row and nobody reads their table entry (pk.vin_of)
// dps (optional): the n_dp = 2 E B DropPath factors of the step, drawn by the first blocks of this launch (was its own 5 us launch)
// Round 6: one WAVE per (b, t) row, four rows per block (was: a 256-thread block per row whose first wave walked the sample's mask in a rolled loop, the
// others waiting at a barrier -- a load -> sum -> barrier -> sincos -> store chain per block, 19200 blocks in 9 rounds: 36 us at config 2, 105 us at
// L_v = 1200).  The mask walk is the wave's own (unrolled loads, DPP-free shuffles, no barrier); sums of 0 / 1 are exact in any order.
__global__ __launch_bounds__(256) void seq_prep_kernel(const float* vid_mask, const float* txt_mask, int B, int Lv, int Lt, int d,
                                const float* dim_t, float* pos, unsigned char* kvalid, const int* skip,
                                float* dps, int n_dp, float p_path, unsigned long long seed, unsigned* zero_words, int n_zero) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;  // (b, t) over B*Lv
  if (blockIdx.x == 0 && zero_words)      // (the split-K tickets of the forward's small GEMMs: zero whatever an aborted call left behind)
    for (int i = threadIdx.x; i < n_zero; i += blockDim.x) zero_words[i] = 0u;
  if (dps) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_dp; i += gridDim.x * blockDim.x) {
      unsigned r[4];
      philox4(seed, (unsigned long long)i, UVTG_RNG_PATH, r);
      const float keep = 1.0f - p_path;
      dps[i] = floorf(keep + u01(r[0])) / keep;     // drop_path(): mask = floor(keep + U[0,1)), x / keep * mask
    }
  }
  if (row >= B * Lv) return;
  const int b = row / Lv, t = row % Lv;
  if (t == 0) {
    const int S = Lv + Lt;
    for (int s = lane; s < S; s += 64)
      kvalid[b * S + s] = (s < Lv ? vid_mask[b * Lv + s] : txt_mask[b * Lt + (s - Lv)]) != 0.f;
  }
  if (skip && skip[row] < 0) return;
  float c = 0.f, tot = 0.f;
#pragma unroll 4
  for (int i = lane; i < Lv; i += 64) {
    const float m = vid_mask[b * Lv + i];
    tot += m;
    if (i <= t) c += m;
  }
  c = wave_sum(c); tot = wave_sum(tot);
  // x_embed / (x_embed[:, -1:] + eps) * scale, all in fp32 (position_encoding.py:70-73)
  const float e = c / (tot + 1e-6f) * 6.283185307179586f;
  if ((d & 1) == 0) {
    // columns 2j (sin) and 2j + 1 (cos) share their denominator (dim_t[2j] == dim_t[2j + 1], position_encoding.py:75-78): one angle,
    // one sincosf, one 8-byte store per pair
    for (int cc = lane * 2; cc < d; cc += 128) {
      const float a0 = e / dim_t[cc], a1 = e / dim_t[cc + 1];
      float sv, cv;
      if (a0 == a1) sincosf(a0, &sv, &cv); else { sv = sinf(a0); cv = cosf(a1); }
      *(f32x2*)(pos + (size_t)row * d + cc) = (f32x2){sv, cv};
    }
  } else {
    for (int cc = lane; cc < d; cc += 64) {
      const float ang = e / dim_t[cc];
      pos[(size_t)row * d + cc] = (cc & 1) ? cosf(ang) : sinf(ang);
    }
  }
}

__global__ void droppath_kernel(float* scales, int n, int B, float p, unsigned long long seed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * B) return;
  unsigned r[4];
  philox4(seed, (unsigned long long)i, UVTG_RNG_PATH, r);
  const float keep = 1.0f - p;
  scales[i] = floorf(keep + u01(r[0])) / keep;     // drop_path(): mask = floor(keep + U[0,1)), x / keep * mask
}

__global__ __launch_bounds__(256) void zero_ranges_kernel(float* base, const ZeroRanges r, float* extra, int n_extra, int nb_extra, char* frame, int fLv, int frow_bytes) {
  if ((int)blockIdx.x >= r.count + nb_extra) {      // last 2 B blocks: the zero rows b (Lv + 2) and b (Lv + 2) + Lv + 1 of a zero-framed buffer (was its own launch)
    const int which = blockIdx.x - r.count - nb_extra, b = which >> 1;
    u32x4* row = (u32x4*)(frame + (size_t)(b * (fLv + 2) + ((which & 1) ? fLv + 1 : 0)) * frow_bytes);
    const u32x4 z = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < frow_bytes / 16; i += 256) row[i] = z;
    return;
  }
  if ((int)blockIdx.x >= r.count) {       // further blocks: a second buffer (clipping-norm slots + tickets + attention deltas of uvtg_backward; was a memset)
    const int nb = nb_extra, j = blockIdx.x - r.count;
    for (long long i = (long long)j * 256 + threadIdx.x; i < n_extra; i += (long long)nb * 256) extra[i] = 0.f;
    return;
  }
  float* p = base + r.off[blockIdx.x];
  for (int i = threadIdx.x; i < r.n[blockIdx.x]; i += 256) p[i] = 0.f;
}
__global__ __launch_bounds__(256) void sqsum_ranges_kernel(const float* base, const ZeroRanges r, float* out) {
  __shared__ float red[4];
  float acc = 0.f;
  if ((int)blockIdx.x == r.count) {          // last block: the 64 slots the reduce passes added into (128 bytes apart)
    if (threadIdx.x < 64) acc = out[32 + threadIdx.x * 32];
  } else {
    const float* p = base + r.off[blockIdx.x];
    for (int i = threadIdx.x; i < r.n[blockIdx.x]; i += 256) acc += p[i] * p[i];
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));
}
__global__ __launch_bounds__(256) void sqsum_kernel(const float* x, long long n, float* out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) acc += x[i] * x[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));
}
__global__ void cast_bf16_kernel(const float* src, bf16_t* dst, long long n) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    const f32x4 v = *(const f32x4*)(src + i);
    u32x2 t; t[0] = pack_bf2(v[0], v[1]); t[1] = pack_bf2(v[2], v[3]);
    *(u32x2*)(dst + i) = t;
  } else {
    for (long long j = i; j < n; j++) dst[j] = f2bf(src[j]);
  }
}
__global__ void cast_f32_kernel(const bf16_t* src, float* dst, long long n) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    const u32x2 t = *(const u32x2*)(src + i);
    const f32x4 v = {__uint_as_float(t[0] << 16), __uint_as_float(t[0] & 0xffff0000u), __uint_as_float(t[1] << 16), __uint_as_float(t[1] & 0xffff0000u)};
    *(f32x4*)(dst + i) = v;
  } else {
    for (long long j = i; j < n; j++) dst[j] = bf2f(src[j]);
  }
}
template <typename T> __device__ __forceinline__ T cvt(float v);
template <> __device__ __forceinline__ float cvt<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t cvt<bf16_t>(float v) { return f2bf(v); }

template <typename T>
__global__ void cast_pad_kernel(const float* src, int rows, int cols, T* dst, int ld) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)rows * ld) return;
  const int r = (int)(i / ld), c = (int)(i % ld);
  dst[i] = cvt<T>(c < cols ? src[(size_t)r * cols + c] : 0.f);
}
// two zero-padded casts in one launch (blockIdx.y picks): the bf16 first-projection operands of both modalities
__global__ void cast_pad2_bf16_kernel(const float* src0, int rows0, int cols0, bf16_t* dst0, int ld0,
                                      const float* src1, int rows1, int cols1, bf16_t* dst1, int ld1) {
  const bool second = blockIdx.y != 0;
  const float* src = second ? src1 : src0; bf16_t* dst = second ? dst1 : dst0;
  const int rows = second ? rows1 : rows0, cols = second ? cols1 : cols0, ld = second ? ld1 : ld0;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)rows * ld) return;
  const int r = (int)(i / ld), c = (int)(i % ld);
  dst[i] = f2bf(c < cols ? src[(size_t)r * cols + c] : 0.f);
}
// dst[c][r] = src[r][c]  (bf16), 32x32 LDS tile
__global__ void transpose_bf16_kernel(const float* src, int rows, int cols, bf16_t* dst, int ld) {
  __shared__ float t[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    t[i][threadIdx.x] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (c < cols && r < rows) dst[(size_t)c * ld + r] = f2bf(t[threadIdx.x][i]);
  }
}
// conv weight (N, C, 3) -> forward operand [n][tap*C + c]
__global__ void conv_w_fwd_kernel(const float* w, int N, int C, bf16_t* dstB, float* dstF, int ld) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)N * 3 * C) return;
  const int n = (int)(i / (3 * C)), rem = (int)(i % (3 * C)), tap = rem / C, c = rem % C;
  const float v = w[((size_t)n * C + c) * 3 + tap];
  if (dstB) dstB[(size_t)n * ld + rem] = f2bf(v);
  if (dstF) dstF[(size_t)n * ld + rem] = v;
}
// dgrad operand [c][tap' * Ntot + n_off + n] = w[n][c][2 - tap']
__global__ void conv_w_bwd_kernel(const float* w, int N, int C, bf16_t* dst, int ld, int Ntot, int n_off) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)N * 3 * C) return;
  const int c = (int)(i / (3 * N)), rem = (int)(i % (3 * N)), tp = rem / N, n = rem % N;
  dst[(size_t)c * ld + tp * Ntot + n_off + n] = f2bf(w[((size_t)n * C + c) * 3 + (2 - tp)]);
}

// ---- batched operand-cache rebuild: blockIdx.y (z for transposes) selects the op ----
__global__ __launch_bounds__(256) void cast_bf16_multi_kernel(const CastOps ops) {
  const int op = blockIdx.y;
  const long long n = ops.n[op];
  const float* src = ops.src[op];
  bf16_t* dst = ops.dst[op];
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += (long long)gridDim.x * 2048) {
    if (i + 7 < n) {
      const f32x4 a = *(const f32x4*)(src + i), b = *(const f32x4*)(src + i + 4);
      u32x4 t; t[0] = pack_bf2(a[0], a[1]); t[1] = pack_bf2(a[2], a[3]); t[2] = pack_bf2(b[0], b[1]); t[3] = pack_bf2(b[2], b[3]);
      *(u32x4*)(dst + i) = t;
    } else {
      for (long long j = i; j < n; j++) dst[j] = f2bf(src[j]);
    }
  }
}
// 64 x 64 tiles: 16-byte fp32 reads, 8-byte bf16 writes on both outputs (4 consecutive elements of a row of the plain copy, 4
// consecutive source rows of a row of the transposed copy); shapes whose rows are not 16-byte aligned take the element path
__global__ __launch_bounds__(256) void transpose_bf16_multi_kernel(const TransposeOps ops) {
  __shared__ float t[64][65];
  const int op = blockIdx.z, tid = threadIdx.x;
  const int rows = ops.rows[op], cols = ops.cols[op], ld = ops.ld[op], colsP = ops.cols_pad[op] > cols ? ops.cols_pad[op] : cols;
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  if (c0 >= colsP || r0 >= rows) return;
  const float* src = ops.src[op];
  bf16_t* dst = ops.dst[op];
  bf16_t* plain = ops.plain[op];
  const bool vec = (cols % 4 == 0) && (ld % 4 == 0) && ((((uintptr_t)src) & 15) == 0) && ((((uintptr_t)dst) & 7) == 0) &&
                   (!plain || (((uintptr_t)plain) & 7) == 0);
  if (vec) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int r = r0 + (tid >> 4) + 16 * k, c = c0 + (tid & 15) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < rows && c < cols) {
        v = *(const f32x4*)(src + (size_t)r * cols + c);
        if (plain) { u32x2 o; o[0] = pack_bf2(v[0], v[1]); o[1] = pack_bf2(v[2], v[3]); *(u32x2*)(plain + (size_t)r * cols + c) = o; }
      }
      const int rl = (tid >> 4) + 16 * k, cl = (tid & 15) * 4;
      t[rl][cl] = v[0]; t[rl][cl + 1] = v[1]; t[rl][cl + 2] = v[2]; t[rl][cl + 3] = v[3];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int cl = (tid >> 4) + 16 * k, rl = (tid & 15) * 4, c = c0 + cl, r = r0 + rl;
      if (c >= colsP || r >= rows) continue;                 // (columns [cols, colsP) hold the zeros the guarded reads left in the tile)
      if (r + 3 < rows) {
        u32x2 o; o[0] = pack_bf2(t[rl][cl], t[rl + 1][cl]); o[1] = pack_bf2(t[rl + 2][cl], t[rl + 3][cl]);
        *(u32x2*)(dst + (size_t)c * ld + r) = o;
      } else {
        for (int e = 0; e < 4 && r + e < rows; e++) dst[(size_t)c * ld + r + e] = f2bf(t[rl + e][cl]);
      }
    }
  } else {
    for (int i = tid; i < 64 * 64; i += 256) {
      const int rl = i >> 6, cl = i & 63, r = r0 + rl, c = c0 + cl;
      float v = 0.f;
      if (r < rows && c < cols) { v = src[(size_t)r * cols + c]; if (plain) plain[(size_t)r * cols + c] = f2bf(v); }
      t[rl][cl] = v;
    }
    __syncthreads();
    for (int i = tid; i < 64 * 64; i += 256) {
      const int cl = i >> 6, rl = i & 63, r = r0 + rl, c = c0 + cl;
      if (c < colsP && r < rows) dst[(size_t)c * ld + r] = f2bf(t[rl][cl]);
    }
  }
}
__global__ __launch_bounds__(256) void conv_w_multi_kernel(const ConvWOps ops, int tiled_bwd) {
  const int op = blockIdx.y, N = ops.N, C = ops.C;
  if (tiled_bwd && ops.kind[op] != 0) return;          // dgrad operands are written by conv_w_bwd_tiled_kernel
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)N * 3 * C) return;
  const float* w = ops.w[op];
  bf16_t* dst = ops.dst[op];
  if (ops.kind[op] == 0) {
    // one thread per (n, c): its three taps are 12 contiguous bytes; the three stores are each contiguous across the wave
    if (i >= (long long)N * C) return;
    const int n = (int)(i / C), c = (int)(i % C);
    const float* src = w + (size_t)i * 3;
    bf16_t* d = dst + (size_t)n * ops.ld[op] + c;
    const float t0 = src[0], t1 = src[1], t2 = src[2];
    d[0] = f2bf(t0); d[C] = f2bf(t1); d[2 * C] = f2bf(t2);
  } else if ((N % 64) || (C % 32)) {
    const int c = (int)(i / (3 * N)), rem = (int)(i % (3 * N)), tp = rem / N, n = rem % N;
    dst[(size_t)c * ops.ld[op] + tp * ops.ntot[op] + ops.n_off[op] + n] = f2bf(w[((size_t)n * C + c) * 3 + (2 - tp)]);
  }
}
// dgrad operand (kind 1) as a tiled transpose: a block moves a 64 (n) x 32 (c) x 3 (tap) tile -- 384-byte contiguous reads per
// output channel, 128-byte contiguous bf16 writes per (c, tap) row (the element-wise mapping above reads at a 12 KB stride)
__global__ __launch_bounds__(256) void conv_w_bwd_tiled_kernel(const ConvWOps ops) {
  __shared__ float tile[3][32][65];
  const int op = blockIdx.y, N = ops.N, C = ops.C;
  if (ops.kind[op] != 1) return;
  const int tiles_c = C / 32;
  const int n0 = (blockIdx.x / tiles_c) * 64, c0 = (blockIdx.x % tiles_c) * 32;
  const float* w = ops.w[op];
  bf16_t* dst = ops.dst[op];
#pragma unroll
  for (int k = 0; k < 24; k++) {
    const int j = threadIdx.x + 256 * k, nl = j / 96, q = j % 96;
    tile[q % 3][q / 3][nl] = w[((size_t)(n0 + nl) * C + c0) * 3 + q];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 24; k++) {
    const int j = threadIdx.x + 256 * k, nl = j % 64, cl = (j / 64) % 32, tp = j / 2048;
    dst[(size_t)(c0 + cl) * ops.ld[op] + tp * ops.ntot[op] + ops.n_off[op] + n0 + nl] = f2bf(tile[2 - tp][cl][nl]);
  }
}

template <typename T> __device__ __forceinline__ float ldf(const T* p);
// ---------------- wire format: ragged rows -> zero-padded batch + mask ----------------
// (pad_sequences_1d, utils/tensor_utils.py:5-53, on device: only the valid rows cross PCIe)
template <typename T>
__global__ __launch_bounds__(256) void ragged_to_padded_kernel(const T* packed, const int* offsets, int B, int Lmax, int D, float* out, float* mask) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);     // (b, t) over B * Lmax
  if (row >= (long long)B * Lmax) return;
  const int b = (int)(row / Lmax), t = (int)(row % Lmax);
  const int len = offsets[b + 1] - offsets[b];
  float* o = out + (size_t)row * D;
  if (mask && lane == 0) mask[row] = t < len ? 1.f : 0.f;
  if (t < len) {
    const T* src = packed + (size_t)(offsets[b] + t) * D;
    for (int c = lane; c < D; c += 64) o[c] = ldf<T>(src + c);
  } else {
    for (int c = lane; c < D; c += 64) o[c] = 0.f;
  }
}

// ---------------- packed (ragged) encoder stream ----------------
// Every padded clip of a sample enters the encoder with the same value (LN(0) projected + type embedding, and the sine PE
// is constant beyond the last valid clip), is masked as a key, and only ever acts as a query.  Its encoder output is
// therefore identical for all padded clips of the sample, and -- backward being linear in the upstream gradient -- the sum of
// their gradients equals the gradient of ONE such row fed with the summed upstream gradient.  Padded text tokens influence
// nothing (masked keys, outputs never read).  So the encoder runs on: the valid clips, one representative padded clip (when
// the sample is shorter than the batch), the valid text tokens -- exactly the reference's results with ~25 % fewer rows on
// ragged batches.  Row order inside a sample: valid clips, representative, valid text.
// per-sample valid lengths out of the 0/1 prefix masks (utils/tensor_utils.py:49-52): lens[b] clips, lens[B + b] text tokens
__global__ __launch_bounds__(64) void lens_from_masks_kernel(const float* vmask, const float* tmask, int B, int Lv, int Lt, int* lens) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float a = 0.f, c = 0.f;
  for (int i = lane; i < Lv; i += 64) a += vmask[(size_t)b * Lv + i];
  for (int i = lane; i < Lt; i += 64) c += tmask[(size_t)b * Lt + i];
  a = wave_sum(a); c = wave_sum(c);
  if (lane == 0) { lens[b] = (int)(a + 0.5f); lens[B + b] = (int)(c + 0.5f); }
}
// keep_pad >= 0: the valid clips and the first keep_pad padded clips of the sample stay rows of the stream, each with its own dropout
// realisation (keep_pad >= Lv: all of them); further padded clips are dropped altogether (training-only: see engine.hip PACK_HALO);
// padded text tokens are always dropped.  keep_pad < 0: one representative row stands for all padded clips.
__global__ __launch_bounds__(128) void pack_tables_kernel(const int* lens, int B, int Lv, int Lt, int keep_pad, PackTables t) {
  const int b = blockIdx.x, S = Lv + Lt;
  __shared__ int s_start;
  auto kept = [&](int lvr) { return keep_pad < 0 ? lvr : min(Lv, lvr + keep_pad); };
  __shared__ int s_vstart, s_f, s_red[2][3];
  {   // exclusive prefix sums over the samples before b (packed rows, compact clip rows, frame rows): every thread sums a strided
      // share, two wave sums -- a single thread walking up to B - 1 samples made this kernel 25 us of pure latency
    int st = 0, vs = 0, f = 0;
    for (int i = threadIdx.x; i < b; i += blockDim.x) {
      const int lvi = lens[i], lt = lens[B + i], nvi = kept(lvi) + ((keep_pad < 0 && lvi < Lv) ? 1 : 0);
      st += nvi + lt; vs += nvi; f += kept(lvi) + 2;
    }
    for (int o = 32; o > 0; o >>= 1) { st += __shfl_xor(st, o, 64); vs += __shfl_xor(vs, o, 64); f += __shfl_xor(f, o, 64); }
    if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6][0] = st; s_red[threadIdx.x >> 6][1] = vs; s_red[threadIdx.x >> 6][2] = f; }
    __syncthreads();
    if (threadIdx.x == 0) { s_start = s_red[0][0] + s_red[1][0]; s_vstart = s_red[0][1] + s_red[1][1]; s_f = s_red[0][2] + s_red[1][2]; }
  }
  __syncthreads();
  const int lvr = lens[b], lt = lens[B + b];           // lvr: real number of valid clips
  const int lv = kept(lvr);                            // clip rows kept one-to-one
  const int rep = (keep_pad < 0 && lv < Lv) ? 1 : 0, n = lv + rep + lt, st = s_start;
  if (threadIdx.x == 0) { t.seq_start[b] = st; t.seq_count[b] = n; }
  if (keep_pad >= 0) {                                 // conv-head frames (ragged on the loss-only stream): kept clips + 2 zero rows per sample
    if (threadIdx.x == 0) { t.fstart[b] = s_f; t.kept[b] = lv; }
    for (int i = threadIdx.x; i < lv + 2; i += blockDim.x) t.frame_valid[s_f + i] = (i >= 1 && i <= lv) ? 1.f : 0.f;
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int r = st + i;
    int ps;                                            // position in the padded layout
    if (i < lv) ps = i; else if (i < lv + rep) ps = lv; else ps = Lv + (i - lv - rep);
    t.row_sample[r] = b;
    t.row_src[r] = b * S + ps;
    t.row_pos[r] = ps < Lv ? b * Lv + ps : -1;
    t.kvalid[r] = (ps < Lv && ps >= lvr) ? 0 : 1;      // padded clip positions (the representative included) are never keys
  }
  // compact clip rows of the input projection: the sample's lv + rep leading clips, in order
  for (int i = threadIdx.x; i < Lv; i += blockDim.x) {
    const bool in = i < lv + rep;
    t.vin_of[b * Lv + i] = in ? s_vstart + i : -1;
    if (in) { t.vin_src[s_vstart + i] = b * Lv + i; t.vin_dst[s_vstart + i] = st + i; t.vin_x0[s_vstart + i] = b * S + i; t.vin_sample[s_vstart + i] = b; }
  }
  if (threadIdx.x == 0) t.vin_cnt[b] = lv + rep;
  for (int i = threadIdx.x; i < Lt; i += blockDim.x) t.tin_dst[b * Lt + i] = i < lt ? st + lv + rep + i : -1;
  for (int s = threadIdx.x; s < S; s += blockDim.x) {
    int pk, gm;
    if (s < lv) { pk = st + s; gm = pk; }
    else if (s < Lv) { pk = rep ? st + lv : -1; gm = (rep && s == lv) ? pk : -1; }   // padded clip -> representative (its gradient on the first one), or dropped
    else if (s - Lv < lt) { pk = st + lv + rep + (s - Lv); gm = pk; }
    else { pk = -1; gm = -1; }
    t.pad2pack[b * S + s] = pk; t.grad_map[b * S + s] = gm;
  }
}
// x0 rows of the clips the compact input projection drops (a suffix of each sample's clips; the saliency kernels read every row of
// x0): without dropout every padded clip projects to the same vector, so they take the representative's row -- bit-identical to
// the padded execution; on the loss-only stream (dropout: the padded clips differ, nothing reads them) zeros
__global__ __launch_bounds__(256) void fill_dropped_rows_kernel(float* x0, const int* vin_of, const int* vin_cnt, int copy_rep, int B, int S, int Lv, int d) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);      // (b, t) over B * Lv
  if (row >= B * Lv || vin_of[row] >= 0) return;
  const int b = row / Lv;
  float* o = x0 + ((size_t)b * S + row % Lv) * d;
  const float* rep = x0 + ((size_t)b * S + max(vin_cnt[b] - 1, 0)) * d;
  for (int c = lane * 4; c < d; c += 256) *(f32x4*)(o + c) = copy_rep ? *(const f32x4*)(rep + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
}
// encoder output (packed) -> zero-framed conv input: every clip position of the padded layout, padded clips from the representative
__global__ __launch_bounds__(256) void unpack_vm_kernel(const bf16_t* packed, const int* pad2pack, const int* fstart, const int* kept, int B, int S, int Lv,
                                                        int d, bf16_t* vm_pad) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);      // (b, s) over B * Lv
  if (row >= B * Lv) return;
  const int b = row / Lv, s = row % Lv;
  const u32x4 z = {0, 0, 0, 0};                 // dropped padded clips (beyond the conv halo): zero rows
  {   // the two zero rows that frame the sample (the k = 3 convolution's padding): written here instead of by a zero_frame launch
    const int fs0 = fstart ? fstart[b] : b * (Lv + 2), nk = kept ? kept[b] : Lv;
    if (s == 0) for (int c = lane * 8; c < d; c += 512) *(u32x4*)(vm_pad + (size_t)fs0 * d + c) = z;
    if (s == max(nk, 1) - 1) for (int c = lane * 8; c < d; c += 512)      /* (kept == 0 cannot reach here -- packed_rows() refuses len_v < 1 -- but the bottom row is written even then) */ *(u32x4*)(vm_pad + (size_t)(fs0 + nk + 1) * d + c) = z;
  }
  if (kept && s >= kept[b]) return;                         // ragged frames: a dropped clip has no frame row
  const int pk = pad2pack[b * S + s];
  const size_t src = (size_t)(pk < 0 ? 0 : pk) * d, dst = (size_t)((fstart ? fstart[b] : b * (Lv + 2)) + s + 1) * d;
  for (int c = lane * 8; c < d; c += 512) *(u32x4*)(vm_pad + dst + c) = pk < 0 ? z : *(const u32x4*)(packed + src + c);
}
// conv-head gradient wrt the clip rows (padded layout) -> packed rows: valid clips copy, the representative gets the SUM over the
// sample's padded clips, text rows zero
__global__ __launch_bounds__(256) void pack_reduce_dvm_kernel(const bf16_t* dvm, PackTables t, int B, int S, int Lv, int Mp, int d, int keep_pad, int ragged,
                                                              bf16_t* out) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= Mp) return;
  const int b = t.row_sample[r], src = t.row_src[r], ps = src - b * S;
  bf16_t* o = out + (size_t)r * d;
  const bool rep = !keep_pad && t.kvalid[r] == 0;
  for (int c = lane * 8; c < d; c += 512) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (ps < Lv) {
      const int s1 = rep ? Lv : ps + 1;
      for (int sv = ps; sv < s1; sv++) {
        const u32x4 v = *(const u32x4*)(dvm + (size_t)(ragged ? t.fstart[b] + sv + 1 : b * Lv + sv) * d + c);
#pragma unroll
        for (int e = 0; e < 4; e++) { acc[2 * e] += __uint_as_float(v[e] << 16); acc[2 * e + 1] += __uint_as_float(v[e] & 0xffff0000u); }
      }
    }
    u32x4 w; w[0] = pack_bf2(acc[0], acc[1]); w[1] = pack_bf2(acc[2], acc[3]); w[2] = pack_bf2(acc[4], acc[5]); w[3] = pack_bf2(acc[6], acc[7]);
    *(u32x4*)(o + c) = w;
  }
}

// ---------------- heads: last conv layer + activations ----------------
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return bf2f(*p); }

// 8 consecutive channels of a hidden row as floats (16-byte load in bf16 mode, 2 x 16 bytes in fp32 mode)
template <typename T> __device__ __forceinline__ void ld8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void ld8<float>(const float* p, float (&v)[8]) {
  const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
template <> __device__ __forceinline__ void ld8<bf16_t>(const bf16_t* p, float (&v)[8]) {
  const u32x4 t = *(const u32x4*)p;
#pragma unroll
  for (int e = 0; e < 4; e++) { v[2 * e] = __uint_as_float(t[e] << 16); v[2 * e + 1] = __uint_as_float(t[e] & 0xffff0000u); }
}
// the 24 weights w[c..c+7][tap 0..2] of one output channel (contiguous: (d, 3) layout)
__device__ __forceinline__ void ldw24(const float* w, float (&o)[24]) {
#pragma unroll
  for (int q = 0; q < 6; q++) { const f32x4 t = *(const f32x4*)(w + 4 * q); o[4 * q] = t[0]; o[4 * q + 1] = t[1]; o[4 * q + 2] = t[2]; o[4 * q + 3] = t[3]; }
}

// one wave per clip (b, t); lane owns 8 consecutive channels per pass (d % 8 == 0)
template <typename T>
__global__ __launch_bounds__(256) void heads_final_fwd_kernel(const HeadsFinalArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);      // (b, t)
  if (row >= a.B * a.Lv) return;
  const int b = row / a.Lv, t = row % a.Lv, d = a.d;
  const int fs = a.fstart ? a.fstart[b] : b * (a.Lv + 2);
  if (a.kept && t >= a.kept[b]) {       // no frame row (loss-only stream): constants nobody reads
    if (lane == 0) { a.pred_spans[(size_t)row * 2] = -0.5f; a.pred_spans[(size_t)row * 2 + 1] = 0.5f; a.pred_logits[row] = 0.5f; }
    return;
  }
  const T* h2 = (const T*)a.h2;
  float z0 = 0.f, z1 = 0.f, zc = 0.f;
  for (int c = lane * 8; c < d; c += 512) {
    float w0[24], w1[24], wc[24];
    ldw24(a.w_span + (size_t)c * 3, w0); ldw24(a.w_span + ((size_t)d + c) * 3, w1); ldw24(a.w_cls + (size_t)c * 3, wc);
#pragma unroll
    for (int tap = 0; tap < 3; tap++) {
      const T* hr = h2 + (size_t)(fs + t + tap) * a.ldh;
      float hs[8], hc[8];
      ld8<T>(hr + c, hs); ld8<T>(hr + d + c, hc);
#pragma unroll
      for (int e = 0; e < 8; e++) { z0 += hs[e] * w0[e * 3 + tap]; z1 += hs[e] * w1[e * 3 + tap]; zc += hc[e] * wc[e * 3 + tap]; }
    }
  }
  z0 = wave_sum(z0); z1 = wave_sum(z1); zc = wave_sum(zc);
  if (lane == 0) {
    a.pred_spans[(size_t)row * 2 + 0] = -1.0f / (1.0f + expf(-(z0 + a.b_span[0])));
    a.pred_spans[(size_t)row * 2 + 1] = 1.0f / (1.0f + expf(-(z1 + a.b_span[1])));
    a.pred_logits[row] = 1.0f / (1.0f + expf(-(zc + a.b_cls[0])));
  }
}

// pre-activation gradients of the three head outputs for clip (b, t); zero outside [0, Lv)
__device__ __forceinline__ void head_dz(const HeadsFinalArgs& a, int b, int t, float& d0, float& d1, float& dc) {
  if (t < 0 || t >= a.Lv) { d0 = d1 = dc = 0.f; return; }
  const size_t r = (size_t)b * a.Lv + t;
  const float s0 = -a.pred_spans[r * 2], s1 = a.pred_spans[r * 2 + 1], p = a.pred_logits[r];
  d0 = a.g_spans ? -a.g_spans[r * 2] * s0 * (1.f - s0) : 0.f;
  d1 = a.g_spans ? a.g_spans[r * 2 + 1] * s1 * (1.f - s1) : 0.f;
  dc = a.g_logits ? a.g_logits[r] * p * (1.f - p) : 0.f;
}

// dh2[b, u, :] (zero-framed, relu' applied) : dh[u][c] = sum_tap sum_j w[j][c][tap] * dz_j[u - tap + 1]
// one wave per (b, u) row, lane owns 8 channels of the 2d-wide row per pass
// NP > 0: 2 d == 512 NP -- the row's NP passes are unrolled with every hidden-row load issued before the first store (the output may
// alias nothing, but the compiler cannot know: with the rolled loop each pass waited for the previous pass's store)
template <int NP>
__global__ __launch_bounds__(256) void heads_final_bwd_dh_kernel(const HeadsFinalArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);      // (b, u)
  if (row >= a.B * a.Lv) return;
  const int b = row / a.Lv, u = row % a.Lv, d = a.d;
  const int fs = a.fstart ? a.fstart[b] : b * (a.Lv + 2);
  {   // the two zero rows framing the sample's gradient rows (was a zero_frame launch)
    const int nk = a.kept ? a.kept[b] : a.Lv;
    const u32x4 z = {0, 0, 0, 0};
    if (u == 0) for (int c = lane * 8; c < 2 * d; c += 512) *(u32x4*)(a.dh2 + (size_t)fs * a.lddh + c) = z;
    if (u == max(nk, 1) - 1) for (int c = lane * 8; c < 2 * d; c += 512) *(u32x4*)(a.dh2 + (size_t)(fs + nk + 1) * a.lddh + c) = z;
  }
  if (a.kept && u >= a.kept[b]) return;
  const bf16_t* h2 = (const bf16_t*)a.h2 + (size_t)(fs + u + 1) * a.ldh;
  bf16_t* out = a.dh2 + (size_t)(fs + u + 1) * a.lddh;
  constexpr int NPP = NP > 0 ? NP : 1;
  u32x4 hraw[NPP];
  if constexpr (NP > 0) {
#pragma unroll
    for (int k = 0; k < NP; k++) hraw[k] = *(const u32x4*)(h2 + lane * 8 + 512 * k);
  }
  float dz0[3], dz1[3], dzc[3];
#pragma unroll
  for (int tap = 0; tap < 3; tap++) head_dz(a, b, u - tap + 1, dz0[tap], dz1[tap], dzc[tap]);
  auto pass = [&](int c2, const float (&h)[8]) {
    const bool cls = c2 >= d;
    const int c = cls ? c2 - d : c2;
    float g[8];
    if (!cls) {
      float w0[24], w1[24];
      ldw24(a.w_span + (size_t)c * 3, w0); ldw24(a.w_span + ((size_t)d + c) * 3, w1);
#pragma unroll
      for (int e = 0; e < 8; e++) {
        float t = 0.f;
#pragma unroll
        for (int tap = 0; tap < 3; tap++) t += w0[e * 3 + tap] * dz0[tap] + w1[e * 3 + tap] * dz1[tap];
        g[e] = h[e] > 0.f ? t : 0.f;
      }
    } else {
      float wc[24];
      ldw24(a.w_cls + (size_t)c * 3, wc);
#pragma unroll
      for (int e = 0; e < 8; e++) {
        float t = 0.f;
#pragma unroll
        for (int tap = 0; tap < 3; tap++) t += wc[e * 3 + tap] * dzc[tap];
        g[e] = h[e] > 0.f ? t : 0.f;
      }
    }
    u32x4 o; o[0] = pack_bf2(g[0], g[1]); o[1] = pack_bf2(g[2], g[3]); o[2] = pack_bf2(g[4], g[5]); o[3] = pack_bf2(g[6], g[7]);
    *(u32x4*)(out + c2) = o;
  };
  if constexpr (NP > 0) {
#pragma unroll
    for (int k = 0; k < NP; k++) {
      float h[8];
#pragma unroll
      for (int e = 0; e < 4; e++) { h[2 * e] = __uint_as_float(hraw[k][e] << 16); h[2 * e + 1] = __uint_as_float(hraw[k][e] & 0xffff0000u); }
      pass(lane * 8 + 512 * k, h);
    }
  } else {
    for (int c2 = lane * 8; c2 < 2 * d; c2 += 512) {
      float h[8];
      ld8<bf16_t>(h2 + c2, h);
      pass(c2, h);
    }
  }
}

// dW[j][c][tap] = sum_{b,t} dz_j[b,t] * h2[b, t + tap - 1][c] = sum over hidden rows u of h2[b, u][c] * dz_j[b, u - tap + 1].
// One 1024-thread block per sample: 4 row groups x 256 threads, a thread owns 8 channels of the 2d-wide hidden row
// (16-byte loads; every hidden row is loaded once and feeds all three taps), the 4 groups' partial sums are folded
// through LDS and leave as one set of atomics per sample.
// gridDim.y > 1 (round 6): the sample's hidden rows are cut into gridDim.y ranges, one block each, every block with its own partial slot -- with one
// block per sample a batch of 32 samples x 1200 clips walked 300 rows per thread on 32 of the chip's 256 CUs (177 us at config 4).
__global__ __launch_bounds__(1024) void heads_final_bwd_dw_kernel(const HeadsFinalArgs a) {
  extern __shared__ float sm[];                 // [3][Lv + 2] dz (zero at both ends) | [2][256][49] partial sums
  const int b = blockIdx.x, d = a.d, tid = threadIdx.x, Lv = a.Lv, grp = tid >> 8, t8 = tid & 255;
  const int nsplit = gridDim.y, ys = blockIdx.y;
  float* s_dz = sm;
  float* s_part = sm + 3 * (Lv + 2);
  const bf16_t* h2 = (const bf16_t*)a.h2;
  float bs0 = 0.f, bs1 = 0.f, bsc = 0.f;
  for (int i = tid; i < Lv + 2; i += 1024) {
    float d0, d1, dc;
    head_dz(a, b, i - 1, d0, d1, dc);
    s_dz[i] = d0; s_dz[(Lv + 2) + i] = d1; s_dz[2 * (Lv + 2) + i] = dc;
    bs0 += d0; bs1 += d1; bsc += dc;
  }
  bs0 = wave_sum(bs0); bs1 = wave_sum(bs1); bsc = wave_sum(bsc);
  if (ys == 0 && (tid & 63) == 0 && (bs0 != 0.f || bs1 != 0.f || bsc != 0.f)) { atomicAdd(a.db_span, bs0); atomicAdd(a.db_span + 1, bs1); atomicAdd(a.db_cls, bsc); }
  __syncthreads();
  const int kc = a.kept ? a.kept[b] : Lv, fs = a.fstart ? a.fstart[b] : b * (Lv + 2);
  const int perb = (kc + nsplit - 1) / nsplit, ub0 = ys * perb, ub1 = min(kc, ub0 + perb);       // this block's rows
  const int per = (max(ub1 - ub0, 0) + 3) / 4, u0 = ub0 + grp * per, u1 = min(ub1, u0 + per);
  for (int c2 = t8 * 8; c2 < 2 * d + 2047; c2 += 2048) {        // block-uniform trip count; inactive threads only join the barriers
    const bool live = c2 < 2 * d;
    const bool cls = c2 >= d;
    const int c = cls ? c2 - d : c2;
    float acc0[24], acc1[24];
#pragma unroll
    for (int i = 0; i < 24; i++) { acc0[i] = 0.f; acc1[i] = 0.f; }
    if (live) {
      const float* z0 = s_dz + (cls ? 2 * (Lv + 2) : 0);
      const float* z1 = s_dz + (Lv + 2);
#pragma unroll 4
      for (int u = u0; u < u1; u++) {
        float h[8];
        ld8<bf16_t>(h2 + (size_t)(fs + u + 1) * a.ldh + c2, h);
#pragma unroll
        for (int tap = 0; tap < 3; tap++) {
          const float g0 = z0[u - tap + 2], g1 = z1[u - tap + 2];      // clip t = u - tap + 1 -> index 1 + t
#pragma unroll
          for (int e = 0; e < 8; e++) { acc0[e * 3 + tap] += g0 * h[e]; acc1[e * 3 + tap] += g1 * h[e]; }
        }
      }
    }
    // fold groups 2,3 into 0,1, then 1 into 0
    if (grp >= 2) {
#pragma unroll
      for (int i = 0; i < 24; i++) { s_part[((grp - 2) * 256 + t8) * 49 + i] = acc0[i]; s_part[((grp - 2) * 256 + t8) * 49 + 24 + i] = acc1[i]; }
    }
    __syncthreads();
    if (grp < 2) {
#pragma unroll
      for (int i = 0; i < 24; i++) { acc0[i] += s_part[(grp * 256 + t8) * 49 + i]; acc1[i] += s_part[(grp * 256 + t8) * 49 + 24 + i]; }
    }
    __syncthreads();
    if (grp == 1) {
#pragma unroll
      for (int i = 0; i < 24; i++) { s_part[t8 * 49 + i] = acc0[i]; s_part[t8 * 49 + 24 + i] = acc1[i]; }
    }
    __syncthreads();
    if (grp == 0 && live) {
#pragma unroll
      for (int i = 0; i < 24; i++) { acc0[i] += s_part[t8 * 49 + i]; acc1[i] += s_part[t8 * 49 + 24 + i]; }
      if (a.scratch) {        // per-sample partial in the final (j, c, tap) order; heads_final_dw_reduce_kernel sums over samples
        float* part = a.scratch + ((size_t)b * nsplit + ys) * 9 * d;
        float* p0 = part + (cls ? (size_t)6 * d : 0) + (size_t)c * 3;
#pragma unroll
        for (int q = 0; q < 6; q++) *(f32x4*)(p0 + 4 * q) = (f32x4){acc0[4 * q], acc0[4 * q + 1], acc0[4 * q + 2], acc0[4 * q + 3]};
        if (!cls) {
          float* p1 = part + ((size_t)d + c) * 3;
#pragma unroll
          for (int q = 0; q < 6; q++) *(f32x4*)(p1 + 4 * q) = (f32x4){acc1[4 * q], acc1[4 * q + 1], acc1[4 * q + 2], acc1[4 * q + 3]};
        }
      } else if (!cls) {      // (256 same-address atomics per element serialise in L2: only the fallback without scratch)
#pragma unroll
        for (int i = 0; i < 24; i++) { atomicAdd(a.dw_span + (size_t)c * 3 + i, acc0[i]); atomicAdd(a.dw_span + ((size_t)d + c) * 3 + i, acc1[i]); }
      } else {
#pragma unroll
        for (int i = 0; i < 24; i++) atomicAdd(a.dw_cls + (size_t)c * 3 + i, acc0[i]);
      }
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void heads_final_dw_reduce_kernel(const HeadsFinalArgs a, const int nslots) {
  __shared__ float red[16][17];
  const int il = threadIdx.x & 15, bl = threadIdx.x >> 4;        // 16 outputs x 16 slot lanes per block
  const int idx = blockIdx.x * 16 + il, n = 9 * a.d;
  float s = 0.f;
  if (idx < n) for (int b = bl; b < nslots; b += 16) s += a.scratch[(size_t)b * n + idx];
  red[bl][il] = s;
  __syncthreads();
  if (bl == 0 && idx < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) t += red[k][il];
    if (idx < 6 * a.d) a.dw_span[idx] += t; else a.dw_cls[idx - 6 * a.d] += t;
  }
}

// ---------------- weighted pooling + cosine saliency ----------------
// forward, pass 1: one 1024-thread block per sample -- pooling logits (one wave per text row), masked softmax,
// pooled text vector (one thread per column) and its norm
__global__ __launch_bounds__(1024) void saliency_pool_kernel(const SaliencyArgs a) {
  extern __shared__ float sm[];                 // [Lt] logits/alpha | [16] scratch
  float* s_alpha = sm;
  float* s_red = sm + a.Lt;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, d = a.d;
  const float* xt = a.x0 + ((size_t)b * a.S + a.Lv) * d;      // text rows
  for (int t = wave; t < a.Lt; t += 16) {
    float acc = 0.f;
    for (int c = lane; c < d; c += 64) acc += xt[(size_t)t * d + c] * a.w_pool[c];
    acc = wave_sum(acc);
    if (lane == 0) s_alpha[t] = acc + (1.0f - a.txt_mask[b * a.Lt + t]) * (-1e30f);
  }
  __syncthreads();
  if (wave == 0) {
    float mx = -INFINITY;
    for (int t = lane; t < a.Lt; t += 64) mx = fmaxf(mx, s_alpha[t]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int t = lane; t < a.Lt; t += 64) sum += expf(s_alpha[t] - mx);
    sum = wave_sum(sum);
    for (int t = lane; t < a.Lt; t += 64) {
      const float al = expf(s_alpha[t] - mx) / sum;
      s_alpha[t] = al;
      if (a.alpha) a.alpha[b * a.Lt + t] = al;
    }
  }
  __syncthreads();
  float nsq = 0.f;
  for (int c = tid; c < d; c += 1024) {
    float acc = 0.f;
#pragma unroll 8
    for (int t = 0; t < a.Lt; t++) acc += s_alpha[t] * xt[(size_t)t * d + c];
    a.pooled[(size_t)b * d + c] = acc;
    nsq += acc * acc;
  }
  nsq = wave_sum(nsq);
  if (lane == 0) s_red[wave] = nsq;
  __syncthreads();
  if (tid == 0 && a.qnorm) {
    float t = 0.f;
    for (int i = 0; i < 16; i++) t += s_red[i];
    a.qnorm[b] = sqrtf(t);
  }
}
// forward, pass 2: one wave per video row -- |v|, cos(v, pooled), saliency = cos + log-mask
__global__ __launch_bounds__(256) void saliency_cos_kernel(const SaliencyArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.B * a.Lv) return;
  const int b = row / a.Lv, t = row % a.Lv, d = a.d;
  const float* v = a.x0 + ((size_t)b * a.S + t) * d;
  const float* q = a.pooled + (size_t)b * d;
  float dot = 0.f, vs = 0.f;
  for (int c = lane * 4; c < d; c += 256) {
    const f32x4 x = *(const f32x4*)(v + c), y = *(const f32x4*)(q + c);
    dot += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
    vs += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
  }
  dot = wave_sum(dot); vs = wave_sum(vs);
  if (lane == 0) {
    const float vn = sqrtf(vs);
    const float cs = dot / (fmaxf(vn, 1e-8f) * fmaxf(a.qnorm[b], 1e-8f));
    if (a.vnorm) a.vnorm[row] = vn;
    if (a.cosv) a.cosv[row] = cs;
    a.sal[row] = cs + (a.vid_mask[row] != 0.f ? 0.f : UVTG_LOG_TINY);
  }
}

// ---------------- the fused per-clip head pass (north_star: "(saliency, fg-prob, span offsets) in one pass") ----------------
// ONE launch for everything behind the second conv layer: one 512-thread block per sample pools its text tokens (weighted softmax pooling,
// model/univtg.py:36-49) and then walks its clips: the last Conv1d(k = 3) tap sums of both heads + sigmoid / sign (model/univtg.py:129-136)
// AND cosine(vid_mem_proj, txt_mem_proj) + log-mask (model/univtg.py:143-147) -- pred_logits, pred_spans and saliency_scores of a clip leave
// together.  Replaces heads_final_fwd + saliency_pool + saliency_cos (three launches, 88 us at config 2).
// Each wave owns a CONTIGUOUS run of clips and every lane the same 8 channels per pass for all of them, so that (1) the lane's 72 tap weights
// per pass are fetched ONCE into registers (the per-clip kernel re-reads 288 B per lane and clip: 700 MB of L1/L2 traffic at config 2 against
// 66 MB of hidden rows) and (2) the three hidden rows of a clip slide: one new row per clip.  (A first version -- 16 waves, weights re-read
// per clip -- took 93 us: no faster than the three launches.)  Used when the batch gives every CU a block (B >= 128) in bf16 mode; else the
// per-clip grids keep the chip full (L_v = 1200, B = 32).
template <int NPASS>     // d = 512 * NPASS
__global__ __launch_bounds__(512) void heads_saliency_fwd_kernel(const HeadsFinalArgs h, const SaliencyArgs a) {
  extern __shared__ float sm[];                 // [Lt] logits / alpha | [d] pooled | [16] scratch
  float* s_alpha = sm;
  float* s_pool = sm + ((a.Lt + 3) & ~3);      // 16-byte aligned for the f32x4 reads below, whatever the (odd) text length
  float* s_red = s_pool + a.d;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, d = a.d;
  const float* xt = a.x0 + ((size_t)b * a.S + a.Lv) * d;      // text rows
  {   // pooling logits: the lane's 2 NPASS pieces of w_pool once, the pieces of a text row issued together (same sums in the same order as the rolled loop)
    f32x4 wp[2 * NPASS];
#pragma unroll
    for (int i = 0; i < 2 * NPASS; i++) wp[i] = *(const f32x4*)(a.w_pool + lane * 4 + 256 * i);
    for (int t = wave; t < a.Lt; t += 8) {
      f32x4 x[2 * NPASS];
#pragma unroll
      for (int i = 0; i < 2 * NPASS; i++) x[i] = *(const f32x4*)(xt + (size_t)t * d + lane * 4 + 256 * i);
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < 2 * NPASS; i++) acc += x[i][0] * wp[i][0] + x[i][1] * wp[i][1] + x[i][2] * wp[i][2] + x[i][3] * wp[i][3];
      acc = wave_sum(acc);
      if (lane == 0) s_alpha[t] = acc + (1.0f - a.txt_mask[b * a.Lt + t]) * (-1e30f);
    }
  }
  __syncthreads();
  if (wave == 0) {
    float mx = -INFINITY;
    for (int t = lane; t < a.Lt; t += 64) mx = fmaxf(mx, s_alpha[t]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int t = lane; t < a.Lt; t += 64) sum += expf(s_alpha[t] - mx);
    sum = wave_sum(sum);
    for (int t = lane; t < a.Lt; t += 64) {
      const float al = expf(s_alpha[t] - mx) / sum;
      s_alpha[t] = al;
      if (a.alpha) a.alpha[b * a.Lt + t] = al;
    }
  }
  __syncthreads();
  float nsq = 0.f;
  for (int c = tid; c < d; c += 512) {
    float acc = 0.f;
#pragma unroll 16
    for (int t = 0; t < a.Lt; t++) acc += s_alpha[t] * xt[(size_t)t * d + c];
    a.pooled[(size_t)b * d + c] = acc;
    s_pool[c] = acc;
    nsq += acc * acc;
  }
  nsq = wave_sum(nsq);
  if (lane == 0) s_red[wave] = nsq;
  __syncthreads();
  float qn = 0.f;
  for (int i = 0; i < 8; i++) qn += s_red[i];
  qn = sqrtf(qn);
  if (tid == 0 && a.qnorm) a.qnorm[b] = qn;
  // ---- the clips of this sample: wave w owns clips [t0, t1) ----
  const int fs = h.fstart ? h.fstart[b] : b * (h.Lv + 2);
  const int kept = h.kept ? h.kept[b] : h.Lv;
  const bf16_t* h2 = (const bf16_t*)h.h2;
  // gridDim.y blocks per sample (experiment; default 1): each pools the text tokens itself (identical values; 128 KB of L2 reads) and walks its
  // share of the clips -- the walk is a per-clip latency chain, 10 clips per wave with one block per sample, 5 with two
  const int nw = 8 * gridDim.y, wslot = blockIdx.y * 8 + wave;
  const int per = (a.Lv + nw - 1) / nw, t0 = wslot * per, t1 = min(a.Lv, t0 + per);
  float w0[NPASS][24], w1[NPASS][24], wc[NPASS][24];
#pragma unroll
  for (int ps = 0; ps < NPASS; ps++) {
    const int c = lane * 8 + 512 * ps;
    ldw24(h.w_span + (size_t)c * 3, w0[ps]); ldw24(h.w_span + ((size_t)d + c) * 3, w1[ps]); ldw24(h.w_cls + (size_t)c * 3, wc[ps]);
  }
  // hidden rows fs + t - 1 + 1 .. : rows[k] = frame row fs + t + k (k = 0, 1, 2 are the clip's three taps), raw bf16 (span half | class half)
  u32x4 rs[3][NPASS], rc[3][NPASS];
  auto load_row = [&](int fr, u32x4 (&os)[NPASS], u32x4 (&oc)[NPASS]) {
#pragma unroll
    for (int ps = 0; ps < NPASS; ps++) {
      const bf16_t* hr = h2 + (size_t)fr * h.ldh + lane * 8 + 512 * ps;
      os[ps] = *(const u32x4*)hr; oc[ps] = *(const u32x4*)(hr + d);
    }
  };
  // the clip's own fp32 row for the cosine: 2 NPASS 16-byte pieces per lane, all issued together with the hidden row (round 6: with the column loop
  // left rolled every piece was its own load -> use round trip -- 6.3 us per clip, 63 of the kernel's 83 us at config 2; holding the NEXT clip's
  // pieces as well spills: the lane already keeps 144 tap weights and 48 registers of sliding hidden rows)
  // (... and lane 0's three bias words and the clip's mask word were four more serial round trips per clip behind the wave sums: the biases are read
  // once, the mask word with the clip's rows)
  auto uniform = [](float x) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(x))); };      // (scalar registers: the lane's vector file is full)
  const float bs0 = uniform(h.b_span[0]), bs1 = uniform(h.b_span[1]), bc0 = uniform(h.b_cls[0]);
  if (t0 < t1 && t0 < kept) { load_row(fs + t0, rs[0], rc[0]); load_row(fs + t0 + 1, rs[1], rc[1]); }
  for (int t = t0; t < t1; t++) {
    const int row = b * a.Lv + t;
    const bool framed = t < kept;
    if (framed) load_row(fs + t + 2, rs[2], rc[2]);
    const float vmask = a.vid_mask[row];
    f32x4 xv[2 * NPASS];
    {
      const float* v = a.x0 + ((size_t)b * a.S + t) * d + lane * 4;
#pragma unroll
      for (int i = 0; i < 2 * NPASS; i++) xv[i] = *(const f32x4*)(v + 256 * i);
    }
    // cosine saliency (lane owns 4 consecutive fp32 channels per piece)
    float dot = 0.f, vs = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * NPASS; i++) {
      const f32x4 x = xv[i], y = *(const f32x4*)(s_pool + lane * 4 + 256 * i);
      dot += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
      vs += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
    }
    float z0 = 0.f, z1 = 0.f, zc = 0.f;
    if (framed) {
#pragma unroll
      for (int ps = 0; ps < NPASS; ps++)
#pragma unroll
        for (int tap = 0; tap < 3; tap++)
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const float s_lo = __uint_as_float(rs[tap][ps][e] << 16), s_hi = __uint_as_float(rs[tap][ps][e] & 0xffff0000u);
            const float c_lo = __uint_as_float(rc[tap][ps][e] << 16), c_hi = __uint_as_float(rc[tap][ps][e] & 0xffff0000u);
            z0 += s_lo * w0[ps][(2 * e) * 3 + tap] + s_hi * w0[ps][(2 * e + 1) * 3 + tap];
            z1 += s_lo * w1[ps][(2 * e) * 3 + tap] + s_hi * w1[ps][(2 * e + 1) * 3 + tap];
            zc += c_lo * wc[ps][(2 * e) * 3 + tap] + c_hi * wc[ps][(2 * e + 1) * 3 + tap];
          }
#pragma unroll
      for (int ps = 0; ps < NPASS; ps++) { rs[0][ps] = rs[1][ps]; rc[0][ps] = rc[1][ps]; rs[1][ps] = rs[2][ps]; rc[1][ps] = rc[2][ps]; }
    }
    dot = wave_sum_dpp(dot); vs = wave_sum_dpp(vs); z0 = wave_sum_dpp(z0); z1 = wave_sum_dpp(z1); zc = wave_sum_dpp(zc);      // (DPP: no LDS crossbar round trips)
    if (lane == 0) {
      const float vn = sqrtf(vs);
      const float cs = dot / (fmaxf(vn, 1e-8f) * fmaxf(qn, 1e-8f));
      if (a.vnorm) a.vnorm[row] = vn;
      if (a.cosv) a.cosv[row] = cs;
      a.sal[row] = cs + (vmask != 0.f ? 0.f : UVTG_LOG_TINY);
      if (framed) {
        h.pred_spans[(size_t)row * 2 + 0] = -1.0f / (1.0f + expf(-(z0 + bs0)));
        h.pred_spans[(size_t)row * 2 + 1] = 1.0f / (1.0f + expf(-(z1 + bs1)));
        h.pred_logits[row] = 1.0f / (1.0f + expf(-(zc + bc0)));
      } else {
        h.pred_spans[(size_t)row * 2] = -0.5f; h.pred_spans[(size_t)row * 2 + 1] = 0.5f; h.pred_logits[row] = 0.5f;
      }
    }
  }
}

// backward.  Everything that flows into the pre-encoder tokens x0 from saliency / vid_mem_proj / txt_mem_proj is added
// to the encoder's gradient dx0 while the rows are re-packed per modality in bf16 for the input-projection backward
// (three passes: dq per sample and column; softmax gradient per sample; then one wave per token row).
//   video row t: g = dx0 + g_sal (qhat - cos vhat) / |v| + g_vid (+ g_vrow on the positive row)
//   dq           = g_pooled + sum_t g_sal (vhat - cos qhat) / |q|
//   text row s : g = dx0 + alpha dq + dlog w_pool,  dlog = alpha (dq.x_s - sum alpha dq.x);  dw_pool += sum dlog x_s
// block = 64 columns x 4 row phases (one block per (sample, 256 columns) with a serial walk over the clips was latency-bound:
// 535 us at L_v = 1200): thread (cc, ph) walks clips ph, ph + 4, ...; the four phases meet in LDS
// (round 6: blockDim.x / 64 row phases -- 4 at L_v <= 256, 16 above: at L_v = 1200 a thread of the 4-phase block walked 300 clips, 61 us at config 4)
__global__ __launch_bounds__(1024) void saliency_dq_kernel(const SaliencyArgs a) {
  extern __shared__ float sm[];                 // [Lv] gs | [Lv] |v| | [Lv] cos | [blockDim.x] partial sums
  float* s_gs = sm;
  float* s_vn = s_gs + a.Lv;
  float* s_cs = s_vn + a.Lv;
  float* red = s_cs + a.Lv;
  const int b = blockIdx.x, tid = threadIdx.x, d = a.d;
  const int cc = tid & 63, ph = tid >> 6, nph = blockDim.x >> 6;
  const int c = blockIdx.y * 64 + cc;
  const float qn = fmaxf(a.qnorm[b], 1e-8f);
  for (int t = tid; t < a.Lv; t += blockDim.x) {
    s_gs[t] = a.g_sal ? a.g_sal[b * a.Lv + t] : 0.f;
    s_vn[t] = fmaxf(a.vnorm[b * a.Lv + t], 1e-8f);
    s_cs[t] = a.cosv[b * a.Lv + t];
  }
  __syncthreads();
  const float* xv = a.x0 + (size_t)b * a.S * d;
  const float qh = c < d ? a.pooled[(size_t)b * d + c] / qn : 0.f;
  float acc = 0.f;
  if (c < d) {
#pragma unroll 4
    for (int t = ph; t < a.Lv; t += nph) {
      const float gs = s_gs[t];
      acc += gs * (xv[(size_t)t * d + c] / s_vn[t] - s_cs[t] * qh);
    }
  }
  red[tid] = acc;
  __syncthreads();
  if (ph == 0 && c < d) {
    float t4 = red[cc] + red[64 + cc] + red[128 + cc] + red[192 + cc];      // (the four-phase order of rounds 2-5)
    for (int k = 4; k < nph; k++) t4 += red[64 * k + cc];
    a.dq[(size_t)b * d + c] = (a.g_pooled ? a.g_pooled[(size_t)b * d + c] : 0.f) + t4 / qn;
  }
}
__global__ __launch_bounds__(1024) void saliency_dlog_kernel(const SaliencyArgs a) {
  extern __shared__ float sm[];                 // [Lt] da
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, d = a.d;
  const float* xt = a.x0 + ((size_t)b * a.S + a.Lv) * d;
  const float* dq = a.dq + (size_t)b * d;
  for (int t = wave; t < a.Lt; t += 16) {
    float acc = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
      const f32x4 x = *(const f32x4*)(xt + (size_t)t * d + c), y = *(const f32x4*)(dq + c);
      acc += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
    }
    acc = wave_sum(acc);
    if (lane == 0) sm[t] = acc;
  }
  __syncthreads();
  float dot = 0.f;
  for (int t = 0; t < a.Lt; t++) dot += a.alpha[b * a.Lt + t] * sm[t];
  for (int t = tid; t < a.Lt; t += 1024) a.dlog[b * a.Lt + t] = a.alpha[b * a.Lt + t] * (sm[t] - dot);
}
// one block per (sample, SAL_CHUNK-row chunk); 4 waves, each walks rows chunk*SAL_CHUNK + wave, +4, ...; lane owns columns 4*lane + 256*k
// (16-row chunks: twice the blocks of the 32-row version, all still resident at once -- the kernel is a per-row latency chain)
#ifndef UVTG_SAL_CHUNK
#define UVTG_SAL_CHUNK 16
#define UVTG_SAL_TXT_CHUNK 16     // text rows per block (0: all text rows of a sample in ONE block = one set of pooling-weight atomics per sample, but an 8-row chain per wave: the launch's tail -- 71.9 vs 65.4 us at config 2, round 5)
#endif
constexpr int SAL_CHUNK = UVTG_SAL_CHUNK, SAL_TXT = UVTG_SAL_TXT_CHUNK;
template <int KC>    // d = 256 * KC
__global__ __launch_bounds__(256) void saliency_rows_kernel(const SaliencyArgs a) {
  __shared__ float s_dw[4][256 * KC];
  const int b = blockIdx.x, chunk = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, d = a.d;
  const float qn = fmaxf(a.qnorm[b], 1e-8f);
  const int prow = (a.g_vrow && a.pos_idx) ? (int)a.pos_idx[b] : -1;
  float dw[KC][4];
#pragma unroll
  for (int k = 0; k < KC; k++)
#pragma unroll
    for (int e = 0; e < 4; e++) dw[k][e] = 0.f;
  bool any_txt = false;
  // clip chunks of SAL_CHUNK rows, then the text rows of the sample in their own chunks of SAL_TXT rows (the pooling-weight gradient leaves
  // as one set of atomics per text chunk, none from the clip chunks)
  const int nvc = (a.Lv + SAL_CHUNK - 1) / SAL_CHUNK;
  const int s_begin = chunk < nvc ? chunk * SAL_CHUNK : a.Lv + (SAL_TXT ? (chunk - nvc) * SAL_TXT : 0);
  const int s_end = chunk < nvc ? min(a.Lv, chunk * SAL_CHUNK + SAL_CHUNK) : (SAL_TXT ? min(a.S, s_begin + SAL_TXT) : a.S);
  for (int srow = s_begin + wave; srow < s_end; srow += 4) {
    const size_t row = (size_t)b * a.S + srow;
    const float* x = a.x0 + row * d;
    const long long grow = a.dx0_map ? (long long)a.dx0_map[row] : (long long)row;      // row of the (possibly packed) encoder gradient
    const float* g0 = a.dx0 ? a.dx0 + grow * d : nullptr;
    const bf16_t* g0b = a.dx0 ? nullptr : a.dx0B + grow * d;
    auto ldg = [&](int c) -> f32x4 {
      if (grow < 0) return (f32x4){0.f, 0.f, 0.f, 0.f};
      if (g0) return *(const f32x4*)(g0 + c);
      const u32x2 t = *(const u32x2*)(g0b + c);
      return (f32x4){__uint_as_float(t[0] << 16), __uint_as_float(t[0] & 0xffff0000u), __uint_as_float(t[1] << 16), __uint_as_float(t[1] & 0xffff0000u)};
    };
    if (srow < a.Lv) {
      const int t = srow;
      const float gs = a.g_sal ? a.g_sal[b * a.Lv + t] : 0.f;
      const float vn = fmaxf(a.vnorm[b * a.Lv + t], 1e-8f), cs = a.cosv[b * a.Lv + t];
      const int vo = a.vout_map ? a.vout_map[b * a.Lv + t] : b * a.Lv + t;
      if (vo < 0) continue;                       // a clip the compact input projection dropped: no gradient row to write
      bf16_t* out = a.out_vid + (size_t)vo * d;
#pragma unroll
      for (int k = 0; k < KC; k++) {
        const int c = k * 256 + lane * 4;
        f32x4 g = ldg(c);
        if (gs != 0.f) {
          const f32x4 xv = *(const f32x4*)(x + c), q = *(const f32x4*)(a.pooled + (size_t)b * d + c);
#pragma unroll
          for (int e = 0; e < 4; e++) g[e] += gs * (q[e] / qn - cs * (xv[e] / vn)) / vn;
        }
        if (a.g_vid) { const f32x4 t4 = *(const f32x4*)(a.g_vid + (size_t)b * a.gv_sb + (size_t)t * a.gv_st + c); g += t4; }
        if (t == prow) { const f32x4 t4 = *(const f32x4*)(a.g_vrow + (size_t)b * d + c); g += t4; }
        u32x2 o; o[0] = pack_bf2(g[0], g[1]); o[1] = pack_bf2(g[2], g[3]);
        *(u32x2*)(out + c) = o;
      }
    } else {
      const int t = srow - a.Lv;
      const float al = a.alpha[b * a.Lt + t], dl = a.dlog[b * a.Lt + t];
      bf16_t* out = a.out_txt + ((size_t)b * a.Lt + t) * d;
      any_txt = true;
#pragma unroll
      for (int k = 0; k < KC; k++) {
        const int c = k * 256 + lane * 4;
        f32x4 g = ldg(c);
        const f32x4 xt = *(const f32x4*)(x + c), dq = *(const f32x4*)(a.dq + (size_t)b * d + c), wp = *(const f32x4*)(a.w_pool + c);
        if (a.g_txt_rows) { const f32x4 t4 = *(const f32x4*)(a.g_txt_rows + ((size_t)b * a.Lt + t) * d + c); g += t4; }
#pragma unroll
        for (int e = 0; e < 4; e++) { g[e] += al * dq[e] + dl * wp[e]; dw[k][e] += dl * xt[e]; }
        u32x2 o; o[0] = pack_bf2(g[0], g[1]); o[1] = pack_bf2(g[2], g[3]);
        *(u32x2*)(out + c) = o;
      }
    }
  }
  if (!a.dw_pool) return;
  const bool blk_txt = s_end > a.Lv;          // block-uniform: does this chunk contain text rows
  if (!blk_txt) return;
#pragma unroll
  for (int k = 0; k < KC; k++)
#pragma unroll
    for (int e = 0; e < 4; e++) s_dw[wave][k * 256 + lane * 4 + e] = any_txt ? dw[k][e] : 0.f;
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += 256) atomicAdd(a.dw_pool + c, s_dw[0][c] + s_dw[1][c] + s_dw[2][c] + s_dw[3][c]);
}
// generic width fallback of the row pass (one wave per row, scalar columns)
__global__ __launch_bounds__(256) void saliency_rows_generic_kernel(const SaliencyArgs a) {
  const int lane = threadIdx.x & 63, d = a.d;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)a.B * a.S) return;
  const int b = (int)(row / a.S), srow = (int)(row % a.S);
  const float qn = fmaxf(a.qnorm[b], 1e-8f);
  const int prow = (a.g_vrow && a.pos_idx) ? (int)a.pos_idx[b] : -1;
  const float* x = a.x0 + (size_t)row * d;
  const long long grow = a.dx0_map ? (long long)a.dx0_map[row] : row;
  const float* g0f = a.dx0 ? a.dx0 + grow * d : nullptr;
  const bf16_t* g0b = a.dx0 ? nullptr : a.dx0B + grow * d;
  auto g0 = [&](int c) -> float { return grow < 0 ? 0.f : (g0f ? g0f[c] : bf2f(g0b[c])); };
  if (srow < a.Lv) {
    const int t = srow;
    const float gs = a.g_sal ? a.g_sal[b * a.Lv + t] : 0.f;
    const float vn = fmaxf(a.vnorm[b * a.Lv + t], 1e-8f), cs = a.cosv[b * a.Lv + t];
    const int vo = a.vout_map ? a.vout_map[b * a.Lv + t] : b * a.Lv + t;
    if (vo < 0) return;
    bf16_t* out = a.out_vid + (size_t)vo * d;
    for (int c = lane; c < d; c += 64) {
      float g = g0(c);
      if (gs != 0.f) g += gs * (a.pooled[(size_t)b * d + c] / qn - cs * (x[c] / vn)) / vn;
      if (a.g_vid) g += a.g_vid[(size_t)b * a.gv_sb + (size_t)t * a.gv_st + c];
      if (t == prow) g += a.g_vrow[(size_t)b * d + c];
      out[c] = f2bf(g);
    }
  } else {
    const int t = srow - a.Lv;
    const float al = a.alpha[b * a.Lt + t], dl = a.dlog[b * a.Lt + t];
    bf16_t* out = a.out_txt + ((size_t)b * a.Lt + t) * d;
    for (int c = lane; c < d; c += 64) {
      out[c] = f2bf(g0(c) + al * a.dq[(size_t)b * d + c] + dl * a.w_pool[c] + (a.g_txt_rows ? a.g_txt_rows[((size_t)b * a.Lt + t) * d + c] : 0.f));
      if (a.dw_pool) atomicAdd(a.dw_pool + c, dl * x[c]);
    }
  }
}

}  // namespace

int launch_ragged_to_padded(const void* packed, int src_bf16, const int* offsets, int B, int Lmax, int D, float* out, float* mask, hipStream_t s) {
  if (B <= 0 || Lmax <= 0 || D <= 0) return -11;
  const unsigned blocks = (unsigned)(((long long)B * Lmax + 3) / 4);
  if (src_bf16) hipLaunchKernelGGL(ragged_to_padded_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)packed, offsets, B, Lmax, D, out, mask);
  else hipLaunchKernelGGL(ragged_to_padded_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)packed, offsets, B, Lmax, D, out, mask);
  UVTG_CHECK_LAUNCH();
  return 0;
}
int launch_pack_tables(const float* vid_mask, const float* txt_mask, int* lens_dev, int B, int Lv, int Lt, int keep_pad, const PackTables& t,
                       hipStream_t s) {
  hipLaunchKernelGGL(lens_from_masks_kernel, dim3(B), dim3(64), 0, s, vid_mask, txt_mask, B, Lv, Lt, lens_dev);
  hipLaunchKernelGGL(pack_tables_kernel, dim3(B), dim3(128), 0, s, lens_dev, B, Lv, Lt, keep_pad, t);
  UVTG_CHECK_LAUNCH();
  return 0;
}
int launch_fill_dropped_rows(float* x0, const PackTables& t, int copy_rep, int B, int S, int Lv, int d, hipStream_t s) {
  if (d % 4) return -2;
  hipLaunchKernelGGL(fill_dropped_rows_kernel, dim3(cdiv(B * Lv, 4)), dim3(256), 0, s, x0, t.vin_of, t.vin_cnt, copy_rep, B, S, Lv, d);
  UVTG_CHECK_LAUNCH();
  return 0;
}
int launch_unpack_vm(const bf16_t* packed, const PackTables& t, bool ragged_frames, int B, int S, int Lv, int d, bf16_t* vm_pad, hipStream_t s) {
  if (d % 8) return -2;
  hipLaunchKernelGGL(unpack_vm_kernel, dim3(cdiv(B * Lv, 4)), dim3(256), 0, s, packed, t.pad2pack, ragged_frames ? t.fstart : nullptr,
                     ragged_frames ? t.kept : nullptr, B, S, Lv, d, vm_pad);
  UVTG_CHECK_LAUNCH();
  return 0;
}
int launch_pack_reduce_dvm(const bf16_t* dvm, const PackTables& t, int B, int S, int Lv, int Mp, int d, bool keep_pad, bool ragged_frames, bf16_t* out,
                           hipStream_t s) {
  if (d % 8) return -2;
  hipLaunchKernelGGL(pack_reduce_dvm_kernel, dim3(cdiv(Mp, 4)), dim3(256), 0, s, dvm, t, B, S, Lv, Mp, d, keep_pad ? 1 : 0, ragged_frames ? 1 : 0, out);
  UVTG_CHECK_LAUNCH();
  return 0;
}
int launch_seq_prep(const float* vid_mask, const float* txt_mask, int B, int Lv, int Lt, int d,
                    const float* dim_t, float* pos, unsigned char* kvalid, const int* skip, hipStream_t s,
                    float* dps, int n_dp, float p_path, unsigned long long seed, unsigned* zero_words, int n_zero) {
  hipLaunchKernelGGL(seq_prep_kernel, dim3(cdiv(B * Lv, 4)), dim3(256), 0, s, vid_mask, txt_mask, B, Lv, Lt, d, dim_t, pos, kvalid, skip, dps, n_dp, p_path, seed, zero_words, n_zero);
  UVTG_CHECK_LAUNCH();
  return 0;
}
int launch_droppath_scales(float* scales, int n, int B, float p, unsigned long long seed, hipStream_t s) {
  hipLaunchKernelGGL(droppath_kernel, dim3(cdiv(n * B, 256)), dim3(256), 0, s, scales, n, B, p, seed);
  UVTG_CHECK_LAUNCH();
  return 0;
}
int launch_zero_ranges(float* base, const ZeroRanges& r, hipStream_t s, float* extra, int n_extra, void* frame, int fB, int fLv, int frow_bytes) {
  if (r.count <= 0 && !extra && !frame) return 0;
  const int nb_extra = extra ? (n_extra > 65536 ? 256 : 1) : 0;
  hipLaunchKernelGGL(zero_ranges_kernel, dim3(r.count + nb_extra + (frame ? 2 * fB : 0)), dim3(256), 0, s, base, r, extra, n_extra, nb_extra, (char*)frame, fLv, frow_bytes);
  UVTG_CHECK_LAUNCH();
  return 0;
}
int launch_sqsum_ranges(const float* base, const ZeroRanges& r, float* sqsum, hipStream_t s) {
  if (!sqsum) return 0;
  hipLaunchKernelGGL(sqsum_ranges_kernel, dim3(r.count + 1), dim3(256), 0, s, base, r, sqsum);
  UVTG_CHECK_LAUNCH();
  return 0;
}
int launch_sqsum(const float* x, long long n, float* sqsum, hipStream_t s) {
  if (n <= 0 || !sqsum) return 0;
  const long long b = (n + 2047) / 2048;
  hipLaunchKernelGGL(sqsum_kernel, dim3((unsigned)(b < 1024 ? b : 1024)), dim3(256), 0, s, x, n, sqsum);
  UVTG_CHECK_LAUNCH();
  return 0;
}
int launch_cast_bf16(const float* src, bf16_t* dst, long long n, hipStream_t s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, s, src, dst, n);
  UVTG_CHECK_LAUNCH();
  return 0;
}
int launch_cast_f32(const bf16_t* src, float* dst, long long n, hipStream_t s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(cast_f32_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, s, src, dst, n);
  UVTG_CHECK_LAUNCH();
  return 0;
}
int launch_cast_pad_bf16(const float* src, int rows, int cols, bf16_t* dst, int ld, hipStream_t s) {
  const long long n = (long long)rows * ld;
  hipLaunchKernelGGL(cast_pad_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, rows, cols, dst, ld);
  UVTG_CHECK_LAUNCH();
  return 0;
}
int launch_cast_pad2_bf16(const float* src0, int rows0, int cols0, bf16_t* dst0, int ld0, const float* src1, int rows1, int cols1, bf16_t* dst1, int ld1,
                          hipStream_t s) {
  const long long n0 = (long long)rows0 * ld0, n1 = (long long)rows1 * ld1, n = n0 > n1 ? n0 : n1;
  hipLaunchKernelGGL(cast_pad2_bf16_kernel, dim3((unsigned)((n + 255) / 256), 2), dim3(256), 0, s, src0, rows0, cols0, dst0, ld0, src1, rows1, cols1, dst1, ld1);
  UVTG_CHECK_LAUNCH();
  return 0;
}
int launch_transpose_bf16(const float* src, int rows, int cols, bf16_t* dst, int ld, hipStream_t s) {
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(32, 8), 0, s, src, rows, cols, dst, ld);
  UVTG_CHECK_LAUNCH();
  return 0;
}
// fp16 hi / lo images of fp32 matrices in the interleaved row layout (split-operand GEMM operands), one thread per 4 real columns of a row
__global__ __launch_bounds__(256) void split_f16_multi_kernel(const SplitOps ops) {
  const int op = blockIdx.y;
  const int rows = ops.rows[op], cols = ops.cols[op], kp = ops.kp[op], conv = ops.conv[op];
  const float* src = ops.src[op];
  unsigned short* dst = ops.dst[op];
  const long long n4 = (long long)rows * (kp / 4);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int r = (int)(i / (kp / 4)), c = (int)(i % (kp / 4)) * 4;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int k = c + e;
      float x = 0.f;
      if (k < cols) x = conv ? src[((size_t)r * conv + (k % conv)) * 3 + k / conv] : src[(size_t)r * cols + k];
      v[e] = x;
    }
    u32x2 hi, lo; split4_f16(v, ops.scale, hi, lo);
    unsigned short* o = dst + (size_t)r * 2 * kp + split_col(c);
    *(u32x2*)o = hi; *(u32x2*)(o + 32) = lo;
  }
}
int launch_split_f16_multi(const SplitOps& ops, hipStream_t s) {
  if (ops.count <= 0) return 0;
  long long mx = 0;
  for (int i = 0; i < ops.count; i++) {
    if (ops.kp[i] % 32 || ops.kp[i] < ops.cols[i]) return -2;
    const long long n4 = (long long)ops.rows[i] * (ops.kp[i] / 4);
    mx = n4 > mx ? n4 : mx;
  }
  const unsigned bx = (unsigned)((mx + 255) / 256);
  hipLaunchKernelGGL(split_f16_multi_kernel, dim3(bx < 2048 ? bx : 2048, ops.count), dim3(256), 0, s, ops);
  UVTG_CHECK_LAUNCH();
  return 0;
}
int launch_cast_bf16_multi(const CastOps& ops, hipStream_t s) {
  if (ops.count <= 0) return 0;
  long long mx = 0;
  for (int i = 0; i < ops.count; i++) mx = ops.n[i] > mx ? ops.n[i] : mx;
  const unsigned bx = (unsigned)((mx + 2047) / 2048);
  hipLaunchKernelGGL(cast_bf16_multi_kernel, dim3(bx < 512 ? bx : 512, ops.count), dim3(256), 0, s, ops);
  UVTG_CHECK_LAUNCH();
  return 0;
}
int launch_transpose_bf16_multi(const TransposeOps& ops, hipStream_t s) {
  if (ops.count <= 0) return 0;
  int mr = 0, mc = 0;
  for (int i = 0; i < ops.count; i++) { mr = ops.rows[i] > mr ? ops.rows[i] : mr; mc = ops.cols[i] > mc ? ops.cols[i] : mc; mc = ops.cols_pad[i] > mc ? ops.cols_pad[i] : mc; }
  hipLaunchKernelGGL(transpose_bf16_multi_kernel, dim3(cdiv(mc, 64), cdiv(mr, 64), ops.count), dim3(256), 0, s, ops);
  UVTG_CHECK_LAUNCH();
  return 0;
}
int launch_conv_w_multi(const ConvWOps& ops, hipStream_t s) {
  if (ops.count <= 0) return 0;
  const long long n = (long long)ops.N * 3 * ops.C;
  const int tiled = (ops.N % 64 == 0 && ops.C % 32 == 0) ? 1 : 0;
  const long long nblk = tiled ? ((long long)ops.N * ops.C + 255) / 256 : (n + 255) / 256;    // forward operands: one thread per (n, c)
  hipLaunchKernelGGL(conv_w_multi_kernel, dim3((unsigned)nblk, ops.count), dim3(256), 0, s, ops, tiled);
  UVTG_CHECK_LAUNCH();
  if (tiled) {
    hipLaunchKernelGGL(conv_w_bwd_tiled_kernel, dim3((ops.N / 64) * (ops.C / 32), ops.count), dim3(256), 0, s, ops);
    UVTG_CHECK_LAUNCH();
  }
  return 0;
}
int launch_conv_w_fwd(const float* w, int N, int C, bf16_t* dstB, float* dstF, int ld, hipStream_t s) {
  const long long n = (long long)N * 3 * C;
  hipLaunchKernelGGL(conv_w_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, N, C, dstB, dstF, ld);
  UVTG_CHECK_LAUNCH();
  return 0;
}
int launch_conv_w_bwd(const float* w, int N, int C, bf16_t* dst, int ld, int Ntot, int n_off, hipStream_t s) {
  const long long n = (long long)N * 3 * C;
  hipLaunchKernelGGL(conv_w_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, N, C, dst, ld, Ntot, n_off);
  UVTG_CHECK_LAUNCH();
  return 0;
}
int launch_heads_final_fwd(const HeadsFinalArgs& a, hipStream_t s) {
  const int rows = a.B * a.Lv;
  if (a.precise) hipLaunchKernelGGL(heads_final_fwd_kernel<float>, dim3(cdiv(rows, 4)), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(heads_final_fwd_kernel<bf16_t>, dim3(cdiv(rows, 4)), dim3(256), 0, s, a);
  UVTG_CHECK_LAUNCH();
  return 0;
}
int launch_heads_final_bwd(const HeadsFinalArgs& a, hipStream_t s) {
  if (a.precise) return -6;
  if (a.d == 1024) hipLaunchKernelGGL(heads_final_bwd_dh_kernel<4>, dim3(cdiv(a.B * a.Lv, 4)), dim3(256), 0, s, a);
  else if (a.d == 512) hipLaunchKernelGGL(heads_final_bwd_dh_kernel<2>, dim3(cdiv(a.B * a.Lv, 4)), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(heads_final_bwd_dh_kernel<0>, dim3(cdiv(a.B * a.Lv, 4)), dim3(256), 0, s, a);
  UVTG_CHECK_LAUNCH();
  static bool attr = false;
  if (!attr) {
    if (hipError_t e = hipFuncSetAttribute((const void*)heads_final_bwd_dw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) return (int)e;
    attr = true;
  }
  const size_t sh = (3 * (size_t)(a.Lv + 2) + 2 * 256 * 49) * sizeof(float);
  if (sh > 160 * 1024) return -11;
  HeadsFinalArgs b = a;
  // blocks per sample: enough for one block per CU when the batch alone does not give that (B = 32, L_v = 1200: 8), never ranges under 16 rows;
  // B >= 256 keeps one block per sample (and its summation order)
  static const int split_env = uvtg_dev_env("UVTG_HEADS_DW_SPLIT") ? atoi(uvtg_dev_env("UVTG_HEADS_DW_SPLIT")) : 0;      // experiment: force
  int nsplit = split_env > 0 ? split_env : max(1, min(cdiv(256, max(b.B, 1)), cdiv(b.Lv, 16)));
  while (nsplit > 1 && b.scratch && b.scratch_floats < (long long)b.B * nsplit * 9 * b.d) nsplit--;
  if (b.scratch && b.scratch_floats < (long long)b.B * nsplit * 9 * b.d) b.scratch = nullptr;
  hipLaunchKernelGGL(heads_final_bwd_dw_kernel, dim3(b.B, nsplit), dim3(1024), sh, s, b);
  if (b.scratch) hipLaunchKernelGGL(heads_final_dw_reduce_kernel, dim3(cdiv(9 * b.d, 16)), dim3(256), 0, s, b, b.B * nsplit);
  UVTG_CHECK_LAUNCH();
  return 0;
}
int launch_saliency_fwd(const SaliencyArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(saliency_pool_kernel, dim3(a.B), dim3(1024), (a.Lt + 16) * sizeof(float), s, a);
  hipLaunchKernelGGL(saliency_cos_kernel, dim3(cdiv(a.B * a.Lv, 4)), dim3(256), 0, s, a);
  UVTG_CHECK_LAUNCH();
  return 0;
}
// heads' last layer + activations + text pooling + cosine saliency: ONE launch when every CU gets a sample, else the three per-stage launches
int launch_heads_saliency_fwd(const HeadsFinalArgs& h, const SaliencyArgs& a, hipStream_t s) {
  static const bool off = uvtg_dev_env("UVTG_HEADFUSE_OFF") != nullptr;      // experiment: always the separate launches
  const size_t sh = ((size_t)((a.Lt + 3) & ~3) + a.d + 16) * sizeof(float);
  if (off || h.precise || a.B < 128 || (a.d != 512 && a.d != 1024) || sh > 60 * 1024) {
    if (int e = launch_heads_final_fwd(h, s)) return e;
    return launch_saliency_fwd(a, s);
  }
  // blocks per sample: 1.  (Round 5 measured 2 and 4 -- each block pooling the text itself, half / a quarter of the clips per wave -- in-box:
  // 9.150 / 9.178 / 9.220 ms per step, profiles/r05_ab_head_pass_blocks.txt: the second pooling costs what the shorter clip walk saves.)
  static const int split_env = uvtg_dev_env("UVTG_HEADFUSE_SPLIT") ? atoi(uvtg_dev_env("UVTG_HEADFUSE_SPLIT")) : 1;
  const int ny = (split_env >= 1 && split_env <= 4) ? split_env : 1;
  if (a.d == 1024) hipLaunchKernelGGL(heads_saliency_fwd_kernel<2>, dim3(a.B, ny), dim3(512), sh, s, h, a);
  else hipLaunchKernelGGL(heads_saliency_fwd_kernel<1>, dim3(a.B, ny), dim3(512), sh, s, h, a);
  UVTG_CHECK_LAUNCH();
  return 0;
}
int launch_saliency_bwd(const SaliencyArgs& a, hipStream_t s) {
  const int dq_threads = a.Lv > 256 ? 1024 : 256;
  hipLaunchKernelGGL(saliency_dq_kernel, dim3(a.B, cdiv(a.d, 64)), dim3(dq_threads), (3 * a.Lv + dq_threads) * sizeof(float), s, a);
  hipLaunchKernelGGL(saliency_dlog_kernel, dim3(a.B), dim3(1024), a.Lt * sizeof(float), s, a);
  const dim3 grid(a.B, cdiv(a.Lv, SAL_CHUNK) + (SAL_TXT ? cdiv(a.Lt, SAL_TXT) : 1));
  if (a.d == 1024) hipLaunchKernelGGL(saliency_rows_kernel<4>, grid, dim3(256), 0, s, a);
  else if (a.d == 512) hipLaunchKernelGGL(saliency_rows_kernel<2>, grid, dim3(256), 0, s, a);
  else if (a.d == 256) hipLaunchKernelGGL(saliency_rows_kernel<1>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(saliency_rows_generic_kernel, dim3((unsigned)(((long long)a.B * a.S + 3) / 4)), dim3(256), 0, s, a);
  UVTG_CHECK_LAUNCH();
  return 0;
}
