// Masked multi-head self-attention for the UniVTG encoder (model/transformer_encoder_droppath.py:117-118
// -> torch F.multi_head_attention_forward): softmax(q k^T + key_padding(-inf)) v per (sample, head),
// S = L_v + L_t tokens, flash-style (no S x S matrix in HBM), on v_mfma_f32_32x32x16_bf16.
//
// Layout trick used throughout: every score tile is computed TRANSPOSED (keys along MFMA rows, queries
// along lanes, or vice versa) so that the softmax statistics are lane-local and the probability tile
// can be fed back to the next MFMA straight from registers; the operand that has to be transposed is
// fetched with ds_read_b64_tr_b16.
//   fwd : S^T = K Q^T (K: LDS b128, Q: registers)      O^T += V^T P^T (V^T: LDS tr-read, P^T: registers)
//   dKdV: S   = Q K^T, dP = dO V^T                      dV^T += dO^T P, dK^T += Q^T dS
//   dQ  : S^T = K Q^T, dP^T = V dO^T                    dQ^T += K^T dS^T
#include "uvtg_kernels.h"

namespace {

constexpr float NEG_BIG = -1e30f;

template <int HD> struct KSwz {      // swizzle for [rows][HD] bf16 tiles read with ds_read_b128
  static constexpr int CPR = HD / 8;
  static constexpr int RPB = (128 / HD) > 0 ? (128 / HD) : 1;
  __device__ static __forceinline__ int off(int r, int chunk) { return r * HD + ((chunk ^ ((r / RPB) % CPR)) * 8); }
};

// XCD-aware decode of the 1-D grids of the tiled kernels.  Workgroups are dispatched round-robin over the 8 XCDs, each with its own L2.  All
// `nblk` 128-row blocks of one (sample, head) stream the SAME K / V rows (forward, dQ) or Q / dO rows (dK / dV): with the (block, head, sample)
// grid their ids were consecutive, i.e. one block per XCD, and every XCD pulled every head's rows into its own L2 (8 x the traffic, S = 1232:
// 1.3 GB per launch).  Here XCD x takes a contiguous range of the block list, so that the blocks of a head run on ONE XCD, back to back.
__device__ __forceinline__ void attn_block_id(int nblk, int H, int B, int& blk, int& h, int& b, int sample_major) {
  const int total = gridDim.x, q = total / 8, r = total % 8, xcd = blockIdx.x % 8, idx = blockIdx.x / 8;
  const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  blk = v % nblk;
  const int bh = v / nblk;
  // head-major over the (sample, head) pairs (round 5): an XCD's contiguous share is then ONE head of (nearly) every sample rather than every head of
  // a few samples -- all heads of a sample cost the same, samples of a ragged batch do not (S_b^2: 4x between 600 and 1200 clips), and a launch
  // ends with its slowest XCD.  UVTG_ATTN_SAMPLE_MAJOR (experiment; AttnArgs::sample_major) restores b = bh / H.
  const int dv = sample_major ? H : B, lo = bh % dv, hi = bh / dv;
  h = sample_major ? lo : hi; b = sample_major ? hi : lo;
}

// Counter = position in the PADDED [B, H, S, S] probability tensor (S = a.S also on the packed stream, whose in-sample row order under
// dropout is the padded layout's): padded and packed executions draw identical masks.
__device__ __forceinline__ float keep_scale(unsigned long long seed, unsigned stream, int bh, int q, int k, int S, float p) {
  unsigned r[4];
  philox4(seed, ((unsigned long long)bh * S + q) * (unsigned long long)S + k, stream, r);
  return (u01(r[0]) >= p) ? 1.0f / (1.0f - p) : 0.0f;
}

__device__ __forceinline__ s16x8 pack8(const float* v) {
  s16x8 o;
#pragma unroll
  for (int e = 0; e < 8; e++) o[e] = (short)f2bf(v[e]);
  return o;
}
__device__ __forceinline__ s16x8 pack8_lo(const float* v) {   // residual after bf16 rounding
  s16x8 o;
#pragma unroll
  for (int e = 0; e < 8; e++) o[e] = (short)f2bf(v[e] - bf2f(f2bf(v[e])));
  return o;
}
// fp16 hi / lo images of 8 values times a power-of-two scale (the precise mode's operand split, uvtg_common.h split_f16)
__device__ __forceinline__ void pack8_split(const float* v, float scale, s16x8& hi, s16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; e++) { unsigned short h, l; split_f16(v[e] * scale, h, l); hi[e] = (short)h; lo[e] = (short)l; }
}
// operand scales of the precise attention: q, k, v x16, probabilities x1024 (p <= 1); folded back into the score scale / the final 1 / l
constexpr float ATT_QKV_S = 16.0f, ATT_P_S = 1024.0f;
__device__ __forceinline__ s16x8 cat4(s16x4 a, s16x4 b) { return (s16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }

// The backward kernels evaluate p = exp(s - lse) as exp2(s * log2(e) - lse * log2(e)): one v_fma + one v_exp per element.  A query row
// beyond the sequence gets the "lse" ROW_OFF (exp2 underflows to an exact 0), a padded / absent key the bias -ROW_OFF: no per-element
// compare, branch or LDS look-up is left in the softmax sections (round 3: the per-element `sL[ql]` / `sD[ql]` / `sValid[kl]` reads were 32
// dependent LDS round trips per query block, each behind its own s_waitcnt -- more time than the block's 32 MFMAs).
constexpr float LOG2E = 1.4426950408889634f;
constexpr float ROW_OFF = 1e30f;
__device__ __forceinline__ float exp2_raw(float x) { return __builtin_amdgcn_exp2f(x); }

// [rows][HD] bf16 tile that is read both by rows (ds_read_b128: one row per lane) and transposed (ds_read_b64_tr_b16: a 32-lane group
// covers 4 rows x 64 bytes).  Padded rows (HD + 8 elements = 4 banks of skew per row) keep the row reads conflict-free but put the four
// rows of a transposing read on overlapping banks (4-way).  SWZ (HD = 128 only: 16 chunks of 16 bytes = one 256-byte bank row per tile
// row): chunk c of row r is stored at chunk c ^ (4 (r & 3) + ((r >> 2) & 3)) -- the 16 rows of a ds_read_b128 lane group land on 16
// different chunks, and the 4 rows of a transposing read on 4 different 64-byte quarters (tools/lds_conflict_probe.hip).
template <int HD, bool SWZ> struct TileRT {
  static_assert(!SWZ || HD == 128, "the chunk swizzle assumes 16 chunks per row");
  static constexpr int STR = SWZ ? HD : HD + 8;
  __device__ static __forceinline__ int off(int r, int col) {
    if constexpr (SWZ) return r * HD + ((((col >> 3) ^ (((r & 3) << 2) | ((r >> 2) & 3)))) << 3) + (col & 7);
    else return r * STR + col;
  }
  // offset of (r, col + dcol) from base = off(r, col) when dcol is a multiple of 16 elements that only sets bits col has clear (head-dim
  // blocks / k-steps): the chunk permutation is an XOR, so the step is an XOR of the offset (one VALU op instead of one register per step)
  __device__ static __forceinline__ int step(int base, int dcol) {
    if constexpr (SWZ) return base ^ dcol; else return base + dcol;
  }
  // offset of (r + 8, col) from off(r, col) for r % 16 < 8, and of (r + 4, col) for r % 8 < 4: bits 2..3 of the row are bits 0..1 of the
  // permutation, so the row step flips one chunk bit
  __device__ static __forceinline__ int rows8(int base) { if constexpr (SWZ) return (base ^ 16) + 8 * STR; else return base + 8 * STR; }
  __device__ static __forceinline__ int rows4(int base) { if constexpr (SWZ) return (base ^ 8) + 4 * STR; else return base + 4 * STR; }
};

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// min 2 workgroups per CU: without the bound hipcc spreads over 276 VGPR+AGPR (HD=128) and ONE workgroup per CU stays resident --
// the kernel then runs at a single wave per SIMD with every global-load latency exposed (measured: SQ_WAVE_CYCLES == SQ_BUSY_CU_CYCLES)
template <int HD, bool PRECISE>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const AttnArgs a) {
  constexpr int VSTR = HD + (HD >= 64 ? 32 : 0);
  constexpr int NP = PRECISE ? 2 : 1;
  __shared__ __attribute__((aligned(16))) bf16_t sK[NP][64 * HD];
  __shared__ __attribute__((aligned(16))) bf16_t sV[NP][64 * VSTR];
  __shared__ __attribute__((aligned(16))) float sBias[64];     // 0 on a real key, NEG_BIG on padding and beyond S (added in the log2 domain)
  using KS = KSwz<HD>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, l31 = lane & 31;
  const int i16 = lane & 15, qd = (lane >> 4) & 1;
  int qblk, h, b;
  attn_block_id((a.S + 127) / 128, a.H, a.B, qblk, h, b, a.sample_major);
  const int S = a.seq_count ? a.seq_count[b] : a.S;
  if (qblk * 128 >= S) return;
  const int q_raw = qblk * 128 + wave * 32 + l31;
  const int qrow = min(q_raw, S - 1);
  const size_t rowbase = a.seq_start ? (size_t)a.seq_start[b] : (size_t)b * S;

  // Q fragments (B operand of S^T = K Q^T): lane (query, g) holds q[16ks + 8g .. +7]
  s16x8 qh[HD / 16], ql[PRECISE ? HD / 16 : 1];
#pragma unroll
  for (int ks = 0; ks < HD / 16; ks++) {
    const size_t off = (rowbase + qrow) * a.ldqkv + h * HD + 16 * ks + 8 * g;
    if constexpr (!PRECISE) qh[ks] = *(const s16x8*)((const bf16_t*)a.qkv + off);
    else {
      float v[8];
      const float* p = (const float*)a.qkv + off;
      *(f32x4*)v = *(const f32x4*)p; *(f32x4*)(v + 4) = *(const f32x4*)(p + 4);
      pack8_split(v, ATT_QKV_S, qh[ks], ql[ks]);
    }
  }
  f32x16 oacc[HD / 32];
#pragma unroll
  for (int i = 0; i < HD / 32; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) oacc[i][r] = 0.f;
  float m_run = NEG_BIG, l_run = 0.f;
  const unsigned rng_stream = UVTG_RNG_ATTN + a.layer;

  const int ntiles = (S + 63) / 64;
  // bf16 mode: the next K / V tile (and its key-validity bytes) is fetched into registers while the current one is consumed
  constexpr int CH = HD / 8, NPF = PRECISE ? 1 : (64 * CH) / 256;     // 16-byte chunks per row; pieces per thread
  u32x4 pk[NPF], pvv[NPF];
  unsigned char pval = 0;
  auto fetch = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NPF; i++) {
      const int q = tid + 256 * i, r = q / CH, c = q % CH;
      const int key = kt * 64 + r;
      pk[i] = (u32x4){0, 0, 0, 0}; pvv[i] = (u32x4){0, 0, 0, 0};
      if (key < S) {
        const bf16_t* base = (const bf16_t*)a.qkv + (rowbase + key) * a.ldqkv + h * HD + c * 8;
        pk[i] = *(const u32x4*)(base + a.H * HD);
        pvv[i] = *(const u32x4*)(base + 2 * a.H * HD);
      }
    }
    if (tid < 64) { const int key = kt * 64 + tid; pval = (key < S) ? a.kvalid[rowbase + key] : 0; }
  };
  if constexpr (!PRECISE) fetch(0);
  for (int kt = 0; kt < ntiles; kt++) {
    __syncthreads();
    // ---- stage K / V tile (keys kt*64 .. +63), zero-filled beyond S ----
    if constexpr (!PRECISE) {
#pragma unroll
      for (int i = 0; i < NPF; i++) {
        const int q = tid + 256 * i, r = q / CH, c = q % CH;
        *(u32x4*)(&sK[0][KS::off(r, c)]) = pk[i];
        *(u32x4*)(&sV[0][r * VSTR + c * 8]) = pvv[i];
      }
      if (tid < 64) sBias[tid] = pval ? 0.f : NEG_BIG;
      if (kt + 1 < ntiles) fetch(kt + 1);
    } else {
      constexpr int PC = HD / 4;                       // float4 pieces per row
#pragma unroll
      for (int i = 0; i < (64 * PC) / 256; i++) {
        const int q = tid + 256 * i, r = q / PC, c = q % PC;
        const int key = kt * 64 + r;
        float kf[4] = {0, 0, 0, 0}, vf[4] = {0, 0, 0, 0};
        if (key < S) {
          const float* base = (const float*)a.qkv + (rowbase + key) * a.ldqkv + h * HD + c * 4;
          *(f32x4*)kf = *(const f32x4*)(base + a.H * HD);
          *(f32x4*)vf = *(const f32x4*)(base + 2 * a.H * HD);
        }
        const int ko = KS::off(r, c >> 1) + (c & 1) * 4, vo = r * VSTR + c * 4;
#pragma unroll
        for (int e = 0; e < 4; e++) {       // fp16 hi / lo images (x16): the three-product split carries ~22 bits (uvtg_common.h)
          unsigned short kh_, kl_, vh_, vl_;
          split_f16(kf[e] * ATT_QKV_S, kh_, kl_); split_f16(vf[e] * ATT_QKV_S, vh_, vl_);
          sK[0][ko + e] = kh_; sK[NP - 1][ko + e] = kl_;
          sV[0][vo + e] = vh_; sV[NP - 1][vo + e] = vl_;
        }
      }
    }
    if constexpr (PRECISE) {
      if (tid < 64) { const int key = kt * 64 + tid; sBias[tid] = (key < S && a.kvalid[rowbase + key]) ? 0.f : NEG_BIG; }
    }
    __syncthreads();

    // ---- S^T = K Q^T : sc[kb][r] <-> key kb*32 + (r&3)+8(r>>2)+4g, query l31 ----
    f32x16 sc[2];
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
#pragma unroll
      for (int r = 0; r < 16; r++) sc[kb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < HD / 16; ks++) {
        const int o = KS::off(kb * 32 + l31, 2 * ks + g);
        const s16x8 kf = *(const s16x8*)(&sK[0][o]);
        if constexpr (PRECISE) {
          const s16x8 kl = *(const s16x8*)(&sK[NP - 1][o]);
          sc[kb] = mfma32h(kl, qh[ks], sc[kb]);
          sc[kb] = mfma32h(kf, ql[ks], sc[kb]);
          sc[kb] = mfma32h(kf, qh[ks], sc[kb]);
        } else sc[kb] = mfma32(kf, qh[ks], sc[kb]);
      }
    }
    // online softmax in the log2 domain: t = s log2(e) + bias (one v_fma per element; the running maximum m_run is a log2-domain value too)
    float mx = NEG_BIG;
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const f32x4 bk = *(const f32x4*)(&sBias[kb * 32 + 8 * j + 4 * g]);      // keys kb*32 + 8 j + 4 g + (0..3) = registers 4 j .. 4 j + 3
#pragma unroll
        for (int e = 0; e < 4; e++) {
          sc[kb][4 * j + e] = fmaf(sc[kb][4 * j + e], PRECISE ? LOG2E / (ATT_QKV_S * ATT_QKV_S) : LOG2E, bk[e]);
          mx = fmaxf(mx, sc[kb][4 * j + e]);
        }
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = exp2_raw(m_run - m_new);
    const float m_use = fmaxf(m_new, -1e20f);          // nothing but padding so far: t - m_use stays at -1e30, p = 0 (not exp2(0))
    float rs = 0.f;
    float pv[2][16];
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float p = exp2_raw(sc[kb][r] - m_use);
        rs += p;
        pv[kb][r] = p;
      }
    rs += __shfl_xor(rs, 32, 64);
    l_run = l_run * alpha + rs;
    m_run = m_new;
    if (a.p_drop > 0.f) {
#pragma unroll
      for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int key = kt * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
          pv[kb][r] *= keep_scale(a.seed, rng_stream, b * a.H + h, q_raw, key, a.S, a.p_drop);
        }
    }
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {     // (the running maximum settles after the first tiles: HD/2 multiplies saved per tile)
#pragma unroll
      for (int i = 0; i < HD / 32; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) oacc[i][r] *= alpha;
    }
    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int hf = 0; hf < 2; hf++) {
        s16x8 pb, pl;
        if constexpr (PRECISE) pack8_split(&pv[kb][8 * hf], ATT_P_S, pb, pl);
        else pb = pack8(&pv[kb][8 * hf]);
        const int kr = kb * 32 + 16 * hf + 4 * g + (i16 >> 2);
#pragma unroll
        for (int dvb = 0; dvb < HD / 32; dvb++) {
          const int col = dvb * 32 + 16 * qd + 4 * (i16 & 3);
          const s16x8 vf = cat4(lds_tr16(&sV[0][kr * VSTR + col]), lds_tr16(&sV[0][(kr + 8) * VSTR + col]));
          if constexpr (PRECISE) {
            const s16x8 vl = cat4(lds_tr16(&sV[NP - 1][kr * VSTR + col]), lds_tr16(&sV[NP - 1][(kr + 8) * VSTR + col]));
            oacc[dvb] = mfma32h(vl, pb, oacc[dvb]);
            oacc[dvb] = mfma32h(vf, pl, oacc[dvb]);
            oacc[dvb] = mfma32h(vf, pb, oacc[dvb]);
          } else oacc[dvb] = mfma32(vf, pb, oacc[dvb]);
        }
      }
  }
  if constexpr (!PRECISE && HD >= 64) {
    // O^T fragments hold (channel, query): a direct store is 8 B per lane at a 2 KB stride (16-byte pieces of 32 different
    // cache lines per instruction).  Transpose through a wave-private slab in the dead V stage instead, half the channels per
    // pass, and store 16 B per lane with HD/16 lanes covering one contiguous half row.
    constexpr int OSTR = HD / 2 + 8, CPR = HD / 16;
    static_assert(4 * 32 * OSTR <= 64 * VSTR, "output slab must fit the V stage");
    __syncthreads();                                   // every wave is done reading sV
    bf16_t* slab = &sV[0][0] + wave * 32 * OSTR;
    const float inv = 1.0f / l_run;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
#pragma unroll
      for (int dvh = 0; dvh < HD / 64; dvh++)
#pragma unroll
        for (int rq = 0; rq < 4; rq++) {
          const int dvb = pass * (HD / 64) + dvh;
          u32x2 t;
          t[0] = pack_bf2(oacc[dvb][4 * rq] * inv, oacc[dvb][4 * rq + 1] * inv);
          t[1] = pack_bf2(oacc[dvb][4 * rq + 2] * inv, oacc[dvb][4 * rq + 3] * inv);
          *(u32x2*)(slab + l31 * OSTR + dvh * 32 + 8 * rq + 4 * g) = t;
        }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < (32 * CPR) / 64; i++) {
        const int idx = lane + 64 * i, row = idx / CPR, ch = idx % CPR;
        const int q = qblk * 128 + wave * 32 + row;
        const u32x4 v = *(const u32x4*)(slab + row * OSTR + ch * 8);
        if (q < S) *(u32x4*)((bf16_t*)a.o + (rowbase + q) * a.ldo + h * HD + pass * (HD / 2) + ch * 8) = v;
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (q_raw < S && a.lse && g == 0) a.lse[((size_t)b * a.H + h) * a.S + q_raw] = m_run * 0.6931471805599453f + __logf(l_run);
  } else if (q_raw < S) {
    const float inv = PRECISE ? 1.0f / (l_run * (ATT_QKV_S * ATT_P_S)) : 1.0f / l_run;
#pragma unroll
    for (int dvb = 0; dvb < HD / 32; dvb++)
#pragma unroll
      for (int rq = 0; rq < 4; rq++) {
        const int dv = dvb * 32 + 8 * rq + 4 * g;
        const size_t off = (rowbase + q_raw) * a.ldo + h * HD + dv;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = oacc[dvb][4 * rq + e] * inv;
        if constexpr (!PRECISE) {
          u32x2 t; t[0] = pack_bf2(v[0], v[1]); t[1] = pack_bf2(v[2], v[3]);
          *(u32x2*)((bf16_t*)a.o + off) = t;
        } else {
          if (a.o) { f32x4 t = {v[0], v[1], v[2], v[3]}; *(f32x4*)((float*)a.o + off) = t; }
          if (a.oS) {          // fp16 hi | lo images for the split-operand out-projection GEMM
            u32x2 hi, lo; split4_f16(v, UVTG_SPLIT_A_SCALE, hi, lo);
            unsigned short* o = a.oS + (rowbase + q_raw) * a.ldoS + split_col(h * HD + dv);
            *(u32x2*)o = hi; *(u32x2*)(o + 32) = lo;
          }
        }
      }
    if (a.lse && g == 0) a.lse[((size_t)b * a.H + h) * a.S + q_raw] = m_run * 0.6931471805599453f + __logf(l_run);
  }
}

// ------------------------------------------------------------------------------------------------
// backward: delta = rowsum(dO * O)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_delta_kernel(const AttnArgs a) {
  // one wave per token row; lane owns 8 consecutive channels; heads are reduced inside groups of hd/8 lanes
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (a.row_sample ? (long long)a.total_rows : (long long)a.B * a.S)) return;
  const int b = a.row_sample ? a.row_sample[row] : (int)(row / a.S);
  const int s = a.row_sample ? (int)(row - a.seq_start[b]) : (int)(row % a.S);
  const int d = a.H * a.hd, lph = a.hd / 8;
  const bf16_t* o = (const bf16_t*)a.o + row * a.ldo;
  const bf16_t* g = a.dO + row * a.lddo;
  for (int c = lane * 8; c < d; c += 512) {
    const s16x8 ov = *(const s16x8*)(o + c), dv = *(const s16x8*)(g + c);
    float acc = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) acc += bf2f((bf16_t)ov[e]) * bf2f((bf16_t)dv[e]);
    for (int off = lph >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((lane % lph) == 0) a.delta[((size_t)b * a.H + c / a.hd) * a.S + s] = acc;
  }
}

// ------------------------------------------------------------------------------------------------
// backward: dK, dV   (one wave = 32 keys, loops over 32-query blocks) -- the long-sequence path (S > 128: BASELINE config 4)
// Round 2: (1) the wave's K and V fragments are loop invariants: they live in registers (fetched once, straight from HBM) instead of
// being re-read from a 64 KB LDS copy every query block; (2) the next query block's Q / dO rows, lse and delta are fetched into
// registers while the current one is multiplied (round 1 loaded, stored to LDS and synchronised with the load latency exposed), and
// the Q / dO tiles are double-buffered, so one barrier per query block; (3) 34 KB of LDS and <= 256 registers: two workgroups per CU.
// (head_dim 128: one workgroup per CU -- the fragments and the 128 accumulator registers do not fit 256 -- but no LDS re-reads.
//  Splitting it into a dV pass and a dK pass (208 / 256 registers, two workgroups per CU, 40 instead of 32 MFMAs per query block and
//  Q / dO streamed twice) measured SLOWER: 2117 vs 1847 us for the whole backward at B=32, S=1232.)
// ------------------------------------------------------------------------------------------------
// (head_dim 128: ~330 registers, one workgroup per CU.  Re-fetching the V fragments per query block -- what keeps the fused kernel at 256 --
//  measured 30 % SLOWER here: 39 query blocks x 2560 workgroups x 32 KB = 3.2 GB of L2 reads per launch, and the kernel spilled.)
// ABL (measurement builds only, -DUVTG_ATTN_ABLATE; results are garbage): 1 = no product MFMAs / transposing reads, 2 = no softmax arithmetic,
// 3 = no staging / prefetch after the first block, 4 = no barrier in the loop, 5 = no score MFMAs / row reads
template <int HD, bool DROP, bool SWZ, int ABL = 0>
__global__ __launch_bounds__(256, (HD < 128 && !DROP) ? 2 : 1) void attn_bwd_dkdv_kernel(const AttnArgs a) {
  using QT = TileRT<HD, SWZ>;
  // Software pipeline over the query blocks (round 3): iteration qb computes S / dP and the softmax section of block qb AND the dV / dK
  // products of block qb - 1 -- two independent instruction streams in one basic block, so that the 16 MFMAs of the older block run under
  // the exp2 / pack VALU work of the newer one (at head_dim 128 this kernel is alone on its SIMD: nothing else hides them).  Three tile
  // buffers, one barrier per iteration: the buffer written at the top of iteration qb was last read in iteration qb - 2.
  __shared__ __attribute__((aligned(16))) bf16_t sQ[3][32 * QT::STR];
  __shared__ __attribute__((aligned(16))) bf16_t sO[3][32 * QT::STR];
  __shared__ __attribute__((aligned(16))) float sL[3][32], sD[3][32];     // lse * log2(e) (ROW_OFF beyond S), delta
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, l31 = lane & 31;
  const int i16 = lane & 15, qd = (lane >> 4) & 1;
  int kblk, h, b;
  attn_block_id((a.S + 127) / 128, a.H, a.B, kblk, h, b, a.sample_major);
  const int S = a.seq_count ? a.seq_count[b] : a.S, d = a.H * HD;
  if (kblk * 128 >= S) return;
  const size_t rowbase = a.seq_start ? (size_t)a.seq_start[b] : (size_t)b * S;
  const int key0 = kblk * 128;
  const bf16_t* qkv = (const bf16_t*)a.qkv;
  constexpr int CH = HD / 8;
  const int key = key0 + wave * 32 + l31;            // this lane's key (lane <-> key in S, dP tiles)
  const bool kin = key < S;
  const bool kok = kin && a.kvalid[rowbase + min(key, S - 1)];
  // this lane's K / V row as MFMA B fragments: k-step ks covers head-dim columns 16 ks + 8 g .. + 7
  // (keys beyond S: clamped duplicates whose rows are not stored)
  s16x8 kf[HD / 16], vf[HD / 16];
  {
    const bf16_t* kvbase = qkv + (rowbase + min(key, S - 1)) * a.ldqkv + h * HD + 8 * g;
#pragma unroll
    for (int ks = 0; ks < HD / 16; ks++) {
      kf[ks] = *(const s16x8*)(kvbase + d + 16 * ks);
      vf[ks] = *(const s16x8*)(kvbase + 2 * d + 16 * ks);
    }
  }
  f32x16 dk[HD / 32], dv[HD / 32];
#pragma unroll
  for (int i = 0; i < HD / 32; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) { dk[i][r] = 0.f; dv[i][r] = 0.f; }
  [[maybe_unused]] const unsigned rng_stream = UVTG_RNG_ATTN + a.layer;

  constexpr int NPT = (32 * CH + 255) / 256;         // staging pieces per thread per operand
  u32x4 pq[NPT], po[NPT];
  float pl = 0.f, pdl = 0.f;
  auto prefetch = [&](int qb) {
#pragma unroll
    for (int i = 0; i < NPT; i++) {
      const int q = tid + 256 * i, r = q / CH, c = q % CH;
      const int qi = min(qb * 32 + r, S - 1);          // clamped rows are loaded; their probabilities are exact zeros (ROW_OFF)
      const bool on = q < 32 * CH;
      const u32x4 z = {0, 0, 0, 0};
      pq[i] = on ? *(const u32x4*)(qkv + (rowbase + qi) * a.ldqkv + h * HD + c * 8) : z;
      po[i] = on ? *(const u32x4*)(a.dO + (rowbase + qi) * a.lddo + h * HD + c * 8) : z;
    }
    if (tid < 32) {          // (raw values: anything computed from them HERE would wait out the whole load latency right behind the request)
      const int qi = min(qb * 32 + tid, S - 1);
      pl = a.lse[((size_t)b * a.H + h) * a.S + qi];
      pdl = a.delta[((size_t)b * a.H + h) * a.S + qi];
    }
  };
  auto stage = [&](int buf, int qb) {                 // the prefetched rows of block qb -> tile buffer `buf`
#pragma unroll
    for (int i = 0; i < NPT; i++) {
      const int q = tid + 256 * i;
      if (q < 32 * CH) {
        const int r = q / CH, c = q % CH;
        *(u32x4*)(&sQ[buf][QT::off(r, c * 8)]) = pq[i];
        *(u32x4*)(&sO[buf][QT::off(r, c * 8)]) = po[i];
      }
    }
    if (tid < 32) { sL[buf][tid] = (qb * 32 + tid < S) ? pl * LOG2E : ROW_OFF; sD[buf][tid] = pdl; }
  };
  // LDS offsets of this lane's fragments inside a tile (with SWZ the chunk index is lane dependent: steps are XORs, not immediates)
  const int qoff = QT::off(l31, 8 * g);                                        // + k-step: QT::step(qoff, 16 ks)
  const int toff0 = QT::off(4 * g + (i16 >> 2), 16 * qd + 4 * (i16 & 3));      // + head-dim block: QT::step(toff, 32 blk); + 16 rows for the
  const int toff1 = QT::rows8(toff0);                                          // second half of the block (the same chunk permutation)
  // S = Q K^T, dP = dO V^T of the block in buffer `buf`: reg r <-> query (r&3)+8(r>>2)+4g, lane <-> key.  The row fragments are fetched
  // AHEAD k-steps in front of their MFMAs, in fenced steps (left alone the compiler keeps ONE fragment pair in flight and every MFMA waits
  // out an LDS round trip); `extra(ks)` issues further LDS reads behind the MFMAs of k-step ks (the caller's fragments for what follows)
  auto scores = [&](int buf, f32x16& sc, f32x16& dp, auto&& extra) {
    const bf16_t* bq = sQ[buf];
    const bf16_t* bo = sO[buf];
    constexpr int KS = HD / 16, AHEAD = KS < 4 ? KS : 4;
    s16x8 qf[KS], of[KS];
#pragma unroll
    for (int r = 0; r < 16; r++) { sc[r] = 0.f; dp[r] = 0.f; }
    if constexpr (ABL != 5) {
#pragma unroll
      for (int ks = 0; ks < AHEAD; ks++) { qf[ks] = *(const s16x8*)(&bq[QT::step(qoff, 16 * ks)]); of[ks] = *(const s16x8*)(&bo[QT::step(qoff, 16 * ks)]); }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      if constexpr (ABL != 5) {
        sc = mfma32(qf[ks], kf[ks], sc);
        dp = mfma32(of[ks], vf[ks], dp);
      }
      if (ABL != 5 && ks + AHEAD < KS) { qf[ks + AHEAD] = *(const s16x8*)(&bq[QT::step(qoff, 16 * (ks + AHEAD))]); of[ks + AHEAD] = *(const s16x8*)(&bo[QT::step(qoff, 16 * (ks + AHEAD))]); }
      extra(ks);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // P (after dropout) and dS of the block, packed as the B operands of the dV / dK products: [hf] = queries 16 hf .. 16 hf + 15
  auto softmax = [&](int qb, int buf, const f32x16& sc, const f32x16& dp, s16x8 (&pb)[2], s16x8 (&db)[2]) {
    f32x4 Lq[4], Dq[4];     // row statistics of this lane's 16 accumulator registers: queries 8 j + 4 g + (0..3), j = 0..3
#pragma unroll
    for (int j = 0; j < 4; j++) { Lq[j] = *(const f32x4*)(&sL[buf][8 * j + 4 * g]); Dq[j] = *(const f32x4*)(&sD[buf][8 * j + 4 * g]); }
    float pd[16], ds[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const float p = exp2_raw(fmaf(sc[r], LOG2E, -Lq[r >> 2][r & 3]));      // (padded keys: see the stores at the end)
      if constexpr (DROP) {
        const int qi = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        const float ksc = keep_scale(a.seed, rng_stream, b * a.H + h, qi, key, a.S, a.p_drop);
        pd[r] = p * ksc;
        ds[r] = p * (dp[r] * ksc - Dq[r >> 2][r & 3]);
      } else {
        pd[r] = p;
        ds[r] = p * (dp[r] - Dq[r >> 2][r & 3]);
      }
    }
#pragma unroll
    for (int hf = 0; hf < 2; hf++) { pb[hf] = pack8(&pd[8 * hf]); db[hf] = pack8(&ds[8 * hf]); }
  };
  // dV^T += dO^T P, dK^T += Q^T dS of the block in buffer `buf`
  auto products = [&](int buf, const s16x8 (&pb)[2], const s16x8 (&db)[2]) {
    const bf16_t* bq = sQ[buf];
    const bf16_t* bo = sO[buf];
#pragma unroll
    for (int hf = 0; hf < 2; hf++)
#pragma unroll
      for (int blk = 0; blk < HD / 32; blk++) {
        const int o0 = QT::step(toff0, 32 * blk) + 16 * hf * QT::STR, o1 = QT::step(toff1, 32 * blk) + 16 * hf * QT::STR;
        const s16x8 ot = cat4(lds_tr16(&bo[o0]), lds_tr16(&bo[o1]));
        const s16x8 qt = cat4(lds_tr16(&bq[o0]), lds_tr16(&bq[o1]));
        dv[blk] = mfma32(ot, pb[hf], dv[blk]);
        dk[blk] = mfma32(qt, db[hf], dk[blk]);
      }
  };
  const int nqb = (S + 31) / 32;
  s16x8 pb[2], db[2];
  prefetch(0);
  stage(0, 0);
  __syncthreads();
  if (1 < nqb) prefetch(1);
  {
    f32x16 sc, dp;
    scores(0, sc, dp, [](int) {});
    softmax(0, 0, sc, dp, pb, db);
  }
  int bc = 1, bp = 0;                                  // tile buffers of the current / the previous block
  for (int qb = 1; qb < nqb; qb++) {
    if constexpr (ABL != 3) stage(bc, qb);
    if constexpr (ABL != 4) __syncthreads();          // (buffer bc was last read two iterations ago; every wave passed the barrier in between)
    if constexpr (ABL != 3) { if (qb + 1 < nqb) prefetch(qb + 1); }
    // The older block's 16 product MFMAs go out two at a time, each pair followed by the softmax arithmetic of two score elements of the
    // newer block (~12 VALU operations: they issue while the pair runs), fenced so that the order survives the scheduler -- left alone it
    // issues all 32 MFMAs first and the ~190 VALU operations behind them, and with in-order issue nothing overlaps.  The transposed
    // fragments of a pair are fetched two pairs ahead; the first two pairs' ride under the score MFMAs.
    const bf16_t* pq_ = sQ[bp];
    const bf16_t* po_ = sO[bp];
    auto frag = [&](int step, s16x8& ot, s16x8& qt) {       // step = (HD / 32) hf + blk
      const int hf = step / (HD / 32), blk = step % (HD / 32);
      const int o0 = QT::step(toff0, 32 * blk) + 16 * hf * QT::STR, o1 = QT::step(toff1, 32 * blk) + 16 * hf * QT::STR;
      ot = cat4(lds_tr16(&po_[o0]), lds_tr16(&po_[o1]));
      qt = cat4(lds_tr16(&pq_[o0]), lds_tr16(&pq_[o1]));
    };
    constexpr int NSTEP = 2 * (HD / 32), EPS = 16 / NSTEP;   // product steps (pairs of MFMAs), score elements per step
    s16x8 fo[2], fq[2];
    f32x4 Lq[4], Dq[4];     // row statistics of this lane's 16 accumulator registers: queries 8 j + 4 g + (0..3), j = 0..3
    f32x16 sc, dp;
    constexpr int KS = HD / 16;
    scores(bc, sc, dp, [&](int ks) {      // behind the last four k-steps (no row fragments left to fetch): the first two product steps' fragments, the statistics
      if (ABL != 1 && (ks == KS - 4 || KS < 4 && ks == 0)) frag(0, fo[0], fq[0]);
      if (ABL != 1 && (ks == KS - 3 || KS < 4 && ks == 0)) frag(1, fo[1], fq[1]);
      if (ks == KS - 2 || KS < 4 && ks == KS - 1) {
#pragma unroll
        for (int j = 0; j < 4; j++) Lq[j] = *(const f32x4*)(&sL[bc][8 * j + 4 * g]);
      }
      if (ks == KS - 1) {
#pragma unroll
        for (int j = 0; j < 4; j++) Dq[j] = *(const f32x4*)(&sD[bc][8 * j + 4 * g]);
      }
    });
    unsigned pw[8], dw[8];                                   // the newer block's P / dS, packed pair by pair as they are produced
#pragma unroll
    for (int step = 0; step < NSTEP; step++) {
      const int hf = step / (HD / 32), blk = step % (HD / 32);
      if constexpr (ABL != 1) {
        dv[blk] = mfma32(fo[step & 1], pb[hf], dv[blk]);
        dk[blk] = mfma32(fq[step & 1], db[hf], dk[blk]);
        if (step + 2 < NSTEP) frag(step + 2, fo[step & 1], fq[step & 1]);
      }
#pragma unroll
      for (int e = 0; e < EPS; e += 2) {
        float pd[2], ds[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int r = step * EPS + e + u;
          if constexpr (ABL == 2) { pd[u] = sc[r]; ds[u] = dp[r]; continue; }
          const float p = exp2_raw(fmaf(sc[r], LOG2E, -Lq[r >> 2][r & 3]));
          if constexpr (DROP) {
            const int qi = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
            const float ksc = keep_scale(a.seed, rng_stream, b * a.H + h, qi, key, a.S, a.p_drop);
            pd[u] = p * ksc;
            ds[u] = p * (dp[r] * ksc - Dq[r >> 2][r & 3]);
          } else {
            pd[u] = p;
            ds[u] = p * (dp[r] - Dq[r >> 2][r & 3]);
          }
        }
        pw[(step * EPS + e) / 2] = pack_bf2(pd[0], pd[1]);
        dw[(step * EPS + e) / 2] = pack_bf2(ds[0], ds[1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int hf = 0; hf < 2; hf++) {
      pb[hf] = __builtin_bit_cast(s16x8, (u32x4){pw[4 * hf], pw[4 * hf + 1], pw[4 * hf + 2], pw[4 * hf + 3]});
      db[hf] = __builtin_bit_cast(s16x8, (u32x4){dw[4 * hf], dw[4 * hf + 1], dw[4 * hf + 2], dw[4 * hf + 3]});
    }
    bp = bc; bc = bc == 2 ? 0 : bc + 1;
  }
  products(bp, pb, db);
  // A lane's key contributes to nobody's sums but its own dK / dV row: the key-padding mask is applied HERE (a padded key's row is zero)
  // instead of as a select per probability in the loop
  if (kin) {
    const u32x2 z = {0u, 0u};
#pragma unroll
    for (int blk = 0; blk < HD / 32; blk++)
#pragma unroll
      for (int rq = 0; rq < 4; rq++) {
        const int c = blk * 32 + 8 * rq + 4 * g;
        bf16_t* base = a.dqkv + (rowbase + key) * a.lddqkv + h * HD + c;
        u32x2 t;
        t[0] = pack_bf2(dk[blk][4 * rq], dk[blk][4 * rq + 1]); t[1] = pack_bf2(dk[blk][4 * rq + 2], dk[blk][4 * rq + 3]);
        *(u32x2*)(base + d) = kok ? t : z;
        t[0] = pack_bf2(dv[blk][4 * rq], dv[blk][4 * rq + 1]); t[1] = pack_bf2(dv[blk][4 * rq + 2], dv[blk][4 * rq + 3]);
        *(u32x2*)(base + 2 * d) = kok ? t : z;
      }
  }
}

// ------------------------------------------------------------------------------------------------
// backward: dK, dV at head_dim 128, ROLE-SPLIT (round 5; BASELINE config 4, S = 1232).  attn_bwd_dkdv_kernel above needs ~360-420 registers at
// head_dim 128 (128 dK / dV accumulators + 64 K / V fragment registers + scores + fragments in flight): ONE wave per SIMD, so the ~190 VALU
// operations of a query block's softmax and the 32 MFMAs of the block cannot overlap however they are interleaved in the one instruction stream
// (in-order issue; the matrix pipe was busy 18-24 % of the time).  Here the work of a 128-key block is cut between two kinds of waves that meet
// on the same SIMD.  S-wave kg (four of them) keeps the K / V fragments of keys 32 kg .. 32 kg + 31 and computes S = Q K^T, dP = dO V^T and the
// whole softmax section of query block t: P and dS leave as the packed bf16 B operands the product MFMAs take -- 64 bytes per lane and block,
// lane-order image in LDS (4 x ds_write_b128, conflict-free by construction), double buffered.  The P-waves keep the dK / dV accumulators
// and compute dV += dO^T P, dK += Q^T dS of block t - 1.  Both roles fit 256 registers: eight waves per workgroup, two per SIMD, and the S-wave's
// VALU section runs under the P-wave's MFMAs.  One barrier per query block, as before; Q / dO tiles: the same three-buffer ring (tile t + 1
// staged while t is read by rows and t - 1 transposed).  The two roles are two separate loops (not one loop with a role branch): the register
// allocator then sees that the K / V fragments and the accumulators are never live together.
// Which accumulators a P-wave owns (HDP):
//   false -- the keys of S-wave kg x all 128 head-dim columns (first version).  Every P-wave then fetches the WHOLE transposed Q / dO tile: 32
//            ds_read_b64_tr_b16 per wave and block, the slowest LDS read (profiles/r03_lds_conflict_probe.txt), four times over for the same bytes.
//            In-box (profiles/r05_ab_attn_dkdv_role_split_modes.txt): attention backward of config 4 3.27 -> 2.94 ms; moving dP / dS work from the
//            S-wave to the P-wave made it SLOWER (3.23 / 3.01): the P-wave, not the S-wave, was the long stream.
//   true  -- head-dim block w (32 columns) x all 128 keys: 8 transposing reads per wave and block instead of 32, and 16 ds_read_b128 of the
//            hand-off image (all four key groups) instead of 4.  Same MFMAs on the same operands in the same order per accumulator.
// Either way the results are bit-identical to attn_bwd_dkdv_kernel's (tests/test_gpu_kernels.py::test_attention_bwd_role_split_dkdv_bit_identical).
// ------------------------------------------------------------------------------------------------
template <bool DROP, bool SWZ, bool HDP, bool PF2>
__global__ __launch_bounds__(512, 2) void attn_bwd_dkdv_ws_kernel(const AttnArgs a) {
  constexpr int HD = 128;
  using QT = TileRT<HD, SWZ>;
  __shared__ __attribute__((aligned(16))) bf16_t sQ[3][32 * QT::STR];
  __shared__ __attribute__((aligned(16))) bf16_t sO[3][32 * QT::STR];
  __shared__ __attribute__((aligned(16))) float sL[3][32], sD[3][32];     // lse * log2(e) (ROW_OFF beyond S), delta
  __shared__ __attribute__((aligned(16))) u32x4 sH[2][4][4][64];          // hand-off: [slot][key group][P q0-15, P q16-31, dS q0-15, dS q16-31][lane]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 5, l31 = lane & 31;
  const int i16 = lane & 15, qd = (lane >> 4) & 1;
  const int role = wave >> 2, kg = wave & 3;          // role 0: S-wave of key group kg; role 1: P-wave (HDP: of head-dim block kg, else of key group kg)
  int kblk, h, b;
  attn_block_id((a.S + 127) / 128, a.H, a.B, kblk, h, b, a.sample_major);
  const int S = a.seq_count ? a.seq_count[b] : a.S, d = a.H * HD;
  if (kblk * 128 >= S) return;
  const size_t rowbase = a.seq_start ? (size_t)a.seq_start[b] : (size_t)b * S;
  const int key0 = kblk * 128;
  const bf16_t* qkv = (const bf16_t*)a.qkv;
  constexpr int CH = HD / 8;
  [[maybe_unused]] const unsigned rng_stream = UVTG_RNG_ATTN + a.layer;
  // staging: 32 rows x 16 chunks of Q and of dO per query block = one 16-byte piece per thread and operand.  PF2: TWO register sets -- the rows of
  // block t + 1 are staged from the set that was requested TWO iterations ago (first build, !PF2: one iteration ago -- an iteration is ~1 us, about
  // one global load round trip under load, and every iteration opened with a wait for it)
  u32x4 aq, ao, bq2, bo2;                 // set A / set B
  float al = 0.f, adl = 0.f, bl = 0.f, bdl = 0.f;
#define WS_PREFETCH(qb_, Q_, O_, L_, D_) do {                                                                  \
    const int r_ = tid / CH, c_ = tid % CH;                                                                    \
    const int qi_ = min((qb_) * 32 + r_, S - 1);      /* clamped rows are loaded; their probabilities are exact zeros (ROW_OFF) */ \
    Q_ = *(const u32x4*)(qkv + (rowbase + qi_) * a.ldqkv + h * HD + c_ * 8);                                   \
    O_ = *(const u32x4*)(a.dO + (rowbase + qi_) * a.lddo + h * HD + c_ * 8);                                   \
    if (tid < 32) {      /* raw values: anything computed from them HERE would wait out the whole load latency right behind the request */ \
      const int q2_ = min((qb_) * 32 + tid, S - 1);                                                            \
      L_ = a.lse[((size_t)b * a.H + h) * a.S + q2_];                                                           \
      D_ = a.delta[((size_t)b * a.H + h) * a.S + q2_];                                                         \
    }                                                                                                          \
  } while (0)
#define WS_STAGE(buf_, qb_, Q_, O_, L_, D_) do {      /* the prefetched rows of block qb_ -> tile buffer buf_ */ \
    const int r_ = tid / CH, c_ = tid % CH;                                                                    \
    *(u32x4*)(&sQ[buf_][QT::off(r_, c_ * 8)]) = Q_;                                                            \
    *(u32x4*)(&sO[buf_][QT::off(r_, c_ * 8)]) = O_;                                                            \
    if (tid < 32) { sL[buf_][tid] = ((qb_) * 32 + tid < S) ? L_ * LOG2E : ROW_OFF; sD[buf_][tid] = D_; }       \
  } while (0)
  const int nqb = (S + 31) / 32;
  WS_PREFETCH(0, aq, ao, al, adl);
  WS_STAGE(0, 0, aq, ao, al, adl);
  if (1 < nqb) WS_PREFETCH(1, aq, ao, al, adl);
  if (PF2 && 2 < nqb) WS_PREFETCH(2, bq2, bo2, bl, bdl);
  __syncthreads();
  // top of iteration t: stage tile t + 1 from the set of its parity, then re-fill that set with tile t + 3 (!PF2: one set, tile t + 2)
  auto advance_a = [&](int t) {      // even t (or !PF2: every t): set A
    if (t + 1 < nqb) WS_STAGE((t + 1) % 3, t + 1, aq, ao, al, adl);
    if (t + (PF2 ? 3 : 2) < nqb) WS_PREFETCH(t + (PF2 ? 3 : 2), aq, ao, al, adl);
  };
  auto advance_b = [&](int t) {      // odd t with PF2: set B
    if (t + 1 < nqb) WS_STAGE((t + 1) % 3, t + 1, bq2, bo2, bl, bdl);
    if (t + 3 < nqb) WS_PREFETCH(t + 3, bq2, bo2, bl, bdl);
  };
  // iteration t = 0 .. nqb: tile t + 1 is staged and tile t + 2 requested by everybody; the S-waves work on block t, the P-waves on block t - 1
  if (a.ws_prio == 1 + role) __builtin_amdgcn_s_setprio(1);      // (experiment: static priority for one role)
  if (role == 0) {
    // ---- S-waves: this lane's K / V row as MFMA B fragments (k-step ks covers head-dim columns 16 ks + 8 g .. + 7) ----
    constexpr int KS = HD / 16, AHEAD = 4;
    const int key = key0 + kg * 32 + l31;              // this lane's key (lane <-> key in the S, dP tiles)
    s16x8 kf[KS], vf[KS];
    {
      const bf16_t* kvbase = qkv + (rowbase + min(key, S - 1)) * a.ldqkv + h * HD + 8 * g;
#pragma unroll
      for (int ks = 0; ks < KS; ks++) {
        kf[ks] = *(const s16x8*)(kvbase + d + 16 * ks);
        vf[ks] = *(const s16x8*)(kvbase + 2 * d + 16 * ks);
      }
      // These loads must be WAITED FOR HERE.  Left to hipcc, the waits sit at the fragments' first use -- inside the loop, as a counted
      // s_waitcnt vmcnt(15) ... vmcnt(0) ladder along the 16 score MFMAs -- and in every later iteration that same ladder drains the rows
      // the iteration has just requested for two blocks ahead: the S-waves stood still for a global round trip per query block with the P-waves
      // parked at the barrier (first builds: 3150 cycles per block at 32 % matrix-pipe occupancy; PMC: 39 % of the wave cycles in s_waitcnt).
#pragma unroll
      for (int ks = 0; ks < KS; ks++) asm volatile("" :: "v"(kf[ks]), "v"(vf[ks]));
    }
    const int qoff = QT::off(l31, 8 * g);                                        // Q / dO row fragments: + k-step: QT::step(qoff, 16 ks)
    auto s_block = [&](int t) {
      {
        const int buf = t % 3;
        const bf16_t* bq = sQ[buf];
        const bf16_t* bo = sO[buf];
        f32x16 sc, dp;
#pragma unroll
        for (int r = 0; r < 16; r++) { sc[r] = 0.f; dp[r] = 0.f; }
        s16x8 qf[KS], of[KS];
#pragma unroll
        for (int ks = 0; ks < AHEAD; ks++) { qf[ks] = *(const s16x8*)(&bq[QT::step(qoff, 16 * ks)]); of[ks] = *(const s16x8*)(&bo[QT::step(qoff, 16 * ks)]); }
        if (a.ws_prio == 3) __builtin_amdgcn_s_setprio(1);      // (experiment: the S-wave's MFMAs first, its VALU section under the P-wave's MFMAs)
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
          sc = mfma32(qf[ks], kf[ks], sc);
          dp = mfma32(of[ks], vf[ks], dp);
          if (ks + AHEAD < KS) { qf[ks + AHEAD] = *(const s16x8*)(&bq[QT::step(qoff, 16 * (ks + AHEAD))]); of[ks + AHEAD] = *(const s16x8*)(&bo[QT::step(qoff, 16 * (ks + AHEAD))]); }
        }
        if (a.ws_prio == 3) __builtin_amdgcn_s_setprio(0);
        f32x4 Lq[4], Dq[4];     // row statistics of this lane's 16 accumulator registers: queries 8 j + 4 g + (0..3), j = 0..3
#pragma unroll
        for (int j = 0; j < 4; j++) { Lq[j] = *(const f32x4*)(&sL[buf][8 * j + 4 * g]); Dq[j] = *(const f32x4*)(&sD[buf][8 * j + 4 * g]); }
        unsigned pw[8], dw[8];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          float pd[2], ds[2];
#pragma unroll
          for (int u = 0; u < 2; u++) {
            const int rr = r + u;
            const float pp = exp2_raw(fmaf(sc[rr], LOG2E, -Lq[rr >> 2][rr & 3]));      // (padded keys: masked at the dK / dV stores)
            if constexpr (DROP) {
              const int qi = t * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * g;
              const float ksc = keep_scale(a.seed, rng_stream, b * a.H + h, qi, key, a.S, a.p_drop);
              pd[u] = pp * ksc;
              ds[u] = pp * (dp[rr] * ksc - Dq[rr >> 2][rr & 3]);
            } else {
              pd[u] = pp;
              ds[u] = pp * (dp[rr] - Dq[rr >> 2][rr & 3]);
            }
          }
          pw[r >> 1] = pack_bf2(pd[0], pd[1]);
          dw[r >> 1] = pack_bf2(ds[0], ds[1]);
        }
        u32x4* hs = &sH[t & 1][kg][0][lane];
        hs[0] = (u32x4){pw[0], pw[1], pw[2], pw[3]};
        hs[64] = (u32x4){pw[4], pw[5], pw[6], pw[7]};
        hs[128] = (u32x4){dw[0], dw[1], dw[2], dw[3]};
        hs[192] = (u32x4){dw[4], dw[5], dw[6], dw[7]};
      }
      __syncthreads();
    };
    // (two iterations per trip: which register set an iteration stages from is then a compile-time fact -- selected by `t & 1` inside one loop body
    //  hipcc kept one set in scratch)
    for (int t = 0; t < nqb; t += 2) {
      advance_a(t);
      s_block(t);
      if (t + 1 < nqb) {
        if constexpr (PF2) advance_b(t + 1); else advance_a(t + 1);
        s_block(t + 1);
      }
    }
    __syncthreads();       // (iteration nqb: the P-waves multiply the last block)
  } else {
    // ---- P-waves: dV^T += dO^T P, dK^T += Q^T dS of the previous block; the accumulators live here ----
    // accumulator i: !HDP head-dim block i of key group kg; HDP key group i of head-dim block kg
    f32x16 dk[4], dv[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) { dk[i][r] = 0.f; dv[i][r] = 0.f; }
    const int toff0 = QT::off(4 * g + (i16 >> 2), 16 * qd + 4 * (i16 & 3));      // + head-dim block: QT::step(toff, 32 blk); + 16 rows for the
    const int toff1 = QT::rows8(toff0);                                          // second half of the block (the same chunk permutation)
    // (iteration 0 peeled: with `if (t >= 1)` around the accumulator updates hipcc kept two copies of the 128 accumulator registers and moved
    //  them every iteration -- 130 v_mov_b32 per block in the first build)
    auto p_block = [&](int t) {
      // (experiment, ws_prio 4 / 5 / 6: the P-wave starts its block 256 / 512 / 768 cycles late -- its MFMAs then run beside the S-wave's VALU
      //  section instead of beside the S-wave's MFMAs)
      if (a.ws_prio == 4) __builtin_amdgcn_s_sleep(4);
      else if (a.ws_prio == 5) __builtin_amdgcn_s_sleep(8);
      else if (a.ws_prio == 6) __builtin_amdgcn_s_sleep(12);
      {
        const int buf = (t - 1) % 3;
        const bf16_t* bq = sQ[buf];
        const bf16_t* bo = sO[buf];
        if constexpr (!HDP) {
          const u32x4* hs = &sH[(t - 1) & 1][kg][0][lane];
          s16x8 pb[2], db[2];
          pb[0] = __builtin_bit_cast(s16x8, hs[0]); pb[1] = __builtin_bit_cast(s16x8, hs[64]);
          db[0] = __builtin_bit_cast(s16x8, hs[128]); db[1] = __builtin_bit_cast(s16x8, hs[192]);
#pragma unroll
          for (int hf = 0; hf < 2; hf++)
#pragma unroll
            for (int blk = 0; blk < HD / 32; blk++) {
              const int o0 = QT::step(toff0, 32 * blk) + 16 * hf * QT::STR, o1 = QT::step(toff1, 32 * blk) + 16 * hf * QT::STR;
              const s16x8 ot = cat4(lds_tr16(&bo[o0]), lds_tr16(&bo[o1]));
              const s16x8 qt = cat4(lds_tr16(&bq[o0]), lds_tr16(&bq[o1]));
              dv[blk] = mfma32(ot, pb[hf], dv[blk]);
              dk[blk] = mfma32(qt, db[hf], dk[blk]);
            }
        } else {
          const u32x4* hs = &sH[(t - 1) & 1][0][0][lane];
          s16x8 ot[2], qt[2];
#pragma unroll
          for (int hf = 0; hf < 2; hf++) {
            const int o0 = QT::step(toff0, 32 * kg) + 16 * hf * QT::STR, o1 = QT::step(toff1, 32 * kg) + 16 * hf * QT::STR;
            ot[hf] = cat4(lds_tr16(&bo[o0]), lds_tr16(&bo[o1]));
            qt[hf] = cat4(lds_tr16(&bq[o0]), lds_tr16(&bq[o1]));
          }
          // ALL sixteen hand-off operands are requested up front and fenced there.  Left alone hipcc read each one into the same four registers
          // right in front of its MFMA (write-after-read on the previous MFMA's operand + an exposed LDS round trip per pair): the P-wave -- 16 MFMAs --
          // was the long stream of the iteration (delaying it by 256 / 512 / 768 cycles cost exactly that: profiles/r05_attn_ws_skew.txt).
          s16x8 pv[4][2], dvv[4][2];
#pragma unroll
          for (int j = 0; j < 4; j++)
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
              pv[j][hf] = __builtin_bit_cast(s16x8, hs[256 * j + 64 * hf]);
              dvv[j][hf] = __builtin_bit_cast(s16x8, hs[256 * j + 128 + 64 * hf]);
            }
          __builtin_amdgcn_sched_barrier(0);
          // (hf outermost: eight independent MFMAs between two uses of an accumulator)
#pragma unroll
          for (int hf = 0; hf < 2; hf++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
              dv[j] = mfma32(ot[hf], pv[j][hf], dv[j]);
              dk[j] = mfma32(qt[hf], dvv[j][hf], dk[j]);
            }
        }
      }
      __syncthreads();
    };
    advance_a(0);
    __syncthreads();
    for (int t = 1; t <= nqb; t += 2) {
      if constexpr (PF2) advance_b(t); else advance_a(t);
      p_block(t);
      if (t + 1 <= nqb) {
        advance_a(t + 1);
        p_block(t + 1);
      }
    }
    // A lane's key contributes to nobody's sums but its own dK / dV row: the key-padding mask is applied HERE (a padded key's row is zero)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int key = key0 + (HDP ? i : kg) * 32 + l31, blk = HDP ? kg : i;
      if (key < S) {
        const bool kok = a.kvalid[rowbase + key] != 0;
        const u32x2 z = {0u, 0u};
#pragma unroll
        for (int rq = 0; rq < 4; rq++) {
          const int c = blk * 32 + 8 * rq + 4 * g;
          bf16_t* base = a.dqkv + (rowbase + key) * a.lddqkv + h * HD + c;
          u32x2 tt;
          tt[0] = pack_bf2(dk[i][4 * rq], dk[i][4 * rq + 1]); tt[1] = pack_bf2(dk[i][4 * rq + 2], dk[i][4 * rq + 3]);
          *(u32x2*)(base + d) = kok ? tt : z;
          tt[0] = pack_bf2(dv[i][4 * rq], dv[i][4 * rq + 1]); tt[1] = pack_bf2(dv[i][4 * rq + 2], dv[i][4 * rq + 3]);
          *(u32x2*)(base + 2 * d) = kok ? tt : z;
        }
      }
    }
  }
}

#undef WS_PREFETCH
#undef WS_STAGE

// ------------------------------------------------------------------------------------------------
// backward: dQ   (one wave = 32 queries, loops over 64-key tiles)
// ------------------------------------------------------------------------------------------------
template <int HD, bool DROP, bool SWZ>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const AttnArgs a) {
  using KT = TileRT<HD, SWZ>;
  constexpr int VSTR = HD + 8;                       // V is read by rows only
  __shared__ __attribute__((aligned(16))) bf16_t sK[64 * KT::STR];
  __shared__ __attribute__((aligned(16))) bf16_t sV[64 * VSTR];
  __shared__ __attribute__((aligned(16))) float sBias[64];     // 0 on a real key, -ROW_OFF on padding and beyond S
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, l31 = lane & 31;
  const int i16 = lane & 15, qd = (lane >> 4) & 1;
  int qblk, h, b;
  attn_block_id((a.S + 127) / 128, a.H, a.B, qblk, h, b, a.sample_major);
  const int S = a.seq_count ? a.seq_count[b] : a.S, d = a.H * HD;
  if (qblk * 128 >= S) return;
  const size_t rowbase = a.seq_start ? (size_t)a.seq_start[b] : (size_t)b * S;
  const int q_raw = qblk * 128 + wave * 32 + l31;
  const int qrow = min(q_raw, S - 1);
  const bf16_t* qkv = (const bf16_t*)a.qkv;
  s16x8 qf[HD / 16], of[HD / 16];
#pragma unroll
  for (int ks = 0; ks < HD / 16; ks++) {
    qf[ks] = *(const s16x8*)(qkv + (rowbase + qrow) * a.ldqkv + h * HD + 16 * ks + 8 * g);
    of[ks] = *(const s16x8*)(a.dO + (rowbase + qrow) * a.lddo + h * HD + 16 * ks + 8 * g);
  }
  const float L2 = a.lse[((size_t)b * a.H + h) * a.S + qrow] * LOG2E;
  const float dl = a.delta[((size_t)b * a.H + h) * a.S + qrow];
  f32x16 dq[HD / 32];
#pragma unroll
  for (int i = 0; i < HD / 32; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) dq[i][r] = 0.f;
  [[maybe_unused]] const unsigned rng_stream = UVTG_RNG_ATTN + a.layer;
  constexpr int CH = HD / 8;
  // this lane's fragment offsets inside the K / V tiles (rows 0..31; + 32 rows for kb = 1, + 16 for hf = 1: the same chunk permutation)
  const int koff = KT::off(l31, 8 * g), voff = l31 * VSTR + 8 * g;
  const int toff0 = KT::off(4 * g + (i16 >> 2), 16 * qd + 4 * (i16 & 3)), toff1 = KT::rows8(toff0);
  const int ntiles = (S + 63) / 64;
  for (int kt = 0; kt < ntiles; kt++) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < (64 * CH) / 256; i++) {
      const int q = tid + 256 * i, r = q / CH, c = q % CH;
      const int key = min(kt * 64 + r, S - 1);         // rows beyond S: clamped duplicates, their dS is an exact zero (bias)
      const bf16_t* base = qkv + (rowbase + key) * a.ldqkv + h * HD + c * 8;
      const u32x4 kv = *(const u32x4*)(base + d);
      const u32x4 vv = *(const u32x4*)(base + 2 * d);
      *(u32x4*)(&sK[KT::off(r, c * 8)]) = kv;
      *(u32x4*)(&sV[r * VSTR + c * 8]) = vv;
    }
    if (tid < 64) { const int key = kt * 64 + tid; sBias[tid] = (key < S && a.kvalid[rowbase + key]) ? 0.f : -ROW_OFF; }
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
      f32x4 bq[4];                                     // bias of this lane's 16 accumulator registers: keys kb*32 + 8 j + 4 g + (0..3)
#pragma unroll
      for (int j = 0; j < 4; j++) bq[j] = *(const f32x4*)(&sBias[kb * 32 + 8 * j + 4 * g]);
      f32x16 sc, dp;
#pragma unroll
      for (int r = 0; r < 16; r++) { sc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < HD / 16; ks++) {
        sc = mfma32(*(const s16x8*)(&sK[KT::step(koff, 16 * ks) + kb * 32 * KT::STR]), qf[ks], sc);
        dp = mfma32(*(const s16x8*)(&sV[voff + 16 * ks + kb * 32 * VSTR]), of[ks], dp);
      }
      float ds[16];
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float p = exp2_raw(fmaf(sc[r], LOG2E, bq[r >> 2][r & 3] - L2));
        if constexpr (DROP) {
          const int kl = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
          const float ksc = keep_scale(a.seed, rng_stream, b * a.H + h, q_raw, kt * 64 + kl, a.S, a.p_drop);
          ds[r] = p * (dp[r] * ksc - dl);
        } else {
          ds[r] = p * (dp[r] - dl);
        }
      }
#pragma unroll
      for (int hf = 0; hf < 2; hf++) {
        const s16x8 db = pack8(&ds[8 * hf]);
#pragma unroll
        for (int blk = 0; blk < HD / 32; blk++) {
          const int rowo = (kb * 32 + 16 * hf) * KT::STR;
          const s16x8 kt_ = cat4(lds_tr16(&sK[KT::step(toff0, 32 * blk) + rowo]), lds_tr16(&sK[KT::step(toff1, 32 * blk) + rowo]));
          dq[blk] = mfma32(kt_, db, dq[blk]);
        }
      }
    }
  }
  if (q_raw < S) {
#pragma unroll
    for (int blk = 0; blk < HD / 32; blk++)
#pragma unroll
      for (int rq = 0; rq < 4; rq++) {
        const int c = blk * 32 + 8 * rq + 4 * g;
        u32x2 t;
        t[0] = pack_bf2(dq[blk][4 * rq] * a.qscale, dq[blk][4 * rq + 1] * a.qscale);
        t[1] = pack_bf2(dq[blk][4 * rq + 2] * a.qscale, dq[blk][4 * rq + 3] * a.qscale);
        *(u32x2*)(a.dqkv + (rowbase + q_raw) * a.lddqkv + h * HD + c) = t;
      }
  }
}

// ------------------------------------------------------------------------------------------------
// backward: dQ at head_dim 128 with the K / V tiles delivered by LDS-DMA (round 3).  The kernel above loads a tile into registers, stores it
// to LDS and synchronises twice per tile with the load latency in the open (PMC: 46 % of its wave cycles wait on a counter; 254 registers
// leave no room for a register prefetch).  Here the 64-key tile kt + 1 is requested straight into the OTHER half of a double buffer while
// tile kt is multiplied: `buffer_load ... lds` (16 bytes per lane, 1 KB = 4 swizzled tile rows per wave-instruction, the chunk permutation
// applied on the SOURCE address), no registers, one barrier per tile.  The requests are issued from inline asm: hipcc answers every transposing
// LDS read that follows an LDS-DMA it knows of with `s_waitcnt vmcnt(0)` (gemm.hip, weight-gradient kernel), which would put the latency back.
// ------------------------------------------------------------------------------------------------
typedef int __attribute__((ext_vector_type(4))) i32x4_t;
__device__ __forceinline__ void attn_dma16(i32x4_t rsrc, unsigned voff, unsigned lds_dst) {      // (M0 is the compiler's: saved and restored)
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}
template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_dma_kernel(const AttnArgs a, unsigned qkv_bytes) {
  constexpr int HD = 128;
  using KT = TileRT<HD, true>;
  __shared__ __attribute__((aligned(1024))) bf16_t sKV[2][2][64 * HD];      // [buffer][K | V][64 keys x 128], swizzled 256-byte rows
  __shared__ __attribute__((aligned(16))) float sBias[2][64];               // 0 on a real key, -ROW_OFF on padding and beyond S
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 5, l31 = lane & 31;
  const int i16 = lane & 15, qd = (lane >> 4) & 1;
  int qblk, h, b;
  attn_block_id((a.S + 127) / 128, a.H, a.B, qblk, h, b, a.sample_major);
  const int S = a.seq_count ? a.seq_count[b] : a.S, d = a.H * HD;
  if (qblk * 128 >= S) return;
  const size_t rowbase = a.seq_start ? (size_t)a.seq_start[b] : (size_t)b * S;
  const int q_raw = qblk * 128 + wave * 32 + l31;
  const int qrow = min(q_raw, S - 1);
  const bf16_t* qkv = (const bf16_t*)a.qkv;
  // buffer resource over the whole qkv matrix (wave-uniform)
  i32x4_t rsrc;
  {
    const unsigned long long pa = (unsigned long long)(uintptr_t)qkv;
    rsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)pa);
    rsrc[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(pa >> 32) & 0xffffu));
    rsrc[2] = __builtin_amdgcn_readfirstlane((int)qkv_bytes);
    rsrc[3] = 0x00020000;
  }
  const unsigned lds0 = (unsigned)(uintptr_t)&sKV[0][0][0];
  // this lane's share of a tile: pieces 4 wave .. 4 wave + 3 of K and of V (a piece = 4 tile rows = 1 KB); lane l supplies physical chunk
  // l & 15 of row 4 p + (l >> 4), i.e. the logical chunk (l & 15) ^ (4 (l >> 4) + (p & 3))
  auto request = [&](int kt, int buf) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int p = wave * 4 + i, row = 4 * p + (lane >> 4);
      const int chunk = (lane & 15) ^ (((lane >> 4) << 2) | (p & 3));
      const int key = min(kt * 64 + row, S - 1);          // rows beyond S: clamped duplicates, their dS is an exact zero (bias)
      const unsigned src = (unsigned)(((rowbase + key) * (size_t)a.ldqkv + (size_t)h * HD + (size_t)chunk * 8) * 2);
      attn_dma16(rsrc, src + (unsigned)d * 2u, lds0 + (unsigned)((buf * 2 + 0) * 64 * HD * 2 + p * 1024));
      attn_dma16(rsrc, src + (unsigned)d * 4u, lds0 + (unsigned)((buf * 2 + 1) * 64 * HD * 2 + p * 1024));
    }
  };
  // key-padding byte of key kt * 64 + tid (tid < 64), RAW: the bias is derived from it where it is stored, at the END of the iteration.  Round 5:
  // computed right behind the request (`nbias = kvalid[..] ? 0 : -ROW_OFF`) hipcc put `s_waitcnt vmcnt(0)` between the tile's LDS-DMA issue and
  // its own compute in wave 0 -- which drained the DMA it had just issued: wave 0 stood still for a global round trip per key tile and the
  // other three waited for it at the next barrier.
  auto valid_of = [&](int kt) -> unsigned char {        // tid < 64
    return a.kvalid[rowbase + min(kt * 64 + tid, S - 1)];
  };
  auto bias_from = [&](int kt, unsigned char v) -> float { return (kt * 64 + tid < S && v) ? 0.f : -ROW_OFF; };
  const int ntiles = (S + 63) / 64;
  unsigned char nvalid = tid < 64 ? valid_of(0) : (unsigned char)0;
  request(0, 0);
  s16x8 qf[HD / 16], of[HD / 16];
#pragma unroll
  for (int ks = 0; ks < HD / 16; ks++) {
    qf[ks] = *(const s16x8*)(qkv + (rowbase + qrow) * a.ldqkv + h * HD + 16 * ks + 8 * g);
    of[ks] = *(const s16x8*)(a.dO + (rowbase + qrow) * a.lddo + h * HD + 16 * ks + 8 * g);
  }
  const float L2 = a.lse[((size_t)b * a.H + h) * a.S + qrow] * LOG2E;
  const float dl = a.delta[((size_t)b * a.H + h) * a.S + qrow];
  f32x16 dq[HD / 32];
#pragma unroll
  for (int i = 0; i < HD / 32; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) dq[i][r] = 0.f;
  [[maybe_unused]] const unsigned rng_stream = UVTG_RNG_ATTN + a.layer;
  // this lane's fragment offsets inside a tile (rows 0..31; + 32 rows for kb = 1, + 16 for hf = 1: the same chunk permutation)
  const int koff = KT::off(l31, 8 * g);
  const int toff0 = KT::off(4 * g + (i16 >> 2), 16 * qd + 4 * (i16 & 3)), toff1 = KT::rows8(toff0);
  if (tid < 64) sBias[0][tid] = bias_from(0, nvalid);
  for (int kt = 0; kt < ntiles; kt++) {
    const int buf = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of tile kt have landed (and everything older)
    __syncthreads();                                     // ... everybody's; and everybody is done with the other buffer (tile kt - 1)
    if (kt + 1 < ntiles) {
      if (tid < 64) nvalid = valid_of(kt + 1);          // (requested BEFORE the DMA pieces, consumed behind the tile's compute)
      request(kt + 1, buf ^ 1);
    }
    const bf16_t* sK = sKV[buf][0];
    const bf16_t* sV = sKV[buf][1];
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
      f32x4 bq[4];                                     // bias of this lane's 16 accumulator registers: keys kb*32 + 8 j + 4 g + (0..3)
#pragma unroll
      for (int j = 0; j < 4; j++) bq[j] = *(const f32x4*)(&sBias[buf][kb * 32 + 8 * j + 4 * g]);
      f32x16 sc, dp;
#pragma unroll
      for (int r = 0; r < 16; r++) { sc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < HD / 16; ks++) {
        const int o = KT::step(koff, 16 * ks) + kb * 32 * KT::STR;
        sc = mfma32(*(const s16x8*)(&sK[o]), qf[ks], sc);
        dp = mfma32(*(const s16x8*)(&sV[o]), of[ks], dp);
      }
      float ds[16];
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float p = exp2_raw(fmaf(sc[r], LOG2E, bq[r >> 2][r & 3] - L2));
        if constexpr (DROP) {
          const int kl = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
          const float ksc = keep_scale(a.seed, rng_stream, b * a.H + h, q_raw, kt * 64 + kl, a.S, a.p_drop);
          ds[r] = p * (dp[r] * ksc - dl);
        } else {
          ds[r] = p * (dp[r] - dl);
        }
      }
#pragma unroll
      for (int hf = 0; hf < 2; hf++) {
        const s16x8 db = pack8(&ds[8 * hf]);
#pragma unroll
        for (int blk = 0; blk < HD / 32; blk++) {
          const int rowo = (kb * 32 + 16 * hf) * KT::STR;
          const s16x8 kt_ = cat4(lds_tr16(&sK[KT::step(toff0, 32 * blk) + rowo]), lds_tr16(&sK[KT::step(toff1, 32 * blk) + rowo]));
          dq[blk] = mfma32(kt_, db, dq[blk]);
        }
      }
    }
    if (kt + 1 < ntiles && tid < 64) sBias[buf ^ 1][tid] = bias_from(kt + 1, nvalid);     // (visible behind the next barrier; its last readers passed this one)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no LDS-DMA may outlive the workgroup's LDS allocation
  if (q_raw < S) {
#pragma unroll
    for (int blk = 0; blk < HD / 32; blk++)
#pragma unroll
      for (int rq = 0; rq < 4; rq++) {
        const int c = blk * 32 + 8 * rq + 4 * g;
        u32x2 t;
        t[0] = pack_bf2(dq[blk][4 * rq] * a.qscale, dq[blk][4 * rq + 1] * a.qscale);
        t[1] = pack_bf2(dq[blk][4 * rq + 2] * a.qscale, dq[blk][4 * rq + 3] * a.qscale);
        *(u32x2*)(a.dqkv + (rowbase + q_raw) * a.lddqkv + h * HD + c) = t;
      }
  }
}

// ------------------------------------------------------------------------------------------------
// backward, S <= 128: dQ, dK, dV of one (sample, head) in ONE pass over the operands (the split dK/dV + dQ kernels
// above read q, k, v, dO twice and recompute the scores twice).  wave w owns keys 32w .. 32w+31: it keeps its V rows in
// registers, accumulates dK / dV for them over the 32-query blocks exactly like attn_bwd_dkdv_kernel, and publishes its
// dS tile (bf16) to LDS; after a barrier wave w computes the head-dim tile 32w .. 32w+31 of dQ^T = K^T dS^T for the
// query block from the full K (LDS resident, padded rows: row reads and transposing reads are both conflict-free).
// No attention dropout in this kernel (the counter-based RNG per element costs ~150 registers when unrolled, which would
// halve the occupancy); with p_drop > 0 -- never the case in the reference scripts, scripts/pretrain.sh:34 -- the split
// kernels run instead.
// ------------------------------------------------------------------------------------------------
// NW = 4: up to 128 keys, two workgroups per CU.  NW = 8 (128 < S <= 256, e.g. the pretraining shape L_v = 128 + 32 text tokens): 256 keys,
// 8 waves, one workgroup per CU (the same two waves per SIMD); the dQ^T tile of a query block is summed over two key halves by two
// wave groups and folded through an fp32 LDS slab.
// ABL (measurement builds only, -DUVTG_ATTN_ABLATE; results are garbage): 1 = no dQ pass, 2 = no product MFMAs / transposing reads, 3 = no score
// MFMAs / row reads, 4 = no softmax arithmetic, 5 = no V re-fetch, 6 = no Q / dO prefetch + staging after the first block, 7 = no dK / dV epilogue
template <int HD, int NW, bool SWZ, int ABL = 0>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void attn_bwd_fused_kernel(const AttnArgs a) {
  constexpr int KROWS = 32 * NW, T = 64 * NW;
  constexpr int DQSTR = HD + 8, CH = HD / 8;
  using KT = TileRT<HD, SWZ>;           // K, Q, dO tiles: read by rows and transposed
  __shared__ __attribute__((aligned(16))) bf16_t sK[KROWS * KT::STR];
  __shared__ float sPart[NW == 8 ? 4 * 16 * 64 : 1];     // NW = 8: dQ^T partials of the upper key half, [hd tile][acc register][lane]
  __shared__ __attribute__((aligned(16))) bf16_t sQ[32 * KT::STR];
  __shared__ __attribute__((aligned(16))) bf16_t sO[32 * KT::STR];
  // dS of the query block as [key][32 queries] (64-byte rows), 8-byte unit u (4 queries) of key k stored at unit u ^ ((k >> 1) & 7): each
  // lane (= key) stores the two packed halves of its dK operand as they are (4 x ds_write_b64, 16 consecutive keys on 16 different bank
  // pairs) and the dQ pass fetches its B fragments with the transposing read (4 keys x 64 bytes per 32-lane group = all 64 banks).  The
  // [query][key] image it replaces took 16 two-byte stores per lane and block.
  __shared__ __attribute__((aligned(16))) bf16_t sDS[KROWS * 32];
  __shared__ __attribute__((aligned(16))) bf16_t sDQ[32 * DQSTR];   // dQ rows of the previous query block, stored out coalesced
  __shared__ __attribute__((aligned(16))) float sL[32], sD[32];     // lse * log2(e) (ROW_OFF beyond S), delta
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, l31 = lane & 31;
  const int i16 = lane & 15, qd = (lane >> 4) & 1;
  const int b = blockIdx.z, h = blockIdx.y, S = a.seq_count ? a.seq_count[b] : a.S, d = a.H * HD;
  const size_t rowbase = a.seq_start ? (size_t)a.seq_start[b] : (size_t)b * S;
  const bf16_t* qkv = (const bf16_t*)a.qkv;
  const int key = wave * 32 + l31;
  const int krow = min(key, S - 1);
  const bool kok = key < S && a.kvalid[rowbase + krow];
  const bf16_t* vrow = qkv + (rowbase + krow) * a.ldqkv + 2 * d + h * HD + 8 * g;   // this lane's V row (A operand of dP = dO V^T)
  f32x16 dk[HD / 32], dv[HD / 32];
#pragma unroll
  for (int i = 0; i < HD / 32; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) { dk[i][r] = 0.f; dv[i][r] = 0.f; }
  constexpr int NP = (32 * CH + T - 1) / T;          // staging pieces per thread per operand
  u32x4 pq[NP], po[NP];
  float pl = 0.f, pdl = 0.f;                         // lse / delta of query qb*32 + tid (tid < 32), fetched with the block's rows
  auto prefetch = [&](int qb) {
    if (tid < 32) {          // (raw values: anything computed from them HERE would wait out the whole load latency right behind the request)
      const int qi = min(qb * 32 + tid, S - 1);
      pl = a.lse[((size_t)b * a.H + h) * a.S + qi];
      pdl = a.delta[((size_t)b * a.H + h) * a.S + qi];
    }
#pragma unroll
    for (int i = 0; i < NP; i++) {
      const int q = tid + T * i, r = q / CH, c = q % CH, qi = min(qb * 32 + r, S - 1);   // clamped rows: exact-zero probabilities (ROW_OFF)
      if (q < 32 * CH) {
        pq[i] = *(const u32x4*)(qkv + (rowbase + qi) * a.ldqkv + h * HD + c * 8);
        po[i] = *(const u32x4*)(a.dO + (rowbase + qi) * a.lddo + h * HD + c * 8);
      }
    }
  };
  const int nqb = (S + 31) / 32;
  prefetch(0);                                       // first query block's rows are in flight while K is staged
  // K (all keys) -> LDS; rows beyond S are clamped duplicates (kok masks them)
#pragma unroll
  for (int i = 0; i < (KROWS * CH) / T; i++) {
    const int q = tid + T * i, r = q / CH, c = q % CH;
    *(u32x4*)(&sK[KT::off(r, c * 8)]) = *(const u32x4*)(qkv + (rowbase + min(r, S - 1)) * a.ldqkv + d + h * HD + c * 8);
  }
  // loop-invariant LDS offsets of this lane's fragments
  // (few bases: the kernel sits at the 256-register line, everything else is derived where it is used)
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int qoff = KT::off(l31, 8 * g);                                         // Q / dO row l31, + k-step: KT::step(., 16 ks); the K row of
                                                                                // this lane's key is 32 wave rows further (same permutation)
  const int toff0 = KT::off(4 * g + (i16 >> 2), 16 * qd + 4 * (i16 & 3));       // transposed Q / dO fragments; + 16 rows (hf), + head-dim block
  const int tile_hd = wave_s & 3, khalf = wave_s >> 2;
  const int ktoff0 = KT::off(8 * g + (i16 >> 2), tile_hd * 32 + 16 * qd + 4 * (i16 & 3)) + khalf * 128 * KT::STR;     // dQ pass: K^T fragments,
                                                                                                                      // + 16 kk rows
  const int dsw = key * 32 + (((key >> 1) & 7) << 2);                           // dS store: unit u of this key at dsw ^ (u << 2)
  // dS fetch (dQ pass): keys 16 kk + 8 g + 4 t + (i16 >> 2); t = 1: four rows further, unit index ^ 2
  const int dsr0 = (khalf * 128 + 8 * g + (i16 >> 2)) * 32 + (((4 * qd + (i16 & 3)) ^ ((4 * g + (i16 >> 3)) & 7)) << 2);
  // the dQ^T fragments hold (head dim, query): 8 bytes per lane at a 6 KB stride.  Each wave drops its tile into sDQ instead and
  // the whole block writes the 32 rows out 16 bytes per lane, 16 lanes per contiguous 256-byte row, one query block later.
  auto flush_dq = [&](int qbp) {
#pragma unroll
    for (int i = 0; i < NP; i++) {
      const int q = tid + T * i, r = q / CH, c = q % CH, qi = qbp * 32 + r;
      if (q < 32 * CH && qi < S) *(u32x4*)(a.dqkv + (rowbase + qi) * a.lddqkv + h * HD + c * 8) = *(const u32x4*)(&sDQ[r * DQSTR + c * 8]);
    }
  };
  for (int qb = 0; qb < nqb; qb++) {
    __syncthreads();                                 // previous block's readers of sQ / sO / sDS are done, its dQ tiles are in sDQ
    if (qb > 0) flush_dq(qb - 1);
    if (ABL != 6 || qb == 0) {
#pragma unroll
    for (int i = 0; i < NP; i++) {
      const int q = tid + T * i, r = q / CH, c = q % CH;
      if (q < 32 * CH) { *(u32x4*)(&sQ[KT::off(r, c * 8)]) = pq[i]; *(u32x4*)(&sO[KT::off(r, c * 8)]) = po[i]; }
    }
    }
    if (tid < 32) { sL[tid] = (qb * 32 + tid < S) ? pl * LOG2E : ROW_OFF; sD[tid] = pdl; }
    // V fragments are re-fetched (L2 hits) per query block instead of living in 32 registers across the whole loop:
    // with the 128 dK / dV accumulators that keeps the kernel at two workgroups per CU without spilling
    // (hoisting these loads above the first barrier measured 11 % slower in round 2; fully resident -- loaded once in front of the loop -- the
    //  compiler parks 25 dwords in scratch and the backward takes 0.59 instead of 0.38 ms per step: profiles/r03_ab_fused_attn_v_resident.txt)
    s16x8 vf[HD / 16];
#pragma unroll
    for (int ks = 0; ks < HD / 16; ks++) { if constexpr (ABL == 5) vf[ks] = (s16x8){1, 2, 3, 4, 5, 6, 7, 8}; else vf[ks] = *(const s16x8*)(vrow + 16 * ks); }
    __syncthreads();
    // (the swizzled fragment offsets are XORs of these bases: behind an opaque move they are recomputed where they are used instead of
    //  being hoisted out of the loop into ~30 registers the kernel does not have)
    int qo = qoff, t0 = toff0, kt0 = ktoff0;
    if constexpr (SWZ) asm volatile("" : "+v"(qo), "+v"(t0), "+v"(kt0));
    const int ko = qo + wave_s * 32 * KT::STR, t1 = KT::rows8(t0), kt1 = KT::rows4(kt0);
    // S = Q K^T, dP = dO V^T : reg r <-> query (r&3)+8(r>>2)+4g, lane <-> key
    f32x16 sc, dp;
#pragma unroll
    for (int r = 0; r < 16; r++) { sc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < HD / 16; ks++) {
      if constexpr (ABL == 3) { sc[ks] += __builtin_bit_cast(float, (int)vf[ks][0]); continue; }
      const s16x8 qf = *(const s16x8*)(&sQ[KT::step(qo, 16 * ks)]);
      const s16x8 of = *(const s16x8*)(&sO[KT::step(qo, 16 * ks)]);
      const s16x8 kf = *(const s16x8*)(&sK[KT::step(ko, 16 * ks)]);
      sc = mfma32(qf, kf, sc);
      dp = mfma32(of, vf[ks], dp);
    }
    // the next query block's rows are requested HERE, when the 32 registers of the V fragments are dead (the kernel sits at the 256-register
    // line: requested before the MFMAs, the compiler parks them in scratch -- behind a vmcnt(0) that exposes the whole load latency)
    if (ABL != 6 && qb + 1 < nqb) prefetch(qb + 1);
    // row statistics of this lane's 16 accumulator registers: queries 8 j + 4 g + (0..3), j = 0..3
    f32x4 Lq[4], Dq[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { Lq[j] = *(const f32x4*)(&sL[8 * j + 4 * g]); Dq[j] = *(const f32x4*)(&sD[8 * j + 4 * g]); }
    float pd[16], ds[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      if constexpr (ABL == 4) { pd[r] = sc[r] * 1e-3f; ds[r] = dp[r] * 1e-3f; continue; }
      float p = exp2_raw(fmaf(sc[r], LOG2E, -Lq[r >> 2][r & 3]));
      p = kok ? p : 0.f;
      pd[r] = p;
      ds[r] = p * (dp[r] - Dq[r >> 2][r & 3]);
    }
#pragma unroll
    for (int hf = 0; hf < 2; hf++) {
      const s16x8 pb = pack8(&pd[8 * hf]);
      const s16x8 db = pack8(&ds[8 * hf]);
      // db = dS of queries 16 hf + 4 g + (0..3) | 16 hf + 8 + 4 g + (0..3): units 4 hf + g and 4 hf + 2 + g of this key's row
      *(s16x4*)(&sDS[dsw ^ ((4 * hf + g) << 2)]) = (s16x4){db[0], db[1], db[2], db[3]};
      *(s16x4*)(&sDS[dsw ^ ((4 * hf + 2 + g) << 2)]) = (s16x4){db[4], db[5], db[6], db[7]};
#pragma unroll
      for (int blk = 0; blk < HD / 32; blk++) {
        const int o0 = KT::step(t0, 32 * blk) + 16 * hf * KT::STR, o1 = KT::step(t1, 32 * blk) + 16 * hf * KT::STR;
        if constexpr (ABL == 2) { dv[blk][0] += __builtin_bit_cast(float, (int)pb[0]); dk[blk][0] += __builtin_bit_cast(float, (int)db[0]); continue; }
        const s16x8 ot = cat4(lds_tr16(&sO[o0]), lds_tr16(&sO[o1]));
        const s16x8 qt = cat4(lds_tr16(&sQ[o0]), lds_tr16(&sQ[o1]));
        dv[blk] = mfma32(ot, pb, dv[blk]);
        dk[blk] = mfma32(qt, db, dk[blk]);
      }
    }
    __syncthreads();                                 // every wave's dS tile of this query block is in sDS
    // dQ^T tile (head dims 32 t .. +31) x (32 queries) = sum over the keys of K^T dS^T: wave w takes tile t = w % 4 over the keys
    // 128 (w / 4) .. + 127; with 8 waves the upper half's partial goes through sPart and the lower half's waves finish the tile
    {
      const bool dq_on = tile_hd * 32 < HD;
      f32x16 dq;
#pragma unroll
      for (int r = 0; r < 16; r++) dq[r] = 0.f;
      if (dq_on && ABL != 1) {
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
          const s16x8 kt_ = cat4(lds_tr16(&sK[kt0 + 16 * kk * KT::STR]), lds_tr16(&sK[kt1 + 16 * kk * KT::STR]));
          const s16x8 db = cat4(lds_tr16(&sDS[dsr0 + 16 * kk * 32]), lds_tr16(&sDS[(dsr0 ^ 8) + 4 * 32 + 16 * kk * 32]));
          dq = mfma32(kt_, db, dq);
        }
      }
      if constexpr (NW == 8) {
        if (dq_on && khalf == 1) {
#pragma unroll
          for (int r = 0; r < 16; r++) sPart[(tile_hd * 16 + r) * 64 + lane] = dq[r];
        }
        __syncthreads();
        if (dq_on && khalf == 0) {
#pragma unroll
          for (int r = 0; r < 16; r++) dq[r] += sPart[(tile_hd * 16 + r) * 64 + lane];
        }
      }
      if (dq_on && khalf == 0) {
#pragma unroll
        for (int rq = 0; rq < 4; rq++) {
          const int c = tile_hd * 32 + 8 * rq + 4 * g;
          u32x2 t;
          t[0] = pack_bf2(dq[4 * rq] * a.qscale, dq[4 * rq + 1] * a.qscale);
          t[1] = pack_bf2(dq[4 * rq + 2] * a.qscale, dq[4 * rq + 3] * a.qscale);
          *(u32x2*)(&sDQ[l31 * DQSTR + c]) = t;
        }
      }
    }
  }
  __syncthreads();                                   // last dQ tiles are in sDQ; nobody reads sK any more
  flush_dq(nqb - 1);
  if constexpr (ABL == 7) { if (dk[0][0] + dv[0][0] == 12345.f) a.dqkv[0] = 1; return; }
  // dK, dV: same transposition through a wave-private slab in the dead K tile (32 keys x HD, the K tile's own row layout)
  bf16_t* slab = sK + wave * 32 * KT::STR;
#pragma unroll
  for (int which = 0; which < 2; which++) {
#pragma unroll
    for (int blk = 0; blk < HD / 32; blk++)
#pragma unroll
      for (int rq = 0; rq < 4; rq++) {
        const int c = blk * 32 + 8 * rq + 4 * g;
        u32x2 t;
        if (which == 0) { t[0] = pack_bf2(dk[blk][4 * rq], dk[blk][4 * rq + 1]); t[1] = pack_bf2(dk[blk][4 * rq + 2], dk[blk][4 * rq + 3]); }
        else { t[0] = pack_bf2(dv[blk][4 * rq], dv[blk][4 * rq + 1]); t[1] = pack_bf2(dv[blk][4 * rq + 2], dv[blk][4 * rq + 3]); }
        *(u32x2*)(slab + KT::off(l31, c)) = t;
      }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < (32 * CH) / 64; i++) {
      const int idx = lane + 64 * i, r = idx / CH, c = idx % CH, kk = wave * 32 + r;
      const u32x4 v = *(const u32x4*)(slab + KT::off(r, c * 8));
      if (kk < S) *(u32x4*)(a.dqkv + (rowbase + kk) * a.lddqkv + (which + 1) * d + h * HD + c * 8) = v;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

void uvtg_prof_begin_launch(int family, double flops, hipStream_t s);
void uvtg_prof_end_launch(int family, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// forward at head_dim 128, bf16, K / V tiles by LDS-DMA (round 5).  attn_fwd_kernel stages a 64-key tile through registers into ONE LDS buffer:
// two barriers per tile, the staging stores in front of the compute, the next tile's request behind them -- at S = 1232 it ran 578 TFLOP/s where the
// dQ kernel, which walks the same K / V tiles with the LDS-DMA double buffer below and ONE barrier per tile, ran 1125 (tools/attn_bench.py).  This is
// that kernel's tile loop (same requests, same swizzled tile image: K by rows, V transposed through ds_read_b64_tr_b16) around the forward's online
// softmax and output transposition.  Same products in the same order as attn_fwd_kernel<128, false>: bit-identical outputs and lse.  No attention
// dropout here (p_drop > 0 takes the other kernel).  Measured (profiles/r05_attn_bench_fwd_dma.txt): 340 -> 317 us per layer at S = 1232
// (628 TFLOP/s) -- the tile delivery was NOT what held the forward: per 64-key tile a wave issues 32 MFMAs (1024 matrix-pipe cycles) against
// ~240 VALU instructions of online softmax, 33 of them quarter-rate v_exp_f32 (~1350 cycles): two waves per SIMD are VALU-bound at ~75 % of the
// matrix rate before any dependency stall.
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256, 2) void attn_fwd_dma_kernel(const AttnArgs a, unsigned qkv_bytes) {
  constexpr int HD = 128;
  using KT = TileRT<HD, true>;
  __shared__ __attribute__((aligned(1024))) bf16_t sKV[2][2][64 * HD];      // [buffer][K | V][64 keys x 128], swizzled 256-byte rows
  __shared__ __attribute__((aligned(16))) float sBias[2][64];               // 0 on a real key, NEG_BIG on padding and beyond S (log2 domain)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 5, l31 = lane & 31;
  const int i16 = lane & 15, qd = (lane >> 4) & 1;
  int qblk, h, b;
  attn_block_id((a.S + 127) / 128, a.H, a.B, qblk, h, b, a.sample_major);
  const int S = a.seq_count ? a.seq_count[b] : a.S, d = a.H * HD;
  if (qblk * 128 >= S) return;
  const size_t rowbase = a.seq_start ? (size_t)a.seq_start[b] : (size_t)b * S;
  const int q_raw = qblk * 128 + wave * 32 + l31;
  const int qrow = min(q_raw, S - 1);
  const bf16_t* qkv = (const bf16_t*)a.qkv;
  i32x4_t rsrc;       // buffer resource over the whole qkv matrix (wave-uniform)
  {
    const unsigned long long pa = (unsigned long long)(uintptr_t)qkv;
    rsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)pa);
    rsrc[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(pa >> 32) & 0xffffu));
    rsrc[2] = __builtin_amdgcn_readfirstlane((int)qkv_bytes);
    rsrc[3] = 0x00020000;
  }
  const unsigned lds0 = (unsigned)(uintptr_t)&sKV[0][0][0];
  // this lane's share of a tile: pieces 4 wave .. 4 wave + 3 of K and of V (a piece = 4 tile rows = 1 KB); lane l supplies physical chunk
  // l & 15 of row 4 p + (l >> 4), i.e. the logical chunk (l & 15) ^ (4 (l >> 4) + (p & 3))
  auto request = [&](int kt, int buf) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int p = wave * 4 + i, row = 4 * p + (lane >> 4);
      const int chunk = (lane & 15) ^ (((lane >> 4) << 2) | (p & 3));
      const int key = min(kt * 64 + row, S - 1);          // rows beyond S: clamped duplicates, their probabilities are exact zeros (bias)
      const unsigned src = (unsigned)(((rowbase + key) * (size_t)a.ldqkv + (size_t)h * HD + (size_t)chunk * 8) * 2);
      attn_dma16(rsrc, src + (unsigned)d * 2u, lds0 + (unsigned)((buf * 2 + 0) * 64 * HD * 2 + p * 1024));
      attn_dma16(rsrc, src + (unsigned)d * 4u, lds0 + (unsigned)((buf * 2 + 1) * 64 * HD * 2 + p * 1024));
    }
  };
  // key-padding byte of key kt * 64 + tid (tid < 64), RAW: requested before the tile's DMA pieces, turned into the bias behind the tile's compute
  auto valid_of = [&](int kt) -> unsigned char { return a.kvalid[rowbase + min(kt * 64 + tid, S - 1)]; };
  auto bias_from = [&](int kt, unsigned char v) -> float { return (kt * 64 + tid < S && v) ? 0.f : NEG_BIG; };
  const int ntiles = (S + 63) / 64;
  unsigned char nvalid = tid < 64 ? valid_of(0) : (unsigned char)0;
  request(0, 0);
  // Q fragments (B operand of S^T = K Q^T): lane (query, g) holds q[16 ks + 8 g .. + 7]
  s16x8 qh[HD / 16];
#pragma unroll
  for (int ks = 0; ks < HD / 16; ks++) qh[ks] = *(const s16x8*)(qkv + (rowbase + qrow) * a.ldqkv + h * HD + 16 * ks + 8 * g);
  f32x16 oacc[HD / 32];
#pragma unroll
  for (int i = 0; i < HD / 32; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) oacc[i][r] = 0.f;
  float m_run = NEG_BIG, l_run = 0.f;
  // this lane's fragment offsets inside a tile (rows 0..31; + 32 rows for kb = 1, + 16 for hf = 1: the same chunk permutation)
  const int koff = KT::off(l31, 8 * g);
  const int toff0 = KT::off(4 * g + (i16 >> 2), 16 * qd + 4 * (i16 & 3)), toff1 = KT::rows8(toff0);
  if (tid < 64) sBias[0][tid] = bias_from(0, nvalid);
  for (int kt = 0; kt < ntiles; kt++) {
    const int buf = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of tile kt have landed (and everything older)
    __syncthreads();                                     // ... everybody's; and everybody is done with the other buffer (tile kt - 1)
    if (kt + 1 < ntiles) {
      if (tid < 64) nvalid = valid_of(kt + 1);
      request(kt + 1, buf ^ 1);
    }
    const bf16_t* sK = sKV[buf][0];
    const bf16_t* sV = sKV[buf][1];
    // ---- S^T = K Q^T : sc[kb][r] <-> key kb*32 + (r&3)+8(r>>2)+4g, query l31 ----
    f32x16 sc[2];
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
#pragma unroll
      for (int r = 0; r < 16; r++) sc[kb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < HD / 16; ks++)
        sc[kb] = mfma32(*(const s16x8*)(&sK[KT::step(koff, 16 * ks) + kb * 32 * KT::STR]), qh[ks], sc[kb]);
    }
    // online softmax in the log2 domain: t = s log2(e) + bias (the running maximum m_run is a log2-domain value too)
    float mx = NEG_BIG;
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const f32x4 bk = *(const f32x4*)(&sBias[buf][kb * 32 + 8 * j + 4 * g]);      // keys kb*32 + 8 j + 4 g + (0..3) = registers 4 j .. 4 j + 3
#pragma unroll
        for (int e = 0; e < 4; e++) {
          sc[kb][4 * j + e] = fmaf(sc[kb][4 * j + e], LOG2E, bk[e]);
          mx = fmaxf(mx, sc[kb][4 * j + e]);
        }
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = exp2_raw(m_run - m_new);
    const float m_use = fmaxf(m_new, -1e20f);          // nothing but padding so far: t - m_use stays at -1e30, p = 0 (not exp2(0))
    float rs = 0.f;
    float pv[2][16];
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float p = exp2_raw(sc[kb][r] - m_use);
        rs += p;
        pv[kb][r] = p;
      }
    rs += __shfl_xor(rs, 32, 64);
    l_run = l_run * alpha + rs;
    m_run = m_new;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {     // (the running maximum settles after the first tiles: HD/2 multiplies saved per tile)
#pragma unroll
      for (int i = 0; i < HD / 32; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) oacc[i][r] *= alpha;
    }
    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int hf = 0; hf < 2; hf++) {
        const s16x8 pb = pack8(&pv[kb][8 * hf]);
        const int rowo = (kb * 32 + 16 * hf) * KT::STR;
#pragma unroll
        for (int dvb = 0; dvb < HD / 32; dvb++) {
          const s16x8 vt = cat4(lds_tr16(&sV[KT::step(toff0, 32 * dvb) + rowo]), lds_tr16(&sV[KT::step(toff1, 32 * dvb) + rowo]));
          oacc[dvb] = mfma32(vt, pb, oacc[dvb]);
        }
      }
    if (kt + 1 < ntiles && tid < 64) sBias[buf ^ 1][tid] = bias_from(kt + 1, nvalid);     // (visible behind the next barrier; its last readers passed this one)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no LDS-DMA may outlive the workgroup's LDS allocation
  // O^T fragments hold (channel, query): transpose through a wave-private slab in the dead tile buffers, half the channels per pass, and store
  // 16 B per lane with HD/16 lanes covering one contiguous half row (as attn_fwd_kernel)
  constexpr int OSTR = HD / 2 + 8, CPR = HD / 16;
  __syncthreads();                                       // every wave is done reading the tiles
  bf16_t* slab = &sKV[0][0][0] + wave * 32 * OSTR;
  const float inv = 1.0f / l_run;
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {
#pragma unroll
    for (int dvh = 0; dvh < HD / 64; dvh++)
#pragma unroll
      for (int rq = 0; rq < 4; rq++) {
        const int dvb = pass * (HD / 64) + dvh;
        u32x2 t;
        t[0] = pack_bf2(oacc[dvb][4 * rq] * inv, oacc[dvb][4 * rq + 1] * inv);
        t[1] = pack_bf2(oacc[dvb][4 * rq + 2] * inv, oacc[dvb][4 * rq + 3] * inv);
        *(u32x2*)(slab + l31 * OSTR + dvh * 32 + 8 * rq + 4 * g) = t;
      }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < (32 * CPR) / 64; i++) {
      const int idx = lane + 64 * i, row = idx / CPR, ch = idx % CPR;
      const int q = qblk * 128 + wave * 32 + row;
      const u32x4 v = *(const u32x4*)(slab + row * OSTR + ch * 8);
      if (q < S) *(u32x4*)((bf16_t*)a.o + (rowbase + q) * a.ldo + h * HD + pass * (HD / 2) + ch * 8) = v;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (q_raw < S && a.lse && g == 0) a.lse[((size_t)b * a.H + h) * a.S + q_raw] = m_run * 0.6931471805599453f + __logf(l_run);
}
}  // namespace

static int g_attn_fwd_dma = -1;      // head_dim-128 bf16 forward: LDS-DMA tile loop (default) / 0: the register-staged kernel (parity tests, A-B)
extern "C" int uvtg_debug_attn_fwd_dma(int on) { g_attn_fwd_dma = on ? 1 : 0; return 0; }
static const bool g_attn_sample_major = uvtg_dev_env("UVTG_ATTN_SAMPLE_MAJOR") != nullptr;      // experiment: the former (sample, head) order of the tiled kernels' block decode
int launch_attn_fwd(const AttnArgs& a0, hipStream_t s) {
  AttnArgs a = a0; a.sample_major = g_attn_sample_major ? 1 : 0;
  if (a.hd != 32 && a.hd != 64 && a.hd != 128) return -5;
  dim3 grid(cdiv(a.S, 128) * a.H * a.B), blk(256);      // decoded by attn_block_id
  uvtg_prof_begin_launch(4, 4.0 * a.B * a.H * (double)a.S * a.S * a.hd, s);
  {   // head_dim 128, bf16, no attention dropout: K / V tiles by LDS-DMA (round 5); the buffer descriptor addresses qkv with 32-bit byte offsets
    static const bool dma_off = uvtg_dev_env("UVTG_ATTN_FWD_DMA_OFF") != nullptr;
    const long long rows_ = a.row_sample ? (long long)a.total_rows : (long long)a.B * a.S;
    const unsigned long long qkv_bytes = (unsigned long long)rows_ * a.ldqkv * 2ull;
    if (a.hd == 128 && !a.precise && a.p_drop <= 0.f && !dma_off && g_attn_fwd_dma != 0 && qkv_bytes < (1ull << 32)) {
      hipLaunchKernelGGL(attn_fwd_dma_kernel, grid, blk, 0, s, a, (unsigned)qkv_bytes);
      uvtg_prof_end_launch(4, s);
      UVTG_CHECK_LAUNCH();
      return 0;
    }
  }
#define FWD(HD_)                                                                                  \
  if (a.hd == HD_) {                                                                              \
    if (a.precise) hipLaunchKernelGGL((attn_fwd_kernel<HD_, true>), grid, blk, 0, s, a);          \
    else hipLaunchKernelGGL((attn_fwd_kernel<HD_, false>), grid, blk, 0, s, a);                   \
  }
  FWD(32) FWD(64) FWD(128)
#undef FWD
  uvtg_prof_end_launch(4, s);
  UVTG_CHECK_LAUNCH();
  return 0;
}

static int g_attn_ws = -1;          // head_dim-128 dK / dV: role-split kernel (default) / 0: the one-wave-per-SIMD kernel (parity tests, A-B)
extern "C" int uvtg_debug_attn_ws(int on) { g_attn_ws = on ? 1 : 0; return 0; }
int launch_attn_bwd(const AttnArgs& a0, hipStream_t s) {
  AttnArgs a = a0; a.sample_major = g_attn_sample_major ? 1 : 0;
  if (a.hd != 32 && a.hd != 64 && a.hd != 128) return -5;
  if (a.precise) return -6;
  const long long rows = a.row_sample ? (long long)a.total_rows : (long long)a.B * a.S;
  uvtg_prof_begin_launch(5, 10.0 * a.B * a.H * (double)a.S * a.S * a.hd, s);
  if (!a.delta_ready) {
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, a);
    UVTG_CHECK_LAUNCH();
  }
  dim3 grid(cdiv(a.S, 128), a.H, a.B), blk(256);
  static const bool swz_off = uvtg_dev_env("UVTG_ATTN_SWZ_OFF") != nullptr;       // experiment: padded rows instead of the chunk swizzle (head_dim 128)
  const bool swz = a.hd == 128 && !swz_off;
#ifdef UVTG_ATTN_ABLATE
  {
    static const int fabl = uvtg_dev_env("UVTG_ATTN_FABL") ? atoi(uvtg_dev_env("UVTG_ATTN_FABL")) : 0;
    if (fabl > 0 && a.S <= 128 && a.hd == 128 && a.p_drop <= 0.f) {
      switch (fabl) {
        case 1: hipLaunchKernelGGL((attn_bwd_fused_kernel<128, 4, false, 1>), grid, blk, 0, s, a); break;
        case 2: hipLaunchKernelGGL((attn_bwd_fused_kernel<128, 4, false, 2>), grid, blk, 0, s, a); break;
        case 3: hipLaunchKernelGGL((attn_bwd_fused_kernel<128, 4, false, 3>), grid, blk, 0, s, a); break;
        case 4: hipLaunchKernelGGL((attn_bwd_fused_kernel<128, 4, false, 4>), grid, blk, 0, s, a); break;
        case 5: hipLaunchKernelGGL((attn_bwd_fused_kernel<128, 4, false, 5>), grid, blk, 0, s, a); break;
        case 6: hipLaunchKernelGGL((attn_bwd_fused_kernel<128, 4, false, 6>), grid, blk, 0, s, a); break;
        default: hipLaunchKernelGGL((attn_bwd_fused_kernel<128, 4, false, 7>), grid, blk, 0, s, a); break;
      }
      uvtg_prof_end_launch(5, s);
      UVTG_CHECK_LAUNCH();
      return 0;
    }
  }
#endif
  if (a.S <= 128 && a.p_drop <= 0.f) {   // whole (sample, head) problem in one workgroup: one pass over q, k, v, dO
    // (4-wave kernel: padded rows -- 0.370 vs 0.390 ms per step at config 2: three or four query blocks per workgroup, the XORs and the 16 bytes
    //  of scratch cost more than the transposing reads gain; the 8-wave kernel and the split kernels gain 5 - 8 % from the swizzle)
    static const bool swz4_on = uvtg_dev_env("UVTG_ATTN_SWZ4_ON") != nullptr;
    if (a.hd == 128) { if (swz && swz4_on) hipLaunchKernelGGL((attn_bwd_fused_kernel<128, 4, true>), grid, blk, 0, s, a); else hipLaunchKernelGGL((attn_bwd_fused_kernel<128, 4, false>), grid, blk, 0, s, a); }
    else if (a.hd == 64) hipLaunchKernelGGL((attn_bwd_fused_kernel<64, 4, false>), grid, blk, 0, s, a);
    else hipLaunchKernelGGL((attn_bwd_fused_kernel<32, 4, false>), grid, blk, 0, s, a);
    uvtg_prof_end_launch(5, s);
    UVTG_CHECK_LAUNCH();
    return 0;
  }
  static const bool fused8_off = uvtg_dev_env("UVTG_ATTN_FUSED8_OFF") != nullptr;      // experiment: the split kernels for 128 < S <= 256
  if (a.S <= 256 && a.p_drop <= 0.f && !fused8_off) {   // the same with 8 waves / 256 keys (one workgroup per CU)
    const dim3 g8(1, a.H, a.B), b8(512);
    if (a.hd == 128) { if (swz) hipLaunchKernelGGL((attn_bwd_fused_kernel<128, 8, true>), g8, b8, 0, s, a); else hipLaunchKernelGGL((attn_bwd_fused_kernel<128, 8, false>), g8, b8, 0, s, a); }
    else if (a.hd == 64) hipLaunchKernelGGL((attn_bwd_fused_kernel<64, 8, false>), g8, b8, 0, s, a);
    else hipLaunchKernelGGL((attn_bwd_fused_kernel<32, 8, false>), g8, b8, 0, s, a);
    uvtg_prof_end_launch(5, s);
    UVTG_CHECK_LAUNCH();
    return 0;
  }
  const dim3 grid1(cdiv(a.S, 128) * a.H * a.B);          // decoded by attn_block_id
  // dQ by LDS-DMA (head_dim 128): the buffer descriptor addresses the qkv matrix with 32-bit byte offsets
  static const bool dq_dma_off = uvtg_dev_env("UVTG_ATTN_DQ_DMA_OFF") != nullptr;
  const unsigned long long qkv_bytes = (unsigned long long)rows * a.ldqkv * 2ull;
  const bool dq_dma = !dq_dma_off && qkv_bytes < (1ull << 32);
#ifdef UVTG_ATTN_ABLATE
  {
    static const int abl = uvtg_dev_env("UVTG_ATTN_ABL") ? atoi(uvtg_dev_env("UVTG_ATTN_ABL")) : 0;
    if (a.hd == 128 && swz && a.p_drop <= 0.f && abl > 0) {
      switch (abl) {
        case 1: hipLaunchKernelGGL((attn_bwd_dkdv_kernel<128, false, true, 1>), grid1, blk, 0, s, a); break;
        case 2: hipLaunchKernelGGL((attn_bwd_dkdv_kernel<128, false, true, 2>), grid1, blk, 0, s, a); break;
        case 3: hipLaunchKernelGGL((attn_bwd_dkdv_kernel<128, false, true, 3>), grid1, blk, 0, s, a); break;
        case 4: hipLaunchKernelGGL((attn_bwd_dkdv_kernel<128, false, true, 4>), grid1, blk, 0, s, a); break;
        case 5: hipLaunchKernelGGL((attn_bwd_dkdv_kernel<128, false, true, 5>), grid1, blk, 0, s, a); break;
        default: hipLaunchKernelGGL((attn_bwd_dkdv_kernel<128, false, true, 0>), grid1, blk, 0, s, a); break;     // 6: the whole kernel, alone
      }
      uvtg_prof_end_launch(5, s);
      UVTG_CHECK_LAUNCH();
      return 0;          // (dK / dV only: the ablation runs time this kernel alone)
    }
  }
#endif
  static const int ws_prio = uvtg_dev_env("UVTG_ATTN_WS_PRIO") ? atoi(uvtg_dev_env("UVTG_ATTN_WS_PRIO")) : 0;
  AttnArgs aw = a; aw.ws_prio = ws_prio;
  static const bool ws_pf1 = uvtg_dev_env("UVTG_ATTN_WS_PF2") == nullptr;      // rows requested one iteration ahead (default; two ahead measured the same and costs the registers the P-wave's operand block needs: UVTG_ATTN_WS_PF2 = experiment)
  static const bool ws_hdp = !uvtg_dev_env("UVTG_ATTN_WS_KEYP");      // product waves own a head-dim block (default) / UVTG_ATTN_WS_KEYP: a key group (first version)
  static const bool ws_off = uvtg_dev_env("UVTG_ATTN_WS_OFF") != nullptr;       // experiment: the one-wave-per-SIMD dK / dV kernel at head_dim 128
#define BWD(HD_, DROP_, SWZ_)                                                                     \
  {                                                                                               \
    if (HD_ == 128 && !ws_off && g_attn_ws != 0) {                                                \
      if (!ws_hdp) hipLaunchKernelGGL((attn_bwd_dkdv_ws_kernel<DROP_, SWZ_, false, true>), grid1, dim3(512), 0, s, aw); \
      else if (ws_pf1) hipLaunchKernelGGL((attn_bwd_dkdv_ws_kernel<DROP_, SWZ_, true, false>), grid1, dim3(512), 0, s, aw); \
      else hipLaunchKernelGGL((attn_bwd_dkdv_ws_kernel<DROP_, SWZ_, true, true>), grid1, dim3(512), 0, s, aw); \
    }                                                                                             \
    else hipLaunchKernelGGL((attn_bwd_dkdv_kernel<HD_, DROP_, SWZ_>), grid1, blk, 0, s, a);       \
    if (HD_ == 128 && dq_dma) {                                                                   \
      hipLaunchKernelGGL((attn_bwd_dq_dma_kernel<DROP_>), grid1, blk, 0, s, a, (unsigned)qkv_bytes); \
    } else hipLaunchKernelGGL((attn_bwd_dq_kernel<HD_, DROP_, SWZ_>), grid1, blk, 0, s, a);       \
  }
  const bool drop = a.p_drop > 0.f;
  if (a.hd == 128) { if (drop) { if (swz) BWD(128, true, true) else BWD(128, true, false) } else { if (swz) BWD(128, false, true) else BWD(128, false, false) } }
  else if (a.hd == 64) { if (drop) BWD(64, true, false) else BWD(64, false, false) }
  else { if (drop) BWD(32, true, false) else BWD(32, false, false) }
#undef BWD
  uvtg_prof_end_launch(5, s);
  UVTG_CHECK_LAUNCH();
  return 0;
}
