// Dense UniVTG criterion on device (model/univtg.py:195-282): SmoothL1 + paired gIoU on the foreground
// clips, weighted BCE on the foreground probability, inter-video and intra-video NCE saliency terms.
// Values and gradients, O(N) paired gIoU instead of the reference's N x N matrix + diag, and no
// device->host synchronisation (the reference's `saliency_scores.sum() == 0` early-out becomes a
// device-side flag that zeroes the two saliency terms and their gradients).
#include "uvtg_kernels.h"

namespace {

constexpr float TAU = 0.07f;          // model/univtg.py:185 (hard-coded)
constexpr float EPS = 1e-8f;          // sim_matrix / cosine_similarity eps

struct WS {                            // carve-up of LossArgs::ws
  float *cosv, *vnorm, *qnorm, *sim, *lse_r, *lse_c, *zr, *zc, *cnt, *gz, *vpn, *dvh, *dqh, *acc;
  __host__ __device__ WS(const LossArgs& a) {
    float* p = a.ws;
    const int B = a.B, Lv = a.Lv, d = a.d;
    cosv = p; p += (size_t)B * Lv;
    vnorm = p; p += (size_t)B * Lv;
    gz = p; p += (size_t)B * Lv;
    sim = p; p += (size_t)B * B;
    qnorm = p; p += B;
    lse_r = p; p += B;
    lse_c = p; p += B;
    zr = p; p += B;
    vpn = p; p += B;
    zc = p; p += Lv;
    p += (4 - ((size_t)(5 * B + Lv) & 3)) & 3;
    dvh = p; p += (size_t)B * d;
    dqh = p; p += (size_t)B * d;
    acc = p; p += 16;                    // 8 accumulators (+ 8 spare), directly followed by cnt: ONE memset zeroes both
    cnt = p; p += Lv;
    // per-clip cosine / norms the model forward already computed (uvtg_forward_saliency_stats): no stats pass over vid_mem_proj
    if (a.cos_c) { cosv = (float*)a.cos_c; vnorm = (float*)a.vnorm_c; qnorm = (float*)a.qnorm_c; }
  }
};

__device__ __forceinline__ const float* vrow(const LossArgs& a, int b, int t) {
  return a.vid + (size_t)b * a.vid_sb + (size_t)t * a.vid_st;
}
__device__ __forceinline__ bool is_neg(const LossArgs& a, int b, int t) {
  // neg_indices_in & mask (model/univtg.py:266-268)
  const int p = (int)a.pos_idx[b];
  const bool n = (a.sal_tgt[b * a.Lv + t] < a.sal_tgt[b * a.Lv + p]) || (t == p);
  return n && (a.ts_mask[b * a.Lv + t] != 0.f);
}
__device__ __forceinline__ float zval(const LossArgs& a, const WS& w, int b, int t) {
  return w.cosv[b * a.Lv + t] + (is_neg(a, b, t) ? 0.f : UVTG_LOG_TINY);
}

// per (b, t): |v|, cos(v, q_b);  per b: |q|
__global__ __launch_bounds__(256) void loss_stats_kernel(const LossArgs a) {
  const WS w(a);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.B * a.Lv) return;
  const int b = row / a.Lv, t = row % a.Lv;
  const float* v = vrow(a, b, t);
  const float* q = a.txt + (size_t)b * a.d;
  float vv = 0.f, qq = 0.f, vq = 0.f;
  for (int c = lane; c < a.d; c += 64) { const float x = v[c], y = q[c]; vv += x * x; qq += y * y; vq += x * y; }
  vv = wave_sum(vv); qq = wave_sum(qq); vq = wave_sum(vq);
  if (lane == 0) {
    const float vn = sqrtf(vv), qn = sqrtf(qq);
    w.vnorm[row] = vn;
    w.cosv[row] = vq / (fmaxf(vn, EPS) * fmaxf(qn, EPS));
    if (t == 0) w.qnorm[b] = qn;
  }
}

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); i++) t += red[i];
  return t;
}

__device__ __forceinline__ void giou_terms(float a0, float a1, float b0, float b1, float& I, float& U, float& H) {
  I = fmaxf(fminf(a1, b1) - fmaxf(a0, b0), 0.f);
  U = (a1 - a0) + (b1 - b0) - I;
  H = fmaxf(fmaxf(a1, b1) - fminf(a0, b0), 0.f);
}

// ---- forward, two launches (round 3; were stats / sim / elem / lse / final = five) ----
// launch 1, by block role: [0, sim blocks) the inter-video similarity rows (loss_sim_kernel's body); then the per-clip terms (SmoothL1, gIoU,
// BCE and the counts, 256 clips per block, ws.acc[8] atomics: nwin, nval, ssum, lb, lg, lf); then the intra-video log-sum-exp tasks, one wave
// each: row b (softmax over the clips of sample b -> zr[b], cnt[pos_b] += 1) and column t (softmax over the batch at clip t -> zc[t]).
// None of these reads another block's output.  ws.acc and ws.cnt are zeroed by ONE memset in front of the launch.
__device__ __forceinline__ void loss_sim_block(const LossArgs& a, const WS& w, int i, int jblock, float* sv) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = (int)a.pos_idx[i];
  const float* v = vrow(a, i, p);
  const float vn = fmaxf(w.vnorm[i * a.Lv + p], EPS);
  for (int c = threadIdx.x; c < a.d; c += 256) sv[c] = v[c] / vn;
  __syncthreads();
  const int jbase = jblock * 64 + wave * 16;
  for (int j0 = jbase; j0 < min(a.B, jbase + 16); j0 += 4) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* q[4];
#pragma unroll
    for (int e = 0; e < 4; e++) q[e] = a.txt + (size_t)min(j0 + e, a.B - 1) * a.d;
    if ((a.d & 255) == 0) {
      for (int c = lane * 4; c < a.d; c += 256) {
        const f32x4 s4 = *(const f32x4*)(sv + c);
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const f32x4 q4 = *(const f32x4*)(q[e] + c);
          acc[e] += s4[0] * q4[0] + s4[1] * q4[1] + s4[2] * q4[2] + s4[3] * q4[3];
        }
      }
    } else {
      for (int c = lane; c < a.d; c += 64) {
        const float sc = sv[c];
#pragma unroll
        for (int e = 0; e < 4; e++) acc[e] += sc * q[e][c];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float t = wave_sum(acc[e]);
      if (lane == 0 && j0 + e < a.B) w.sim[i * a.B + j0 + e] = t / fmaxf(w.qnorm[j0 + e], EPS);
    }
  }
}
__global__ __launch_bounds__(256) void loss_fwd1_kernel(const LossArgs a, int sim_blocks, int elem_blocks, int have_sal) {
  extern __shared__ float sv[];
  __shared__ float red[4];
  const WS w(a);
  const int B = a.B, Lv = a.Lv, n = B * Lv;
  int blk = blockIdx.x;
  if (blk < sim_blocks) {
    const int jb = (B + 63) / 64;
    loss_sim_block(a, w, blk / jb, blk % jb, sv);
    return;
  }
  blk -= sim_blocks;
  if (blk < elem_blocks) {
    const int i = blk * 256 + threadIdx.x;
    float nwin = 0.f, nval = 0.f, ssum = 0.f, lb = 0.f, lg = 0.f, lf = 0.f;
    if (i < n) {
      const float win = a.ts_window[i], msk = a.ts_mask[i];
      nwin = (win != 0.f); nval = (msk != 0.f);
      if (a.sal_tgt) ssum = a.sal_tgt[i];
      if (a.do_spans && win != 0.f) {
        const float s0 = a.timestamp[2 * i] + a.pred_spans[2 * i], s1 = a.timestamp[2 * i + 1] + a.pred_spans[2 * i + 1];
        const float g0 = a.span_nn[2 * i], g1 = a.span_nn[2 * i + 1];
        const float d0 = fabsf(s0 - g0), d1 = fabsf(s1 - g1);
        lb = win * ((d0 < 1.f ? 0.5f * d0 * d0 : d0 - 0.5f) + (d1 < 1.f ? 0.5f * d1 * d1 : d1 - 0.5f));
        float I, U, H;
        giou_terms(s0, s1, g0, g1, I, U, H);
        lg = 1.f - (I / U - (H - U) / H);
      }
      if (a.do_labels && msk != 0.f) {
        const float p = a.pred_logits[i], y = win != 0.f ? 1.f : 0.f, wt = win != 0.f ? 1.f : a.eos_coef;
        lf = -wt * (y * fmaxf(logf(p), -100.f) + (1.f - y) * fmaxf(logf(1.f - p), -100.f));
      }
    }
    float vals[6] = {nwin, nval, ssum, lb, lg, lf};
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const float t = block_sum(vals[k], red);
      if (threadIdx.x == 0) atomicAdd(w.acc + k, t);
    }
    return;
  }
  blk -= elem_blocks;
  if (!have_sal) return;
  const int lane = threadIdx.x & 63;
  const int task = blk * 4 + (threadIdx.x >> 6);
  if (task < B) {
    const int b = task;
    float m = -INFINITY;
    for (int t = lane; t < Lv; t += 64) m = fmaxf(m, zval(a, w, b, t));
    m = wave_max(m);
    float sm = 0.f;
    for (int t = lane; t < Lv; t += 64) sm += expf((zval(a, w, b, t) - m) / TAU);
    sm = wave_sum(sm);
    if (lane == 0) { w.zr[b] = m / TAU + logf(sm); atomicAdd(&w.cnt[(int)a.pos_idx[b]], 1.f); }
  } else if (task < B + Lv) {
    const int t = task - B;
    float m = -INFINITY;
    for (int b = lane; b < B; b += 64) m = fmaxf(m, zval(a, w, b, t));
    m = wave_max(m);
    float sm = 0.f;
    for (int b = lane; b < B; b += 64) sm += expf((zval(a, w, b, t) - m) / TAU);
    sm = wave_sum(sm);
    if (lane == 0) w.zc[t] = m / TAU + logf(sm);
  }
}

// launch 2: one wave per sample i -- the inter-video log-sum-exp of row i and of column i of sim / tau (lanes stride the row / column:
// coalesced, independent loads), the sample's inter- and intra-video NCE terms added into ws.acc[6] / [7]; the block that draws the last
// ticket (ws.acc[8], zeroed with the accumulators) writes the five loss values behind one agent-scope acquire.  (First version: ONE
// 1024-thread block with a thread per row / column -- 94 us of dependent strided loads, measured; the per-block release + acquire here costs
// ~3.5 us and runs concurrently across the blocks.)
__global__ __launch_bounds__(256) void loss_fwd2_kernel(const LossArgs a, int have_sal) {
  __shared__ float red[4];
  __shared__ unsigned s_ticket;
  const WS w(a);
  const int tid = threadIdx.x, lane = tid & 63, B = a.B;
  const float ssum = w.acc[2];
  const bool sal_on = have_sal && ssum != 0.f;
  const int i = blockIdx.x * 4 + (tid >> 6);
  float inter = 0.f, intra = 0.f;
  if (sal_on && i < B) {
    float mr = -INFINITY, mc = -INFINITY;
    for (int j = lane; j < B; j += 64) { mr = fmaxf(mr, w.sim[i * B + j]); mc = fmaxf(mc, w.sim[j * B + i]); }
    mr = wave_max(mr); mc = wave_max(mc);
    float sr = 0.f, sc = 0.f;
    for (int j = lane; j < B; j += 64) { sr += expf((w.sim[i * B + j] - mr) / TAU); sc += expf((w.sim[j * B + i] - mc) / TAU); }
    sr = wave_sum(sr); sc = wave_sum(sc);
    if (lane == 0) {
      const float lr = mr / TAU + logf(sr), lc = mc / TAU + logf(sc);
      w.lse_r[i] = lr; w.lse_c[i] = lc;
      const int p = (int)a.pos_idx[i];
      const float sd = w.sim[i * B + i] / TAU;
      inter = -(sd - lr) - (sd - lc);
      const float z = zval(a, w, i, p) / TAU;
      intra = -(z - w.zr[i]) - (z - w.zc[p]);
    }
  }
  inter = block_sum(inter, red);
  intra = block_sum(intra, red);
  if (tid == 0) {
    if (sal_on) { atomicAdd(w.acc + 6, inter); atomicAdd(w.acc + 7, intra); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_ticket = __hip_atomic_fetch_add((unsigned*)(w.acc + 8), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (s_ticket != gridDim.x - 1) return;
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = __hip_atomic_load(w.acc + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float nwin = v[0], nval = v[1];
    a.losses[0] = a.do_spans ? v[3] / nwin : 0.f;
    a.losses[1] = a.do_spans ? v[4] / nwin : 0.f;
    a.losses[2] = a.do_labels ? v[5] / nval : 0.f;
    a.losses[3] = sal_on ? v[6] / (float)B : 0.f;
    a.losses[4] = sal_on ? v[7] / (float)B : 0.f;
    a.losses[5] = sal_on ? 1.f : 0.f;
    a.losses[6] = nwin;
    a.losses[7] = nval;
  }
}

// ------------------------------------------------------------------------------------------------
// gradients
// ------------------------------------------------------------------------------------------------
// launch 1 of the backward, by block role: [0, dvq blocks) the two small products dvh = dsim qhat, dqh = dsim^T vhat with
// dsim[i][j] = go_inter / (B tau) (softmax_row + softmax_col - 2 delta) evaluated on the fly while the coefficients are staged (sim stays intact:
// no separate dsim pass); then the per-clip gradients (spans, labels, d/d cosine), 256 clips per block.
__device__ __forceinline__ float dsim_at(const LossArgs& a, const WS& w, int i, int j) {
  const float sc = w.sim[i * a.B + j] / TAU, dl = (i == j) ? 1.f : 0.f;
  return a.go[3] / ((float)a.B * TAU) * ((expf(sc - w.lse_r[i]) - dl) + (expf(sc - w.lse_c[j]) - dl));
}
__device__ __forceinline__ void loss_grad_clip(const LossArgs& a, const WS& w, int i) {
  const float nwin = a.losses[6], nval = a.losses[7];
  const bool sal_on = a.losses[5] != 0.f;
  const int b = i / a.Lv, t = i % a.Lv;
  const float win = a.ts_window[i], msk = a.ts_mask[i];
  float gs0 = 0.f, gs1 = 0.f, gl = 0.f;
  if (a.do_spans && win != 0.f) {
    const float s0 = a.timestamp[2 * i] + a.pred_spans[2 * i], s1 = a.timestamp[2 * i + 1] + a.pred_spans[2 * i + 1];
    const float g0 = a.span_nn[2 * i], g1 = a.span_nn[2 * i + 1];
    const float e0 = s0 - g0, e1 = s1 - g1;
    gs0 = a.go[0] * win * fminf(fmaxf(e0, -1.f), 1.f) / nwin;
    gs1 = a.go[0] * win * fminf(fmaxf(e1, -1.f), 1.f) / nwin;
    float I, U, H;
    giou_terms(s0, s1, g0, g1, I, U, H);
    const float dI0 = (I > 0.f && s0 > g0) ? -1.f : ((I > 0.f && s0 == g0) ? -0.5f : 0.f);
    const float dI1 = (I > 0.f && s1 < g1) ? 1.f : ((I > 0.f && s1 == g1) ? 0.5f : 0.f);
    const float dU0 = -1.f - dI0, dU1 = 1.f - dI1;
    const float dH0 = (H > 0.f && s0 < g0) ? -1.f : ((H > 0.f && s0 == g0) ? -0.5f : 0.f);
    const float dH1 = (H > 0.f && s1 > g1) ? 1.f : ((H > 0.f && s1 == g1) ? 0.5f : 0.f);
    const float dg0 = (dI0 * U - I * dU0) / (U * U) + (dU0 * H - U * dH0) / (H * H);
    const float dg1 = (dI1 * U - I * dU1) / (U * U) + (dU1 * H - U * dH1) / (H * H);
    gs0 += -a.go[1] * dg0 / nwin;
    gs1 += -a.go[1] * dg1 / nwin;
  }
  if (a.do_labels && msk != 0.f) {
    const float p = a.pred_logits[i], y = win != 0.f ? 1.f : 0.f, wt = win != 0.f ? 1.f : a.eos_coef;
    gl = a.go[2] * wt * (p - y) / fmaxf((1.f - p) * p, 1e-12f) / nval;
  }
  a.g_spans[2 * i] = gs0; a.g_spans[2 * i + 1] = gs1; a.g_logits[i] = gl;
  float gz = 0.f;
  if (sal_on) {
    const int p = (int)a.pos_idx[b];
    const float z = zval(a, w, b, t) / TAU;
    const float dl = (t == p) ? 1.f : 0.f;
    gz = a.go[4] / ((float)a.B * TAU) * ((expf(z - w.zr[b]) - dl) + (w.cnt[t] * expf(z - w.zc[t]) - dl));
  }
  a.g_cos[i] = gz;
}

// dvh[b][c] = sum_j dsim[b][j] qhat_j[c]   (side 0),   dqh[b][c] = sum_i dsim[i][b] vhat_i[c]   (side 1)
// dvq block = 8 output rows x 256 columns: 64 column groups of 4 (16-byte loads) x 4 j-lanes that split the
// reduction; every source row loaded serves 8 outputs.
__global__ __launch_bounds__(256) void loss_bwd1_kernel(const LossArgs a, int dvq_blocks, int have_sal) {
  const WS w(a);
  const int d = a.d, B = a.B;
  __shared__ float coef[8][260];
  __shared__ float part[4][8][256];
  if ((int)blockIdx.x >= dvq_blocks) {
    const int i = ((int)blockIdx.x - dvq_blocks) * 256 + threadIdx.x;
    if (i < B * a.Lv) loss_grad_clip(a, w, i);
    return;
  }
  if (!have_sal || a.losses[5] == 0.f) return;
  const int nz = (d + 255) / 256;
  const int bx = (int)blockIdx.x / (2 * nz), side = ((int)blockIdx.x / nz) & 1, bz = (int)blockIdx.x % nz;
  const int b0 = bx * 8, c0 = bz * 256;
  const int cg = threadIdx.x & 63, jl = threadIdx.x >> 6;
  const int c = c0 + cg * 4;
  float acc[8][4];
#pragma unroll
  for (int r = 0; r < 8; r++)
#pragma unroll
    for (int e = 0; e < 4; e++) acc[r][e] = 0.f;
  for (int j0 = 0; j0 < B; j0 += 256) {
    const int nj = min(256, B - j0);
    __syncthreads();
    for (int idx = threadIdx.x; idx < 8 * nj; idx += 256) {
      const int r = idx / nj, j = idx % nj, jj = j0 + j, bb = min(b0 + r, B - 1);
      coef[r][j] = side == 0 ? dsim_at(a, w, bb, jj) / fmaxf(w.qnorm[jj], EPS)
                             : dsim_at(a, w, jj, bb) / fmaxf(w.vnorm[jj * a.Lv + (int)a.pos_idx[jj]], EPS);
    }
    __syncthreads();
    if (c < d) {
#pragma unroll 4
      for (int j = jl; j < nj; j += 4) {
        const int jj = j0 + j;
        const float* src = side == 0 ? a.txt + (size_t)jj * d : vrow(a, jj, (int)a.pos_idx[jj]);
        const f32x4 x = *(const f32x4*)(src + c);
#pragma unroll
        for (int r = 0; r < 8; r++) {
          const float cf = coef[r][j];
          acc[r][0] += cf * x[0]; acc[r][1] += cf * x[1]; acc[r][2] += cf * x[2]; acc[r][3] += cf * x[3];
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 8; r++)
#pragma unroll
    for (int e = 0; e < 4; e++) part[jl][r][cg * 4 + e] = acc[r][e];
  __syncthreads();
  for (int idx = threadIdx.x; idx < 8 * 256; idx += 256) {
    const int r = idx >> 8, cc = idx & 255;
    if (b0 + r < B && c0 + cc < d)
      (side == 0 ? w.dvh : w.dqh)[(size_t)(b0 + r) * d + c0 + cc] = part[0][r][cc] + part[1][r][cc] + part[2][r][cc] + part[3][r][cc];
  }
}
// inter-video gradients in input space: g_vrow[b] = (dvh - vhat (vhat.dvh)) / |v_pos|, g_txt[b] = (dqh - qhat (qhat.dqh)) / |q|
__global__ __launch_bounds__(256) void loss_grad_rows_kernel(const LossArgs a) {
  __shared__ float red[8];
  const WS w(a);
  const int b = blockIdx.x, d = a.d, tid = threadIdx.x;
  float* gv = a.g_vrow + (size_t)b * d;
  float* gt = a.g_txt + (size_t)b * d;
  if (a.losses[5] == 0.f) { for (int c = tid; c < d; c += 256) { gv[c] = 0.f; gt[c] = 0.f; } return; }
  const int p = (int)a.pos_idx[b];
  const float* v = vrow(a, b, p);
  const float* q = a.txt + (size_t)b * d;
  const float vn = fmaxf(w.vnorm[b * a.Lv + p], EPS), qn = fmaxf(w.qnorm[b], EPS);
  float dv = 0.f, dq = 0.f;
  for (int c = tid; c < d; c += 256) { dv += w.dvh[(size_t)b * d + c] * v[c] / vn; dq += w.dqh[(size_t)b * d + c] * q[c] / qn; }
  dv = block_sum(dv, red); dq = block_sum(dq, red);
  for (int c = tid; c < d; c += 256) {
    gv[c] = (w.dvh[(size_t)b * d + c] - v[c] / vn * dv) / vn;
    gt[c] = (w.dqh[(size_t)b * d + c] - q[c] / qn * dq) / qn;
  }
}
// dense expansion for generic autograd callers:
// g_vid[b, t, :] = gz * (qhat - cos vhat) / |v|  (+ g_vrow on the positive clip row)
__global__ __launch_bounds__(256) void loss_expand_vid_kernel(const LossArgs a) {
  const WS w(a);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.B * a.Lv) return;
  const int b = row / a.Lv, t = row % a.Lv, d = a.d;
  float* out = a.g_vid + (size_t)row * d;
  if (a.losses[5] == 0.f) { for (int c = lane; c < d; c += 64) out[c] = 0.f; return; }
  const float* v = vrow(a, b, t);
  const float* q = a.txt + (size_t)b * d;
  const float vn = fmaxf(w.vnorm[row], EPS), qn = fmaxf(w.qnorm[b], EPS), cs = w.cosv[row], gz = a.g_cos[row];
  const bool is_pos = (t == (int)a.pos_idx[b]);
  for (int c = lane; c < d; c += 64) {
    float g = gz * (q[c] / qn - cs * v[c] / vn) / vn;
    if (is_pos) g += a.g_vrow[(size_t)b * d + c];
    out[c] = g;
  }
}
// g_txt[b, :] += sum_t gz (vhat_t - cos_t qhat) / |q|     (thread per column, loop over clips: no atomics)
__global__ __launch_bounds__(256) void loss_expand_txt_kernel(const LossArgs a) {
  const WS w(a);
  const int b = blockIdx.x, d = a.d;
  if (a.losses[5] == 0.f) return;
  const float* q = a.txt + (size_t)b * d;
  const float qn = fmaxf(w.qnorm[b], EPS);
  for (int c = threadIdx.x; c < d; c += 256) {
    const float qh = q[c] / qn;
    float g = 0.f;
    for (int t = 0; t < a.Lv; t++) {
      const float gz = a.g_cos[b * a.Lv + t];
      if (gz != 0.f) g += gz * (vrow(a, b, t)[c] / fmaxf(w.vnorm[b * a.Lv + t], EPS) - w.cosv[b * a.Lv + t] * qh);
    }
    a.g_txt[(size_t)b * d + c] += g / qn;
  }
}

}  // namespace

long long loss_ws_floats(int B, int Lv, int d) { return 3LL * B * Lv + (long long)B * B + 5LL * B + 2LL * Lv + 2LL * B * d + 96; }

// ------------------------------------------------------------------------------------------------------------------------------------
// Class term of the 'saliency_cls' loss (model/univtg.py:314-324, the TAL pre-training branch): with v_b = vid_mem_proj[b, pos_b] and the
// pooled class-name features c_j (cls_mem_proj), z_bj = cos(v_b, c_j) / 0.07 (sim_matrix: norms clamped at 1e-8),
// loss = - sum_{(b,j): cls_idx[b,j]} log_softmax_j(z_b)[j] / #{(b,j): cls_idx[b,j]}.
// One block per sample forward (cosines, log-sum-exp, the sample's partial), a one-block fold, and two gather-free backward kernels (block
// per sample: d loss / d v_b; block per class: d loss / d c_j) -- a few hundred samples x classes x d, nowhere near a bound.
// ws: cosine [B, C], lse [B], vnorm [B], cnorm [C], part [B] (sum of picked log-probabilities), npick [B], tot [2] (picked sum, count).
// ------------------------------------------------------------------------------------------------------------------------------------
namespace {
struct ClsWS {
  float *cosm, *lse, *vn, *cn, *part, *npick, *tot;
  __host__ __device__ ClsWS(const ClsNceArgs& a) {
    float* p = a.ws;
    cosm = p; p += (size_t)a.B * a.C;
    lse = p; p += a.B; vn = p; p += a.B; cn = p; p += a.C; part = p; p += a.B; npick = p; p += a.B; tot = p;
  }
};
__device__ __forceinline__ float block_sum256(float v, float* red) {      // 256 threads; red: 4 floats of LDS; every thread gets the sum
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void cls_nce_fwd_kernel(const ClsNceArgs a) {
  extern __shared__ float sm[];                 // z_bj for the block's sample [C], then 4 reduction slots
  float* red = sm + a.C;
  const ClsWS w(a);
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* v = a.vid + (size_t)b * a.vid_sb + (size_t)a.pos_idx[b] * a.vid_st;
  float vv = 0.f;
  for (int c = tid; c < a.d; c += 256) vv += v[c] * v[c];
  const float vnorm = fmaxf(sqrtf(block_sum256(vv, red)), EPS);
  for (int j = 0; j < a.C; j++) {
    const float* cj = a.cls + (size_t)j * a.d;
    float dot = 0.f, cc = 0.f;
    for (int c = tid; c < a.d; c += 256) { const float x = cj[c]; dot += v[c] * x; cc += x * x; }
    dot = block_sum256(dot, red);
    cc = block_sum256(cc, red);
    const float cnorm = fmaxf(sqrtf(cc), EPS);
    if (tid == 0) { const float cs = dot / (vnorm * cnorm); sm[j] = cs / TAU; w.cosm[(size_t)b * a.C + j] = cs; if (b == 0) w.cn[j] = cnorm; }
  }
  __syncthreads();
  if (tid == 0) {                               // C is small (class names): one lane walks the row
    float m = -3.0e38f;
    for (int j = 0; j < a.C; j++) m = fmaxf(m, sm[j]);
    float se = 0.f;
    for (int j = 0; j < a.C; j++) se += __expf(sm[j] - m);
    const float lse = m + __logf(se);
    float part = 0.f, np = 0.f;
    for (int j = 0; j < a.C; j++) if (a.cls_idx[(size_t)b * a.C + j] != 0.f) { part += sm[j] - lse; np += 1.f; }
    w.lse[b] = lse; w.vn[b] = vnorm; w.part[b] = part; w.npick[b] = np;
  }
}
__global__ __launch_bounds__(256) void cls_nce_fold_kernel(const ClsNceArgs a) {
  __shared__ float red[4];
  const ClsWS w(a);
  float ps = 0.f, pn = 0.f;
  for (int b = threadIdx.x; b < a.B; b += 256) { ps += w.part[b]; pn += w.npick[b]; }
  ps = block_sum256(ps, red);
  pn = block_sum256(pn, red);
  if (threadIdx.x == 0) {
    w.tot[0] = ps; w.tot[1] = pn;
    const float act = a.active ? *a.active : 1.f;
    a.loss[0] = (act != 0.f && pn > 0.f) ? -ps / pn : 0.f;
  }
}
// d loss / d z_bj = -(cls_idx_bj - npick_b softmax_bj) / N; z = cos / TAU; d cos / d v = (c^ - cos v^) / |v| (and symmetrically for c)
__device__ __forceinline__ float cls_coef(const ClsNceArgs& a, const ClsWS& w, int b, int j, float scale) {
  const float cs = w.cosm[(size_t)b * a.C + j];
  const float p = __expf(cs / TAU - w.lse[b]);
  return -scale * (a.cls_idx[(size_t)b * a.C + j] - w.npick[b] * p) / TAU;
}
__global__ __launch_bounds__(256) void cls_nce_bwd_v_kernel(const ClsNceArgs a) {
  const ClsWS w(a);
  const int b = blockIdx.x, tid = threadIdx.x;
  const float act = a.active ? *a.active : 1.f;
  const float scale = (act != 0.f && w.tot[1] > 0.f) ? a.go[0] / w.tot[1] : 0.f;
  const float* v = a.vid + (size_t)b * a.vid_sb + (size_t)a.pos_idx[b] * a.vid_st;
  float* gv = a.g_vid + (size_t)b * a.gv_sb + (size_t)a.pos_idx[b] * a.gv_st;
  const float vn = w.vn[b];
  for (int c = tid; c < a.d; c += 256) {
    float acc = 0.f;
    const float vh = v[c] / vn;
    for (int j = 0; j < a.C; j++) {
      const float g = cls_coef(a, w, b, j, scale);
      acc += g * (a.cls[(size_t)j * a.d + c] / w.cn[j] - w.cosm[(size_t)b * a.C + j] * vh);
    }
    gv[c] = acc / vn;
  }
}
__global__ __launch_bounds__(256) void cls_nce_bwd_c_kernel(const ClsNceArgs a) {
  const ClsWS w(a);
  const int j = blockIdx.x, tid = threadIdx.x;
  const float act = a.active ? *a.active : 1.f;
  const float scale = (act != 0.f && w.tot[1] > 0.f) ? a.go[0] / w.tot[1] : 0.f;
  const float cn = w.cn[j];
  for (int c = tid; c < a.d; c += 256) {
    float acc = 0.f;
    const float ch = a.cls[(size_t)j * a.d + c] / cn;
    for (int b = 0; b < a.B; b++) {
      const float g = cls_coef(a, w, b, j, scale);
      const float* v = a.vid + (size_t)b * a.vid_sb + (size_t)a.pos_idx[b] * a.vid_st;
      acc += g * (v[c] / w.vn[b] - w.cosm[(size_t)b * a.C + j] * ch);
    }
    a.g_cls[(size_t)j * a.d + c] = acc / cn;
  }
}
}  // namespace
long long cls_nce_ws_floats(int B, int C) { return (long long)B * C + 4LL * B + C + 8; }
int launch_cls_nce_fwd(const ClsNceArgs& a, hipStream_t s) {
  if (a.B <= 0 || a.C <= 0 || a.d <= 0 || a.C > 8192) return -11;
  hipLaunchKernelGGL(cls_nce_fwd_kernel, dim3(a.B), dim3(256), (size_t)(a.C + 4) * sizeof(float), s, a);
  hipLaunchKernelGGL(cls_nce_fold_kernel, dim3(1), dim3(256), 0, s, a);
  UVTG_CHECK_LAUNCH();
  return 0;
}
int launch_cls_nce_bwd(const ClsNceArgs& a, hipStream_t s) {
  if (a.B <= 0 || a.C <= 0 || a.d <= 0) return -11;
  hipLaunchKernelGGL(cls_nce_bwd_v_kernel, dim3(a.B), dim3(256), 0, s, a);
  hipLaunchKernelGGL(cls_nce_bwd_c_kernel, dim3(a.C), dim3(256), 0, s, a);
  UVTG_CHECK_LAUNCH();
  return 0;
}

int launch_losses_fwd(const LossArgs& a, hipStream_t s) {
  const int n = a.B * a.Lv;
  const bool have_sal = a.do_saliency && a.sal_tgt && a.pos_idx;
  const WS w(a);
  if (hipError_t e = hipMemsetAsync(w.acc, 0, (16 + (size_t)a.Lv) * sizeof(float), s)) return (int)e;      // accumulators + cnt
  if (have_sal && !a.cos_c) hipLaunchKernelGGL(loss_stats_kernel, dim3(cdiv(n, 4)), dim3(256), 0, s, a);
  const int sim_blocks = have_sal ? a.B * cdiv(a.B, 64) : 0, elem_blocks = cdiv(n, 256), lse_blocks = have_sal ? cdiv(a.B + a.Lv, 4) : 0;
  hipLaunchKernelGGL(loss_fwd1_kernel, dim3(sim_blocks + elem_blocks + lse_blocks), dim3(256), a.d * sizeof(float), s, a, sim_blocks, elem_blocks,
                     have_sal ? 1 : 0);
  hipLaunchKernelGGL(loss_fwd2_kernel, dim3(have_sal ? cdiv(a.B, 4) : 1), dim3(256), 0, s, a, have_sal ? 1 : 0);
  UVTG_CHECK_LAUNCH();
  return 0;
}
int launch_losses_bwd(const LossArgs& a, hipStream_t s) {
  const int n = a.B * a.Lv;
  const bool have_sal = a.do_saliency && a.sal_tgt && a.pos_idx;
  const int dvq_blocks = have_sal ? cdiv(a.B, 8) * 2 * cdiv(a.d, 256) : 0;
  hipLaunchKernelGGL(loss_bwd1_kernel, dim3(dvq_blocks + cdiv(n, 256)), dim3(256), 0, s, a, dvq_blocks, have_sal ? 1 : 0);
  if (have_sal) {
    hipLaunchKernelGGL(loss_grad_rows_kernel, dim3(a.B), dim3(256), 0, s, a);
    if (a.g_vid) {     // dense mode: full gradients wrt vid_mem_proj / txt_mem_proj
      hipLaunchKernelGGL(loss_expand_vid_kernel, dim3(cdiv(n, 4)), dim3(256), 0, s, a);
      hipLaunchKernelGGL(loss_expand_txt_kernel, dim3(a.B), dim3(256), 0, s, a);
    }
  }
  UVTG_CHECK_LAUNCH();
  return 0;
}
