// Moment-DETR set criterion on device (model/moment_detr.py:166-365, span_loss_type == "l1"): the losses of one decoder
// layer's outputs given the matcher's index pairs, and their gradients with respect to every prediction tensor.
//   loss_b  = mean |src_cxw - tgt_cxw| over the matched pairs                         (:205-213,230)
//   loss_g  = mean (1 - gIoU(xx(src), xx(tgt)))                                       (:214, utils/span_utils.py:93-122)
//   loss_f  = mean_{b,q} w[cls] * CE(logits, cls),  cls = 0 if matched else 1, w = (1, eos_coef)   (:234-248)
//   class_error = 100 - 100 * #(matched and argmax == 0) / #matched                   (:252, :16-32)
//   loss_s_intra = 2 * sum max(0, margin + s[neg] - s[pos]) / (B * n_pairs)           (:255-270)
//   loss_contrastive_align = mean_b ( -sum_matched l / #matched + logsumexp_q l ),  l[b,q] = <proj_q[b,q], sum_n proj_txt[b,n]> / T   (:272-290)
// One workgroup per sample writes that sample's partial sums; a single wave adds them in sample order, so the six numbers are
// reproducible run to run.  All of it is O(B * (Q + T) * D) memory-bound work -- nothing here touches MFMA.
#include "uvtg_kernels.h"
#include "../../include/uvtg.h"

namespace {

struct DetrArgs {
  const float* logits; const float* spans; const float* tgt; const int* tgt_off;
  const long long* m_pred; const long long* m_tgt; const int* n_match; int max_t;
  const float* sal; const long long* pos; const long long* neg; int n_pairs; int L;
  const float* pq; const float* pt; int T; int D;
  float eos, temp, margin;
  const float* go;
  float* part;      // [B, 8]
  float* d_logits; float* d_spans; float* d_sal; float* d_pq; float* d_pt;
  int B, Q;
};

__device__ __forceinline__ float block_sum(float v, float* red) {   // 256 threads
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void detr_criterion_kernel(DetrArgs a) {
  extern __shared__ float sm[];
  __shared__ float red[4];
  __shared__ int s_tj[256];
  const int b = blockIdx.x, tid = threadIdx.x, Q = a.Q;
  float* tsum = sm;            // [D]
  float* lq = sm + a.D;        // [Q] contrastive logits, then d logit
  // total number of matched pairs (the normaliser of the span losses)
  float cnt = 0.f;
  for (int i = tid; i < a.B; i += 256) cnt += (float)max(a.n_match[i], 0);
  const float N = block_sum(cnt, red);
  const int nm = max(a.n_match[b], 0);
  for (int q = tid; q < Q; q += 256) s_tj[q] = -1;
  __syncthreads();
  for (int k = tid; k < nm; k += 256) s_tj[(int)a.m_pred[(size_t)b * a.max_t + k]] = (int)a.m_tgt[(size_t)b * a.max_t + k];
  __syncthreads();
  const float g_b = a.go ? a.go[0] : 0.f, g_g = a.go ? a.go[1] : 0.f, g_f = a.go ? a.go[2] : 0.f;
  const float g_s = a.go ? a.go[4] : 0.f, g_c = a.go ? a.go[5] : 0.f;

  float l1 = 0.f, gl = 0.f, ce = 0.f, correct = 0.f;
  for (int q = tid; q < Q; q += 256) {
    const size_t i = (size_t)b * Q + q;
    const int tj = s_tj[q];
    const float l0 = a.logits[2 * i], l1v = a.logits[2 * i + 1];
    const float mx = fmaxf(l0, l1v), lse = mx + logf(expf(l0 - mx) + expf(l1v - mx));
    const int cls = tj >= 0 ? 0 : 1;
    const float w = cls ? a.eos : 1.f;
    ce += -w * ((cls ? l1v : l0) - lse);
    if (a.go && a.d_logits) {
      const float s = g_f * w / (float)(a.B * Q), p0 = expf(l0 - lse), p1 = expf(l1v - lse);
      a.d_logits[2 * i] = s * (p0 - (cls == 0));
      a.d_logits[2 * i + 1] = s * (p1 - (cls == 1));
    }
    float dc = 0.f, dw = 0.f;
    if (tj >= 0) {
      correct += l0 >= l1v;
      const float c = a.spans[2 * i], wd = a.spans[2 * i + 1];
      const float* t = a.tgt + 2 * (size_t)(a.tgt_off[b] + tj);
      const float tc = t[0], tw = t[1];
      l1 += fabsf(c - tc) + fabsf(wd - tw);
      const float x1 = c - 0.5f * wd, x2 = c + 0.5f * wd, y1 = tc - 0.5f * tw, y2 = tc + 0.5f * tw;
      const float left = fmaxf(x1, y1), right = fminf(x2, y2);
      const float inter = fmaxf(right - left, 0.f);
      const float uni = (x2 - x1) + (y2 - y1) - inter;
      const float enc = fmaxf(fmaxf(x2, y2) - fminf(x1, y1), 0.f);
      const float giou = inter / uni - (enc - uni) / enc;
      gl += 1.f - giou;
      if (a.go) {
        // d giou / d (x1, x2): inter, union and the hull are piecewise linear in the endpoints
        const float act = (right - left) >= 0.f ? 1.f : 0.f;
        const float di1 = -act * (x1 > y1 ? 1.f : (x1 == y1 ? 0.5f : 0.f)), di2 = act * (x2 < y2 ? 1.f : (x2 == y2 ? 0.5f : 0.f));
        const float du1 = -1.f - di1, du2 = 1.f - di2;
        const float ea = (fmaxf(x2, y2) - fminf(x1, y1)) >= 0.f ? 1.f : 0.f;
        const float de1 = -ea * (x1 < y1 ? 1.f : (x1 == y1 ? 0.5f : 0.f)), de2 = ea * (x2 > y2 ? 1.f : (x2 == y2 ? 0.5f : 0.f));
        // giou = inter/uni - 1 + uni/enc
        const float dg1 = (di1 * uni - inter * du1) / (uni * uni) + (du1 * enc - uni * de1) / (enc * enc);
        const float dg2 = (di2 * uni - inter * du2) / (uni * uni) + (du2 * enc - uni * de2) / (enc * enc);
        const float sg = -g_g / N;
        const float sgn_c = (c > tc) - (c < tc), sgn_w = (wd > tw) - (wd < tw);
        dc = g_b * sgn_c / (2.f * N) + sg * (dg1 + dg2);
        dw = g_b * sgn_w / (2.f * N) + sg * 0.5f * (dg2 - dg1);
      }
    }
    if (a.go && a.d_spans) { a.d_spans[2 * i] = dc; a.d_spans[2 * i + 1] = dw; }
  }
  l1 = block_sum(l1, red); gl = block_sum(gl, red); ce = block_sum(ce, red); correct = block_sum(correct, red);

  // saliency hinge (indices may repeat: one thread walks the pairs)
  float hinge = 0.f;
  if (a.sal) {
    if (a.go && a.d_sal) {
      for (int t = tid; t < a.L; t += 256) a.d_sal[(size_t)b * a.L + t] = 0.f;
      __syncthreads();
    }
    if (tid == 0) {
      const float gs = g_s * 2.f / (float)(a.B * a.n_pairs);
      for (int p = 0; p < a.n_pairs; p++) {
        const int ip = (int)a.pos[(size_t)b * a.n_pairs + p], in = (int)a.neg[(size_t)b * a.n_pairs + p];
        const float h = a.margin + a.sal[(size_t)b * a.L + in] - a.sal[(size_t)b * a.L + ip];
        if (h >= 0.f) {
          hinge += h;
          if (a.go && a.d_sal) { a.d_sal[(size_t)b * a.L + in] += gs; a.d_sal[(size_t)b * a.L + ip] -= gs; }
        }
      }
    }
  }

  // contrastive alignment
  float nce = 0.f;
  if (a.pq) {
    const int D = a.D, T = a.T;
    for (int d = tid; d < D; d += 256) {
      float s = 0.f;
      for (int n = 0; n < T; n++) s += a.pt[((size_t)b * T + n) * D + d];
      tsum[d] = s;
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    for (int q = wave; q < Q; q += 4) {
      float s = 0.f;
      for (int d = lane; d < D; d += 64) s += a.pq[((size_t)b * Q + q) * D + d] * tsum[d];
      for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
      if (lane == 0) lq[q] = s / a.temp;
    }
    __syncthreads();
    float mx = -INFINITY, pos = 0.f;
    for (int q = 0; q < Q; q++) { mx = fmaxf(mx, lq[q]); if (s_tj[q] >= 0) pos += lq[q]; }
    float se = 0.f;
    for (int q = 0; q < Q; q++) se += expf(lq[q] - mx);
    const float lse = mx + logf(se);
    nce = -pos / (float)nm + lse;
    __syncthreads();
    if (a.go) {
      for (int q = tid; q < Q; q += 256)
        lq[q] = g_c / (float)a.B * (expf(lq[q] - lse) - (s_tj[q] >= 0 ? 1.f / (float)nm : 0.f)) / a.temp;
      __syncthreads();
      if (a.d_pq)
        for (int i = tid; i < Q * D; i += 256) a.d_pq[(size_t)b * Q * D + i] = lq[i / D] * tsum[i % D];
      if (a.d_pt)
        for (int d = tid; d < D; d += 256) {
          float s = 0.f;
          for (int q = 0; q < Q; q++) s += lq[q] * a.pq[((size_t)b * Q + q) * D + d];
          for (int n = 0; n < T; n++) a.d_pt[((size_t)b * T + n) * D + d] = s;
        }
    }
  }
  if (tid == 0) {
    float* p = a.part + (size_t)b * 8;
    p[0] = l1; p[1] = gl; p[2] = ce; p[3] = correct; p[4] = hinge; p[5] = nce; p[6] = (float)nm; p[7] = 0.f;
  }
}

__global__ __launch_bounds__(64) void detr_reduce_kernel(const float* part, int B, int Q, int n_pairs, int has_sal, int has_nce,
                                                         float* losses) {
  if (threadIdx.x >= 7) return;
  float s = 0.f;
  for (int b = 0; b < B; b++) s += part[(size_t)b * 8 + threadIdx.x];
  __shared__ float t[8];
  t[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float N = t[6];
    losses[0] = t[0] / (2.f * N);
    losses[1] = t[1] / N;
    losses[2] = t[2] / (float)(B * Q);
    losses[3] = 100.f - t[3] * (100.f / N);
    losses[4] = has_sal ? t[4] / (float)(B * n_pairs) * 2.f : 0.f;
    losses[5] = has_nce ? t[5] / (float)B : 0.f;
  }
}

}  // namespace

extern "C" int uvtg_detr_criterion(const float* pred_logits, const float* pred_spans_cxw, int B, int Q, const float* tgt_cxw,
                                   const int* tgt_off, const long long* match_pred, const long long* match_tgt, const int* n_match,
                                   int max_t, const float* saliency_scores, const long long* pos_idx, const long long* neg_idx,
                                   int n_pairs, int L, const float* proj_queries, const float* proj_txt_mem, int T, int D,
                                   float eos_coef, float temperature, float saliency_margin, const float* go, float* partials,
                                   float* losses, float* d_logits, float* d_spans, float* d_saliency, float* d_proj_queries,
                                   float* d_proj_txt_mem, uvtg_stream_t stream) {
  if (!pred_logits || !pred_spans_cxw || !tgt_cxw || !tgt_off || !match_pred || !match_tgt || !n_match || !partials || !losses)
    return -20;
  if (B <= 0 || Q <= 0 || Q > 256 || max_t <= 0) return -11;
  if (saliency_scores && (!pos_idx || !neg_idx || n_pairs <= 0 || L <= 0)) return -11;
  if (proj_queries && (!proj_txt_mem || T <= 0 || D <= 0 || temperature == 0.f)) return -11;
  DetrArgs a{pred_logits, pred_spans_cxw, tgt_cxw, tgt_off, match_pred, match_tgt, n_match, max_t, saliency_scores, pos_idx, neg_idx,
             n_pairs, L, proj_queries, proj_txt_mem, T, proj_queries ? D : 0, eos_coef, temperature, saliency_margin, go, partials,
             d_logits, d_spans, d_saliency, d_proj_queries, d_proj_txt_mem, B, Q};
  hipStream_t s = (hipStream_t)stream;
  const size_t smem = (size_t)(a.D + Q) * sizeof(float);
  if (smem > 60 * 1024) return -11;
  hipLaunchKernelGGL(detr_criterion_kernel, dim3(B), dim3(256), smem, s, a);
  hipLaunchKernelGGL(detr_reduce_kernel, dim3(1), dim3(64), 0, s, partials, B, Q, n_pairs, saliency_scores != nullptr,
                     proj_queries != nullptr, losses);
  UVTG_CHECK_LAUNCH();
  return 0;
}
