// Index-producing tail of the hot path, on device:
//  * span decode + score masking + stable ranking + greedy hull-IoU temporal NMS
//    (main/inference_mr.py:109-160, utils/temporal_nms.py:6-74) -- one wave per sample;
//  * Hungarian matcher (model/matcher.py:36-100): cost matrix + per-sample rectangular LSAP
//    (shortest augmenting path, Crouse 2016 = scipy.optimize.linear_sum_assignment) -- one thread per sample.
#include "uvtg_kernels.h"
#include "../../include/uvtg.h"

namespace {

__device__ __forceinline__ double round4(float x) { return rint((double)x * 1e4) / 1e4; }   // float(f"{x:.4f}")

// clip_length > 0: the rows additionally go through PostProcessorDETR's round_multiple (eval/postprocessing.py:26-37,46-51), which
// the reference applies to the 4-decimal rows BEFORE the NMS (main/inference_mr.py:183-192 inside compute_mr_results, the NMS in
// eval_epoch_post_processing): torch.tensor(rows) -> fp32, torch.round(w / clip) * clip (half-to-even), score re-rounded to 4 decimals.
// saliency (optional): pred_saliency_scores of main/inference_mr.py:124-136 = fp16(saliency) [+ prob when eval_mode == 'add'], fp32 out.
__global__ __launch_bounds__(64) void decode_rank_nms_kernel(const float* pred_logits, const float* pred_spans,
    const float* timestamp, const float* ts_mask, const float* durations, int B, int Lv, double nms_thd,
    int max_before, int max_after, float clip_length, const float* saliency, int sal_add, float* saliency_out,
    double* windows_out, int* order, int* keep, int* n_keep) {
  extern __shared__ unsigned char smem[];
  const int b = blockIdx.x, lane = threadIdx.x;
  float* sc = (float*)smem;                         // [Lv] masked scores
  double* st = (double*)(sc + ((Lv + 1) & ~1));     // [Lv] ranked, rounded
  double* ed = st + Lv;
  unsigned char* alive = (unsigned char*)(ed + Lv); // [Lv]
  for (int t = lane; t < Lv; t += 64) {
    const int i = b * Lv + t;
    sc[t] = ts_mask[i] != 0.f ? pred_logits[i] : 0.f;                      // scores[~mask] = 0  (:118-119)
  }
  __syncthreads();
  const float dur = durations[b];
  for (int t = lane; t < Lv; t += 64) {
    // stable descending rank: position of clip t in sorted(..., key=score, reverse=True)  (:158)
    const float s = sc[t];
    int rank = 0;
    for (int j = 0; j < Lv; j++) { const float sj = sc[j]; rank += (sj > s) || (sj == s && j < t); }
    const int i = b * Lv + t;
    float w0 = (timestamp[2 * i] + pred_spans[2 * i]) * dur, w1 = (timestamp[2 * i + 1] + pred_spans[2 * i + 1]) * dur;
    w0 = fminf(fmaxf(w0, 0.f), dur); w1 = fminf(fmaxf(w1, 0.f), dur);     // clamp  (:152-153)
    double r0 = round4(w0), r1 = round4(w1), rs = round4(s);               // 4-decimal rounding  (:159)
    if (clip_length > 0.f) {                                                // round_multiple on the fp32 image of the rounded rows
      r0 = (double)(rintf(__fdiv_rn((float)r0, clip_length)) * clip_length);
      r1 = (double)(rintf(__fdiv_rn((float)r1, clip_length)) * clip_length);
      rs = round4((float)rs);
    }
    if (saliency_out) {
      const float h = (float)(_Float16)saliency[i];                        // .half(): round-to-nearest-even
      saliency_out[i] = sal_add ? h + pred_logits[i] : h;                  // half + float promotes to float  (:124-128)
    }
    order[b * Lv + rank] = t;
    st[rank] = r0; ed[rank] = r1;
    double* wo = windows_out + ((size_t)b * Lv + rank) * 3;
    wo[0] = r0; wo[1] = r1; wo[2] = rs;
  }
  __syncthreads();
  // greedy NMS over the first max_before ranked rows (utils/temporal_nms.py:25-74)
  const int n = min(Lv, max_before);
  int* kp = keep + (size_t)b * max_after;
  for (int k = lane; k < max_after; k += 64) kp[k] = -1;
  if (n == 1) { if (lane == 0) { kp[0] = 0; n_keep[b] = 1; } return; }
  for (int t = lane; t < n; t += 64) alive[t] = 1;
  __syncthreads();
  int n_alive = n, kept = 0, head = 0;
  while (n_alive > 1 && kept < max_after) {
    while (!alive[head]) head++;
    const double hs = st[head], he = ed[head];
    int removed = 0;
    for (int j = head + 1 + lane; j < n; j += 64) {
      if (alive[j]) {
        const double inter = fmax(0.0, fmin(he, ed[j]) - fmax(hs, st[j]));
        const double hull = fmax(he, ed[j]) - fmin(hs, st[j]);
        const double iou = hull == 0.0 ? 0.0 : inter / hull;
        if (iou > nms_thd) { alive[j] = 0; removed++; }      // nms_thd is the reference's Python float (a double): 7/10 > 0.7 is False
      }
    }
    removed = (int)wave_sum((float)removed);
    __syncthreads();
    if (lane == 0) { kp[kept] = head; alive[head] = 0; }
    __syncthreads();
    n_alive -= removed + 1;
    kept++;
  }
  if (kept < max_after && n_alive >= 1) {
    while (!alive[head]) head++;
    if (lane == 0) kp[kept] = head;
    kept++;
  }
  if (lane == 0) n_keep[b] = kept;
}

// ---------------- matcher ----------------
__global__ void matcher_cost_kernel(const float* logits, int n_cls, const float* spans, int B, int Q, const float* tgt,
                                    const int* tgt_off, int max_t, float w_class, float w_span, float w_giou, float* cost) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * Q * max_t) return;
  const int j = idx % max_t, q = (idx / max_t) % Q, b = idx / (max_t * Q);
  const int nt = tgt_off[b + 1] - tgt_off[b];
  if (j >= nt) { cost[idx] = 0.f; return; }
  const float* lg = logits + ((size_t)b * Q + q) * n_cls;
  float mx = lg[0];
  for (int c = 1; c < n_cls; c++) mx = fmaxf(mx, lg[c]);
  float se = 0.f;
  for (int c = 0; c < n_cls; c++) se += expf(lg[c] - mx);
  const float prob0 = expf(lg[0] - mx) / se;                              // softmax(-1)[:, foreground_label=0]
  const float pc = spans[((size_t)b * Q + q) * 2], pw = spans[((size_t)b * Q + q) * 2 + 1];
  const float tc = tgt[(size_t)(tgt_off[b] + j) * 2], tw = tgt[(size_t)(tgt_off[b] + j) * 2 + 1];
  const float l1 = fabsf(pc - tc) + fabsf(pw - tw);                        // torch.cdist(p=1) on (cx, w)
  const float a0 = pc - 0.5f * pw, a1 = pc + 0.5f * pw, b0 = tc - 0.5f * tw, b1 = tc + 0.5f * tw;
  const float inter = fmaxf(fminf(a1, b1) - fmaxf(a0, b0), 0.f);
  const float uni = (a1 - a0) + (b1 - b0) - inter;
  const float hull = fmaxf(fmaxf(a1, b1) - fminf(a0, b0), 0.f);
  const float giou = inter / uni - (hull - uni) / hull;
  cost[idx] = w_span * l1 + w_giou * (-giou) + w_class * (-prob0);
}

constexpr int LS_MAXR = 32, LS_MAXC = 256;   // rows = min(Q, nt) side, cols = the larger side
__global__ void lsap_kernel(const float* cost, int B, int Q, const int* tgt_off, int max_t,
                            long long* out_pred, long long* out_tgt, int* n_match) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int nt = tgt_off[b + 1] - tgt_off[b];
  long long* op = out_pred + (size_t)b * max_t;
  long long* ot = out_tgt + (size_t)b * max_t;
  for (int i = 0; i < max_t; i++) { op[i] = -1; ot[i] = -1; }
  const bool transposed = nt < Q;                 // scipy transposes so that nr <= nc
  const int nr = transposed ? nt : Q, nc = transposed ? Q : nt;
  n_match[b] = nr;
  if (nr == 0) return;
  if (nr > LS_MAXR || nc > LS_MAXC) { n_match[b] = -1; return; }
  const float* C = cost + (size_t)b * Q * max_t;
  auto cst = [&](int i, int j) -> double { return transposed ? (double)C[(size_t)j * max_t + i] : (double)C[(size_t)i * max_t + j]; };
  double u[LS_MAXR], v[LS_MAXC], shortest[LS_MAXC];
  int path[LS_MAXC], col4row[LS_MAXR], row4col[LS_MAXC], remaining[LS_MAXC];
  bool SR[LS_MAXR], SC[LS_MAXC];
  for (int i = 0; i < nr; i++) { u[i] = 0.0; col4row[i] = -1; }
  for (int j = 0; j < nc; j++) { v[j] = 0.0; row4col[j] = -1; }
  for (int cur = 0; cur < nr; cur++) {
    for (int j = 0; j < nc; j++) { shortest[j] = INFINITY; path[j] = -1; SC[j] = false; remaining[j] = nc - j - 1; }
    for (int i = 0; i < nr; i++) SR[i] = false;
    int nrem = nc, i = cur, sink = -1;
    double min_val = 0.0;
    while (sink == -1) {
      int index = -1;
      double lowest = INFINITY;
      SR[i] = true;
      for (int it = 0; it < nrem; it++) {
        const int j = remaining[it];
        const double r = min_val + cst(i, j) - u[i] - v[j];
        if (r < shortest[j]) { path[j] = i; shortest[j] = r; }
        if (shortest[j] < lowest || (shortest[j] == lowest && row4col[j] == -1)) { lowest = shortest[j]; index = it; }
      }
      min_val = lowest;
      const int j = remaining[index];
      if (row4col[j] == -1) sink = j; else i = row4col[j];
      SC[j] = true;
      remaining[index] = remaining[--nrem];
    }
    u[cur] += min_val;
    for (int ii = 0; ii < nr; ii++) if (SR[ii] && ii != cur) u[ii] += min_val - shortest[col4row[ii]];
    for (int j = 0; j < nc; j++) if (SC[j]) v[j] -= min_val - shortest[j];
    int j = sink;
    while (true) {
      const int ii = path[j];
      row4col[j] = ii;
      const int tmp = col4row[ii];
      col4row[ii] = j;
      j = tmp;
      if (ii == cur) break;
    }
  }
  if (!transposed) {
    for (int i = 0; i < nr; i++) { op[i] = i; ot[i] = col4row[i]; }
  } else {
    // pairs (pred = col4row[t], tgt = t) sorted by pred index (scipy: argsort of col4row)
    int k = 0;
    for (int q = 0; q < nc; q++) if (row4col[q] != -1) { op[k] = q; ot[k] = row4col[q]; k++; }
  }
}

}  // namespace

extern "C" int uvtg_postprocess_mr(const float* pred_logits, const float* pred_spans, const float* saliency, const float* timestamp,
                                   const float* timestamp_mask, const float* durations, int B, int Lv,
                                   float clip_length, int eval_mode_add, double nms_thd, int max_before, int max_after,
                                   double* windows_out, int* order, int* keep, int* n_keep, float* saliency_out,
                                   uvtg_stream_t stream) {
  if (!pred_logits || !pred_spans || !timestamp || !timestamp_mask || !durations || !windows_out || !order || !keep || !n_keep) return -20;
  if (saliency_out && !saliency) return -20;
  if (B <= 0 || Lv <= 0 || max_after <= 0 || max_before <= 0) return -11;
  const size_t sh = (size_t)((Lv + 1) & ~1) * 4 + (size_t)Lv * 16 + Lv + 16;
  hipLaunchKernelGGL(decode_rank_nms_kernel, dim3(B), dim3(64), sh, (hipStream_t)stream, pred_logits, pred_spans, timestamp,
                     timestamp_mask, durations, B, Lv, nms_thd, max_before, max_after, clip_length, saliency, eval_mode_add ? 1 : 0,
                     saliency_out, windows_out, order, keep, n_keep);
  UVTG_CHECK_LAUNCH();
  return 0;
}
extern "C" int uvtg_decode_rank_nms(const float* pred_logits, const float* pred_spans, const float* timestamp,
                                    const float* timestamp_mask, const float* durations, int B, int Lv,
                                    double nms_thd, int max_before, int max_after,
                                    double* windows_out, int* order, int* keep, int* n_keep, uvtg_stream_t stream) {
  return uvtg_postprocess_mr(pred_logits, pred_spans, nullptr, timestamp, timestamp_mask, durations, B, Lv, 0.f, 0, nms_thd, max_before,
                             max_after, windows_out, order, keep, n_keep, nullptr, stream);
}

extern "C" int uvtg_hungarian(const float* pred_logits, int n_cls, const float* pred_spans_cxw, int B, int Q,
                              const float* tgt_cxw, const int* tgt_off, int max_t, float w_class, float w_span, float w_giou,
                              float* cost, long long* out_pred, long long* out_tgt, int* n_match, uvtg_stream_t stream) {
  if (!pred_logits || !pred_spans_cxw || !tgt_cxw || !tgt_off || !cost || !out_pred || !out_tgt || !n_match) return -20;
  if (B <= 0 || Q <= 0 || max_t <= 0 || n_cls <= 0) return -11;
  hipStream_t s = (hipStream_t)stream;
  const int n = B * Q * max_t;
  hipLaunchKernelGGL(matcher_cost_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, pred_logits, n_cls, pred_spans_cxw, B, Q, tgt_cxw,
                     tgt_off, max_t, w_class, w_span, w_giou, cost);
  hipLaunchKernelGGL(lsap_kernel, dim3(cdiv(B, 64)), dim3(64), 0, s, cost, B, Q, tgt_off, max_t, out_pred, out_tgt, n_match);
  UVTG_CHECK_LAUNCH();
  return 0;
}
