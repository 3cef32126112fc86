// Host-side orchestration of the UniVTG hot path on one MI355X: the whole Model.forward
// (model/univtg.py:105-155) and its backward as ONE C call each, enqueuing ~60 / ~110 kernels on the
// caller's HIP stream (graph-capturable: no allocation, no host sync, all state in caller-owned buffers).
// Token-major layout: row (b*S + s) of every [M, *] activation, s < Lv video clips then Lt text tokens.
#include "uvtg_kernels.h"
#include <mutex>
#include <map>
#include <string>
#include <cstdlib>
#include "../../include/uvtg.h"
#include <cmath>
#include <cstring>

void uvtg_prof_section(int section, int end, hipStream_t s);   // optim.hip: section timing hooks (bench.py)

// ---- developer configuration (include/uvtg_dev.h) --------------------------------------------------------------------------------
// name -> value table behind uvtg_dev_env(); empty unless the developer entry points below fill it
namespace {
std::mutex g_dev_cfg_mu;
std::map<std::string, std::string>& dev_cfg_table() { static std::map<std::string, std::string> t; return t; }
}  // namespace
const char* uvtg_dev_env(const char* name) {
  std::lock_guard<std::mutex> lk(g_dev_cfg_mu);
  auto& t = dev_cfg_table();
  auto it = t.find(name);
  return it == t.end() ? nullptr : it->second.c_str();       // (entries are never erased while a caller may hold the pointer: set replaces the value in place)
}
extern "C" int uvtg_dev_config_set(const char* name, const char* value) {
  if (!name || strncmp(name, "UVTG_", 5) != 0) return -20;
  std::lock_guard<std::mutex> lk(g_dev_cfg_mu);
  if (value) dev_cfg_table()[name] = value; else dev_cfg_table().erase(name);
  return 0;
}
extern char** environ;
extern "C" int uvtg_dev_config_from_env(void) {      // copies every UVTG_* variable of the process environment into the table; returns how many
  int n = 0;
  for (char** e = environ; e && *e; e++) {
    if (strncmp(*e, "UVTG_", 5) != 0) continue;
    const char* eq = strchr(*e, '=');
    if (!eq) continue;
    uvtg_dev_config_set(std::string(*e, eq - *e).c_str(), eq + 1);
    n++;
  }
  return n;
}

namespace {

// ---- parameter table -------------------------------------------------------------------------
enum { IPW = 0, IPB, OPW, OPB, L1W, L1B, L2W, L2B, N1W, N1B, N2W, N2B, PER_LAYER };
// tail of the table: token-type rows, the two conv heads, then n_proj blocks per modality (text first), the pooling vector and -- with
// use_txt_pos -- the trainable text position table and its LayerNorm
enum { TOK = 0, SP0W, SP0B, SP1W, SP1B, SP2W, SP2B, CL0W, CL0B, CL1W, CL1B, CL2W, CL2B, N_FIXED };
enum { PG = 0, PBE, PW, PB, PER_PROJ };      // one LinearLayer: LayerNorm gamma / beta, Linear weight / bias (model/univtg.py:384-406)
constexpr int MAXP = 3;                      // n_input_proj <= 3 (model/univtg.py:89-100)

inline int rup(int x, int m) { return (x + m - 1) / m * m; }

struct Dm {
  uvtg_dims c;
  int S, M, Mv, Mt, Rp, hd, Kpv, Kpt, np, nproj;
  explicit Dm(const uvtg_dims& d) : c(d) {
    S = d.Lv + d.Lt; M = d.B * S; Mv = d.B * d.Lv; Mt = d.B * d.Lt; Rp = d.B * (d.Lv + 2);
    hd = d.H > 0 ? d.d / d.H : 0; Kpv = rup(d.Dv, 64); Kpt = rup(d.Dt, 64); nproj = d.n_proj;
    np = PER_LAYER * d.E + N_FIXED + 2 * PER_PROJ * nproj + 1 + (d.use_txt_pos ? 3 : 0);
  }
  int tail(int k) const { return PER_LAYER * c.E + k; }
  int lay(int l, int k) const { return PER_LAYER * l + k; }
  int proj(int which, int blk, int k) const { return tail(N_FIXED + PER_PROJ * ((which == 0 ? nproj : 0) + blk) + k); }   // which: 0 = video, 1 = text
  int pool() const { return tail(N_FIXED + 2 * PER_PROJ * nproj); }
  int txtpos(int k) const { return pool() + 1 + k; }      // 0: position_embeddings.weight [max_q_l, d], 1 / 2: LayerNorm gamma / beta
  int din(int which, int blk) const { return blk == 0 ? (which == 0 ? c.Dv : c.Dt) : c.d; }
  int kp(int which, int blk) const { return blk == 0 ? (which == 0 ? Kpv : Kpt) : c.d; }
};

int check_dims(const uvtg_dims* d) {
  if (!d) return -10;
  if (d->struct_size != (int)sizeof(uvtg_dims)) return -18;     // caller built against another layout of the struct (uvtg_version)
  if (d->B <= 0 || d->Lv <= 0 || d->Lt <= 0 || d->E <= 0 || d->H <= 0) return -11;
  if (d->n_proj < 1 || d->n_proj > MAXP) return -12;
  if (d->use_txt_pos && d->max_q_l < d->Lt) return -19;
  if (d->d % 32 || d->F % 8) return -13;
  const int hd = d->d / d->H;
  if (hd * d->H != d->d || (hd != 32 && hd != 64 && hd != 128)) return -14;
  if (d->precise != 0 && d->precise != 1) return -25;           // (0 / 1 are the two arithmetic modes; nothing else is defined)
  if (d->precise && d->training) return -15;
  if (d->precise && d->F % 32) return -13;       // split operand rows are made of whole 32-column blocks (uvtg_common.h, split_col)
  if (d->Dv <= 0 || d->Dt <= 0) return -16;
  // (d % 32 == 0 and F % 8 == 0 make every weight MATRIX a multiple of 4 elements: the matrices the weight-gradient launches assign
  // have no alignment padding behind them in the flat gradient buffer, so the clipping norm over the whole buffer sees no unwritten word)
  if (((long long)d->d * d->Dv) % 4 || ((long long)d->d * d->Dt) % 4 || ((long long)d->F * d->d) % 4) return -13;
  return 0;
}

long long pnumel(const Dm& m, int i) {
  const long long d = m.c.d, F = m.c.F;
  if (i < PER_LAYER * m.c.E) {
    switch (i % PER_LAYER) {
      case IPW: return 3 * d * d; case IPB: return 3 * d; case OPW: return d * d; case OPB: return d;
      case L1W: return F * d; case L1B: return F; case L2W: return d * F; case L2B: return d;
      default: return d;
    }
  }
  const int t = i - PER_LAYER * m.c.E;
  switch (t) {
    case TOK: return 2 * d;
    case SP0W: case SP1W: case CL0W: case CL1W: return d * d * 3;
    case SP0B: case SP1B: case CL0B: case CL1B: return d;
    case SP2W: return 2 * d * 3; case SP2B: return 2; case CL2W: return d * 3; case CL2B: return 1;
    default: break;
  }
  const int j = t - N_FIXED;
  if (j < 2 * PER_PROJ * m.nproj) {
    const int which = j < PER_PROJ * m.nproj ? 1 : 0, blk = (j % (PER_PROJ * m.nproj)) / PER_PROJ, k = j % PER_PROJ;
    const long long din = m.din(which, blk);
    return k == PW ? d * din : (k == PB ? d : din);
  }
  const int j2 = j - 2 * PER_PROJ * m.nproj;       // 0: weightedpool.weight, then the text position table + its LayerNorm
  return j2 == 1 ? (long long)m.c.max_q_l * d : d;
}
// is table entry i a weight MATRIX that a weight-gradient launch assigns (everything else is accumulated into zeros)?
bool assigned_matrix(const Dm& m, int i) {
  if (i < PER_LAYER * m.c.E) { const int k = i % PER_LAYER; return k == IPW || k == OPW || k == L1W || k == L2W; }
  const int t = i - PER_LAYER * m.c.E;
  if (t == SP0W || t == SP1W || t == CL0W || t == CL1W) return true;
  const int j = t - N_FIXED;
  return j >= 0 && j < 2 * PER_PROJ * m.nproj && j % PER_PROJ == PW;
}

struct Arena {
  char* base; size_t off;
  explicit Arena(void* b) : base((char*)b), off(0) {}
  template <typename T> T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? (T*)(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

// ---- prepared-operand cache ------------------------------------------------------------------
constexpr int MAXE = 16;
struct WCache {
  bf16_t *wqkv[MAXE], *wo[MAXE], *w1[MAXE], *w2[MAXE], *wqkvT[MAXE], *woT[MAXE], *w1T[MAXE], *w2T[MAXE];
  bf16_t *wc0, *wc1, *wc0T, *wc1T;      // conv operands (fast)
  // precise mode: every weight as fp16 hi | lo images [N, 2 K] scaled by UVTG_SPLIT_W_SCALE (split-operand GEMMs, uvtg_common.h)
  unsigned short *wqkvS[MAXE], *woS[MAXE], *w1S[MAXE], *w2S[MAXE], *wc0S, *wc1S;
  float *bc0, *bc1;                     // merged conv biases [2d]
  // input projections, [modality: 0 = video, 1 = text][block]
  unsigned short* pwS[2][MAXP];         // fp16 hi | lo images [d, 2 Kp] of the (zero-padded) weights: split-operand projections
  bf16_t* pwB[2][MAXP];                 // bf16 weights [d, Kp] (proj_precise == 0)
  bf16_t* pwT[2][MAXP];                 // dgrad operands [Kp, d]
  size_t bytes;
  WCache(const Dm& m, void* base) {
    Arena a(base);
    const size_t d = m.c.d, F = m.c.F;
    const bool fast = !m.c.precise;
    for (int l = 0; l < m.c.E; l++) {
      wqkv[l] = fast ? a.take<bf16_t>(3 * d * d) : nullptr; wo[l] = fast ? a.take<bf16_t>(d * d) : nullptr;
      w1[l] = fast ? a.take<bf16_t>(F * d) : nullptr; w2[l] = fast ? a.take<bf16_t>(d * F) : nullptr;
      const bool tr = fast && m.c.training;
      wqkvT[l] = tr ? a.take<bf16_t>(3 * d * d) : nullptr; woT[l] = tr ? a.take<bf16_t>(d * d) : nullptr;
      w1T[l] = tr ? a.take<bf16_t>(F * d) : nullptr; w2T[l] = tr ? a.take<bf16_t>(d * F) : nullptr;
      wqkvS[l] = !fast ? a.take<unsigned short>(2 * 3 * d * d) : nullptr; woS[l] = !fast ? a.take<unsigned short>(2 * d * d) : nullptr;
      w1S[l] = !fast ? a.take<unsigned short>(2 * F * d) : nullptr; w2S[l] = !fast ? a.take<unsigned short>(2 * d * F) : nullptr;
    }
    wc0 = fast ? a.take<bf16_t>(2 * d * 3 * d) : nullptr; wc1 = fast ? a.take<bf16_t>(2 * d * 3 * d) : nullptr;
    wc0T = (fast && m.c.training) ? a.take<bf16_t>(d * 6 * d) : nullptr;
    wc1T = (fast && m.c.training) ? a.take<bf16_t>(2 * d * 3 * d) : nullptr;
    wc0S = !fast ? a.take<unsigned short>(2 * 2 * d * 3 * d) : nullptr; wc1S = !fast ? a.take<unsigned short>(2 * 2 * d * 3 * d) : nullptr;
    bc0 = a.take<float>(2 * d); bc1 = a.take<float>(2 * d);
    const bool pp = m.c.precise || m.c.proj_precise;
    const bool tr = fast && m.c.training;
    for (int w = 0; w < 2; w++)
      for (int b = 0; b < MAXP; b++) {
        const bool on = b < m.nproj;
        const size_t Kp = on ? m.kp(w, b) : 0;
        pwS[w][b] = (on && pp) ? a.take<unsigned short>(2 * d * Kp) : nullptr;
        pwB[w][b] = (on && !pp) ? a.take<bf16_t>(d * Kp) : nullptr;
        pwT[w][b] = (on && tr) ? a.take<bf16_t>(Kp * d) : nullptr;
      }
    bytes = a.off + 256;
  }
};

// ---- workspace -------------------------------------------------------------------------------
struct WSpace {
  float* pos; unsigned char* kvalid; float* dps;
  // packed (ragged) execution: tables, packed layer-0 operands, packed conv-head gradient
  PackTables pk; int* lens_dev; bf16_t *xb0p, *ub0p, *g2p;
  // input projections, [modality: 0 = video, 1 = text][block]: the block's GEMM operand LN(x) (+dropout), its bf16 copy for the weight
  // gradient when the operand itself is fp32, the LayerNorm statistics, and (blocks before the last) the block's fp32 output
  void* pa[2][MAXP]; bf16_t* paB[2][MAXP]; float *pm[2][MAXP], *pr[2][MAXP], *ph[2][MAXP];
  // trainable text positions (use_txt_pos): pos rows of the text tokens live behind the clip rows of `pos`; row tables; saved LayerNorm input + statistics
  float *pos_txt, *tp_xsum, *tp_mean, *tp_rstd; int *pos_row_all, *tp_src;
  // encoder
  float* xin[MAXE + 1]; void *xb[MAXE + 1], *ub[MAXE + 1];
  void *qkv[MAXE], *o[MAXE], *x1b[MAXE], *h[MAXE]; bf16_t* apre[MAXE];
  float *lse[MAXE], *y1[MAXE], *mean1[MAXE], *rstd1[MAXE], *y2[MAXE], *mean2[MAXE], *rstd2[MAXE], *x1;
  bf16_t *y1b[MAXE], *y2b[MAXE];        // fast mode: the pre-LayerNorm sums live in bf16 (the fp32 y1 / y2 / xin / x1 are precise-mode only)
  // heads
  void *vm_pad, *h1_pad, *h2_pad;
  // saliency
  float *alpha, *cosv, *vnorm, *qnorm, *sal_dq, *sal_dlog;
  // backward scratch
  float *dvm, *gx[2], *dyF, *delta, *dA[2], *tn_scratch; long long tn_scratch_floats;
  float* deltaL[MAXE]; long long delta_floats;      // per-layer attention-backward delta [B, H, S], zeroed with gnorm2 / the tickets (they follow them): filled by the dO GEMM's epilogue
  float* ln_part[2 * MAXE]; long long ln_part_floats;      // per-launch dgamma / dbeta partials of the encoder's LayerNorm backward launches (folded by ONE launch)
  float *dpos_txt, *tp_dx;   // use_txt_pos: gradient wrt the text position rows (summed over the layers' q,k operands), and wrt their LayerNorm input
  float* gnorm2;       // sum of squares of the step's gradients, accumulated by uvtg_backward (uvtg_backward_gradnorm2)
  bf16_t *dh2_pad, *dh1_pad, *dyR, *dvmB, *gxb[2], *dOb, *dyP[2], *dhb[2][MAXP];   // dhb[w][b]: gradient wrt the output of projection block b (b < n_proj - 1)
  // per-layer operands of the encoder's weight gradients (LayerNorm-2 / LayerNorm-1 input gradients, activation gradient, dqkv): kept
  // until the end of the encoder backward so that the weight gradients of ALL layers can run as one launch without a reduce pass
  bf16_t *dy2L[MAXE], *dy1L[MAXE], *daL[MAXE], *dqkvL[MAXE];
  float* tnh_slabs; long long tnh_slab_floats; unsigned* tnh_tickets; int tnh_n_tickets;
  // split-K of the forward's small NT launches (uvtg_kernels.h, GemmArgs::sk_*): partial-tile slabs + tickets (zeroed by the forward's first kernel)
  float* sk_slab; unsigned* sk_tickets;
  size_t bytes;
  WSpace(const Dm& m, void* base, float* x0) {
    Arena a(base);
    const size_t d = m.c.d, F = m.c.F, M = m.M, B = m.c.B, E = m.c.E;
    const bool tr = m.c.training, fast = !m.c.precise;
    const size_t es = fast ? 2 : 4;                       // compute-dtype element size
    const bool pp = m.c.precise || m.c.proj_precise;
    const bool tpos = m.c.use_txt_pos != 0;
    pos = a.take<float>((size_t)(m.Mv + (tpos ? m.Mt : 0)) * d); kvalid = a.take<unsigned char>(M); dps = a.take<float>(2 * E * B);
    pos_txt = tpos ? pos + (size_t)m.Mv * d : nullptr;
    pos_row_all = tpos ? a.take<int>(M) : nullptr; tp_src = tpos ? a.take<int>(m.Mt) : nullptr;
    tp_xsum = (tpos && tr) ? a.take<float>((size_t)m.Mt * d) : nullptr;
    tp_mean = tpos ? a.take<float>(m.Mt) : nullptr; tp_rstd = tpos ? a.take<float>(m.Mt) : nullptr;
    lens_dev = a.take<int>(2 * B);
    sk_tickets = a.take<unsigned>(UVTG_SK_UNITS); sk_slab = a.take<float>((size_t)UVTG_SK_UNITS * UVTG_SK_TILE_FLOATS);
    pk.seq_start = a.take<int>(B); pk.seq_count = a.take<int>(B);
    pk.row_sample = a.take<int>(M); pk.row_src = a.take<int>(M); pk.row_pos = a.take<int>(M);
    pk.pad2pack = a.take<int>(M); pk.grad_map = a.take<int>(M); pk.kvalid = a.take<unsigned char>(M);
    pk.fstart = a.take<int>(B); pk.kept = a.take<int>(B); pk.frame_valid = a.take<float>((size_t)m.Rp + 1);
    pk.vin_src = a.take<int>(m.Mv); pk.vin_dst = a.take<int>(m.Mv); pk.vin_x0 = a.take<int>(m.Mv); pk.vin_of = a.take<int>(m.Mv);
    pk.tin_dst = a.take<int>(m.Mt); pk.vin_cnt = a.take<int>(B); pk.vin_sample = a.take<int>(m.Mv);
    xb0p = fast ? a.take<bf16_t>(M * d) : nullptr; ub0p = fast ? a.take<bf16_t>(M * d) : nullptr;
    g2p = (fast && tr) ? a.take<bf16_t>(M * d) : nullptr;
    for (int w = 0; w < 2; w++)
      for (int b = 0; b < MAXP; b++) {
        const bool on = b < m.nproj;
        const size_t R = w == 0 ? m.Mv : m.Mt, Kp = on ? m.kp(w, b) : 0;
        pa[w][b] = on ? a.take<char>(R * Kp * (pp ? 4 : 2)) : nullptr;
        paB[w][b] = (on && tr && pp) ? a.take<bf16_t>(R * Kp) : nullptr;
        pm[w][b] = on ? a.take<float>(R) : nullptr; pr[w][b] = on ? a.take<float>(R) : nullptr;
        ph[w][b] = (on && b + 1 < m.nproj) ? a.take<float>(R * d) : nullptr;
      }
    x1 = fast ? nullptr : a.take<float>(M * d);
    float* xpp[2] = {nullptr, nullptr}; void* xbpp[2] = {nullptr, nullptr}; void* ubpp[2] = {nullptr, nullptr};
    for (size_t l = 0; l <= E; l++) {       // layer l reads slot l, writes slot l + 1
      if (tr) {
        xin[l] = l == 0 ? x0 : (fast ? nullptr : a.take<float>(M * d));
        xb[l] = fast ? (void*)a.take<char>(M * d * es) : (void*)xin[l];
        ub[l] = a.take<char>(M * d * es);
      } else {                              // eval: ping-pong
        if (l == 0) xin[0] = x0;
        else if (!fast) { if (l <= 2) xpp[l - 1] = a.take<float>(M * d); xin[l] = xpp[(l - 1) % 2]; }
        else xin[l] = nullptr;
        if (l <= 1) xbpp[l] = a.take<char>(M * d * es);      // fast: bf16 rows; precise: fp16 hi | lo images [M, 2d] (the same bytes)
        xb[l] = xbpp[l % 2];
        if (l <= 1) ubpp[l] = a.take<char>(M * d * es);
        ub[l] = ubpp[l % 2];
      }
    }
    for (size_t l = 0; l < E; l++) {
      const bool own = tr || l == 0;
      qkv[l] = own ? (void*)a.take<char>(M * 3 * d * es) : qkv[0];
      o[l] = own ? (void*)a.take<char>(M * d * es) : o[0];
      lse[l] = own ? a.take<float>(B * m.c.H * m.S) : lse[0];
      y1[l] = fast ? nullptr : (own ? a.take<float>(M * d) : y1[0]);
      y1b[l] = fast ? (own ? a.take<bf16_t>(M * d) : y1b[0]) : nullptr;
      mean1[l] = own ? a.take<float>(M) : mean1[0]; rstd1[l] = own ? a.take<float>(M) : rstd1[0];
      x1b[l] = own ? (void*)a.take<char>(M * d * es) : x1b[0];      // fast: bf16 LN1 output; precise: its fp16 hi | lo images (x1 keeps the fp32 residual)
      apre[l] = tr ? a.take<bf16_t>(M * F) : nullptr;
      h[l] = own ? (void*)a.take<char>(M * F * es) : h[0];
      y2[l] = fast ? nullptr : (own ? a.take<float>(M * d) : y2[0]);
      y2b[l] = fast ? (own ? a.take<bf16_t>(M * d) : y2b[0]) : nullptr;
      mean2[l] = own ? a.take<float>(M) : mean2[0]; rstd2[l] = own ? a.take<float>(M) : rstd2[0];
    }
    vm_pad = a.take<char>((size_t)(m.Rp + 1) * d * es);
    h1_pad = a.take<char>((size_t)(m.Rp + 1) * 2 * d * es);
    h2_pad = a.take<char>((size_t)(m.Rp + 1) * 2 * d * es);
    alpha = a.take<float>((size_t)B * m.c.Lt); cosv = a.take<float>(m.Mv); vnorm = a.take<float>(m.Mv); qnorm = a.take<float>(B);
    if (tr) {
      dvm = nullptr; gx[0] = gx[1] = nullptr; dyF = nullptr;       // (fp32 gradient stream: not used by the bf16 training path)
      dvmB = a.take<bf16_t>((size_t)(m.Rp + 1) * d); gxb[0] = a.take<bf16_t>(M * d); gxb[1] = a.take<bf16_t>(M * d);     // (frame-row space on the loss-only stream)
      dyR = a.take<bf16_t>(M * d); delta = a.take<float>(B * m.c.H * m.S);
      sal_dq = a.take<float>(B * d); sal_dlog = a.take<float>(B * (size_t)m.c.Lt);
      {   // clipping-norm slots, directly followed by the tickets of the hybrid weight-gradient launch: ONE memset zeroes both
        const int dt = (int)((d + 255) / 256), ft = (int)((F + 255) / 256);
        tnh_n_tickets = (int)E * (2 * dt * ft + 4 * dt * dt) + 12 * dt * dt;      // (+ the four conv-head weight gradients' tap tiles, round 5)
        const size_t dfl = ((size_t)B * m.c.H * m.S + 3) & ~(size_t)3;
        delta_floats = (long long)(dfl * E);
        gnorm2 = a.take<float>(UVTG_SQSUM_FLOATS + (size_t)tnh_n_tickets + dfl * E);
        tnh_tickets = (unsigned*)(gnorm2 + UVTG_SQSUM_FLOATS);
        for (size_t l = 0; l < MAXE; l++) deltaL[l] = l < E ? gnorm2 + UVTG_SQSUM_FLOATS + tnh_n_tickets + dfl * l : nullptr;
        tnh_slab_floats = gemm_tn_multi_slab_floats(tnh_n_tickets, 320);
        tnh_slabs = a.take<float>((size_t)tnh_slab_floats);
      }
      {  // split-partial slabs of the 256-tile weight-gradient kernel: the largest requirement over the shapes backward launches
        long long need = 0;
        const int shapes[][3] = {{m.M, (int)d, (int)F}, {m.M, (int)F, (int)d}, {m.M, (int)d, (int)d}, {m.M, 2 * (int)d, (int)d}, {m.Rp, (int)d, (int)d}, {m.Rp, (int)d, 3 * (int)d},
                                 {m.Mv, (int)d, (int)d}, {m.Mt, (int)d, (int)d}, {m.Mv, (int)d, m.c.Dv}, {m.Mt, (int)d, m.c.Dt}};
        for (auto& sh : shapes) { const long long f = gemm_tn_scratch_floats(sh[0], sh[1], sh[2]); if (f > need) need = f; }
        { const long long grouped = 320LL * 65536 + 64LL * 8 * (2 * (long long)d + (long long)F + 256); if (grouped > need) need = grouped; }   // grouped launches: <= ~1 unit per CU + bias partials
        tn_scratch_floats = need; tn_scratch = a.take<float>((size_t)need);
      }
      ln_part_floats = ln_bwd_partial_floats(m.M, (int)d);
      for (size_t l = 0; l < 2 * MAXE; l++) ln_part[l] = l < 2 * E ? a.take<float>((size_t)ln_part_floats) : nullptr;
      dh2_pad = a.take<bf16_t>((size_t)(m.Rp + 1) * 2 * d); dh1_pad = a.take<bf16_t>((size_t)(m.Rp + 1) * 2 * d);
      dOb = a.take<bf16_t>(M * d);
      for (size_t l = 0; l < E; l++) { dy2L[l] = a.take<bf16_t>(M * d); dy1L[l] = a.take<bf16_t>(M * d); daL[l] = a.take<bf16_t>(M * F); dqkvL[l] = a.take<bf16_t>(M * 3 * d); }
      for (int w = 0; w < 2; w++) {
        const size_t R = w == 0 ? m.Mv : m.Mt, Kp = w == 0 ? m.Kpv : m.Kpt;
        dyP[w] = a.take<bf16_t>(R * d);
        for (int b = 0; b < MAXP; b++) dhb[w][b] = (b + 1 < m.nproj) ? a.take<bf16_t>(R * d) : nullptr;
        dA[w] = a.take<float>(R * (Kp > d ? Kp : d));        // fp32 dgrad output of the block being processed (blocks run one after the other)
      }
      dpos_txt = tpos ? a.take<float>((size_t)m.Mt * d) : nullptr; tp_dx = tpos ? a.take<float>((size_t)m.Mt * d) : nullptr;
    } else {
      dpos_txt = tp_dx = nullptr;
      for (int l = 0; l < 2 * MAXE; l++) ln_part[l] = nullptr;
      ln_part_floats = 0;
      for (int l = 0; l < MAXE; l++) deltaL[l] = nullptr;
      delta_floats = 0;
      dvm = gx[0] = gx[1] = dyF = delta = nullptr; dyR = dvmB = gxb[0] = gxb[1] = nullptr; sal_dq = sal_dlog = nullptr; tn_scratch = nullptr; tn_scratch_floats = 0; dh2_pad = dh1_pad = dOb = nullptr; gnorm2 = nullptr; tnh_slabs = nullptr; tnh_slab_floats = 0; tnh_tickets = nullptr; tnh_n_tickets = 0;
      for (int l = 0; l < MAXE; l++) dy2L[l] = dy1L[l] = daL[l] = dqkvL[l] = nullptr;
      for (int w = 0; w < 2; w++) { dyP[w] = nullptr; dA[w] = nullptr; for (int b = 0; b < MAXP; b++) dhb[w][b] = nullptr; }
    }
    bytes = a.off + 256;
  }
};

// ---- tiny helper kernels ----------------------------------------------------------------------
// up to three frame buffers in ONE launch (blockIdx.y = buffer; the forward's heads zero the frames of vm_pad, h1_pad and h2_pad: three 5 us launches)
struct ZeroFrames { char* p[3]; int row_bytes[3]; const int* fstart[3]; const int* kept[3]; };
__global__ void zero_frame_rows_multi_kernel(ZeroFrames z, int B, int Lv) {
  const int which = blockIdx.x, b = which >> 1, k = blockIdx.y;
  const int* fs = z.fstart[k];
  const int r = fs ? fs[b] + ((which & 1) ? z.kept[k][b] + 1 : 0) : b * (Lv + 2) + ((which & 1) ? Lv + 1 : 0);
  u32x4* row = (u32x4*)(z.p[k] + (size_t)r * z.row_bytes[k]);
  const u32x4 zero = {0, 0, 0, 0};
  for (int i = threadIdx.x; i < z.row_bytes[k] / 16; i += blockDim.x) row[i] = zero;
}
int zero_frames(const ZeroFrames& z, int count, int B, int Lv, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(zero_frame_rows_multi_kernel, dim3(2 * B, count), dim3(256), 0, s, z, B, Lv);
  UVTG_CHECK_LAUNCH();
  return 0;
}
// ---- the LAST encoder layer's FFN half on the clip rows only (round 5) ------------------------------------------------------------------------
// The text rows of the encoder output are read by nobody (`vid_mem = memory[:, :L_v]`, model/univtg.py:127; the saliency branch reads the
// PROJECTED text rows x0).  In the last layer everything behind the attention block's LayerNorm is row-wise, so LayerNorm 1 reads the clip
// rows out of the token-major stream (LnFwdArgs::x_seg) and writes them compact; linear1 / GELU / linear2 / LayerNorm 2 -- and in the backward
// their gradients -- run on B Lv instead of B (Lv + Lt) rows; LayerNorm 1's backward scatters its input gradient back into the token-major
// stream, whose text rows are zero (their only gradient arrives through the attention as keys / values).  On the packed (ragged) stream the
// clip rows are the compact rows of the video input projection (PackTables::vin_dst: valid clips + representative / halo clips), read and
// scattered through that table.  Exact: the dropped rows' values never reach an output, and their gradient contributions are exact zeros.  
// uvtg_backward reads the rows the way the forward on the same workspace wrote them (clip_record below); a training call that asks for `memory` is refused (-24), an eval call with
// `memory` runs all rows.
static int g_conv_defer = -1;
extern "C" int uvtg_debug_tn_conv_defer(int on) { g_conv_defer = on ? 1 : 0; return 0; }
static int g_last_clip = -1;
extern "C" int uvtg_debug_last_layer_clip(int on) { g_last_clip = on ? 1 : 0; return 0; }
static bool last_layer_clip(const Dm& m) {
  if (g_last_clip < 0) g_last_clip = uvtg_dev_env("UVTG_LAST_CLIP_OFF") ? 0 : 1;
  return g_last_clip == 1 && !m.c.precise && m.c.Lt > 0 && ln_clip_rows_ok(m.c.d);
}
// The choice depends on developer knobs (uvtg_debug_last_layer_clip, uvtg_debug_ln_fwd_lean), so uvtg_backward must not re-derive it: a knob
// flipped between a training forward and its backward would make it read compact clip-row buffers as full ones.  uvtg_forward
// remembers what it did per workspace (host side, a handful of entries: one per live workspace) and uvtg_backward reads the buffers that way.
namespace {
struct ClipRecord { const void* ws; int clip; };
ClipRecord g_clip_rec[64];
int g_clip_rec_n = 0, g_clip_rec_next = 0;
std::mutex g_clip_rec_mu;
void clip_record(const void* ws, bool clip) {
  std::lock_guard<std::mutex> lk(g_clip_rec_mu);
  for (int i = 0; i < g_clip_rec_n; i++) if (g_clip_rec[i].ws == ws) { g_clip_rec[i].clip = clip; return; }
  const int slot = g_clip_rec_n < 64 ? g_clip_rec_n++ : (g_clip_rec_next++ & 63);
  g_clip_rec[slot] = {ws, clip ? 1 : 0};
}
int clip_recorded(const void* ws) {        // -1: this workspace has seen no forward
  std::lock_guard<std::mutex> lk(g_clip_rec_mu);
  for (int i = 0; i < g_clip_rec_n; i++) if (g_clip_rec[i].ws == ws) return g_clip_rec[i].clip;
  return -1;
}
}  // namespace
__global__ void concat2_kernel(const float* a, const float* b, float* dst, const float* a2, const float* b2, float* dst2, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;       // two concatenations per launch (the merged conv biases of both conv layers)
  if (i < n) { dst[i] = a[i]; dst[n + i] = b[i]; dst2[i] = a2[i]; dst2[n + i] = b2[i]; }
}
// gather token rows (b*S + off + t) of a fp32 [B*S, d] tensor into a compact bf16 [B*L, d] one
__global__ void gather_rows_bf16_kernel(const float* src, int S, int off, int L, int d, bf16_t* dst, long long n4) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int dq = d / 4;
  const long long row = i / dq; const int c = (int)(i % dq) * 4;
  const int b = (int)(row / L), t = (int)(row % L);
  const f32x4 v = *(const f32x4*)(src + ((size_t)b * S + off + t) * d + c);
  u32x2 o; o[0] = pack_bf2(v[0], v[1]); o[1] = pack_bf2(v[2], v[3]);
  *(u32x2*)(dst + (size_t)row * d + c) = o;
}
__global__ void add_vec_kernel(float* dst, const float* src, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}
__global__ void add_vec2_kernel(float* dst_a, const float* src_a, float* dst_b, const float* src_b, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { dst_a[i] += src_a[i]; dst_b[i] += src_b[i]; }
}

// --use_txt_pos tables: pos_row_all[b*S + s] = row of the [B*Lv + B*Lt, d] pos table that token (b, s) adds to its q,k operand;
// tp_src[b*Lt + t] = row b*S + Lv + t of the padded token layout (the text rows' LayerNorm input and dropout counter key)
__global__ void txt_pos_tables_kernel(int B, int Lv, int Lt, int* pos_row_all, int* tp_src) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, S = Lv + Lt;
  if (i >= B * S) return;
  const int b = i / S, sidx = i - b * S;
  pos_row_all[i] = sidx < Lv ? b * Lv + sidx : B * Lv + b * Lt + (sidx - Lv);
  if (sidx >= Lv) tp_src[b * Lt + (sidx - Lv)] = i;
}
// dE[t][c] = sum_b g[(b*Lt + t)][c]: gradient of the text position table rows t < Lt (position_encoding.py:33-36)
__global__ void txt_pos_table_grad_kernel(const float* g, int B, int Lt, int d, float* dE) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Lt * d) return;
  const int t = i / d, c = i - t * d;
  float acc = 0.f;
  for (int b = 0; b < B; b++) acc += g[((size_t)b * Lt + t) * d + c];
  dE[i] = acc;
}

#define TRY(x) do { int e__ = (x); if (e__) return e__; } while (0)

// Packed (ragged) encoder stream, decided identically by uvtg_forward and uvtg_backward from (dims, lens_host):
//   PACK_FULL : valid clips + ONE representative padded clip + valid text tokens per sample.  Exact whenever no per-row randomness
//               touches the padded clips (eval, or training with input_dropout == 0 and attention dropout == 0; DropPath is per sample).
//   PACK_TEXT : every clip row (padded ones included: under input / attention dropout each of them carries its own mask in the
//               reference, model/univtg.py:392-404) + valid text tokens.  Padded text tokens are masked keys whose outputs nobody
//               reads, so dropping them is exact under every dropout.
//   PACK_HALO : (dims.loss_only, training, p_attn == 0) valid clips + the first three padded clips + valid text.  Padded clips are
//               never keys; they reach valid positions only through the k = 3 x 3-layer conv heads (receptive field +-3), and every
//               loss masks padded positions: losses and all parameter gradients stay exact, only pred_* at padded positions beyond
//               the halo differ (the heads see zero rows there).
enum { PACK_NONE = 0, PACK_FULL = 1, PACK_TEXT = 2, PACK_HALO = 3 };
constexpr int HALO = 3;
int pack_mode(const uvtg_dims& c, const int* lens_host) {
  if (!lens_host || c.precise || c.use_txt_pos) return PACK_NONE;      // (trainable text positions: padded execution only)
  if (c.training && (c.p_in > 0.f || c.p_attn > 0.f)) return (c.loss_only && c.p_attn <= 0.f) ? PACK_HALO : PACK_TEXT;
  return PACK_FULL;
}
// rows of the ragged conv-head frames of the loss-only stream: kept clips + 2 zero rows per sample
int halo_frame_rows(const Dm& m, const int* lens) {
  long long n = 0;
  for (int b = 0; b < m.c.B; b++) n += (lens[b] + HALO < m.c.Lv ? lens[b] + HALO : m.c.Lv) + 2;
  return (int)n;
}
// clip rows that have a packed row (= rows of the compact video input projection): valid clips + representative / halo / all
int compact_clip_rows(const Dm& m, const int* lens, int mode) {
  long long n = 0;
  for (int b = 0; b < m.c.B; b++) {
    const int lv = lens[b];
    n += mode == PACK_TEXT ? m.c.Lv : (mode == PACK_HALO ? (lv + HALO < m.c.Lv ? lv + HALO : m.c.Lv) : lv + (lv < m.c.Lv ? 1 : 0));
  }
  return (int)n;
}
// rows of the packed encoder stream for these host-side lengths (lens[0..B) clips, lens[B..2B) text tokens per sample)
int packed_rows(const Dm& m, const int* lens, int mode, int* out) {
  long long n = 0;
  for (int b = 0; b < m.c.B; b++) {
    const int lv = lens[b], lt = lens[m.c.B + b];
    if (lv < 1 || lv > m.c.Lv || lt < 1 || lt > m.c.Lt) return -23;
    n += (mode == PACK_TEXT ? m.c.Lv : (mode == PACK_HALO ? (lv + HALO < m.c.Lv ? lv + HALO : m.c.Lv) : lv + (lv < m.c.Lv ? 1 : 0))) + lt;
  }
  *out = (int)n;
  return 0;
}

GemmArgs gemm_base(const void* A, int lda, const void* B, int ldb, int M, int N, int K) {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.B = B; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ktap = K; g.groups = 1; g.rs_seg = 1;
  return g;
}

}  // namespace

// =================================================================================================
// public: sizes / tables
// =================================================================================================
// 300 (round 4): uvtg_dims gained use_txt_pos / max_q_l (trainable text positions; three more table entries when set) and n_proj accepts
// 1..3 (the table holds 4 n_proj entries per modality); precise = 1 means fp16 hi + lo operand images (precise takes 0 / 1 only);
// REMOVED in this step: uvtg_linear_f32x3 (replaced by uvtg_split_f16 + uvtg_linear_split: the kernel-level entry takes pre-split rows).
// 100: round 1 ABI.  200: uvtg_dims gained struct_size (first field, validated) and loss_only; uvtg_decode_rank_nms / uvtg_postprocess_mr take
// nms_thd as double; uvtg_debug_force_nt_wn / uvtg_set_dynamic_tiles removed ("ABI history").
// 301: additive -- uvtg_linear_bf16_sk / uvtg_linear_split_sk / uvtg_linear_sk_ws_floats, uvtg_debug_nt_small / _splitk / _splitk_parts / _small_tile / _loader_waves;
// uvtg_workspace_bytes grew by the forward's split-K slabs (32 MB) and tickets.
// 302 (round 5): additive -- developer-side symbols only (include/uvtg_dev.h, where the uvtg_debug_* / uvtg_profile_* PROTOTYPES now live):
// uvtg_debug_nt_plan2 / _nt_plan_override / _nt_cgw / _layernorm_fwd_bf16 / _ln_fwd_lean / _delta_fuse / _attn_ws / _attn_fwd_dma; uvtg_workspace_bytes grew by the
// per-layer attention-delta buffers and LayerNorm partial slabs (E x (B H S + 2 x 4 MB)); check_dims rejects precise outside {0, 1}: -25.  Nothing in
// include/uvtg.h changed.
// 303 (round 5): the last encoder layer's FFN half runs on the clip rows only (last_layer_clip below); a bf16 training call with memory != NULL is
// refused (-24) on the unpacked stream as it already was on the packed one; developer switches uvtg_debug_last_layer_clip, uvtg_debug_tn_conv_defer.
// 304 (round 6): nothing in include/uvtg.h changed shape; additive: uvtg_cls_nce_fwd / _bwd / _ws_floats.  Behaviour: no entry point reads the process environment any more (developer switches:
// uvtg_dev_config_set / uvtg_dev_config_from_env, include/uvtg_dev.h); uvtg_backward keeps its deferred weight-gradient launch under ready_events
// and records them in groups (uvtg_backward_event_groups, additive) and remembers the forward's last-layer row layout per workspace.
extern "C" int uvtg_version(void) { return 304; }

extern "C" const char* uvtg_strerror(int code) {
  if (code == 0) return "ok";
  if (code > 0) return hipGetErrorString((hipError_t)code);
  switch (code) {
    case -1: return "gemm: non-positive dimension";
    case -2: return "gemm: leading dimension / tap width not a multiple of the vector width";
    case -3: return "gemm: operand pointer not 16-byte aligned";
    case -4: return "layernorm: unsupported row width";
    case -5: return "attention: head_dim must be 32, 64 or 128";
    case -6: return "backward is only available in bf16 mode (precise == 0)";
    case -7: return "gemm: N and every epilogue leading dimension must be multiples of 4";
    case -10: return "null dims";
    case -11: return "dims: non-positive size";
    case -12: return "dims: n_proj must be 1, 2 or 3";
    case -19: return "dims: use_txt_pos needs max_q_l >= Lt (rows of txt_position_embed.position_embeddings)";
    case -13: return "dims: hidden_dim must be a multiple of 32 and dim_feedforward of 8 (of 32 in the precise mode)";
    case -14: return "dims: hidden_dim / nheads must be 32, 64 or 128";
    case -15: return "dims: precise mode is forward-only (training must be 0)";
    case -16: return "dims: feature dims must be positive";
    case -17: return "dims: too many encoder layers";
    case -18: return "dims: struct_size does not match this library's uvtg_dims (caller built against another uvtg.h; see uvtg_version)";
    case -20: return "null pointer argument";
    case -21: return "force_nt_tile: tile must be 0, 128 or 256";
    case -22: return "backward: ready_events must hold enc_layers + 1 events (or n_events = 0)";
    case -23: return "lens_host: every sample needs 1 <= len_v <= Lv clips and 1 <= len_t <= Lt text tokens";
    case -25: return "dims: precise must be 0 (bf16 operands) or 1 (fp16 hi + lo operand images)";
    case -24: return "forward: the memory output cannot be requested in a bf16 training call (neither the packed stream nor the last layer's clip-row tail computes its text rows)";
    default: return "invalid argument";
  }
}

extern "C" int uvtg_param_count(const uvtg_dims* dm) { return dm ? Dm(*dm).np : -10; }
extern "C" int uvtg_param_numel(const uvtg_dims* dm, int i, long long* numel) {
  if (int e = check_dims(dm)) return e;
  Dm m(*dm);
  if (i < 0 || i >= m.np || !numel) return -20;
  *numel = pnumel(m, i);
  return 0;
}
extern "C" int uvtg_param_offsets(const uvtg_dims* dm, long long* off) {
  if (int e = check_dims(dm)) return e;
  if (!off) return -20;
  Dm m(*dm);
  long long o = 0;
  for (int i = 0; i < m.np; i++) { off[i] = o; o += (pnumel(m, i) + 3) / 4 * 4; }
  off[m.np] = o;
  return 0;
}
extern "C" size_t uvtg_workspace_bytes(const uvtg_dims* dm) {
  if (check_dims(dm) || dm->E > MAXE) return 0;
  Dm m(*dm);
  return WSpace(m, nullptr, nullptr).bytes;
}
extern "C" size_t uvtg_wcache_bytes(const uvtg_dims* dm) {
  if (check_dims(dm) || dm->E > MAXE) return 0;
  Dm m(*dm);
  return WCache(m, nullptr).bytes;
}
extern "C" long long uvtg_loss_ws_floats(int B, int Lv, int d) { return loss_ws_floats(B, Lv, d); }

// =================================================================================================
// weight preparation
// =================================================================================================
extern "C" int uvtg_prepare_weights(const uvtg_dims* dm, const float* const* P, void* wcache, uvtg_stream_t stream) {
  if (int e = check_dims(dm)) return e;
  if (dm->E > MAXE) return -17;
  if (!P || !wcache) return -20;
  hipStream_t s = (hipStream_t)stream;
  Dm m(*dm);
  WCache w(m, wcache);
  const int d = m.c.d, F = m.c.F;
  const bool fast = !m.c.precise, tr = fast && m.c.training;
  // every cast / transpose / conv re-layout of the step goes out in ONE launch per kind (the rebuild runs after each
  // optimizer step: ~45 five-microsecond launches otherwise)
  CastOps co; co.count = 0;
  TransposeOps to; to.count = 0;
  ConvWOps cv; cv.count = 0; cv.N = d; cv.C = d;
  SplitOps so; so.count = 0; so.scale = UVTG_SPLIT_W_SCALE;
  auto split = [&](const float* src, unsigned short* dst, int rows, int cols, int kp, int conv = 0) {
    so.src[so.count] = src; so.dst[so.count] = dst; so.rows[so.count] = rows; so.cols[so.count] = cols; so.kp[so.count] = kp; so.conv[so.count] = conv; so.count++;
  };
  auto cast = [&](const float* src, bf16_t* dst, long long n) { co.src[co.count] = src; co.dst[co.count] = dst; co.n[co.count] = n; co.count++; };
  auto transp = [&](const float* src, int rows, int cols, bf16_t* dst, int ld, bf16_t* plain = nullptr, int cols_pad = 0) {
    to.src[to.count] = src; to.dst[to.count] = dst; to.plain[to.count] = plain; to.rows[to.count] = rows; to.cols[to.count] = cols; to.ld[to.count] = ld;
    to.cols_pad[to.count] = cols_pad; to.count++;
  };
  auto convw = [&](const float* wsrc, bf16_t* dst, int ld, int ntot, int n_off, int kind) {
    cv.w[cv.count] = wsrc; cv.dst[cv.count] = dst; cv.ld[cv.count] = ld; cv.ntot[cv.count] = ntot; cv.n_off[cv.count] = n_off; cv.kind[cv.count] = kind; cv.count++;
  };
  if (4 * m.c.E + 4 + 2 * MAXP > UVTG_MAX_PREP_OPS) return -17;
  for (int l = 0; l < m.c.E && !fast; l++) {
    split(P[m.lay(l, IPW)], w.wqkvS[l], 3 * d, d, d);
    split(P[m.lay(l, OPW)], w.woS[l], d, d, d);
    split(P[m.lay(l, L1W)], w.w1S[l], F, d, d);
    split(P[m.lay(l, L2W)], w.w2S[l], d, F, F);
  }
  for (int l = 0; l < m.c.E && fast; l++) {
    if (tr) {        // training: ONE pass over the fp32 master writes the forward operand and the dgrad operand
      transp(P[m.lay(l, IPW)], 3 * d, d, w.wqkvT[l], 3 * d, w.wqkv[l]);
      transp(P[m.lay(l, OPW)], d, d, w.woT[l], d, w.wo[l]);
      transp(P[m.lay(l, L1W)], F, d, w.w1T[l], F, w.w1[l]);
      transp(P[m.lay(l, L2W)], d, F, w.w2T[l], d, w.w2[l]);
    } else {
      cast(P[m.lay(l, IPW)], w.wqkv[l], 3LL * d * d);
      cast(P[m.lay(l, OPW)], w.wo[l], (long long)d * d);
      cast(P[m.lay(l, L1W)], w.w1[l], (long long)F * d);
      cast(P[m.lay(l, L2W)], w.w2[l], (long long)d * F);
    }
  }
  // conv heads: layer 0 of both heads merged along N (span rows then class rows), layer 1 grouped
  const size_t cw = (size_t)d * 3 * d;
  if (fast) {
    convw(P[m.tail(SP0W)], w.wc0, 3 * d, 0, 0, 0);
    convw(P[m.tail(CL0W)], w.wc0 + cw, 3 * d, 0, 0, 0);
    convw(P[m.tail(SP1W)], w.wc1, 3 * d, 0, 0, 0);
    convw(P[m.tail(CL1W)], w.wc1 + cw, 3 * d, 0, 0, 0);
  } else {        // tap-major forward operands [n][tap * d + c] as split images (rows of 2 x 3d)
    split(P[m.tail(SP0W)], w.wc0S, d, 3 * d, 3 * d, d);
    split(P[m.tail(CL0W)], w.wc0S + 2 * cw, d, 3 * d, 3 * d, d);
    split(P[m.tail(SP1W)], w.wc1S, d, 3 * d, 3 * d, d);
    split(P[m.tail(CL1W)], w.wc1S + 2 * cw, d, 3 * d, 3 * d, d);
  }
  hipLaunchKernelGGL(concat2_kernel, dim3(cdiv(d, 256)), dim3(256), 0, s, P[m.tail(SP0B)], P[m.tail(CL0B)], w.bc0, P[m.tail(SP1B)], P[m.tail(CL1B)], w.bc1, d);
  UVTG_CHECK_LAUNCH();
  if (tr) {
    // dgrad operands: conv0 merged [d, 3 * 2d] (taps flipped), conv1 per head [d, 3d]
    convw(P[m.tail(SP0W)], w.wc0T, 6 * d, 2 * d, 0, 1);
    convw(P[m.tail(CL0W)], w.wc0T, 6 * d, 2 * d, d, 1);
    convw(P[m.tail(SP1W)], w.wc1T, 3 * d, d, 0, 1);
    convw(P[m.tail(CL1W)], w.wc1T + cw, 3 * d, d, 0, 1);
  }
  // input projections: block 0 is zero-padded to a multiple of 64 columns (Dv = 2818 -> 2880), later blocks are d x d
  if (4 * m.c.E + 4 + 4 * m.nproj > UVTG_MAX_PREP_OPS) return -17;
  for (int wm = 0; wm < 2; wm++) {
    const float* W0 = P[m.proj(wm, 0, PW)];
    const int D0 = m.din(wm, 0), K0 = m.kp(wm, 0);
    for (int b = 0; b < m.nproj; b++)
      if (w.pwS[wm][b]) split(P[m.proj(wm, b, PW)], w.pwS[wm][b], d, m.din(wm, b), m.kp(wm, b));
    if (w.pwB[wm][0] && wm == 1)       // (both modalities' first blocks in one launch)
      TRY(launch_cast_pad2_bf16(P[m.proj(0, 0, PW)], d, m.c.Dv, w.pwB[0][0], m.Kpv, P[m.proj(1, 0, PW)], d, m.c.Dt, w.pwB[1][0], m.Kpt, s));
    if (tr) {
      // the padding rows [D0, K0) of the transposed operand are zero: written by the transpose launch itself (cols_pad)
      transp(W0, d, D0, w.pwT[wm][0], d, nullptr, K0);
    }
    for (int b = 1; b < m.nproj; b++) {
      const float* Wb = P[m.proj(wm, b, PW)];
      if (tr) transp(Wb, d, d, w.pwT[wm][b], d, w.pwB[wm][b]);       // (pwB is null with split-operand projections: no plain copy then)
      else if (w.pwB[wm][b]) cast(Wb, w.pwB[wm][b], (long long)d * d);
    }
  }
  TRY(launch_split_f16_multi(so, s));
  TRY(launch_cast_bf16_multi(co, s));
  TRY(launch_transpose_bf16_multi(to, s));
  TRY(launch_conv_w_multi(cv, s));
  return 0;
}

// =================================================================================================
// forward
// =================================================================================================
namespace {

struct Fwd {
  const Dm& m; const float* const* P; WCache& w; WSpace& ws; hipStream_t s;
  bool fast, tr, pp;
  bool packed = false; int Mrows = 0;     // packed (ragged) encoder stream: Mrows <= B * S rows (see misc.hip)
  bool halo = false; int Rf = 0;          // loss-only stream: ragged conv-head frames of Rf rows in all (else B * (Lv + 2))
  int Rv = 0;                             // packed: clip rows of the compact video input projection (pk.vin_*)
  bool clip = false;                      // last layer's FFN half on the clip rows only (last_layer_clip)
  // x3: split-operand launch.  The callers describe the operands in REAL columns; a split row holds two elements per real column (hi / lo
  // images interleaved in 32-column blocks, uvtg_common.h), so every K-side size and offset doubles.
  int run_gemm(GemmArgs& g, bool x3) {
    g.sk_slab = ws.sk_slab; g.sk_tickets = ws.sk_tickets; g.sk_cap_units = UVTG_SK_UNITS;
    if (!x3) return launch_gemm_nt_bf16(g, s);
    g.K *= 2; g.ktap *= 2; g.lda *= 2; g.ldb *= 2; g.gA *= 2; g.gB *= 2;
    return launch_gemm_nt_split(g, s);
  }
  void set_out(GemmArgs& g, void* p, int ld) { if (fast) { g.outB = (bf16_t*)p; g.ldoB = ld; } else { g.outF = (float*)p; g.ldoF = ld; } }
  // precise mode: the output as a split row (the next GEMM's operand), `ld` real columns
  void set_split(GemmArgs& g, void* p, int ld) { g.outS = (unsigned short*)p; g.ldoS = 2 * ld; }

  // one modality of the input projection (model/univtg.py:91-100,399-406) -> rows of x0 / xb[0] / ub[0]
  int project(int which, const float* src, float* x0) {
    // packed stream: the video projection runs on the clips that HAVE a packed row only (compact rows, tables pk.vin_*: the feature
    // LayerNorm gathers them from src, dropout counters stay keyed by the padded row), and both modalities write x / x + pos
    // straight into the packed layer-0 operands (text rows through pk.tin_dst); x0 keeps the padded layout.
    // n_input_proj blocks of LayerNorm -> Dropout -> Linear, ReLU after every block but the last (model/univtg.py:89-100)
    const bool cv = packed && which == 0;
    const int R = which == 0 ? (packed ? Rv : m.Mv) : m.Mt;
    const int L = which == 0 ? m.c.Lv : m.c.Lt, d = m.c.d, nb = m.nproj;
    const unsigned rs = which == 0 ? UVTG_RNG_IN_VID : UVTG_RNG_IN_TXT;
    const float p_in = tr ? m.c.p_in : 0.f;
    for (int b = 0; b < nb; b++) {
      const int Din = m.din(which, b), Kp = m.kp(which, b);
      const bool lastb = b == nb - 1;
      LnFwdArgs ln; memset(&ln, 0, sizeof(ln));
      if (cv) { ln.src_rows = ws.pk.vin_src; ln.gather_x = b == 0; }
      ln.x = b == 0 ? src : ws.ph[which][b - 1]; ln.ldx = Din; ln.rows = R; ln.D = Din;
      ln.gamma = P[m.proj(which, b, PG)]; ln.beta = P[m.proj(which, b, PBE)]; ln.eps = 1e-5f;
      ln.mean = ws.pm[which][b]; ln.rstd = ws.pr[which][b]; ln.p_drop = p_in; ln.seed = m.c.seed; ln.stream_id = rs + b; ln.Dpad = Kp;
      if (pp) { ln.yS = (unsigned short*)ws.pa[which][b]; ln.ldyS = 2 * Kp; ln.sscale = UVTG_SPLIT_A_SCALE; ln.yB = ws.paB[which][b]; ln.ldyB = Kp; }
      else { ln.yB = (bf16_t*)ws.pa[which][b]; ln.ldyB = Kp; }
      TRY(launch_ln_fwd(ln, s));
      const void* W = pp ? (const void*)w.pwS[which][b] : (const void*)w.pwB[which][b];
      GemmArgs g = gemm_base(ws.pa[which][b], Kp, W, Kp, R, d, Kp);
      g.bias = P[m.proj(which, b, PB)];
      if (!lastb) {
        g.act = 1; g.outF = ws.ph[which][b]; g.ldoF = d;
        TRY(run_gemm(g, pp));
        continue;
      }
      g.bias2 = P[m.tail(TOK)] + (which == 0 ? d : 0);          // token-type row 1 = video, 0 = text (univtg.py:114-115)
      if (cv) { g.o_rows = ws.pk.vin_dst; g.f_rows = ws.pk.vin_x0; g.pos_map = ws.pk.vin_src; }
      else { g.o_seg = L; g.o_seg_stride = m.S; g.o_off = which == 0 ? 0 : m.c.Lv; if (packed) g.o_rows = ws.pk.tin_dst; }
      g.outF = x0; g.ldoF = d;
      if (fast) { g.outB = packed ? ws.xb0p : (bf16_t*)ws.xb[0]; g.ldoB = d; g.outU = packed ? ws.ub0p : (bf16_t*)ws.ub[0]; }
      else { set_split(g, ws.xb[0], d); g.outUS = (unsigned short*)ws.ub[0]; }       // precise: x and x + pos as split images
      g.ldoU = d;
      if (which == 0) { g.pos = ws.pos; g.ldpos = d; g.pos_rows = R; }
      TRY(run_gemm(g, pp));
    }
    return 0;
  }

  // --use_txt_pos (model/univtg.py:123, model/position_encoding.py:19-41): pos of the text rows = Dropout(LayerNorm(x0_txt + E[0..Lt))), the
  // same tensor in every layer.  One LayerNorm launch over the projected text rows of x0 writes it behind the clip rows of the pos table and
  // rewrites the text rows of layer 0's q,k operand as x + pos; later layers pick it up through the pos_row table.
  int text_positions(const float* x0) {
    const int d = m.c.d;
    hipLaunchKernelGGL(txt_pos_tables_kernel, dim3(cdiv(m.M, 256)), dim3(256), 0, s, m.c.B, m.c.Lv, m.c.Lt, ws.pos_row_all, ws.tp_src);
    UVTG_CHECK_LAUNCH();
    LnFwdArgs ln; memset(&ln, 0, sizeof(ln));
    ln.x = x0; ln.ldx = d; ln.rows = m.Mt; ln.D = d; ln.Dpad = d; ln.src_rows = ws.tp_src; ln.gather_x = 1;
    ln.addtab = P[m.txtpos(0)]; ln.add_L = m.c.Lt; ln.xsum = ws.tp_xsum;
    ln.gamma = P[m.txtpos(1)]; ln.beta = P[m.txtpos(2)]; ln.eps = 1e-5f; ln.mean = ws.tp_mean; ln.rstd = ws.tp_rstd;
    ln.p_drop = tr ? m.c.p_in : 0.f; ln.seed = m.c.seed; ln.stream_id = UVTG_RNG_TXT_POS;
    ln.yF = ws.pos_txt; ln.ldyF = d;
    ln.u_from_x = 1; ln.ldyU = d;
    if (fast) ln.yU = (bf16_t*)ws.ub[0];
    else { ln.yUS = (unsigned short*)ws.ub[0]; ln.ldyS = 2 * d; ln.sscale = UVTG_SPLIT_A_SCALE; }
    return launch_ln_fwd(ln, s);
  }

  int layer(int l, float* memory_out) {
    const int d = m.c.d, F = m.c.F, M = packed ? Mrows : m.M, S = m.S;
    const void* xb_in = (packed && l == 0) ? (const void*)ws.xb0p : ws.xb[l];
    const void* ub_in = (packed && l == 0) ? (const void*)ws.ub0p : ws.ub[l];
    const bool last = l == m.c.E - 1;
    const void* Wqkv = fast ? (const void*)w.wqkv[l] : (const void*)w.wqkvS[l];
    const size_t es = fast ? 2 : 4;          // bytes per element of the operand rows (precise: two fp16 images)
    // q,k from (x + pos); v from x  (transformer_encoder_droppath.py:116-117)
    const bool one_launch = (2 * d) % 256 == 0;      // A-operand switch at a tile boundary: q,k columns read x + pos, v columns read x
    GemmArgs g = gemm_base(ub_in, d, Wqkv, d, M, one_launch ? 3 * d : 2 * d, d);
    if (one_launch) { g.A2 = xb_in; g.a2_n0 = 2 * d; }
    g.bias = P[m.lay(l, IPB)]; g.colscale = 1.0f / sqrtf((float)m.hd); g.colscale_n = d;
    set_out(g, ws.qkv[l], 3 * d);            // (precise: fp32 q | k | v -- the attention kernel splits them itself)
    TRY(run_gemm(g, !fast));
    if (!one_launch) {
      g = gemm_base(xb_in, d, (const char*)Wqkv + (size_t)2 * d * d * es, d, M, d, d);
      g.bias = P[m.lay(l, IPB)] + 2 * d;
      set_out(g, (char*)ws.qkv[l] + (size_t)2 * d * es, 3 * d);
      TRY(run_gemm(g, !fast));
    }
    AttnArgs at; memset(&at, 0, sizeof(at));
    at.qkv = ws.qkv[l]; at.ldqkv = 3 * d; at.lse = ws.lse[l]; at.kvalid = packed ? ws.pk.kvalid : ws.kvalid;
    if (fast) { at.o = ws.o[l]; at.ldo = d; }
    else { at.oS = (unsigned short*)ws.o[l]; at.ldoS = 2 * d; at.ldo = d; }      // precise: o leaves as a split row
    if (packed) { at.seq_start = ws.pk.seq_start; at.seq_count = ws.pk.seq_count; at.row_sample = ws.pk.row_sample; at.total_rows = M; }
    at.B = m.c.B; at.S = S; at.H = m.c.H; at.hd = m.hd; at.p_drop = tr ? m.c.p_attn : 0.f; at.seed = m.c.seed; at.layer = l;
    at.precise = !fast;
    TRY(launch_attn_fwd(at, s));
    // out-proj + DropPath + residual -> y1 ; LN1
    g = gemm_base(ws.o[l], d, fast ? (const void*)w.wo[l] : (const void*)w.woS[l], d, M, d, d);
    g.bias = P[m.lay(l, OPB)];
    if (fast) { g.residB = (const bf16_t*)xb_in; g.ldrB = d; g.outB = ws.y1b[l]; g.ldoB = d; }
    else { g.resid = ws.xin[l]; g.ldr = d; g.outF = ws.y1[l]; g.ldoF = d; }
    if (tr && m.c.p_path > 0.f) { g.rowscale = ws.dps + (size_t)(2 * l) * m.c.B; g.rs_seg = S; if (packed) g.row_sample = ws.pk.row_sample; }
    TRY(run_gemm(g, !fast));
    // (last layer, clip mode: from here on the rows are the B Lv clip rows, compact -- see last_layer_clip)
    const bool cl = last && clip;
    const int Mf = cl ? (packed ? Rv : m.Mv) : M, Sf = cl ? m.c.Lv : S;
    LnFwdArgs ln; memset(&ln, 0, sizeof(ln));
    ln.rows = Mf; ln.D = d; ln.gamma = P[m.lay(l, N1W)]; ln.beta = P[m.lay(l, N1B)]; ln.eps = 1e-5f;
    ln.mean = ws.mean1[l]; ln.rstd = ws.rstd1[l]; ln.Dpad = d;
    if (cl) { if (packed) ln.x_rows = ws.pk.vin_dst; else { ln.x_seg = m.c.Lv; ln.x_seg_stride = S; } }
    if (fast) { ln.xB = ws.y1b[l]; ln.ldxB = d; ln.yB = (bf16_t*)ws.x1b[l]; ln.ldyB = d; }
    else { ln.x = ws.y1[l]; ln.ldx = d; ln.yF = ws.x1; ln.ldyF = d; ln.yS = (unsigned short*)ws.x1b[l]; ln.ldyS = 2 * d; ln.sscale = UVTG_SPLIT_A_SCALE; }
    TRY(launch_ln_fwd(ln, s));
    // FFN: linear1 + GELU, linear2 + DropPath + residual -> y2 ; LN2
    g = gemm_base(ws.x1b[l], d, fast ? (const void*)w.w1[l] : (const void*)w.w1S[l], d, Mf, F, d);
    g.bias = P[m.lay(l, L1B)]; g.act = 2;
    if (tr) { g.outPre = ws.apre[l]; g.ldpre_out = F; }
    if (fast) set_out(g, ws.h[l], F); else set_split(g, ws.h[l], F);
    TRY(run_gemm(g, !fast));
    g = gemm_base(ws.h[l], F, fast ? (const void*)w.w2[l] : (const void*)w.w2S[l], F, Mf, d, F);
    g.bias = P[m.lay(l, L2B)];
    if (fast) { g.residB = (const bf16_t*)ws.x1b[l]; g.ldrB = d; g.outB = ws.y2b[l]; g.ldoB = d; }
    else { g.resid = ws.x1; g.ldr = d; g.outF = ws.y2[l]; g.ldoF = d; }
    if (tr && m.c.p_path > 0.f) { g.rowscale = ws.dps + (size_t)(2 * l + 1) * m.c.B; g.rs_seg = Sf; if (packed) g.row_sample = cl ? ws.pk.vin_sample : ws.pk.row_sample; }
    TRY(run_gemm(g, !fast));
    memset(&ln, 0, sizeof(ln));
    ln.rows = Mf; ln.D = d; ln.gamma = P[m.lay(l, N2W)]; ln.beta = P[m.lay(l, N2B)]; ln.eps = 1e-5f;
    ln.mean = ws.mean2[l]; ln.rstd = ws.rstd2[l]; ln.Dpad = d; ln.S = Sf; ln.Lv = m.c.Lv;
    if (fast) { ln.xB = ws.y2b[l]; ln.ldxB = d; } else { ln.x = ws.y2[l]; ln.ldx = d; }
    if (packed && cl) { ln.S = 0; ln.yB_rows = ws.pk.vin_dst; }      // (compact clip rows in, the packed stream's rows out: launch_unpack_vm below)
    else if (packed) ln.pos_row = ws.pk.row_pos;
    else if (m.c.use_txt_pos && !last) ln.pos_row = ws.pos_row_all;      // text rows add their trainable positions too (the last layer has no next q,k operand)
    if (!fast) { ln.ldyS = 2 * d; ln.sscale = UVTG_SPLIT_A_SCALE; }
    if (!last) {
      if (!fast) { ln.yF = ws.xin[l + 1]; ln.ldyF = d; }
      ln.pos = ws.pos; ln.ldyU = d;
      if (fast) { ln.yB = (bf16_t*)ws.xb[l + 1]; ln.ldyB = d; ln.yU = (bf16_t*)ws.ub[l + 1]; }
      else { ln.yS = (unsigned short*)ws.xb[l + 1]; ln.yUS = (unsigned short*)ws.ub[l + 1]; }
    } else if (packed) {
      ln.yB = (bf16_t*)ws.xb[l + 1]; ln.ldyB = d;             // packed encoder output; expanded into the conv frame below
    } else {
      if (memory_out) { ln.yF = memory_out; ln.ldyF = d; }
      ln.ldyP = d;
      if (fast) ln.yP = (bf16_t*)ws.vm_pad; else ln.yPS = (unsigned short*)ws.vm_pad;
    }
    TRY(launch_ln_fwd(ln, s));
    if (last && packed) TRY(launch_unpack_vm((const bf16_t*)ws.xb[l + 1], ws.pk, halo, m.c.B, S, m.c.Lv, d, (bf16_t*)ws.vm_pad, s));
    return 0;
  }

  int heads(float* pred_logits, float* pred_spans, HeadsFinalArgs& hf) {
    const int d = m.c.d, Lv = m.c.Lv, B = m.c.B;
    const size_t es = fast ? 2 : 4;
    const int* fs = halo ? ws.pk.fstart : nullptr;
    const int* kc = halo ? ws.pk.kept : nullptr;
    {   // zero rows of the conv frames, one launch (packed stream: unpack_vm wrote vm_pad's; ragged frames: the epilogue's 0/1 row factor re-establishes h1 / h2's)
      ZeroFrames z; memset(&z, 0, sizeof(z));
      int n = 0;
      if (!packed) { z.p[n] = (char*)ws.vm_pad; z.row_bytes[n] = (int)(d * es); z.fstart[n] = fs; z.kept[n] = kc; n++; }
      if (!halo) {
        z.p[n] = (char*)ws.h1_pad; z.row_bytes[n] = (int)(2 * d * es); n++;
        z.p[n] = (char*)ws.h2_pad; z.row_bytes[n] = (int)(2 * d * es); n++;
      }
      TRY(zero_frames(z, n, B, Lv, s));
    }
    // Frame addressing of the 3-tap GEMMs.  Uniform frames: output row m = (b, t) reads frame rows b (Lv + 2) + t + tap and lands on
    // frame row b (Lv + 2) + t + 1 (the zero rows are never written).  Ragged frames (loss-only stream): the GEMM runs over ALL frame
    // rows, row f reads f - 1 + tap, and the zero rows are re-established by the 0/1 row factor of the epilogue.
    auto frame = [&](GemmArgs& g) {
      if (halo) { g.M = Rf; g.a_off = -1; g.rowscale = ws.pk.frame_valid; g.rs_seg = 1; }
      else { g.a_seg = Lv; g.a_seg_stride = Lv + 2; g.a_off = 0; g.o_seg = Lv; g.o_seg_stride = Lv + 2; g.o_off = 1; }
    };
    // conv layer 0 of both heads as one 3-tap GEMM, N = 2d (model/univtg.py:84-85,375-382)
    GemmArgs g = gemm_base(ws.vm_pad, d, fast ? (const void*)w.wc0 : (const void*)w.wc0S, 3 * d, m.Mv, 2 * d, 3 * d);
    g.ktap = d; frame(g);
    g.bias = w.bc0; g.act = 1;
    if (fast) set_out(g, ws.h1_pad, 2 * d); else set_split(g, ws.h1_pad, 2 * d);
    TRY(run_gemm(g, !fast));
    // conv layer 1: two groups (span | class), each d -> d
    g = gemm_base(ws.h1_pad, 2 * d, fast ? (const void*)w.wc1 : (const void*)w.wc1S, 3 * d, m.Mv, d, 3 * d);
    g.ktap = d; frame(g);
    g.groups = 2; g.gA = d; g.gB = (long long)d * 3 * d; g.gBias = d; g.gOut = d;
    g.bias = w.bc1; g.act = 1;
    set_out(g, ws.h2_pad, 2 * d);
    TRY(run_gemm(g, !fast));
    memset(&hf, 0, sizeof(hf));                      // (the last conv layer runs fused with the saliency pass: launch_heads_saliency_fwd)
    hf.h2 = ws.h2_pad; hf.ldh = 2 * d; hf.precise = !fast; hf.fstart = fs; hf.kept = kc;
    hf.w_span = P[m.tail(SP2W)]; hf.b_span = P[m.tail(SP2B)]; hf.w_cls = P[m.tail(CL2W)]; hf.b_cls = P[m.tail(CL2B)];
    hf.B = B; hf.Lv = Lv; hf.d = d; hf.pred_logits = pred_logits; hf.pred_spans = pred_spans;
    return 0;
  }
};

SaliencyArgs sal_args(const Dm& m, const float* const* P, WSpace& ws, const float* x0, const float* tmask, const float* vmask,
                      float* pooled, float* sal) {
  SaliencyArgs a; memset(&a, 0, sizeof(a));
  a.x0 = x0; a.S = m.S; a.Lv = m.c.Lv; a.Lt = m.c.Lt; a.B = m.c.B; a.d = m.c.d; a.txt_mask = tmask; a.vid_mask = vmask;
  a.w_pool = P[m.pool()]; a.alpha = ws.alpha; a.pooled = pooled; a.cosv = ws.cosv; a.sal = sal; a.vnorm = ws.vnorm; a.qnorm = ws.qnorm;
  return a;
}

}  // namespace

extern "C" int uvtg_forward(const uvtg_dims* dm, const float* const* P, const void* wcache,
                            const float* src_txt, const float* src_txt_mask, const float* src_vid, const float* src_vid_mask,
                            const float* dim_t, float* x0, float* pred_logits, float* pred_spans, float* txt_mem_proj,
                            float* saliency, float* memory, void* workspace, uvtg_stream_t stream, const int* lens_host) {
  if (int e = check_dims(dm)) return e;
  if (dm->E > MAXE) return -17;
  if (!P || !wcache || !src_txt || !src_txt_mask || !src_vid || !src_vid_mask || !dim_t || !x0 || !pred_logits ||
      !pred_spans || !txt_mem_proj || !saliency || !workspace) return -20;
  hipStream_t s = (hipStream_t)stream;
  Dm m(*dm);
  WCache w(m, (void*)wcache);
  WSpace ws(m, workspace, x0);
  Fwd f{m, P, w, ws, s, !m.c.precise, m.c.training != 0, m.c.precise || m.c.proj_precise};
  int pmode = pack_mode(m.c, lens_host);
  if (pmode != PACK_NONE && memory) {           // the packed stream has no [B, S, d] encoder output to hand out
    if (m.c.training) return -24;               // (uvtg_backward could not know: refuse instead of silently diverging from it)
    pmode = PACK_NONE;
  }
  f.clip = last_layer_clip(m);
  if (f.clip && memory) {                       // the clip-row tail of the last layer has no text rows of the encoder output to hand out either
    if (m.c.training) return -24;
    f.clip = false;
  }
  if (m.c.training) clip_record(workspace, f.clip);
  if (pmode != PACK_NONE) {                     // packed (ragged) encoder stream
    int mp = 0;
    TRY(packed_rows(m, lens_host, pmode, &mp));
    f.packed = true; f.Mrows = mp;
    f.halo = pmode == PACK_HALO; f.Rf = f.halo ? halo_frame_rows(m, lens_host) : m.Rp;
    f.Rv = compact_clip_rows(m, lens_host, pmode);
    // the device-side tables are built from the MASKS (no copy out of the caller's pageable lens_host, which is only read here,
    // synchronously, for the row count): lens_host must be the masks' prefix lengths
    TRY(launch_pack_tables(src_vid_mask, src_txt_mask, ws.lens_dev, m.c.B, m.c.Lv, m.c.Lt, pmode == PACK_TEXT ? m.c.Lv : (pmode == PACK_HALO ? HALO : -1),
                           ws.pk, s));
  }
  uvtg_prof_section(2, 0, s);
  {
    const bool dp = f.tr && m.c.p_path > 0.f;       // (the DropPath factors of the step are drawn by the same launch)
    TRY(launch_seq_prep(src_vid_mask, src_txt_mask, m.c.B, m.c.Lv, m.c.Lt, m.c.d, dim_t, ws.pos, ws.kvalid, f.packed ? ws.pk.vin_of : nullptr, s,
                        dp ? ws.dps : nullptr, 2 * m.c.E * m.c.B, m.c.p_path, m.c.seed, ws.sk_tickets, UVTG_SK_UNITS));
  }
  TRY(f.project(0, src_vid, x0));
  if (f.packed && f.Rv < m.Mv) TRY(launch_fill_dropped_rows(x0, ws.pk, pmode == PACK_FULL, m.c.B, m.S, m.c.Lv, m.c.d, s));
  TRY(f.project(1, src_txt, x0));
  if (m.c.use_txt_pos) TRY(f.text_positions(x0));
  uvtg_prof_section(0, 0, s);
  for (int l = 0; l < m.c.E; l++) TRY(f.layer(l, memory));
  uvtg_prof_section(0, 1, s);
  HeadsFinalArgs hf;
  TRY(f.heads(pred_logits, pred_spans, hf));
  SaliencyArgs sa = sal_args(m, P, ws, x0, src_txt_mask, src_vid_mask, txt_mem_proj, saliency);
  TRY(launch_heads_saliency_fwd(hf, sa, s));
  uvtg_prof_section(2, 1, s);
  return 0;
}

// =================================================================================================
// backward
// =================================================================================================
extern "C" int uvtg_backward(const uvtg_dims* dm, const float* const* P, const void* wcache,
                             const float* src_txt, const float* src_txt_mask, const float* src_vid, const float* src_vid_mask,
                             const float* x0, const float* pred_logits, const float* pred_spans, const float* txt_mem_proj,
                             const float* g_logits, const float* g_spans, const float* g_saliency,
                             const float* g_txt_mem, const float* g_vid_mem, long long g_vid_sb, long long g_vid_st,
                             const float* g_vrow, const long long* pos_idx,
                             float* grads, void* workspace, uvtg_stream_t stream, void* const* ready_events, int n_events,
                             const int* lens_host) {
  if (int e = check_dims(dm)) return e;
  if (dm->E > MAXE) return -17;
  if (dm->precise || !dm->training) return -6;
  if (n_events != 0 && (n_events != dm->E + 1 || !ready_events)) return -22;
  if (!P || !wcache || !x0 || !pred_logits || !pred_spans || !txt_mem_proj || !grads || !workspace || !src_txt || !src_vid ||
      !src_txt_mask || !src_vid_mask) return -20;
  hipStream_t s = (hipStream_t)stream;
  Dm m(*dm);
  WCache w(m, (void*)wcache);
  WSpace ws(m, workspace, (float*)x0);
  const int d = m.c.d, F = m.c.F, S = m.S, Lv = m.c.Lv, B = m.c.B, E = m.c.E;
  // packed (ragged) encoder stream: must match the forward call (same lens_host); the tables are still in the workspace
  const int pmode = pack_mode(m.c, lens_host);
  const bool packed = pmode != PACK_NONE;
  int M = m.M;
  if (packed) TRY(packed_rows(m, lens_host, pmode, &M));
  long long off[PER_LAYER * MAXE + N_FIXED + 2 * PER_PROJ * MAXP + 4 + 1];
  { long long o = 0; for (int i = 0; i < m.np; i++) { off[i] = o; o += (pnumel(m, i) + 3) / 4 * 4; } off[m.np] = o; }
  auto G = [&](int idx) { return grads + off[idx]; };
  uvtg_prof_section(3, 0, s);
  ZeroRanges zr_keep; zr_keep.count = 0;
  // The weight-gradient launches ASSIGN their matrices (99.6 % of the buffer: no zero fill, no read-modify-write in the reduce pass);
  // every other gradient (biases, LayerNorm, token-type rows, pooling vector, the heads' last layer) is accumulated into zeros.
  {
    ZeroRanges zr; zr.count = 0;
    for (int i = 0; i < m.np; i++) {
      if (assigned_matrix(m, i)) continue;
      if (zr.count >= UVTG_MAX_ZERO_RANGES) return -17;
      zr.off[zr.count] = off[i]; zr.n[zr.count] = (int)(off[i + 1] - off[i]); zr.count++;
    }
    // (the same launch zeroes the clipping-norm slots and the hybrid weight-gradient launch's tickets: was a memset)
    // (... and, on uniform conv-head frames, dh1_pad's zero rows: was a launch of its own)
    const bool uf = pmode != PACK_HALO;
    TRY(launch_zero_ranges(grads, zr, s, ws.gnorm2, UVTG_SQSUM_FLOATS + ws.tnh_n_tickets + (int)ws.delta_floats, uf ? ws.dh1_pad : nullptr, B, Lv, 2 * d * 2));
    zr_keep = zr;
  }
  const int splits_M = 8, splits_v = 8;
  auto wgrad = [&](const bf16_t* Pm, int ldp, const bf16_t* Q, int ldq, int rows, int N, int K, float* out, int ldo, int cs,
                   float* dbias, int q_off, int Mq, int splits) {
    GemmTNArgs t; memset(&t, 0, sizeof(t));
    t.P = Pm; t.ldp = ldp; t.Q = Q; t.ldq = ldq; t.M = rows; t.N = N; t.K = K; t.q_row_off = q_off; t.Mq = Mq;
    t.out = out; t.ldo = ldo; t.col_stride = cs; t.dbias = dbias; t.splits = splits; t.assign = cs == 1; t.sqsum = ws.gnorm2;
    t.scratch = ws.tn_scratch; t.scratch_floats = ws.tn_scratch_floats;
    return launch_gemm_tn_bf16(t, s);
  };

  // several weight gradients over the same rows in one launch (more tiles per launch -> fewer M splits -> less partial traffic)
  auto tn_group = [&](const bf16_t* Pm, int ldp, const bf16_t* Q, int ldq, int rows, int N, int K, float* out, int ldo, float* dbias) {
    GemmTNArgs t; memset(&t, 0, sizeof(t));
    t.P = Pm; t.ldp = ldp; t.Q = Q; t.ldq = ldq; t.M = rows; t.N = N; t.K = K; t.Mq = rows;
    t.out = out; t.ldo = ldo; t.col_stride = 1; t.dbias = dbias; t.splits = splits_M; t.assign = 1; t.sqsum = ws.gnorm2;
    t.scratch = ws.tn_scratch; t.scratch_floats = ws.tn_scratch_floats;
    return t;
  };
  auto tn_batch = [&](const GemmTNBatch& b) -> int {
    if (gemm_tn_batch_ok(b)) return launch_gemm_tn_batch(b, s);
    for (int i = 0; i < b.count; i++) TRY(launch_gemm_tn_bf16(b.g[i], s));      // small / odd shapes: one launch each
    return 0;
  };
  // Encoder weight gradients.  Without per-layer readiness events (single rank: nobody waits for a layer's gradients) they are DEFERRED: every
  // layer's operands stay in their own buffers and all 5 E gradients go out as ONE launch behind the encoder loop -- 384 tiles over the same
  // rows at config 2, whole tiles per workgroup + half tiles for the remainder, no partial-slab reduce pass (gemm.hip, gemm_tn256h_kernel).
  // With events (data-parallel overlap) each layer's two batches run in place, as before, and their gradients are final at the event.
  static const bool defer_off = uvtg_dev_env("UVTG_TN_DEFER_OFF") != nullptr;
    static const bool defer_events = uvtg_dev_env("UVTG_TN_EVENT_GROUPS") == nullptr;      // default: ONE launch behind the loop, every event there
  // Round 6: the deferral stays under events.  Default: the ONE hybrid launch of the single-rank step, every event recorded behind it -- the
  // N > 1 step then runs exactly the single-rank step's kernels (+1.2 % on one rank with the coalesced exchange, bench.py --overlap force) and
  // the gradient exchange of the heads + encoder (150 MB) overlaps the saliency branch and the input projections (~1 ms).
  // UVTG_TN_EVENT_GROUPS=1: TWO groups -- the conv heads and layers E-1 .. 1 go out as one hybrid launch (and one LayerNorm fold) right behind
  // layer 1's dgrad, their E events recorded together there, layer 0's own group follows the loop: +2.8 % compute for ~2.3 ms of cover (the
  // choice for a slow interconnect).  UVTG_TN_EVENTS_PER_LAYER=1: the per-layer slab + reduce batches of rounds 2-5 (+5.3 %).
  static const bool events_per_layer = uvtg_dev_env("UVTG_TN_EVENTS_PER_LAYER") != nullptr;
  const bool defer = (n_events == 0 || !events_per_layer) && !defer_off;
  const int flush_layer = (n_events && defer && !defer_events && E >= 2) ? 1 : -1;      // events: group A is flushed behind this layer (uvtg_backward_event_groups mirrors this)
  int events_done = 0;                           // ready_events[0 .. events_done) are recorded
  GemmTNBatch deferred[2 * MAXE + 1]; int n_deferred = 0;
  auto tn_encoder = [&](const GemmTNBatch& b) -> int {
    if (!defer) return tn_batch(b);
    deferred[n_deferred++] = b;
    return 0;
  };
  auto tn_flush = [&]() -> int {
    if (!n_deferred) return 0;
    GemmTNMulti mu; mu.count = 0; mu.slabs = ws.tnh_slabs; mu.slab_floats = ws.tnh_slab_floats; mu.tickets = ws.tnh_tickets; mu.n_tickets = ws.tnh_n_tickets;
    bool fits = true;
    for (int i = 0; i < n_deferred && fits; i++)
      for (int j = 0; j < deferred[i].count; j++) {
        if (mu.count >= UVTG_TNH_MAX_GROUPS) { fits = false; break; }
        mu.g[mu.count++] = deferred[i].g[j];
      }
    // (the clip-row groups of the last layer reduce over fewer rows: their tiles end early; moving them behind the others -- among the split tiles --
    // measured the same, profiles/r05_ab_last_layer_clip_rows.txt)
    if (fits) {      // groups over fewer rows first (stable): their tiles land on the workgroups that also carry a split part (gemm.hip, hybrid plan)
      for (int i = 1; i < mu.count; i++) {
        const GemmTNArgs t = mu.g[i]; int j = i - 1;
        while (j >= 0 && mu.g[j].M > t.M) { mu.g[j + 1] = mu.g[j]; j--; }
        mu.g[j + 1] = t;
      }
    }
    if (fits && gemm_tn_multi_ok(mu)) return launch_gemm_tn_multi(mu, s);
    if (fits) {
      // groups over too few rows for the hybrid kernel (mid-size batches: B Lv < 2048 <= B S puts the last layer's clip-row FFN groups and the
      // conv-head groups below its row floor) leave on their own launches; the long groups keep the hybrid launch
      GemmTNMulti lng = mu; lng.count = 0;
      GemmTNArgs shrt[UVTG_TNH_MAX_GROUPS]; int n_short = 0;
      for (int i = 0; i < mu.count; i++) { if (mu.g[i].M >= 2048) lng.g[lng.count++] = mu.g[i]; else shrt[n_short++] = mu.g[i]; }
      bool short_ok = true;       // (a conv-tap group below the 256-tile kernel's row floor has no single launch: the whole set then takes the fallback below)
      for (int i = 0; i < n_short; i++) short_ok = short_ok && (shrt[i].ktap == 0 || gemm_tn_taps_ok(shrt[i]));
      if (n_short && lng.count && short_ok && gemm_tn_multi_ok(lng)) {
        TRY(launch_gemm_tn_multi(lng, s));
        for (int i = 0; i < n_short; i++) TRY(launch_gemm_tn_bf16(shrt[i], s));
        return 0;
      }
    }
    for (int i = 0; i < n_deferred; i++) TRY(tn_batch(deferred[i]));            // shapes the hybrid launch does not take: the split + reduce path
    return 0;
  };
  auto tn_flush_group = [&]() -> int { const int r = tn_flush(); n_deferred = 0; return r; };
  // (The conv-head and input-projection gradients stay on the batched slab + reduce launches.  A stream-K generalisation of the hybrid kernel
  // -- tiles of all groups end to end, equal pieces per workgroup, conv taps and ragged K in its epilogue -- was built and measured in round 3:
  // correct, and SLOWER on both launches (encoder 1.27 vs 0.89 ms, tail 0.71 vs 0.61 ms): cutting every third tile breaks up the sets of tiles
  // that share a dY / X row panel through one XCD's L2 at the same time, which the whole-groups-first plan keeps together.)
  // weight gradient of one Conv1d(k=3): dW[n][c][tap] = sum_rows dY[row][n] * X[row + tap - 1][c] over the zero-framed rows.
  // One launch over K = 3 d (the k tiles pick their tap's row offset) when the 256-tile kernel takes it, else one per tap.
  auto conv_wgrad = [&](const bf16_t* dY, int ldp, const bf16_t* X, int ldq, float* dW, float* dBi, int rows) -> int {
    GemmTNArgs t; memset(&t, 0, sizeof(t));
    t.P = dY; t.ldp = ldp; t.Q = X; t.ldq = ldq; t.M = rows; t.N = d; t.K = 3 * d; t.q_row_off = -1; t.Mq = rows;
    t.out = dW; t.ldo = 3 * d; t.col_stride = 3; t.dbias = dBi; t.splits = splits_v; t.ktap = d; t.assign = 1; t.sqsum = ws.gnorm2;
    t.scratch = ws.tn_scratch; t.scratch_floats = ws.tn_scratch_floats;
    if (gemm_tn_taps_ok(t)) return launch_gemm_tn_bf16(t, s);
    if (hipError_t e = hipMemsetAsync(dW, 0, (size_t)d * 3 * d * sizeof(float), s)) return (int)e;      // per-tap launches accumulate (stride-3 outputs)
    for (int tap = 0; tap < 3; tap++)
      TRY(wgrad(dY, ldp, X, ldq, rows, d, d, dW + tap, 3 * d, 3, tap == 1 ? dBi : nullptr, tap - 1, rows, splits_v));
    return launch_sqsum(dW, (long long)d * 3 * d, ws.gnorm2 + 32, s);
  };
  // ---------------- heads ----------------
  const bool halo = pmode == PACK_HALO;          // ragged conv-head frames (see Fwd::heads)
  const int Rf = halo ? halo_frame_rows(m, lens_host) : m.Rp;
  const int* fs = halo ? ws.pk.fstart : nullptr;
  const int* kc = halo ? ws.pk.kept : nullptr;
  auto frame = [&](GemmArgs& g, bool scatter_out) {
    if (halo) { g.M = Rf; g.a_off = -1; g.rowscale = ws.pk.frame_valid; g.rs_seg = 1; }
    else {
      g.a_seg = Lv; g.a_seg_stride = Lv + 2; g.a_off = 0;
      if (scatter_out) { g.o_seg = Lv; g.o_seg_stride = Lv + 2; g.o_off = 1; }
    }
  };
  // (dh1_pad's frame rows were zeroed by the zero-ranges launch above, dh2_pad's are by heads_final_bwd_dh itself)
  HeadsFinalArgs hf; memset(&hf, 0, sizeof(hf));
  hf.h2 = ws.h2_pad; hf.ldh = 2 * d; hf.w_span = P[m.tail(SP2W)]; hf.b_span = P[m.tail(SP2B)];
  hf.w_cls = P[m.tail(CL2W)]; hf.b_cls = P[m.tail(CL2B)]; hf.B = B; hf.Lv = Lv; hf.d = d; hf.fstart = fs; hf.kept = kc;
  hf.pred_logits = (float*)pred_logits; hf.pred_spans = (float*)pred_spans; hf.g_logits = g_logits; hf.g_spans = g_spans;
  hf.dh2 = ws.dh2_pad; hf.lddh = 2 * d;
  hf.dw_span = G(m.tail(SP2W)); hf.db_span = G(m.tail(SP2B)); hf.dw_cls = G(m.tail(CL2W)); hf.db_cls = G(m.tail(CL2B));
  hf.scratch = ws.tn_scratch; hf.scratch_floats = ws.tn_scratch_floats;
  TRY(launch_heads_final_bwd(hf, s));
  {                                             // conv layer 1 dgrad (+ relu' of h1) -> dh1_pad
    GemmArgs g = gemm_base(ws.dh2_pad, 2 * d, w.wc1T, 3 * d, m.Mv, d, 3 * d);
    g.ktap = d; frame(g, true);
    g.groups = 2; g.gA = d; g.gB = (long long)d * 3 * d; g.gOut = d; g.gPre = d;
    g.gradPre = (const bf16_t*)ws.h1_pad; g.ldgp = 2 * d; g.actgrad = 1;
    g.outB = ws.dh1_pad; g.ldoB = 2 * d;
    TRY(launch_gemm_nt_bf16(g, s));
  }
  bool conv_deferred = false;
  {   // weight gradients of the four d -> d convolutions (layer 1 and layer 0 of both heads): the same frame rows, ONE launch over 4 x 12
      // tap tiles (5 row splits instead of 4 launches of 12 tiles x 21 splits each) + one reduce pass; their operands all exist by now
    GemmTNBatch cb; cb.count = 0;
    struct { const bf16_t* dY; int ldp; const bf16_t* X; int ldq; int w, b; } cw[4] = {
      {ws.dh2_pad, 2 * d, (const bf16_t*)ws.h1_pad, 2 * d, SP1W, SP1B}, {ws.dh2_pad + d, 2 * d, (const bf16_t*)ws.h1_pad + d, 2 * d, CL1W, CL1B},
      {ws.dh1_pad, 2 * d, (const bf16_t*)ws.vm_pad, d, SP0W, SP0B}, {ws.dh1_pad + d, 2 * d, (const bf16_t*)ws.vm_pad, d, CL0W, CL0B}};
    bool all_ok = true;
    for (auto& c : cw) {
      GemmTNArgs t; memset(&t, 0, sizeof(t));
      t.P = c.dY; t.ldp = c.ldp; t.Q = c.X; t.ldq = c.ldq; t.M = Rf; t.N = d; t.K = 3 * d; t.q_row_off = -1; t.Mq = Rf;
      t.out = G(m.tail(c.w)); t.ldo = 3 * d; t.col_stride = 3; t.dbias = G(m.tail(c.b)); t.splits = splits_v; t.ktap = d; t.assign = 1; t.sqsum = ws.gnorm2;
      t.scratch = ws.tn_scratch; t.scratch_floats = ws.tn_scratch_floats;
      all_ok = all_ok && gemm_tn_taps_ok(t);
      cb.g[cb.count++] = t;
    }
    static const bool cbatch_off = uvtg_dev_env("UVTG_TN_CONVBATCH_OFF") != nullptr;
    // round 5: without readiness events they join the encoder's deferred launch (the hybrid kernel takes conv taps and the stride-3 weight
    // layout now: no slab + reduce pass, 1160 instead of 850 TFLOP/s) -- UVTG_TN_CONV_DEFER_OFF / uvtg_debug_tn_conv_defer(0): their own launch, as before
    if (g_conv_defer < 0) g_conv_defer = uvtg_dev_env("UVTG_TN_CONV_DEFER_OFF") ? 0 : 1;
    if (all_ok && !cbatch_off && defer && g_conv_defer == 1 && gemm_tn_batch_ok(cb)) { TRY(tn_encoder(cb)); conv_deferred = true; }      // (under events: ready_events[0] is recorded behind the group's launch)
    else if (all_ok && !cbatch_off && gemm_tn_batch_ok(cb)) TRY(launch_gemm_tn_batch(cb, s));
    else for (auto& c : cw) TRY(conv_wgrad(c.dY, c.ldp, c.X, c.ldq, G(m.tail(c.w)), G(m.tail(c.b)), Rf));
  }
  {                                             // conv layer 0 dgrad -> dvm (bf16; clip rows [B * Lv], or frame rows on the loss-only stream)
    GemmArgs g = gemm_base(ws.dh1_pad, 2 * d, w.wc0T, 6 * d, m.Mv, d, 6 * d);
    g.ktap = 2 * d; frame(g, false);
    g.outB = ws.dvmB; g.ldoB = d;
    TRY(launch_gemm_nt_bf16(g, s));
  }
  if (n_events && !conv_deferred) { if (hipError_t e = hipEventRecord((hipEvent_t)ready_events[0], s)) return (int)e; events_done = 1; }   // span_embed / class_embed gradients final
  // ---------------- encoder ----------------
  // The gradient stream is bf16 (like the activation stream): gin = gradient wrt the layer output; dyB = LayerNorm input
  // gradient scaled by the DropPath factor (operand of the branch GEMMs), dyR = the same unscaled (residual branch).
  const bf16_t* gin = nullptr;                  // null = zero
  // LayerNorm gamma / beta gradients of the encoder: like the weight gradients, without per-layer readiness events every launch keeps its
  // per-block partials in its own buffer and ONE launch folds all 2 E of them behind the loop (was: a 5 us reduce launch behind each)
  static const bool lnred_off = uvtg_dev_env("UVTG_LN_DEFER_OFF") != nullptr;
  LnReduceMulti lnm; memset(&lnm, 0, sizeof(lnm)); lnm.D = d;
  int ln_nb = 0;
  auto ln_partials = [&](LnBwdArgs& lb, int slot) {
    if (defer && !lnred_off && ws.ln_part[slot]) { lb.partial = ws.ln_part[slot]; lb.partial_floats = ws.ln_part_floats; lb.defer_blocks = &ln_nb; }
    else { lb.partial = ws.tn_scratch; lb.partial_floats = ws.tn_scratch_floats; }
  };
  auto ln_deferred = [&](const LnBwdArgs& lb) {
    if (!lb.defer_blocks || ln_nb <= 0) return;
    lnm.partial[lnm.count] = lb.partial; lnm.dgamma[lnm.count] = lb.dgamma; lnm.dbeta[lnm.count] = lb.dbeta; lnm.nblocks[lnm.count] = ln_nb; lnm.count++;
  };
  uvtg_prof_section(1, 0, s);
  if (packed) TRY(launch_pack_reduce_dvm(ws.dvmB, ws.pk, B, S, Lv, M, d, pmode == PACK_TEXT || pmode == PACK_HALO, pmode == PACK_HALO, ws.g2p, s));   // conv-head gradient onto the packed rows
  const int* row_sample = packed ? ws.pk.row_sample : nullptr;
  const int clip_rec = clip_recorded(workspace);     // the forward's choice, as the forward on THIS workspace recorded it
  const bool clip = clip_rec >= 0 ? clip_rec == 1 : last_layer_clip(m);
  const int Rv_clip = packed ? compact_clip_rows(m, lens_host, pmode) : m.Mv;
  for (int l = E - 1; l >= 0; l--) {
    const bf16_t* xb_in = (packed && l == 0) ? ws.xb0p : (const bf16_t*)ws.xb[l];
    const bf16_t* ub_in = (packed && l == 0) ? ws.ub0p : (const bf16_t*)ws.ub[l];
    const bool last = l == E - 1;
    const float* dp_attn = (m.c.p_path > 0.f) ? ws.dps + (size_t)(2 * l) * B : nullptr;
    const float* dp_ffn = (m.c.p_path > 0.f) ? ws.dps + (size_t)(2 * l + 1) * B : nullptr;
    bf16_t* const dy2 = ws.dy2L[l]; bf16_t* const dy1 = ws.dy1L[l]; bf16_t* const da = ws.daL[l]; bf16_t* const dqkv = ws.dqkvL[l];
    const bf16_t* dyRes = dp_ffn ? ws.dyR : dy2;
    // (last layer, clip mode: the FFN half of the layer holds the B Lv clip rows, compact -- see last_layer_clip)
    const bool cl = last && clip;
    const int Mf = cl ? Rv_clip : M, Sf = cl ? Lv : S;
    const int* rsamp = cl ? (packed ? ws.pk.vin_sample : nullptr) : row_sample;
    LnBwdArgs lb; memset(&lb, 0, sizeof(lb));
    lb.gB = gin; lb.ldgB = d;
    if (cl && packed) { lb.g2B = ws.g2p; lb.ldg2B = d; lb.g2_rows = ws.pk.vin_dst; }      // the conv-head gradient of the packed rows, gathered
    else if (cl) { lb.gB = ws.dvmB; lb.ldgB = d; }          // the heads' gradient IS the layer-output gradient, row for row
    else if (last && packed) { lb.g2B = ws.g2p; lb.ldg2B = d; }
    else if (last) { lb.g2B = ws.dvmB; lb.ldg2B = d; lb.g2_S = S; lb.g2_Lv = Lv; }
    lb.xB = ws.y2b[l]; lb.ldxB = d; lb.mean = ws.mean2[l]; lb.rstd = ws.rstd2[l]; lb.gamma = P[m.lay(l, N2W)];
    lb.rows = Mf; lb.D = d; lb.dgamma = G(m.lay(l, N2W)); lb.dbeta = G(m.lay(l, N2B));
    lb.dxB = dy2; lb.lddxB = d; lb.rowscale = dp_ffn; lb.rs_seg = Sf; lb.row_sample = rsamp;
    if (dp_ffn) { lb.dxB2 = ws.dyR; lb.lddxB2 = d; }
    ln_partials(lb, 2 * l + 1);
    TRY(launch_ln_bwd(lb, s));
    ln_deferred(lb);
    GemmArgs g = gemm_base(dy2, d, w.w2T[l], d, Mf, F, d);            // d h = dy2 W2 ; da = dh * gelu'(a)
    g.gradPre = ws.apre[l]; g.ldgp = F; g.actgrad = 2; g.outB = da; g.ldoB = F;
    TRY(launch_gemm_nt_bf16(g, s));
    {   // FFN weight gradients, one launch: dW2 = dy2^T h, dW1 = da^T x1
      GemmTNBatch tb; tb.count = 2;
      tb.g[0] = tn_group(dy2, d, (const bf16_t*)ws.h[l], F, Mf, d, F, G(m.lay(l, L2W)), F, G(m.lay(l, L2B)));
      tb.g[1] = tn_group(da, F, (const bf16_t*)ws.x1b[l], d, Mf, F, d, G(m.lay(l, L1W)), d, G(m.lay(l, L1B)));
      TRY(tn_encoder(tb));
    }
    g = gemm_base(da, F, w.w1T[l], F, Mf, d, F);                       // dx1 = da W1 + dy2
    g.residB = dyRes; g.ldrB = d; g.outB = ws.gxb[0]; g.ldoB = d;
    TRY(launch_gemm_nt_bf16(g, s));
    memset(&lb, 0, sizeof(lb));
    lb.gB = ws.gxb[0]; lb.ldgB = d; lb.xB = ws.y1b[l]; lb.ldxB = d; lb.mean = ws.mean1[l]; lb.rstd = ws.rstd1[l];
    lb.gamma = P[m.lay(l, N1W)]; lb.rows = Mf; lb.D = d; lb.dgamma = G(m.lay(l, N1W)); lb.dbeta = G(m.lay(l, N1B));
    lb.dxB = dy1; lb.lddxB = d; lb.rowscale = dp_attn; lb.rs_seg = Sf; lb.row_sample = rsamp;
    if (dp_attn) { lb.dxB2 = ws.dyR; lb.lddxB2 = d; }
    if (cl) {      // back into the token-major stream: the clip rows scatter, the text rows (no gradient behind the attention block) are zero
      // (the text rows are zeroed by the same launch: LnBwdArgs::zero_*)
      lb.zero_n = m.Mt;
      if (packed) { lb.x_rows = ws.pk.vin_dst; lb.zero_tab = ws.pk.tin_dst; }
      else { lb.x_seg = Lv; lb.x_seg_stride = S; lb.zero_seg = m.c.Lt; lb.zero_stride = S; lb.zero_off = Lv; }
    }
    ln_partials(lb, 2 * l);
    TRY(launch_ln_bwd(lb, s));
    ln_deferred(lb);
    g = gemm_base(dy1, d, w.woT[l], d, M, d, d);                       // dO = dy1 Wo
    g.outB = ws.dOb; g.ldoB = d;
    // ... and delta = rowsum_head(dO * O) in the same epilogue where the launch can carry it (round 5): attn_delta_kernel's pass over dO and O is gone
    bool delta_fused = false;
    if (ws.deltaL[l] && gemm_nt_delta_ok(g)) {
      g.deltaO = (const bf16_t*)ws.o[l]; g.ldDO = d; g.delta = ws.deltaL[l]; g.delta_S = S; g.delta_H = m.c.H; g.delta_hd = m.hd;
      if (packed) { g.delta_row_sample = ws.pk.row_sample; g.delta_seq_start = ws.pk.seq_start; }
      delta_fused = true;
    }
    TRY(launch_gemm_nt_bf16(g, s));
    AttnArgs at; memset(&at, 0, sizeof(at));
    at.qkv = ws.qkv[l]; at.ldqkv = 3 * d; at.o = ws.o[l]; at.ldo = d; at.lse = ws.lse[l]; at.kvalid = packed ? ws.pk.kvalid : ws.kvalid;
    if (packed) { at.seq_start = ws.pk.seq_start; at.seq_count = ws.pk.seq_count; at.row_sample = ws.pk.row_sample; at.total_rows = M; }
    at.B = B; at.S = S; at.H = m.c.H; at.hd = m.hd; at.p_drop = m.c.p_attn; at.seed = m.c.seed; at.layer = l;
    at.dO = ws.dOb; at.lddo = d; at.delta = delta_fused ? ws.deltaL[l] : ws.delta; at.delta_ready = delta_fused ? 1 : 0; at.dqkv = dqkv; at.lddqkv = 3 * d; at.qscale = 1.0f / sqrtf((float)m.hd);
    TRY(launch_attn_bwd(at, s));
    {   // attention-block weight gradients, one launch: dWo = dy1^T o, dWq|dWk = dqk^T (x + pos), dWv = dv^T x
      GemmTNBatch tb; tb.count = 3;
      tb.g[0] = tn_group(dy1, d, (const bf16_t*)ws.o[l], d, M, d, d, G(m.lay(l, OPW)), d, G(m.lay(l, OPB)));
      tb.g[1] = tn_group(dqkv, 3 * d, ub_in, d, M, 2 * d, d, G(m.lay(l, IPW)), d, G(m.lay(l, IPB)));
      tb.g[2] = tn_group(dqkv + 2 * d, 3 * d, xb_in, d, M, d, d, G(m.lay(l, IPW)) + (size_t)2 * d * d, d, G(m.lay(l, IPB)) + 2 * d);
      TRY(tn_encoder(tb));
    }
    g = gemm_base(dqkv, 3 * d, w.wqkvT[l], 3 * d, M, d, 3 * d);        // dx = dqkv Wqkv + dy1
    g.residB = dp_attn ? ws.dyR : dy1; g.ldrB = d; g.outB = ws.gxb[1]; g.ldoB = d;
    TRY(launch_gemm_nt_bf16(g, s));
    if (m.c.use_txt_pos) {   // d pos_txt += dq,dk (text rows) Wq,k: the text rows' positions enter every layer's q,k operand (univtg.py:123)
      g = gemm_base(dqkv, 3 * d, w.wqkvT[l], 3 * d, m.Mt, d, 2 * d);
      g.a_seg = m.c.Lt; g.a_seg_stride = S; g.a_off = Lv;
      if (l != E - 1) { g.resid = ws.dpos_txt; g.ldr = d; }
      g.outF = ws.dpos_txt; g.ldoF = d;
      TRY(launch_gemm_nt_bf16(g, s));
    }
    gin = ws.gxb[1];  // consumed by the next (lower) layer's LN2 backward before gxb[0] / gxb[1] are rewritten
    if (n_events && !defer) { if (hipError_t e = hipEventRecord((hipEvent_t)ready_events[1 + (E - 1 - l)], s)) return (int)e; events_done = 2 + (E - 1 - l); }   // layer l gradients final
    if (l == flush_layer) {                      // group A: the heads + layers E-1 .. l, one LayerNorm fold + one hybrid launch, their events together
      TRY(launch_ln_bwd_reduce_multi(lnm, s));
      lnm.count = 0;
      TRY(tn_flush_group());
      for (; events_done < 1 + (E - l); events_done++) { if (hipError_t e = hipEventRecord((hipEvent_t)ready_events[events_done], s)) return (int)e; }
    }
  }
  TRY(launch_ln_bwd_reduce_multi(lnm, s));
  TRY(tn_flush_group());                         // the deferred weight gradients of the (remaining) encoder layers, inside the encoder section
  for (; n_events && events_done < n_events; events_done++) { if (hipError_t e = hipEventRecord((hipEvent_t)ready_events[events_done], s)) return (int)e; }
  uvtg_prof_section(1, 1, s);
  const bf16_t* dx0 = ws.gxb[1];                 // d loss / d x0 from the encoder, bf16 [M, d]
  // ---------------- trainable text positions ----------------
  if (m.c.use_txt_pos) {       // pos_txt = Dropout(LayerNorm(x0_txt + E)): LayerNorm backward -> d(x0_txt + E), table rows summed over the batch
    LnBwdArgs lb; memset(&lb, 0, sizeof(lb));
    lb.g = ws.dpos_txt; lb.ldg = d; lb.x = ws.tp_xsum; lb.ldx = d; lb.mean = ws.tp_mean; lb.rstd = ws.tp_rstd; lb.gamma = P[m.txtpos(1)];
    lb.rows = m.Mt; lb.D = d; lb.p_drop = m.c.p_in; lb.seed = m.c.seed; lb.stream_id = UVTG_RNG_TXT_POS; lb.src_rows = ws.tp_src;
    lb.dgamma = G(m.txtpos(1)); lb.dbeta = G(m.txtpos(2)); lb.dxF = ws.tp_dx; lb.lddxF = d; lb.rs_seg = 1;
    lb.partial = ws.tn_scratch; lb.partial_floats = ws.tn_scratch_floats;
    TRY(launch_ln_bwd(lb, s));
    hipLaunchKernelGGL(txt_pos_table_grad_kernel, dim3(cdiv(m.c.Lt * d, 256)), dim3(256), 0, s, ws.tp_dx, B, m.c.Lt, d, G(m.txtpos(0)));
    UVTG_CHECK_LAUNCH();
  }
  // ---------------- saliency branch ----------------
  SaliencyArgs sa = sal_args(m, P, ws, x0, src_txt_mask, src_vid_mask, (float*)txt_mem_proj, nullptr);
  sa.g_txt_rows = m.c.use_txt_pos ? ws.tp_dx : nullptr;
  sa.g_sal = g_saliency; sa.g_pooled = g_txt_mem; sa.g_vid = g_vid_mem; sa.gv_sb = g_vid_sb; sa.gv_st = g_vid_st; sa.g_vrow = g_vrow; sa.pos_idx = pos_idx; sa.dx0B = dx0; sa.dw_pool = G(m.pool());
  if (packed) { sa.dx0_map = ws.pk.grad_map; sa.vout_map = ws.pk.vin_of; }
  sa.dq = ws.sal_dq; sa.dlog = ws.sal_dlog; sa.out_vid = ws.dyP[0]; sa.out_txt = ws.dyP[1];
  TRY(launch_saliency_bwd(sa, s));
  // ---------------- input projections ----------------
  for (int which = 0; which < 2; which++) {
    const bool cv = packed && which == 0;      // compact clip rows (see Fwd::project)
    const int R = which == 0 ? (packed ? compact_clip_rows(m, lens_host, pmode) : m.Mv) : m.Mt, nb = m.nproj;
    const unsigned rs = which == 0 ? UVTG_RNG_IN_VID : UVTG_RNG_IN_TXT;
    const float* src = which == 0 ? src_vid : src_txt;
    const bool pp = m.c.proj_precise != 0;
    // (ws.dyP[which] = bf16(dx0 + saliency-branch gradients), packed per modality by launch_saliency_bwd)
    // Blocks from the last to the first: gout = gradient wrt the block's Linear output (ReLU mask applied for the blocks that have one);
    // dgrad -> fp32 gradient wrt the block's LayerNorm output -> LayerNorm backward (same dropout mask) -> parameter gradients and, for
    // blocks > 0, the gradient wrt the previous block's output.  The weight gradients of a modality reduce over the same rows: they go out
    // as ONE launch at the end (more tiles -> fewer row splits, one reduce pass); small / odd shapes keep the per-gradient launches.
    const bf16_t* gout[MAXP];
    gout[nb - 1] = ws.dyP[which];
    for (int b = nb - 1; b >= 0; b--) {
      const int Din = m.din(which, b), Kp = m.kp(which, b);
      GemmArgs g = gemm_base(gout[b], d, w.pwT[which][b], d, R, Kp, d);
      // the wide feature LayerNorm has no upstream: its backward is the dgamma / dbeta column reduction alone, which takes the dgrad as a bf16
      // stream like every other gradient stream of the step (round 6: the GEMM writes and the reduction reads half the bytes -- 2 x 110 MB at config 2)
      static const bool dgrad_f32 = uvtg_dev_env("UVTG_PROJ_DGRAD_F32") != nullptr;      // experiment: fp32 as in rounds 1-5
      const bool gbf = b == 0 && Din > 2048 && (Kp % 2 == 0) && !dgrad_f32;
      if (gbf) { g.outB = (bf16_t*)ws.dA[which]; g.ldoB = Kp; } else { g.outF = ws.dA[which]; g.ldoF = Kp; }
      TRY(launch_gemm_nt_bf16(g, s));
      LnBwdArgs lb; memset(&lb, 0, sizeof(lb));
      if (gbf) { lb.gB = (const bf16_t*)ws.dA[which]; lb.ldgB = Kp; } else { lb.g = ws.dA[which]; lb.ldg = Kp; }
      lb.x = b == 0 ? src : ws.ph[which][b - 1]; lb.ldx = Din; lb.mean = ws.pm[which][b]; lb.rstd = ws.pr[which][b];
      lb.gamma = P[m.proj(which, b, PG)]; lb.rows = R; lb.D = Din; lb.p_drop = m.c.p_in; lb.seed = m.c.seed; lb.stream_id = rs + b;
      lb.dgamma = G(m.proj(which, b, PG)); lb.dbeta = G(m.proj(which, b, PBE)); lb.rs_seg = 1;
      if (b > 0) { lb.dxB = ws.dhb[which][b - 1]; lb.lddxB = d; lb.relu_from_x = 1; gout[b - 1] = ws.dhb[which][b - 1]; }
      if (cv) { lb.src_rows = ws.pk.vin_src; lb.gather_x = b == 0; }
      lb.partial = ws.tn_scratch; lb.partial_floats = ws.tn_scratch_floats;
      TRY(launch_ln_bwd(lb, s));
    }
    GemmTNBatch pb; pb.count = nb;
    for (int b = 0; b < nb; b++) {
      const bf16_t* ab = pp ? ws.paB[which][b] : (const bf16_t*)ws.pa[which][b];
      pb.g[b] = tn_group(gout[b], d, ab, m.kp(which, b), R, d, m.din(which, b), G(m.proj(which, b, PW)), m.din(which, b), G(m.proj(which, b, PB)));
      pb.g[b].splits = splits_v;
    }
    static const bool pbatch_off = uvtg_dev_env("UVTG_TN_PROJBATCH_OFF") != nullptr;
    if (!pbatch_off && nb > 1 && R >= 2048 && gemm_tn_batch_ok(pb)) TRY(launch_gemm_tn_batch(pb, s));
    else for (int b = 0; b < nb; b++) TRY(launch_gemm_tn_bf16(pb.g[b], s));
  }
  // token-type rows: row 1 (video) / row 0 (text) receive the bias gradient of their modality's second projection (univtg.py:114-115), one launch
  hipLaunchKernelGGL(add_vec2_kernel, dim3(cdiv(d, 256)), dim3(256), 0, s, G(m.tail(TOK)) + d, G(m.proj(0, m.nproj - 1, PB)), G(m.tail(TOK)), G(m.proj(1, m.nproj - 1, PB)), d);
  UVTG_CHECK_LAUNCH();
  TRY(launch_sqsum_ranges(grads, zr_keep, ws.gnorm2, s));      // the gradients no weight-gradient launch assigned
  uvtg_prof_section(3, 1, s);
  return 0;
}

// How uvtg_backward batches its ready_events (include/uvtg.h): last_event[g] = index of the LAST event of group g; the events of a group are
// recorded at the same point of the stream, so a caller can wait for the last one and exchange the group's ranges in one coalesced collective.
extern "C" int uvtg_backward_event_groups(int E, int* last_event) {
  if (E <= 0 || E > MAXE || !last_event) return -20;
  const bool defer_off = uvtg_dev_env("UVTG_TN_DEFER_OFF") != nullptr, per_layer = uvtg_dev_env("UVTG_TN_EVENTS_PER_LAYER") != nullptr;
  const bool one_launch = uvtg_dev_env("UVTG_TN_EVENT_GROUPS") == nullptr;
  if (defer_off || per_layer) { for (int i = 0; i <= E; i++) last_event[i] = i; return E + 1; }      // one event behind each range's own launches
  if (one_launch || E < 2) { last_event[0] = E; return 1; }                                            // everything behind the loop
  last_event[0] = E - 1; last_event[1] = E;                                                            // heads + layers E-1 .. 1 | layer 0
  return 2;
}

// device address of the squared L2 norm of ALL gradients of the last uvtg_backward on this workspace (single-rank steps hand it to
// uvtg_adamw_clip_step_prenorm instead of re-reading the gradient buffer; after a gradient all-reduce it is stale)
extern "C" const float* uvtg_backward_gradnorm2(const uvtg_dims* dm, void* workspace) {
  if (check_dims(dm) || !workspace || !dm->training || dm->precise) return nullptr;
  Dm m(*dm);
  WSpace ws(m, workspace, nullptr);
  return ws.gnorm2;
}

// device addresses of what the saliency pass of the last uvtg_forward on this workspace left behind: cosine(vid_mem_proj, txt_mem_proj)
// [B, Lv], |vid_mem_proj| [B, Lv], |txt_mem_proj| [B] -- the criterion takes them instead of recomputing them from vid_mem_proj
extern "C" int uvtg_forward_saliency_stats(const uvtg_dims* dm, void* workspace, const float** cosv, const float** vnorm, const float** qnorm) {
  if (int e = check_dims(dm)) return e;
  if (!workspace || !cosv || !vnorm || !qnorm) return -20;
  Dm m(*dm);
  WSpace ws(m, workspace, nullptr);
  *cosv = ws.cosv; *vnorm = ws.vnorm; *qnorm = ws.qnorm;
  return 0;
}

// =================================================================================================
// criterion
// =================================================================================================
namespace {
LossArgs loss_args(int B, int Lv, int d, int which, float eos_coef, const float* pred_logits, const float* pred_spans,
                   const float* vid, long long vid_sb, long long vid_st, const float* txt_mem, const float* timestamp,
                   const float* ts_mask, const float* ts_window, const float* span_nn, const float* sal, const long long* pos_idx,
                   float* ws, float* losses, const float* cos_c, const float* vnorm_c, const float* qnorm_c) {
  LossArgs a; memset(&a, 0, sizeof(a));
  if (cos_c && vnorm_c && qnorm_c) { a.cos_c = cos_c; a.vnorm_c = vnorm_c; a.qnorm_c = qnorm_c; }
  a.B = B; a.Lv = Lv; a.d = d; a.pred_logits = pred_logits; a.pred_spans = pred_spans; a.vid = vid; a.vid_sb = vid_sb; a.vid_st = vid_st;
  a.txt = txt_mem; a.timestamp = timestamp; a.ts_mask = ts_mask; a.ts_window = ts_window; a.span_nn = span_nn; a.sal_tgt = sal;
  a.pos_idx = pos_idx; a.eos_coef = eos_coef; a.do_spans = which & 1; a.do_labels = (which >> 1) & 1; a.do_saliency = (which >> 2) & 1;
  a.ws = ws; a.losses = losses;
  return a;
}
}  // namespace

extern "C" long long uvtg_cls_nce_ws_floats(int B, int C) { return (B > 0 && C > 0) ? cls_nce_ws_floats(B, C) : 0; }
extern "C" int uvtg_cls_nce_fwd(int B, int C, int d, const float* vid, long long vid_sb, long long vid_st, const long long* pos_idx,
                                const float* cls, const float* cls_idx, const float* active, float* ws, float* loss_out, uvtg_stream_t stream) {
  if (B <= 0 || C <= 0 || d <= 0) return -11;
  if (!vid || !pos_idx || !cls || !cls_idx || !ws || !loss_out) return -20;
  ClsNceArgs a; memset(&a, 0, sizeof(a));
  a.B = B; a.C = C; a.d = d; a.vid = vid; a.vid_sb = vid_sb; a.vid_st = vid_st; a.pos_idx = pos_idx; a.cls = cls; a.cls_idx = cls_idx;
  a.active = active; a.ws = ws; a.loss = loss_out;
  return launch_cls_nce_fwd(a, (hipStream_t)stream);
}
extern "C" int uvtg_cls_nce_bwd(int B, int C, int d, const float* vid, long long vid_sb, long long vid_st, const long long* pos_idx,
                                const float* cls, const float* cls_idx, const float* active, float* ws, const float* go,
                                float* g_vid, long long gv_sb, long long gv_st, float* g_cls, uvtg_stream_t stream) {
  if (B <= 0 || C <= 0 || d <= 0) return -11;
  if (!vid || !pos_idx || !cls || !cls_idx || !ws || !go || !g_vid || !g_cls) return -20;
  ClsNceArgs a; memset(&a, 0, sizeof(a));
  a.B = B; a.C = C; a.d = d; a.vid = vid; a.vid_sb = vid_sb; a.vid_st = vid_st; a.pos_idx = pos_idx; a.cls = cls; a.cls_idx = cls_idx;
  a.active = active; a.ws = ws; a.go = go; a.g_vid = g_vid; a.gv_sb = gv_sb; a.gv_st = gv_st; a.g_cls = g_cls;
  return launch_cls_nce_bwd(a, (hipStream_t)stream);
}

extern "C" int uvtg_criterion_fwd(int B, int Lv, int d, int which, float eos_coef, const float* pred_logits, const float* pred_spans,
                                  const float* vid, long long vid_sb, long long vid_st, const float* txt_mem,
                                  const float* timestamp, const float* timestamp_mask, const float* timestamp_window,
                                  const float* span_labels_nn, const float* saliency_scores, const long long* pos_idx,
                                  float* loss_ws, float* losses_out, const float* cos_cached, const float* vnorm_cached,
                                  const float* qnorm_cached, uvtg_stream_t stream) {
  if (B <= 0 || Lv <= 0 || d <= 0) return -11;
  if (!pred_logits || !pred_spans || !timestamp || !timestamp_mask || !timestamp_window || !span_labels_nn || !loss_ws || !losses_out) return -20;
  if ((which & 4) && saliency_scores && pos_idx && (!vid || !txt_mem)) return -20;
  LossArgs a = loss_args(B, Lv, d, which, eos_coef, pred_logits, pred_spans, vid, vid_sb, vid_st, txt_mem, timestamp, timestamp_mask,
                         timestamp_window, span_labels_nn, saliency_scores, pos_idx, loss_ws, losses_out, cos_cached, vnorm_cached, qnorm_cached);
  return launch_losses_fwd(a, (hipStream_t)stream);
}

extern "C" int uvtg_criterion_bwd(int B, int Lv, int d, int which, float eos_coef, const float* pred_logits, const float* pred_spans,
                                  const float* vid, long long vid_sb, long long vid_st, const float* txt_mem,
                                  const float* timestamp, const float* timestamp_mask, const float* timestamp_window,
                                  const float* span_labels_nn, const float* saliency_scores, const long long* pos_idx,
                                  float* loss_ws, const float* losses_out, const float* go,
                                  float* g_logits, float* g_spans, float* g_vid, float* g_txt, float* g_cos, float* g_vrow,
                                  const float* cos_cached, const float* vnorm_cached, const float* qnorm_cached, uvtg_stream_t stream) {
  if (B <= 0 || Lv <= 0 || d <= 0) return -11;
  if (!pred_logits || !pred_spans || !timestamp || !timestamp_mask || !timestamp_window || !span_labels_nn || !loss_ws ||
      !losses_out || !go || !g_logits || !g_spans || !g_cos || !g_vrow || !g_txt) return -20;
  LossArgs a = loss_args(B, Lv, d, which, eos_coef, pred_logits, pred_spans, vid, vid_sb, vid_st, txt_mem, timestamp, timestamp_mask,
                         timestamp_window, span_labels_nn, saliency_scores, pos_idx, loss_ws, (float*)losses_out, cos_cached, vnorm_cached, qnorm_cached);
  a.go = go; a.g_logits = g_logits; a.g_spans = g_spans; a.g_vid = g_vid; a.g_txt = g_txt; a.g_cos = g_cos; a.g_vrow = g_vrow;
  hipStream_t s = (hipStream_t)stream;
  const bool sal = (which & 4) && saliency_scores && pos_idx;
  if (!sal) {
    if (g_vid) hipMemsetAsync(g_vid, 0, (size_t)B * Lv * d * sizeof(float), s);
    hipMemsetAsync(g_txt, 0, (size_t)B * d * sizeof(float), s);
    hipMemsetAsync(g_vrow, 0, (size_t)B * d * sizeof(float), s);
  }
  return launch_losses_bwd(a, s);
}

// =================================================================================================
// kernel-level entry points
// =================================================================================================
extern "C" int uvtg_linear_bf16(const void* A, const void* W, const float* bias, float* C, int M, int N, int K, int act, uvtg_stream_t st) {
  if (!A || !W || !C) return -20;
  GemmArgs g = gemm_base(A, K, W, K, M, N, K);
  g.bias = bias; g.act = act; g.outF = C; g.ldoF = N;
  return launch_gemm_nt_bf16(g, (hipStream_t)st);
}
// split-operand ("fp32x3") GEMM at kernel level: uvtg_split_f16 builds the fp16 hi | lo images [rows, 2 kp] of an fp32 matrix (columns
// [cols, kp) zero; kp a multiple of 64; activations scaled by 16, weights by 64 -- uvtg_common.h), uvtg_linear_split multiplies two of them
extern "C" int uvtg_split_f16(const float* src, void* dst, int rows, int cols, int kp, int is_weight, uvtg_stream_t st) {
  if (!src || !dst) return -20;
  if (rows <= 0 || cols <= 0 || kp < cols || kp % 64) return -11;
  SplitOps so; so.count = 1; so.scale = is_weight ? UVTG_SPLIT_W_SCALE : UVTG_SPLIT_A_SCALE;
  so.src[0] = src; so.dst[0] = (unsigned short*)dst; so.rows[0] = rows; so.cols[0] = cols; so.kp[0] = kp; so.conv[0] = 0;
  return launch_split_f16_multi(so, (hipStream_t)st);
}
extern "C" int uvtg_linear_split(const void* A, const void* W, const float* bias, float* C, int M, int N, int Kp, int act, uvtg_stream_t st) {
  if (!A || !W || !C) return -20;
  GemmArgs g = gemm_base(A, 2 * Kp, W, 2 * Kp, M, N, 2 * Kp);
  g.bias = bias; g.act = act; g.outF = C; g.ldoF = N;
  return launch_gemm_nt_split(g, (hipStream_t)st);
}
// the same two GEMMs with the split-K workspace the engine's forward gives its small launches: sk_ws = uvtg_linear_sk_ws_floats() floats,
// 16-byte aligned, its first 256 words (the tickets) ZERO before the first call (the kernel leaves them zero)
static void set_sk_ws(GemmArgs& g, float* sk_ws) {
  g.sk_tickets = (unsigned*)sk_ws; g.sk_slab = sk_ws + UVTG_SK_UNITS; g.sk_cap_units = UVTG_SK_UNITS;
}
extern "C" long long uvtg_linear_sk_ws_floats(void) { return UVTG_SK_UNITS + (long long)UVTG_SK_UNITS * UVTG_SK_TILE_FLOATS; }
extern "C" int uvtg_linear_bf16_sk(const void* A, const void* W, const float* bias, float* C, int M, int N, int K, int act, float* sk_ws, uvtg_stream_t st) {
  if (!A || !W || !C || !sk_ws || ((uintptr_t)sk_ws & 15)) return -20;
  GemmArgs g = gemm_base(A, K, W, K, M, N, K);
  g.bias = bias; g.act = act; g.outF = C; g.ldoF = N; set_sk_ws(g, sk_ws);
  return launch_gemm_nt_bf16(g, (hipStream_t)st);
}
extern "C" int uvtg_linear_split_sk(const void* A, const void* W, const float* bias, float* C, int M, int N, int Kp, int act, float* sk_ws, uvtg_stream_t st) {
  if (!A || !W || !C || !sk_ws || ((uintptr_t)sk_ws & 15)) return -20;
  GemmArgs g = gemm_base(A, 2 * Kp, W, 2 * Kp, M, N, 2 * Kp);
  g.bias = bias; g.act = act; g.outF = C; g.ldoF = N; set_sk_ws(g, sk_ws);
  return launch_gemm_nt_split(g, (hipStream_t)st);
}
extern "C" int uvtg_wgrad_bf16(const void* dY, const void* X, float* dW, float* dbias, int M, int N, int K, int splits, uvtg_stream_t st) {
  if (!dY || !X || !dW) return -20;
  GemmTNArgs t; memset(&t, 0, sizeof(t));
  t.P = (const bf16_t*)dY; t.ldp = N; t.Q = (const bf16_t*)X; t.ldq = K; t.M = M; t.N = N; t.K = K; t.Mq = M;
  t.out = dW; t.ldo = K; t.col_stride = 1; t.dbias = dbias; t.splits = splits;
  return launch_gemm_tn_bf16(t, (hipStream_t)st);
}
extern "C" long long uvtg_wgrad_scratch_floats(int M, int N, int K) { return gemm_tn_scratch_floats(M, N, K); }
extern "C" int uvtg_wgrad_bf16_ws(const void* dY, const void* X, float* dW, float* dbias, int M, int N, int K, float* scratch,
                                  long long scratch_floats, uvtg_stream_t st) {
  if (!dY || !X || !dW) return -20;
  GemmTNArgs t; memset(&t, 0, sizeof(t));
  t.P = (const bf16_t*)dY; t.ldp = N; t.Q = (const bf16_t*)X; t.ldq = K; t.M = M; t.N = N; t.K = K; t.Mq = M;
  t.out = dW; t.ldo = K; t.col_stride = 1; t.dbias = dbias; t.splits = 8; t.scratch = scratch; t.scratch_floats = scratch_floats;
  return launch_gemm_tn_bf16(t, (hipStream_t)st);
}
extern "C" long long uvtg_wgrad_multi_slab_floats(int total_tiles) { return gemm_tn_multi_slab_floats(total_tiles, 320); }
extern "C" int uvtg_wgrad_bf16_multi(int count, const void* const* dY, const int* N, const void* const* X, const int* K, float* const* dW,
                                     float* const* dbias, int M, float* slabs, long long slab_floats, unsigned* tickets, int n_tickets,
                                     uvtg_stream_t st) {
  if (!dY || !N || !X || !K || !dW || !slabs || !tickets) return -20;
  if (count < 1 || count > UVTG_TNH_MAX_GROUPS) return -11;
  GemmTNMulti mu; mu.count = count; mu.slabs = slabs; mu.slab_floats = slab_floats; mu.tickets = tickets; mu.n_tickets = n_tickets;
  for (int i = 0; i < count; i++) {
    GemmTNArgs& t = mu.g[i]; memset(&t, 0, sizeof(t));
    t.P = (const bf16_t*)dY[i]; t.ldp = N[i]; t.Q = (const bf16_t*)X[i]; t.ldq = K[i]; t.M = M; t.N = N[i]; t.K = K[i]; t.Mq = M;
    t.out = dW[i]; t.ldo = K[i]; t.col_stride = 1; t.dbias = dbias ? dbias[i] : nullptr; t.splits = 8; t.assign = 1;
  }
  if (!gemm_tn_multi_ok(mu)) return -2;
  return launch_gemm_tn_multi(mu, (hipStream_t)st);
}
extern "C" int uvtg_cast_bf16(const float* src, void* dst, long long n, uvtg_stream_t st) {
  if (!src || !dst) return -20;
  return launch_cast_bf16(src, (bf16_t*)dst, n, (hipStream_t)st);
}
extern "C" int uvtg_cast_f32(const void* src, float* dst, long long n, uvtg_stream_t st) {
  if (!src || !dst) return -20;
  return launch_cast_f32((const bf16_t*)src, dst, n, (hipStream_t)st);
}
extern "C" int uvtg_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                  int rows, int D, uvtg_stream_t st) {
  if (!x || !gamma || !beta || !y) return -20;
  LnFwdArgs a; memset(&a, 0, sizeof(a));
  a.x = x; a.ldx = D; a.rows = rows; a.D = D; a.gamma = gamma; a.beta = beta; a.eps = 1e-5f; a.mean = mean; a.rstd = rstd;
  a.yF = y; a.ldyF = D; a.Dpad = D;
  return launch_ln_fwd(a, (hipStream_t)st);
}
// kernel-level entry of the encoder's bf16 LayerNorm launches (include/uvtg_dev.h; parity tests): x bf16 [rows, D]; y (bf16, optional), y + pos
// (bf16, optional: row (b, s) with s < Lv of each S-row sample adds pos[b * Lv + s], fp32 [rows / S * Lv, D]; pos may be null), mean / rstd.
extern "C" int uvtg_debug_layernorm_fwd_bf16(const void* xB, const float* gamma, const float* beta, void* yB, void* yU, const float* pos, int S,
                                             int Lv, float* mean, float* rstd, int rows, int D, uvtg_stream_t st) {
  if (!xB || !gamma || !beta) return -20;
  LnFwdArgs a; memset(&a, 0, sizeof(a));
  a.xB = (const bf16_t*)xB; a.ldxB = D; a.rows = rows; a.D = D; a.gamma = gamma; a.beta = beta; a.eps = 1e-5f; a.mean = mean; a.rstd = rstd;
  a.yB = (bf16_t*)yB; a.ldyB = D; a.Dpad = D; a.yU = (bf16_t*)yU; a.ldyU = D; a.pos = pos; a.S = S; a.Lv = Lv;
  return launch_ln_fwd(a, (hipStream_t)st);
}
extern "C" int uvtg_layernorm_bwd(const float* g, const float* x, const float* mean, const float* rstd, const float* gamma,
                                  float* dx, float* dgamma, float* dbeta, int rows, int D, uvtg_stream_t st) {
  if (!g || !x || !mean || !rstd || !gamma) return -20;
  LnBwdArgs a; memset(&a, 0, sizeof(a));
  a.g = g; a.ldg = D; a.x = x; a.ldx = D; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.rows = rows; a.D = D;
  a.dgamma = dgamma; a.dbeta = dbeta; a.dxF = dx; a.lddxF = D; a.rs_seg = 1;
  return launch_ln_bwd(a, (hipStream_t)st);
}
extern "C" int uvtg_attention_fwd(const void* qkv, const unsigned char* kvalid, void* o, float* lse, int B, int S, int H, int hd,
                                  int precise, uvtg_stream_t st) {
  if (!qkv || !kvalid || !o) return -20;
  AttnArgs a; memset(&a, 0, sizeof(a));
  a.qkv = qkv; a.ldqkv = 3 * H * hd; a.o = o; a.ldo = H * hd; a.lse = lse; a.kvalid = kvalid; a.B = B; a.S = S; a.H = H; a.hd = hd;
  a.precise = precise;
  return launch_attn_fwd(a, (hipStream_t)st);
}
extern "C" int uvtg_attention_bwd(const void* qkv, const unsigned char* kvalid, const void* o, const float* lse, const void* dO,
                                  float* delta, void* dqkv, float qscale, int B, int S, int H, int hd, uvtg_stream_t st) {
  if (!qkv || !kvalid || !o || !lse || !dO || !delta || !dqkv) return -20;
  AttnArgs a; memset(&a, 0, sizeof(a));
  a.qkv = qkv; a.ldqkv = 3 * H * hd; a.o = (void*)o; a.ldo = H * hd; a.lse = (float*)lse; a.kvalid = kvalid; a.B = B; a.S = S; a.H = H;
  a.hd = hd; a.dO = (const bf16_t*)dO; a.lddo = H * hd; a.delta = delta; a.dqkv = (bf16_t*)dqkv; a.lddqkv = 3 * H * hd; a.qscale = qscale;
  return launch_attn_bwd(a, (hipStream_t)st);
}
extern "C" int uvtg_ragged_to_padded(const void* packed, int src_bf16, const int* offsets, int B, int Lmax, int D, float* out, float* mask,
                                     uvtg_stream_t st) {
  if (!packed || !offsets || !out) return -20;
  return launch_ragged_to_padded(packed, src_bf16, offsets, B, Lmax, D, out, mask, (hipStream_t)st);
}
extern "C" int uvtg_sine_position(const float* vid_mask, const float* txt_mask, const float* dim_t, float* pos, unsigned char* kvalid,
                                  int B, int Lv, int Lt, int d, uvtg_stream_t st) {
  if (!vid_mask || !txt_mask || !dim_t || !pos || !kvalid) return -20;
  return launch_seq_prep(vid_mask, txt_mask, B, Lv, Lt, d, dim_t, pos, kvalid, nullptr, (hipStream_t)st);
}
