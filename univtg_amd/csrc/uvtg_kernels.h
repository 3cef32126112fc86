// Internal launcher interface shared by the kernel translation units and the engine.
// (Public C-ABI: include/uvtg.h.)  All launchers enqueue on `stream`, never allocate, return 0/hipError_t.
#pragma once
#include "uvtg_common.h"

// ---------------------------------------------------------------------------------------------
// GEMM  C[M,N] = A[M,K] * B[N,K]^T   (both operands K-contiguous), fused epilogue.
// ---------------------------------------------------------------------------------------------
struct GemmArgs {
  const void* A;      // bf16 [rows, lda], or (launch_gemm_nt_split) fp16 split rows
  // optional second A operand (same layout) for the output columns n >= a2_n0 (a multiple of 256): one launch computes
  // [A | A2-columns] -- the q,k projections read x + pos and the v projection reads x (transformer_encoder_droppath.py:116-117)
  const void* A2; int a2_n0;
  const void* B;      // bf16 / fp32 [N, ldb]
  int M, N, K, lda, ldb;
  // row window of the persistent 256-wide kernel: tiles cover rows [m_begin, M) (set by its launcher when it cuts a launch into a whole
  // number of CU rounds of tall tiles + a tail of short ones; all row addressing stays absolute).  0 everywhere else.
  int m_begin;
  // A-row gather: arow(m) = (m / a_seg) * a_seg_stride + (m % a_seg) + a_off   (a_seg == 0: identity)
  // K is split in taps of `ktap` columns; tap t reads row arow(m) + t, columns k % ktap (3-tap conv).
  int a_seg, a_seg_stride, a_off, ktap;
  // output-row scatter, same form
  int o_seg, o_seg_stride, o_off;
  // batched groups over blockIdx.z: element offsets added per group
  long long gA, gB, gBias, gOut, gPre;
  int groups;
  // epilogue (applied in this order)
  const float* bias;       // [N]
  const float* bias2;      // [N] second bias (token-type embedding row)
  float colscale; int colscale_n;     // columns < colscale_n are multiplied (q scaling)
  bf16_t* outPre; int ldpre_out;      // bf16 copy before the activation (pre-GELU, saved for backward)
  int act;                 // 0 none, 1 relu, 2 gelu(erf)
  const bf16_t* gradPre; int ldgp; int actgrad;   // 1: zero where gradPre<=0 (relu'), 2: *= gelu'(gradPre)
  const float* rowscale; int rs_seg;  // per-sample factor rowscale[m / rs_seg] (DropPath)
  const int* row_sample;              // packed rows: rowscale[row_sample[m]] instead
  const float* resid; int ldr;        // fp32 residual, indexed by the OUTPUT row
  const bf16_t* residB; int ldrB;     // bf16 residual (the bf16 activation stream of the fast mode), same indexing
  float* outF; int ldoF;              // fp32 output
  bf16_t* outB; int ldoB;             // bf16 output
  const float* pos; int ldpos; int pos_rows;   // fp32 positional table indexed by m (< pos_rows)
  bf16_t* outU; float* outUF; int ldoU;        // (value + pos) in bf16 / fp32
  // output-row tables (replace the affine scatter; no epilogue operand with these): row m's outPre / outB / outU / outUF go to row
  // o_rows[m] (< 0: not written); outF goes to f_rows[m] (null: the affine o_seg row); pos is read at row pos_map[m] (null: m)
  const int* o_rows; const int* f_rows; const int* pos_map;
  // ---- split-operand ("fp32x3") mode: launch_gemm_nt_split -----------------------------------------------------------------------
  // A and B rows hold the fp16 hi / lo images of the fp32 operand interleaved in 32-column blocks (uvtg_common.h, split_col): K, lda, ldb,
  // ktap, gA, gB count ELEMENTS of such rows (two per real column).  Every 64-element K tile yields hi.hi + hi.lo + lo.hi on
  // v_mfma_f32_32x32x16_f16.  Epilogue: the accumulator is multiplied by accscale (1 / (A scale x W scale)) first; fp32 residual / outputs
  // as above, plus outS / outUS: the value (/ value + pos) re-split (x sscale) for the next GEMM, row m at outS[m * ldoS + split_col(n)].
  float accscale;
  unsigned short* outS; unsigned short* outUS; int ldoS; float sscale;
  // ---- split-K of small launches (persistent 256-wide kernel, 128-row tiles): when a launch has at most half as many tiles as the chip has
  // CUs, its launcher may cut K into `sk` parts (one workgroup each): every part publishes its fp32 partial tile to sk_slab (write-through),
  // takes a ticket, and the part that draws the last one sums ALL parts in part order (bit-reproducible) and runs the epilogue.  The caller
  // provides sk_slab (sk_cap_units x 128 x 256 floats) and sk_tickets (sk_cap_units words, ZERO before the launch; the kernel leaves them zero);
  // null = never split.  `sk` is set by the launcher.
  float* sk_slab; unsigned* sk_tickets; int sk_cap_units; int sk;
  // ---- tile order of the persistent 256-wide kernel (set by its launcher): the column tiles are walked in GROUPS of `cgw` (0 = all of them:
  // row-block-major over the whole width).  Wide outputs (N = 3 d: 12 column tiles) in groups of cgw keep cgw weight panels -- not all
  // twelve, 6 MB against a 4 MB L2 -- in front of an XCD while its row blocks stream past them; same tiles, same K order, same results.
  int cgw;
  // ---- attention backward's delta = rowsum_head(dO * O), produced where dO is (round 5; persistent 256-wide kernel, plain row mapping): the
  // out-projection dgrad launch takes O [M, ldDO] (bf16) as its epilogue operand and adds, per output row m and head, the dot product of the
  // ROUNDED (bf16) dO values it stores with the O values into delta[(b * delta_H + head) * delta_S + s] -- (b, s) = (m / delta_S, m % delta_S), or
  // (delta_row_sample[m], m - delta_seq_start[b]) on the packed stream.  delta must be ZERO before the launch (two fp32 atomics per element at
  // head_dim 128, one below: order-independent).  Replaces attn_delta_kernel's pass over dO and O (gemm_nt_delta_ok() says whether a launch takes it).
  const bf16_t* deltaO; int ldDO; float* delta; int delta_S, delta_H, delta_hd; const int* delta_row_sample; const int* delta_seq_start;
};
bool gemm_nt_delta_ok(const GemmArgs& a);
constexpr int UVTG_SK_UNITS = 256;                 // capacity the engine's workspace provides
constexpr int UVTG_SK_TILE_FLOATS = 128 * 256;
int launch_gemm_nt_bf16(const GemmArgs& a, hipStream_t s);
int launch_gemm_nt_split(const GemmArgs& a, hipStream_t s);

// C[N,K] (+)= P[M,N]^T * Q[M,K]  (reduction over rows; fp32 atomic accumulation, split over M)
struct GemmTNArgs {
  const bf16_t* P; int ldp;     // [M, ldp]
  const bf16_t* Q; int ldq;     // [Mq, ldq]
  int M, N, K;
  int q_row_off, Mq;            // Q row = m + q_row_off, rows outside [0, Mq) read as zero
  float* out; int ldo, col_stride;   // out[n * ldo + k * col_stride]
  float* dbias;                 // optional: dbias[n] += sum_m P[m][n]
  int splits;                   // M splits of the atomic 128-tile kernel
  // optional scratch for the 256-tile kernel (split partial tiles as plain fp32 slabs + a reduce pass instead of
  // fp32 atomics); null / too small -> the atomic kernel is used
  float* scratch; long long scratch_floats;
  // 256-tile kernel only: K is `K / ktap` conv taps of ktap columns; tap t reads Q rows m + q_row_off + t, columns k % ktap,
  // and lands at out[n * ldo + (k % ktap) * col_stride + t]   (0: no taps)
  int ktap;
  // 1: out = the product (the caller does NOT zero `out`; dbias is still accumulated); 0: out += the product
  int assign;
  // optional (assign only): the sum of squares of the assigned matrix is added into the slot array sqsum[32 + 32 s], s < 64 (the clipping
  // norm without a pass over the gradients; UVTG_SQSUM_FLOATS floats, zeroed by the caller, folded by launch_sqsum_ranges into sqsum[0])
  float* sqsum;
};
// several weight gradients over the SAME reduction rows (M, splits plan) in one launch: the tiles of all groups share the M splits,
// so there are more tiles per launch, fewer splits, and far less split-partial traffic than one launch per gradient
constexpr int UVTG_TN_MAX_GROUPS = 8;
struct GemmTNBatch { GemmTNArgs g[UVTG_TN_MAX_GROUPS]; int count; };
bool gemm_tn_batch_ok(const GemmTNBatch& b);       // all groups eligible for the 256-tile kernel, same M, scratch large enough
long long gemm_tn_batch_scratch_floats(const GemmTNBatch& b);
int launch_gemm_tn_batch(const GemmTNBatch& b, hipStream_t s);
// the same WITHOUT a reduce pass ("hybrid" plan, gemm.hip: gemm_tn256h_kernel): whole tiles per workgroup + the remainder cut into <= 3 row
// ranges whose parts meet through write-through slabs and a ticket per tile.  Contiguous assigned outputs, N and K multiples of 256, no taps.
// slabs: gemm_tn_multi_slab_floats(total 256 x 256 tiles) floats (16-byte aligned); tickets: one zeroed unsigned per tile.
constexpr int UVTG_TNH_MAX_GROUPS = 32;
struct GemmTNMulti { GemmTNArgs g[UVTG_TNH_MAX_GROUPS]; int count; float* slabs; long long slab_floats; unsigned* tickets; int n_tickets; };
bool gemm_tn_multi_ok(const GemmTNMulti& b);
long long gemm_tn_multi_slab_floats(int total_tiles, int cus_hint);
int launch_gemm_tn_multi(const GemmTNMulti& b, hipStream_t s);
bool gemm_tn_taps_ok(const GemmTNArgs& a);   // can this call (with ktap set) run as ONE launch?
long long gemm_tn_scratch_floats(int M, int N, int K);   // scratch that makes every (M, N, K) eligible for the 256-tile kernel
int launch_gemm_tn_bf16(const GemmTNArgs& a, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// LayerNorm
// ---------------------------------------------------------------------------------------------
struct LnFwdArgs {
  const float* x; int ldx;      // [rows, D] fp32
  const bf16_t* xB; int ldxB;   // alternative bf16 input (used when x == nullptr)
  int rows, D;
  const float* gamma; const float* beta; float eps;
  float* mean; float* rstd;     // optional [rows]
  // dropout on the normalised output (input projections): keep prob 1-p, Philox stream id
  float p_drop; unsigned long long seed; unsigned stream_id;
  // outputs (any may be null); Dpad: columns [D, Dpad) of the bf16/f32 GEMM operand are zero-filled
  float* yF; int ldyF;
  bf16_t* yB; int ldyB; int Dpad;
  // split-operand GEMM operand (precise mode): fp16 hi / lo images of y * sscale in the interleaved row layout (uvtg_common.h split_col):
  // column c of row r at yS[r * ldyS + split_col(c)] (hi) and + 32 (lo), zero padding to Dpad; yUS = the same for y + pos, yPS for the
  // zero-framed conv layout
  unsigned short* yS; unsigned short* yUS; unsigned short* yPS; int ldyS; float sscale;
  // (y + pos) for rows that are video tokens: row -> (b = row / S, s = row % S), s < Lv
  const float* pos; int S, Lv;  // pos [B*Lv, D]
  const int* pos_row;           // packed rows: row -> row of the pos table (or -1); replaces the (S, Lv) arithmetic, no yP / yPF
  bf16_t* yU; float* yUF; int ldyU;
  // copy of the video rows into the zero-framed conv layout [(b*(Lv+2) + s + 1), ldyP]
  bf16_t* yP; float* yPF; int ldyP;
  // compact row streams (the input projection of the kept clips only): src_rows[row] = the row's index in the padded layout.  It keys
  // the dropout counters (same mask as the padded execution) and, with gather_x, is the row of x that is read; outputs stay compact.
  const int* src_rows; int gather_x;
  // trainable text positions (model/position_encoding.py:19-41; generic kernel only): the LayerNorm input is x + addtab[row % add_L];
  // xsum (optional, fp32 [rows, D]) receives that sum (saved for backward); with u_from_x the yU / yUF output is y + x (the row as it was
  // read, without the table) written at row src_rows[row] -- the text rows' q,k operand x + pos of layer 0
  const float* addtab; int add_L; float* xsum; int u_from_x;
  // clip-row stream of the LAST encoder layer (lean kernel only, else -4): row r READS input row x_rows[r] (packed stream: the clip rows'
  // table) or (r / x_seg) * x_seg_stride + r % x_seg (the clip rows of the token-major stream); mean, rstd and every output stay indexed by
  // r, except yB with yB_rows: row yB_rows[r] (the packed stream's encoder output, which launch_unpack_vm expands)
  int x_seg, x_seg_stride; const int* x_rows; const int* yB_rows;
};
int launch_ln_fwd(const LnFwdArgs& a, hipStream_t s);

struct LnBwdArgs {
  const float* g; int ldg;      // upstream gradient wrt LN output [rows, D] fp32
  const bf16_t* gB; int ldgB;   // ... or bf16 (used when g == nullptr)
  const float* g2; int ldg2;    // optional second upstream gradient, added (video rows only if g2_Lv>0)
  const bf16_t* g2B; int ldg2B; // ... or bf16
  int g2_S, g2_Lv;              // g2 row index = b*g2_Lv + s for s < g2_Lv when g2_S > 0; else same row
  const float* x; int ldx;      // LN input (saved)
  const bf16_t* xB; int ldxB;   // ... or bf16 (used when x == nullptr)
  const float* mean; const float* rstd; const float* gamma;
  int rows, D;
  float p_drop; unsigned long long seed; unsigned stream_id;   // same dropout mask as forward
  float* dgamma; float* dbeta;  // atomically accumulated [D]
  float* dxF; int lddxF;        // fp32 dx
  bf16_t* dxB; int lddxB;       // bf16 dx (optionally scaled per sample)
  bf16_t* dxB2; int lddxB2;     // bf16 dx, never scaled (the residual branch of the gradient stream)
  const float* rowscale; int rs_seg;
  const int* row_sample;        // packed rows: rowscale[row_sample[row]]
  int relu_from_x;              // 1: dx is masked by (x > 0): x is the output of a ReLU (input projections)
  float* partial; long long partial_floats;   // optional scratch for per-block dgamma / dbeta partials (else atomics)
  const int* src_rows; int gather_x;          // as in LnFwdArgs (generic kernel and the wide dgamma / dbeta kernel)
  // defer_blocks (host pointer, optional): the launch writes its per-block partials to `partial` and does NOT fold them -- *defer_blocks
  // receives the number of partial rows; the caller folds several launches' partials with ONE launch_ln_bwd_reduce_multi (round 5: the
  // encoder's 2 E LayerNorm backward launches were each followed by their own 5 us reduce launch)
  int* defer_blocks;
  // clip-row stream of the LAST encoder layer (lean kernel only, else -4): g / mean / rstd / rowscale are indexed by the compact row r; the
  // rows of xB and of every dx output are x_rows[r] (packed stream) or (r / x_seg) * x_seg_stride + r % x_seg (token-major stream);
  // g2_rows: g2B is read at row g2_rows[r] (the packed stream's conv-head gradient)
  int x_seg, x_seg_stride; const int* x_rows; const int* g2_rows;
  // ... and the rows in between, zeroed by the same launch: zero_n rows of every dx output, row i = zero_tab[i] (< 0: none) or
  // (i / zero_seg) * zero_stride + zero_off + i % zero_seg
  int zero_n, zero_seg, zero_stride, zero_off; const int* zero_tab;
};
int launch_ln_bwd(const LnBwdArgs& a, hipStream_t s);
// Developer configuration (include/uvtg_dev.h): the experiment switches the launch heuristics consult by name.  The table is EMPTY -- every
// switch at its shipped default -- until uvtg_dev_config_set / uvtg_dev_config_from_env is called: no entry point of the library reads the
// process environment on its own.  (A switch is looked up on first use and most call sites cache the answer: configure before the first launch.)
const char* uvtg_dev_env(const char* name);
bool ln_clip_rows_ok(int D);     // x_seg launches (forward and backward) will be taken at this width
long long ln_bwd_partial_floats(int rows, int D);      // floats of `partial` a launch of this shape needs (its per-block dgamma / dbeta rows)
constexpr int UVTG_LN_MULTI_MAX = 32;      // (2 x the engine's MAXE)
struct LnReduceMulti { const float* partial[UVTG_LN_MULTI_MAX]; float* dgamma[UVTG_LN_MULTI_MAX]; float* dbeta[UVTG_LN_MULTI_MAX]; int nblocks[UVTG_LN_MULTI_MAX]; int D, count; };
int launch_ln_bwd_reduce_multi(const LnReduceMulti& m, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// attention
// ---------------------------------------------------------------------------------------------
struct AttnArgs {
  const void* qkv; int ldqkv;   // bf16 (or fp32 when precise) [B*S, 3d]: q | k | v, q pre-scaled
  void* o; int ldo;             // bf16 / fp32 [B*S, d] (precise: optional when oS is given)
  unsigned short* oS; int ldoS;   // precise: fp16 hi / lo images of o (x UVTG_SPLIT_A_SCALE, interleaved row layout) for the split-operand out-projection
  float* lse;                   // [B, H, S]
  const unsigned char* kvalid;  // [B, S] 1 = real key
  int B, S, H, hd;
  // packed (ragged) batches: sample b owns rows seq_start[b] .. + seq_count[b] (<= S); lse / delta keep the stride S.
  // row_sample [total_rows]: sample of each packed row (attn_delta).  All null: sample b owns rows b*S .. b*S + S.
  const int* seq_start; const int* seq_count; const int* row_sample; int total_rows;
  float p_drop; unsigned long long seed; unsigned layer;
  int precise;
  // backward
  const bf16_t* dO; int lddo;   // [B*S, d]
  float* delta;                 // [B, H, S] scratch: rowsum(dO * O)
  int sample_major;             // tiled kernels' block decode: 0 = head-major over the (sample, head) pairs (XCD balance on ragged batches), 1 = sample-major (experiment)
  int ws_prio;                  // role-split dK / dV kernel: wave-priority experiment (0 none, 1 S-waves high, 2 P-waves high, 3 S-waves high during their MFMAs only)
  int delta_ready;              // 1: delta was already produced (by the dO GEMM's epilogue, GemmArgs::delta): launch_attn_bwd skips its own pass
  bf16_t* dqkv; int lddqkv;     // [B*S, 3d]
  float qscale;                 // dq is multiplied by the forward q scale
};
int launch_attn_fwd(const AttnArgs& a, hipStream_t s);
int launch_attn_bwd(const AttnArgs& a, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// small fused kernels (misc.hip)
// ---------------------------------------------------------------------------------------------
// ---- packed (ragged) execution of the encoder: valid rows of every sample + ONE representative padded clip per sample ----
struct PackTables {
  int* seq_start; int* seq_count;     // [B]
  int* row_sample; int* row_src; int* row_pos;   // [Mp]: sample, padded-layout source row (b*S + s), pos-table row or -1
  int* pad2pack;                      // [B*S]: token row -> packed row (padded clips -> the representative, padded text -> -1)
  int* grad_map;                      // [B*S]: token row -> packed row carrying its gradient (valid rows, first padded clip), else -1
  unsigned char* kvalid;              // [Mp]
  // conv-head frames (loss-only stream: ragged): sample b owns frame rows fstart[b] .. fstart[b] + kept[b] + 1 (first and last are the
  // zero rows of the k = 3 convolution); frame_valid [sum(kept + 2)] = 1 on clip rows, 0 on zero rows
  int* fstart; int* kept; float* frame_valid;
  // compact clip rows of the input projection (the clips that have a packed row: valid + representative / halo, sample after sample):
  // vin_src [Rv] = clip row b*Lv + t, vin_dst [Rv] = its packed-stream row, vin_x0 [Rv] = its row b*S + t of the padded x0;
  // vin_of [B*Lv] = compact row of a clip or -1; tin_dst [B*Lt] = packed-stream row of a text token or -1 (padded)
  int* vin_src; int* vin_dst; int* vin_x0; int* vin_of; int* tin_dst;
  int* vin_cnt;                       // [B] clips of the sample that have a packed row (a prefix of its clips)
  int* vin_sample;                    // [Rv] sample of a compact clip row (DropPath factor of the last layer's clip-row launches)
};
// keep_pad < 0: valid clips + ONE representative padded clip per sample; keep_pad >= 0: valid clips + the first keep_pad padded clips,
// each its own row (keep_pad >= Lv: every clip row), no representative; padded clips beyond that are dropped (pad2pack = -1)
int launch_pack_tables(const float* vid_mask, const float* txt_mask, int* lens_dev /* scratch [2B] */, int B, int Lv, int Lt, int keep_pad,
                       const PackTables& t, hipStream_t s);
// the fp32 rows of x0 [B*S, d] whose clip has no packed row (the compact projection never writes them): copy_rep = the sample's
// last kept clip row (the representative padded clip: every padded clip projects to the same vector without dropout), else zeros
int launch_fill_dropped_rows(float* x0, const PackTables& t, int copy_rep, int B, int S, int Lv, int d, hipStream_t s);
// frames: null = uniform frames of Lv + 2 rows per sample; else t.fstart / t.kept
int launch_unpack_vm(const bf16_t* packed, const PackTables& t, bool ragged_frames, int B, int S, int Lv, int d, bf16_t* vm_pad, hipStream_t s);
// dvm: gradient wrt the clip rows, [B * Lv, d] (uniform frames) or in frame-row space (ragged_frames)
int launch_pack_reduce_dvm(const bf16_t* dvm, const PackTables& t, int B, int S, int Lv, int Mp, int d, bool keep_pad, bool ragged_frames, bf16_t* out,
                           hipStream_t s);

int launch_ragged_to_padded(const void* packed, int src_bf16, const int* offsets, int B, int Lmax, int D, float* out, float* mask, hipStream_t s);
int launch_seq_prep(const float* vid_mask, const float* txt_mask, int B, int Lv, int Lt, int d,
                    const float* dim_t, float* pos, unsigned char* kvalid, const int* skip /* [B*Lv], < 0: row not needed; may be NULL */, hipStream_t s,
                    float* dps = nullptr, int n_dp = 0, float p_path = 0.f, unsigned long long seed = 0 /* optional: also draw the DropPath factors */,
                    unsigned* zero_words = nullptr, int n_zero = 0 /* optional: words the launch zeroes (split-K tickets) */);
int launch_droppath_scales(float* scales, int n_layers2, int B, float p, unsigned long long seed, hipStream_t s);
int launch_cast_bf16(const float* src, bf16_t* dst, long long n, hipStream_t s);
int launch_cast_f32(const bf16_t* src, float* dst, long long n, hipStream_t s);
// zero the float ranges [off[i], off[i] + n[i]) of base (the gradients no weight-gradient launch assigns)
constexpr int UVTG_MAX_ZERO_RANGES = 224;
struct ZeroRanges { long long off[UVTG_MAX_ZERO_RANGES]; int n[UVTG_MAX_ZERO_RANGES]; int count; };
// extra: one more float buffer zeroed by the same launch; frame: the two zero rows per sample of a zero-framed [fB (fLv + 2), frow_bytes] buffer too
int launch_zero_ranges(float* base, const ZeroRanges& r, hipStream_t s, float* extra = nullptr, int n_extra = 0, void* frame = nullptr, int fB = 0, int fLv = 0, int frow_bytes = 0);
constexpr int UVTG_SQSUM_FLOATS = 32 + 64 * 32;
int launch_sqsum_ranges(const float* base, const ZeroRanges& r, float* sqsum, hipStream_t s);     // sqsum[0] += sum of squares over the ranges + the 64 slots
int launch_sqsum(const float* x, long long n, float* sqsum, hipStream_t s);
int launch_cast_pad_bf16(const float* src, int rows, int cols, bf16_t* dst, int ld, hipStream_t s);
int launch_cast_pad2_bf16(const float* src0, int rows0, int cols0, bf16_t* dst0, int ld0, const float* src1, int rows1, int cols1, bf16_t* dst1, int ld1,
                          hipStream_t s);
int launch_transpose_bf16(const float* src, int rows, int cols, bf16_t* dst, int ld, hipStream_t s);  // dst[c][r]
// batched forms for the per-step operand-cache rebuild: all casts / transposes / conv re-layouts of a step in one launch each
constexpr int UVTG_MAX_PREP_OPS = 80;
struct CastOps { const float* src[UVTG_MAX_PREP_OPS]; bf16_t* dst[UVTG_MAX_PREP_OPS]; long long n[UVTG_MAX_PREP_OPS]; int count; };
// dst[c][r] = bf16(src[r][c]) (leading dimension ld); plain (optional) additionally receives the untransposed bf16 copy [rows, cols]:
// one read of the fp32 master feeds both GEMM operands (the forward's W and the dgrad's W^T)
// (cols_pad >= cols: rows [cols, cols_pad) of the transposed copy are written as zeros -- the zero padding of a K-padded dgrad operand, round 5: was a memset)
struct TransposeOps { const float* src[UVTG_MAX_PREP_OPS]; bf16_t* dst[UVTG_MAX_PREP_OPS]; bf16_t* plain[UVTG_MAX_PREP_OPS]; int rows[UVTG_MAX_PREP_OPS], cols[UVTG_MAX_PREP_OPS], ld[UVTG_MAX_PREP_OPS], cols_pad[UVTG_MAX_PREP_OPS]; int count; };
struct ConvWOps {     // kind 0: forward operand (dst[n][tap*C + c]), 1: dgrad operand (dst[c][tap'*Ntot + n_off + n] = w[n][c][2 - tap'])
  const float* w[16]; bf16_t* dst[16]; int ld[16], ntot[16], n_off[16], kind[16]; int N, C, count;
};
// split-operand weights / standalone operand splits: dst [rows, 2 * kp] fp16 = hi / lo images of src * scale in the interleaved row layout
// (uvtg_common.h split_col), columns [cols, kp) zero.  conv != 0: src is a Conv1d weight (rows, conv, 3) and image column tap * conv + c reads src[n][c][tap] (the tap-major forward operand)
struct SplitOps { const float* src[UVTG_MAX_PREP_OPS]; unsigned short* dst[UVTG_MAX_PREP_OPS]; int rows[UVTG_MAX_PREP_OPS], cols[UVTG_MAX_PREP_OPS],
                  kp[UVTG_MAX_PREP_OPS], conv[UVTG_MAX_PREP_OPS]; float scale; int count; };
int launch_split_f16_multi(const SplitOps& ops, hipStream_t s);
int launch_cast_bf16_multi(const CastOps& ops, hipStream_t s);
int launch_transpose_bf16_multi(const TransposeOps& ops, hipStream_t s);
int launch_conv_w_multi(const ConvWOps& ops, hipStream_t s);
// conv weight (N, C, 3) -> tap-major (N, 3*C) [forward operand]; and its dgrad operand (C, 3*N) with taps flipped
int launch_conv_w_fwd(const float* w, int N, int C, bf16_t* dstB, float* dstF, int ld, hipStream_t s);
int launch_conv_w_bwd(const float* w, int N, int C, bf16_t* dst, int ld, int Ntot, int n_off, hipStream_t s);

struct HeadsFinalArgs {           // last conv layer of both heads + sigmoid/sign (model/univtg.py:129-136)
  const void* h2; int ldh;        // zero-framed [(b*(Lv+2)+t), 2*d]: span half | class half; bf16 or fp32
  int precise;
  const float* w_span; const float* b_span;   // (2, d, 3), (2)
  const float* w_cls; const float* b_cls;     // (1, d, 3), (1)
  int B, Lv, d;
  float* pred_logits;  // [B, Lv, 1]
  float* pred_spans;   // [B, Lv, 2]
  // backward
  const float* g_logits; const float* g_spans;   // upstream grads
  bf16_t* dh2; int lddh;          // zero-framed gradient wrt h2 (pre-activation of layer 3 input), relu' applied
  float* dw_span; float* db_span; float* dw_cls; float* db_cls;
  float* scratch; long long scratch_floats;   // optional [B, 9 d] per-sample weight-gradient partials (else atomics)
  // ragged frames (null: sample b owns frame rows b * (Lv + 2) .. + Lv + 1): clips t >= kept[b] have no frame row; their predictions
  // come out as the constants sigmoid(0) = 0.5 / (-0.5, +0.5) (loss-only stream: no loss reads them)
  const int* fstart; const int* kept;
};
int launch_heads_final_fwd(const HeadsFinalArgs& a, hipStream_t s);
int launch_heads_final_bwd(const HeadsFinalArgs& a, hipStream_t s);

struct SaliencyArgs {             // weighted text pooling + cosine saliency (model/univtg.py:36-49,143-147)
  const float* x0; int S, Lv, Lt, B, d;   // fp32 [B*S, d] projected tokens (video rows then text rows)
  const float* txt_mask; const float* vid_mask;
  const float* w_pool;            // [d]
  float* alpha;                   // [B, Lt] softmax weights (saved)
  float* pooled;                  // [B, d]  (txt_mem_proj)
  float* cosv;                    // [B, Lv] raw cosine (saved for the criterion)
  float* sal;                     // [B, Lv] cosine + log-mask
  float* vnorm; float* qnorm;     // [B, Lv], [B]
  // backward
  const float* g_sal;             // [B, Lv]  d/d saliency_scores
  const float* g_pooled;          // [B, d]   d/d txt_mem_proj
  const float* g_vid; long long gv_sb, gv_st;   // d/d vid_mem_proj: g_vid[b*gv_sb + t*gv_st + c], or null
  const float* g_vrow; const long long* pos_idx; // optional compact extra: g_vrow[b, :] is added on row pos_idx[b]
  const float* dx0;               // [B*S, d] encoder gradient wrt x0 (read only), fp32 ...
  const bf16_t* dx0B;             // ... or bf16 (used when dx0 == nullptr)
  const int* dx0_map;             // packed encoder stream: token row (b*S + s) -> row of dx0 / dx0B, or -1 (no gradient)
  float* dq; float* dlog;         // scratch [B, d], [B, Lt]
  bf16_t* out_vid; bf16_t* out_txt;   // bf16 [B*Lv, d] / [B*Lt, d]: dx0 + saliency-branch gradients, re-packed per modality
  const int* vout_map;            // clip row (b*Lv + t) -> row of out_vid (compact input projection), < 0: not written; null: identity
  float* dw_pool;                 // [d] atomically accumulated
  const float* g_txt_rows;        // optional fp32 [B*Lt, d]: extra gradient on the text rows of x0 (the trainable text positions' LayerNorm input)
};
int launch_saliency_fwd(const SaliencyArgs& a, hipStream_t s);
int launch_heads_saliency_fwd(const HeadsFinalArgs& h, const SaliencyArgs& a, hipStream_t s);   // heads_final_fwd + saliency_fwd, fused when B >= 128
int launch_saliency_bwd(const SaliencyArgs& a, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// dense criterion (model/univtg.py:195-282), values + gradients, no host synchronisation
// ---------------------------------------------------------------------------------------------
struct LossArgs {
  int B, Lv, d;
  const float* pred_logits;   // [B, Lv]
  const float* pred_spans;    // [B, Lv, 2]
  const float* vid; long long vid_sb, vid_st;   // vid_mem_proj[b, t, c] = vid[b*vid_sb + t*vid_st + c]
  const float* txt;           // [B, d] txt_mem_proj
  const float* timestamp;     // [B, Lv, 2]
  const float* ts_mask;       // [B, Lv]
  const float* ts_window;     // [B, Lv]
  const float* span_nn;       // [B, Lv, 2]
  const float* sal_tgt;       // [B, Lv] or null
  const long long* pos_idx;   // [B] or null  (saliency_pos_labels[:, 0])
  float eos_coef;
  int do_spans, do_labels, do_saliency;
  // optional (all three or none): per-clip cosine(vid_mem_proj, txt_mem_proj) [B, Lv], |vid_mem_proj| [B, Lv], |txt_mem_proj| [B] as the model
  // forward's saliency pass already computed them (uvtg_forward_saliency_stats) -- saves the criterion's own pass over vid_mem_proj
  const float* cos_c; const float* vnorm_c; const float* qnorm_c;
  float* ws;                  // workspace, uvtg_loss_ws_floats(B, Lv) floats
  float* losses;              // [8]: loss_b, loss_g, loss_f, loss_s_inter, loss_s_intra, active, Nwin, Nvalid
  // backward
  const float* go;            // [5] upstream gradient per loss (device memory)
  float* g_logits;            // [B, Lv]
  float* g_spans;             // [B, Lv, 2]
  float* g_vid;               // [B, Lv, d] dense, or NULL = compact mode
  float* g_txt;               // [B, d]  (compact mode: inter-video part only)
  float* g_cos;               // [B, Lv] d/d cosine(vid, txt)  (= d/d saliency_scores)
  float* g_vrow;              // [B, d]  inter-video gradient wrt vid_mem_proj[b, pos_b, :]
};
long long loss_ws_floats(int B, int Lv, int d);
// class term of the 'saliency_cls' loss (model/univtg.py:314-324): see losses.hip, cls_nce_*
struct ClsNceArgs {
  int B, C, d;                  // samples, classes, width
  const float* vid;             // vid_mem_proj addressed as vid[b * vid_sb + t * vid_st + c]
  long long vid_sb, vid_st;
  const long long* pos_idx;     // [B] positive clip of every sample
  const float* cls;             // [C, d] pooled class-name features
  const float* cls_idx;         // [B, C] 0 / 1 (multi-hot)
  const float* active;          // device flag (losses_out[5] of uvtg_criterion_fwd: 0 when saliency_scores sums to zero) or NULL
  float* ws;                    // uvtg_cls_nce_ws_floats
  float* loss;                  // [1]
  const float* go;              // backward: upstream gradient of the loss [1]
  float* g_vid;                 // backward: dense [B, Lv, d] gradient buffer, ZERO-FILLED by the caller; row pos_idx[b] of sample b is written
  long long gv_sb, gv_st;
  float* g_cls;                 // backward: [C, d]
};
long long cls_nce_ws_floats(int B, int C);
int launch_cls_nce_fwd(const ClsNceArgs& a, hipStream_t s);
int launch_cls_nce_bwd(const ClsNceArgs& a, hipStream_t s);
int launch_losses_fwd(const LossArgs& a, hipStream_t s);
int launch_losses_bwd(const LossArgs& a, hipStream_t s);
