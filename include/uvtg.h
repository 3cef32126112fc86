/*
 * uvtg.h -- C ABI of libuvtg.so: the MI355X (gfx950) implementation of the UniVTG hot path.
 *
 * The reference (showlab/UniVTG) has no native code and no FFI: its hot path is PyTorch ATen calls
 * issued from model/univtg.py.  This header is the boundary a reference maintainer binds instead
 * (ctypes stub in INTEGRATION.md).  Each entry point names the reference code it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless it says "host"; plain pointers + sizes, no torch types
 *   - the library never allocates or frees device memory: callers (PyTorch) own params, activations,
 *     workspaces and outputs; required sizes come from the *_bytes() / *_floats() queries
 *   - all work is enqueued on `stream` (a hipStream_t) and returns immediately
 *   - return value: 0 = ok, >0 = hipError_t, <0 = invalid argument (see uvtg_strerror)
 *   - threading: one caller thread per process/GPU (the reference runs one process per GPU,
 *     scripts/pretrain.sh:98); re-entrant across processes
 *   - fp32 tensors must be 16-byte aligned, row-major, innermost dimension contiguous
 */
#ifndef UVTG_H
#define UVTG_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* uvtg_stream_t;   /* hipStream_t */

/* Geometry + mode of one forward/backward call.  Mirrors the fields build_model() reads from `args`
 * (model/univtg.py:409-450) plus the batch shape. */
typedef struct uvtg_dims {
  int struct_size;           /* = sizeof(uvtg_dims) of the header the caller was built against; every entry point rejects a
                                mismatch (-18) instead of reading a shorter / differently laid out struct (ABI break guard)  */
  int B, Lv, Lt;             /* batch, padded #clips, padded #text tokens (S = Lv + Lt)            */
  int d, H, F, E;            /* hidden_dim, nheads, dim_feedforward, enc_layers                    */
  int Dv, Dt;                /* v_feat_dim (incl. TEF), t_feat_dim                                 */
  int n_proj;                /* n_input_proj: 1, 2 or 3 LinearLayer blocks per modality (model/univtg.py:89-100)  */
  int precise;               /* 0: bf16 MFMA encoder/heads; 1: split-operand fp16 hi+lo images, three products (fp32-class), forward only.
                                OPERAND RANGE of mode 1 (and of proj_precise): an operand x is held as fp16(x s) + fp16(x s - hi), s = 16 for
                                activations / 64 for weights, clamped at +-65000 -- |activation| > ~4060 (LayerNorm outputs, GELU / ReLU hidden
                                rows, q, k, v, attention probabilities) or |weight| > ~1015 SATURATES silently instead of being fp32-class.
                                Trained UniVTG checkpoints are orders of magnitude below both; any other value than 0 / 1 is rejected (-25)  */
  int training;              /* 1: keep activations for backward, apply dropout / DropPath         */
  int proj_precise;          /* 1: input projections on split operands even when precise==0 (keeps the
                                saliency logits within 1e-4 of the fp32 reference)                 */
  float p_in, p_attn, p_path;/* input_dropout, dropout (attention), droppath                       */
  unsigned long long seed;   /* Philox seed of this step (stochastic ops are counter-based)        */
  int loss_only;             /* 1: the caller consumes only what the dense criterion consumes (a native training step): outputs
                                at PADDED clip positions may differ from the reference's -- see lens_host below               */
  int use_txt_pos;           /* --use_txt_pos (model/univtg.py:123): pos of the text rows = Dropout(LayerNorm(x_txt + E[0..Lt))),
                                model/position_encoding.py:19-41, dropout p = p_in; padded execution only (lens_host ignored)  */
  int max_q_l;               /* rows of txt_position_embed.position_embeddings.weight (>= Lt); read only with use_txt_pos      */
} uvtg_dims;

/* ---- parameter table -------------------------------------------------------------------------
 * Parameters are passed as an array of fp32 device pointers in the reference's state_dict order
 * (model/univtg.py:76-103):
 *   for l in 0..E-1 (base 12*l): in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias,
 *       linear1.weight, linear1.bias, linear2.weight, linear2.bias, norm1.weight, norm1.bias,
 *       norm2.weight, norm2.bias
 *   then (base 12*E): token_type_embeddings.weight,
 *       span_embed.layers.{0,1,2}.{weight,bias}, class_embed.layers.{0,1,2}.{weight,bias},
 *       input_txt_proj.{0..n_proj-1}.{LayerNorm.weight, LayerNorm.bias, net.1.weight, net.1.bias},
 *       input_vid_proj.{0..n_proj-1}.{...}, weightedpool.weight,
 *       and -- only with dims.use_txt_pos -- txt_position_embed.position_embeddings.weight [max_q_l, d],
 *       txt_position_embed.LayerNorm.{weight, bias}
 *   (without --use_txt_pos the text position table is NOT in the table: the reference does not read it then and it receives no
 *    gradient there either -- model/univtg.py:123.)
 * uvtg_param_count() = 12*E + 14 + 8*n_proj (+ 3 with use_txt_pos).  Gradients come back in ONE flat fp32 buffer; parameter i starts
 * at element offsets[i] (uvtg_param_offsets; each start is a multiple of 4 elements). */
int uvtg_param_count(const uvtg_dims* dm);
int uvtg_param_numel(const uvtg_dims* dm, int index, long long* numel);
int uvtg_param_offsets(const uvtg_dims* dm, long long* offsets /* host, [count + 1] */);

/* ---- sizes -------------------------------------------------------------------------------- */
size_t uvtg_workspace_bytes(const uvtg_dims* dm);   /* activations + scratch for forward(+backward) */
size_t uvtg_wcache_bytes(const uvtg_dims* dm);      /* prepared GEMM operands (bf16 copies, transposes) */
long long uvtg_loss_ws_floats(int B, int Lv, int d);

/* Re-layout / down-cast the weights into the MFMA operand cache.  Call after every parameter update
 * (replaces nothing in the reference: ATen reads the fp32 weights directly). */
int uvtg_prepare_weights(const uvtg_dims* dm, const float* const* params, void* wcache, uvtg_stream_t stream);

/* ---- model forward: replaces Model.forward (model/univtg.py:105-155) --------------------------
 * inputs : src_txt [B,Lt,Dt], src_txt_mask [B,Lt], src_vid [B,Lv,Dv], src_vid_mask [B,Lv]  (fp32, 0/1 masks)
 *          dim_t [d]: the sine-embedding denominators 10000^(2*(i/2)/d) (model/position_encoding.py:75-78)
 * outputs: x0 [B,S,d]   projected tokens (vid rows then txt rows); vid_mem_proj = x0[:, :Lv]
 *          pred_logits [B,Lv,1], pred_spans [B,Lv,2], txt_mem_proj [B,1,d], saliency [B,Lv]
 *          memory [B,S,d] encoder output (optional, may be NULL)
 * lens_host (optional): the per-sample valid lengths the caller's collate already knows on the host (the reference's
 *          pad_sequences_1d computes them, utils/tensor_utils.py:34-53); masks must be the matching prefix masks (the device-side
 *          tables are rebuilt from the masks; lens_host is read synchronously, for the row count only, and may be freed on return).
 *          When given (bf16 mode) the encoder runs on a packed row stream that reproduces the padded computation exactly:
 *            - eval, or training with p_in == 0 and p_attn == 0: valid clips + ONE representative padded clip per sample + valid text
 *              tokens (all padded clips of a sample are then identical rows; ~25 % fewer rows on ragged batches);
 *            - training with input or attention dropout: EVERY clip row (each padded clip draws its own mask in the reference,
 *              model/univtg.py:392-404) + the valid text tokens (padded text tokens are masked keys whose outputs nobody reads);
 *            - the same with dims.loss_only (and p_attn == 0): valid clips + the FIRST THREE padded clips of every sample + valid
 *              text.  A padded clip is never a key, so it reaches a valid position only through the 3-layer k = 3 conv heads
 *              (receptive field +-3); every loss masks the padded positions (model/univtg.py:195-282).  Losses and ALL parameter
 *              gradients are therefore exactly the reference's; pred_* at padded positions beyond the halo come out as the heads'
 *              response to zero rows (finite, meaningless -- nothing reads them in training).
 *          In every variant the VIDEO INPUT PROJECTION runs only on the clips that own a packed row (the feature LayerNorm gathers
 *          them; dropout counters stay keyed by the padded row, so the masks are the padded execution's).  x0 keeps its padded
 *          layout: the rows of the other padded clips are copies of the representative's row (first variant: bit-identical to
 *          computing them) or zeros (loss-only variant: saliency at those positions is the masked constant + 0).
 *          The choice is a function of (dims, lens_host != NULL) only, so uvtg_backward -- which must get the same array -- makes the
 *          same one.  With memory != NULL: ignored in eval calls, error -24 in training calls.
 * memory: in the bf16 mode the LAST encoder layer computes its FFN half for the clip rows only (the text rows of the encoder output are read by
 *          nobody, model/univtg.py:127): an eval call that passes memory != NULL runs every row instead, a TRAINING call that does is refused
 *          with -24 (uvtg_backward, which re-derives the row layout from dims alone, could not know). */
int uvtg_forward(const uvtg_dims* dm, const float* const* params, const void* wcache,
                 const float* src_txt, const float* src_txt_mask, const float* src_vid, const float* src_vid_mask,
                 const float* dim_t,
                 float* x0, float* pred_logits, float* pred_spans, float* txt_mem_proj, float* saliency,
                 float* memory, void* workspace, uvtg_stream_t stream,
                 const int* lens_host /* optional, HOST memory [2B]: clips then text tokens per sample */);

/* ---- model backward: replaces autograd through Model.forward ------------------------------------
 * Needs the workspace, inputs and outputs (x0, pred_*, txt_mem_proj) of the matching
 * uvtg_forward(training=1) call.  Upstream gradients (any may be
 * NULL = zero): g_logits [B,Lv,1], g_spans [B,Lv,2], g_saliency [B,Lv], g_txt_mem [B,1,d],
 * g_vid_mem addressed as g_vid_mem[b*g_vid_sb + t*g_vid_st + c] (so the gradient of the whole x0
 * buffer, strides S*d and d, can be passed as is).  g_vrow [B,d] + pos_idx [B] (optional pair): extra gradient
 * on row pos_idx[b] of vid_mem_proj -- the compact form uvtg_criterion_bwd emits instead of a dense g_vid.  grads: flat fp32 buffer (uvtg_param_offsets), overwritten.
 * ready_events (optional, for overlapping the data-parallel gradient exchange with the rest of backward): event 0 is recorded
 * on `stream` once the span_embed / class_embed gradients (table entries 12E+1 .. 12E+12) are final, event 1 + i once those of
 * encoder layer E-1-i are; everything else is final when the call's work completes.  An event is recorded no earlier than its range is
 * final and in index order, but several may be recorded at the same point of the stream: by default every weight gradient of the heads and
 * of the encoder leaves in ONE deferred launch behind the encoder loop and all E + 1 events are recorded there (uvtg_backward_event_groups
 * reports the batching; a developer switch records the heads + layers E-1 .. 1 behind layer 1 instead). */
/* How the ready_events of uvtg_backward are batched: writes into last_event[0 .. n) the index of the LAST event of each group (n = return value,
 * <= E + 1; negative = error) -- the events of one group are recorded at the same point of the stream, so a data-parallel caller waits for
 * the last event of a group and exchanges the group's ranges in ONE coalesced collective. */
int uvtg_backward_event_groups(int E, int* last_event /* [E + 1] */);
int uvtg_backward(const uvtg_dims* dm, const float* const* params, const void* wcache,
                  const float* src_txt, const float* src_txt_mask, const float* src_vid, const float* src_vid_mask,
                  const float* x0, const float* pred_logits, const float* pred_spans, const float* txt_mem_proj,
                  const float* g_logits, const float* g_spans, const float* g_saliency,
                  const float* g_txt_mem, const float* g_vid_mem, long long g_vid_sb, long long g_vid_st,
                  const float* g_vrow, const long long* pos_idx,
                  float* grads, void* workspace, uvtg_stream_t stream,
                  void* const* ready_events /* hipEvent_t[E + 1] or NULL */, int n_events /* E + 1 or 0 */,
                  const int* lens_host /* as passed to the matching uvtg_forward, or NULL */);

/* ---- criterion: replaces SetCriterion.forward + its autograd (model/univtg.py:195-282,338-351) ---
 * vid_mem_proj is addressed as vid[b*vid_sb + t*vid_st + c] so that the strided view of x0 works.
 * losses_out [8] (device): loss_b, loss_g, loss_f, loss_s_inter, loss_s_intra, saliency_active, Nwin, Nvalid.
 * which: bit0 spans, bit1 labels, bit2 saliency (the `losses` list of build_model). */
int uvtg_criterion_fwd(int B, int Lv, int d, int which, float eos_coef,
                       const float* pred_logits, const float* pred_spans,
                       const float* vid, long long vid_sb, long long vid_st, const float* txt_mem,
                       const float* timestamp, const float* timestamp_mask, const float* timestamp_window,
                       const float* span_labels_nn, const float* saliency_scores, const long long* pos_idx,
                       float* loss_ws, float* losses_out,
                       const float* cos_cached, const float* vnorm_cached, const float* qnorm_cached /* optional, all three or NULL:
                       cosine(vid, txt) [B,Lv], |vid| [B,Lv], |txt| [B] from uvtg_forward_saliency_stats (same values the
                       criterion would compute from vid / txt_mem itself; saves its pass over vid_mem_proj) */,
                       uvtg_stream_t stream);
/* go [5] (device): upstream gradient of each of the five losses.  Must follow the matching _fwd call.
 * Outputs: g_logits [B,Lv], g_spans [B,Lv,2], g_cos [B,Lv] (gradient wrt cosine(vid_mem_proj, txt_mem_proj), i.e.
 * wrt saliency_scores), g_vrow [B,d] (inter-video gradient wrt vid_mem_proj[b, pos_b, :]), g_txt [B,d].
 * Dense mode (g_vid != NULL): also g_vid [B,Lv,d] = full gradient wrt vid_mem_proj and g_txt = full gradient wrt
 * txt_mem_proj (what autograd needs).  Compact mode (g_vid == NULL): g_txt holds the inter-video part only; feed
 * g_cos as g_saliency and (g_vrow, pos_idx) to uvtg_backward, which differentiates the cosine itself. */
int uvtg_criterion_bwd(int B, int Lv, int d, int which, float eos_coef,
                       const float* pred_logits, const float* pred_spans,
                       const float* vid, long long vid_sb, long long vid_st, const float* txt_mem,
                       const float* timestamp, const float* timestamp_mask, const float* timestamp_window,
                       const float* span_labels_nn, const float* saliency_scores, const long long* pos_idx,
                       float* loss_ws, const float* losses_out, const float* go,
                       float* g_logits, float* g_spans, float* g_vid, float* g_txt, float* g_cos, float* g_vrow,
                       const float* cos_cached, const float* vnorm_cached, const float* qnorm_cached /* as passed to _fwd */,
                       uvtg_stream_t stream);
/* ---- class term of the 'saliency_cls' loss: replaces model/univtg.py:314-324 and its autograd (the TAL pre-training branch, selected by
 * 'tal' in train_path, model/univtg.py:436-438).  v_b = vid[b*vid_sb + pos_idx[b]*vid_st + .], cls [C,d] = cls_mem_proj (the class-name
 * features pooled by the text path of uvtg_forward), cls_idx [B,C] 0 / 1 (multi-hot, as float):
 * loss = - sum over marked (b,j) of log_softmax_j(cos(v_b, cls_j) / 0.07) / #marked.  `active` (device, may be NULL): losses_out + 5 of
 * uvtg_criterion_fwd -- the saliency_scores.sum() == 0 early-out (model/univtg.py:288-290) zeroes loss and gradients without a host sync.
 * ws: uvtg_cls_nce_ws_floats(B, C) floats, shared by _fwd and _bwd.  _bwd: go [1] = upstream gradient of the loss; g_vid is a gradient
 * buffer for vid_mem_proj ZERO-FILLED by the caller, addressed like vid with (gv_sb, gv_st): row pos_idx[b] of sample b is written;
 * g_cls [C,d] is overwritten. */
long long uvtg_cls_nce_ws_floats(int B, int C);
int uvtg_cls_nce_fwd(int B, int C, int d, const float* vid, long long vid_sb, long long vid_st, const long long* pos_idx,
                     const float* cls, const float* cls_idx, const float* active, float* ws, float* loss_out, uvtg_stream_t stream);
int uvtg_cls_nce_bwd(int B, int C, int d, const float* vid, long long vid_sb, long long vid_st, const long long* pos_idx,
                     const float* cls, const float* cls_idx, const float* active, float* ws, const float* go,
                     float* g_vid, long long gv_sb, long long gv_st, float* g_cls, uvtg_stream_t stream);

/* Where the saliency pass of the last uvtg_forward on `workspace` left cosine(vid_mem_proj, txt_mem_proj) [B,Lv], |vid_mem_proj| [B,Lv] and
 * |txt_mem_proj| [B] (device addresses inside the workspace; valid until the next uvtg_forward on it). */
int uvtg_forward_saliency_stats(const uvtg_dims* dm, void* workspace, const float** cosv, const float** vnorm, const float** qnorm);

/* ---- kernel-level entry points (used by the parity tests; same kernels the engine launches) ----- */
/* C[M,N] = A[M,K] * W[N,K]^T + bias (nn.Linear).  bf16: A,W bf16, C fp32.  act: 0/1 relu/2 gelu */
int uvtg_linear_bf16(const void* A, const void* W, const float* bias, float* C, int M, int N, int K, int act,
                     uvtg_stream_t stream);
/* The same in the precise ("fp32x3") arithmetic: both operands as fp16 hi | lo IMAGES, x = (hi + lo) / scale to ~22 bits, the product as
 * hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation.  uvtg_split_f16 builds the images of an fp32 [rows, cols]
 * matrix: dst is fp16 [rows, 2 * kp] (hi image in columns [0, kp), lo image in [kp, 2 kp), columns [cols, kp) zero, kp a multiple of 64;
 * is_weight selects the operand scale: activations x16, weights x64).  uvtg_linear_split: A [M, 2 Kp] (activation images), W [N, 2 Kp]
 * (weight images), C fp32 [M, N]. */
int uvtg_split_f16(const float* src, void* dst, int rows, int cols, int kp, int is_weight, uvtg_stream_t stream);
int uvtg_linear_split(const void* A, const void* W, const float* bias, float* C, int M, int N, int Kp, int act,
                      uvtg_stream_t stream);
/* Both GEMMs with the split-K workspace uvtg_forward gives its small launches (<= half as many 128 x 256 tiles as the chip has CUs: inference
 * batches): K is cut into <= 4 parts, one workgroup each; the parts meet through fp32 partial-tile slabs and one ticket per tile, the part that
 * arrives last sums ALL parts in part order (bit-reproducible) and runs the epilogue.  sk_ws: uvtg_linear_sk_ws_floats() floats, 16-byte
 * aligned; its first 256 words (the tickets) must be ZERO before the first call (the kernel leaves them zero).  Shapes that do not qualify run
 * exactly as uvtg_linear_bf16 / uvtg_linear_split do.  (Experiment knobs of the split: include/uvtg_dev.h.) */
long long uvtg_linear_sk_ws_floats(void);
int uvtg_linear_bf16_sk(const void* A, const void* W, const float* bias, float* C, int M, int N, int K, int act, float* sk_ws,
                        uvtg_stream_t stream);
int uvtg_linear_split_sk(const void* A, const void* W, const float* bias, float* C, int M, int N, int Kp, int act, float* sk_ws,
                         uvtg_stream_t stream);
/* dW[N,K] += dY[M,N]^T * X[M,K] (bf16 operands, fp32 atomic accumulate), dbias[N] += colsum(dY) (may be NULL) */
int uvtg_wgrad_bf16(const void* dY, const void* X, float* dW, float* dbias, int M, int N, int K, int splits,
                    uvtg_stream_t stream);
/* same, with caller scratch of uvtg_wgrad_scratch_floats(M, N, K) floats: the large shapes then run the 256-tile kernel
 * (split partial tiles as plain fp32 slabs + a reduce pass instead of fp32 atomics) */
long long uvtg_wgrad_scratch_floats(int M, int N, int K);
int uvtg_wgrad_bf16_ws(const void* dY, const void* X, float* dW, float* dbias, int M, int N, int K, float* scratch,
                       long long scratch_floats, uvtg_stream_t stream);
/* `count` (<= 24) weight gradients over the SAME M reduction rows in ONE launch and without a reduce pass: dW[i][N_i,K_i] = dY[i][M,N_i]^T *
 * X[i][M,K_i] (ASSIGNED), dbias[i][N_i] += colsum(dY[i]) (host arrays of device pointers; dbias or its entries may be NULL).  Every N_i, K_i
 * a multiple of 256.  Whole 256 x 256 tiles per workgroup; the tiles that do not fill a last CU round are cut into <= 3 row ranges whose
 * parts meet through write-through slabs and one ticket per tile (the last part to arrive folds the others).  slabs:
 * uvtg_wgrad_multi_slab_floats(total tiles) floats, 16-byte aligned; tickets: >= total tiles unsigned, ZEROED by the caller before the call.
 * -2: shapes / split this path does not take (use uvtg_wgrad_bf16_ws per gradient).  uvtg_backward runs the encoder's 5 E weight gradients
 * through it when no per-layer readiness events are requested. */
long long uvtg_wgrad_multi_slab_floats(int total_tiles);
int uvtg_wgrad_bf16_multi(int count, const void* const* dY, const int* N, const void* const* X, const int* K, float* const* dW,
                          float* const* dbias, int M, float* slabs, long long slab_floats, unsigned* tickets, int n_tickets,
                          uvtg_stream_t stream);
int uvtg_cast_bf16(const float* src, void* dst, long long n, uvtg_stream_t stream);
/* bf16 -> fp32 (exact).  With uvtg_cast_bf16: the optional bf16 gradient buckets of the data-parallel exchange. */
int uvtg_cast_f32(const void* src, float* dst, long long n, uvtg_stream_t stream);
/* LayerNorm rows (eps 1e-5): y fp32, optional mean/rstd */
int uvtg_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                       int rows, int D, uvtg_stream_t stream);
int uvtg_layernorm_bwd(const float* g, const float* x, const float* mean, const float* rstd, const float* gamma,
                       float* dx, float* dgamma, float* dbeta, int rows, int D, uvtg_stream_t stream);
/* masked MHA core on packed qkv [B*S, 3d] (q pre-scaled). bf16: qkv/o bf16; precise: fp32 */
int uvtg_attention_fwd(const void* qkv, const unsigned char* kvalid, void* o, float* lse,
                       int B, int S, int H, int hd, int precise, uvtg_stream_t stream);
int uvtg_attention_bwd(const void* qkv, const unsigned char* kvalid, const void* o, const float* lse,
                       const void* dO, float* delta_scratch, void* dqkv, float qscale,
                       int B, int S, int H, int hd, uvtg_stream_t stream);
int uvtg_sine_position(const float* vid_mask, const float* txt_mask, const float* dim_t, float* pos,
                       unsigned char* kvalid, int B, int Lv, int Lt, int d, uvtg_stream_t stream);

/* ---- wire format: replaces pad_sequences_1d (utils/tensor_utils.py:5-53) + the padded H2D copies of
 * prepare_batch_inputs_mr (main/dataset.py:1071-1100) for one key: only the VALID rows cross PCIe.
 * packed [sum(len), D]: the samples' rows back to back (fp32, or bf16 when src_bf16), offsets [B + 1] row offsets (device int),
 * out [B, Lmax, D] fp32 zero-padded, mask [B, Lmax] fp32 0/1 (may be NULL). */
int uvtg_ragged_to_padded(const void* packed, int src_bf16, const int* offsets, int B, int Lmax, int D, float* out, float* mask,
                          uvtg_stream_t stream);

/* ---- Hungarian matcher: replaces HungarianMatcher.forward (model/matcher.py:36-100) ------------
 * cost[b*Q + q, j] = w_span*L1(cxw) + w_giou*(-gIoU(xx)) + w_class*(-softmax(logits)[.,0]) for the targets
 * of sample b (tgt_off[b] .. tgt_off[b+1]); then a per-sample rectangular LSAP on device.
 * out_pred / out_tgt [B, max_t] int64 index pairs sorted by prediction index (-1 padded); n_match [B]. */
int uvtg_hungarian(const float* pred_logits, int n_cls, const float* pred_spans_cxw, int B, int Q,
                   const float* tgt_cxw, const int* tgt_off, int max_t,
                   float w_class, float w_span, float w_giou,
                   float* cost_out /* [B, Q, max_t] */, long long* out_pred, long long* out_tgt,
                   int* n_match, uvtg_stream_t stream);

/* ---- Moment-DETR set criterion: replaces SetCriterion.get_loss over one decoder layer's outputs
 * (model/moment_detr.py:166-365, span_loss_type "l1"), given the index pairs uvtg_hungarian produced.
 * pred_logits [B,Q,2], pred_spans_cxw [B,Q,2]; tgt_cxw, tgt_off, match_pred, match_tgt, n_match, max_t exactly as in uvtg_hungarian.
 * saliency_scores [B,L] with pos_idx/neg_idx [B,n_pairs] int64 (NULL: loss_s_intra = 0, :257-258);
 * proj_queries [B,Q,D] / proj_txt_mem [B,T,D] (NULL: no contrastive_align term).
 * losses [6] (device) = loss_b, loss_g, loss_f, class_error, loss_s_intra, loss_contrastive_align.
 * go [6] (device) or NULL: upstream gradient of each entry of losses (entry 3 is ignored); when given, the d_* arrays that
 * are non-NULL receive the gradient with respect to the same-shaped prediction tensor (every element written).
 * partials [B,8] float scratch.  Sums run in sample order: results are bit-reproducible. */
int uvtg_detr_criterion(const float* pred_logits, const float* pred_spans_cxw, int B, int Q, const float* tgt_cxw,
                        const int* tgt_off, const long long* match_pred, const long long* match_tgt, const int* n_match, int max_t,
                        const float* saliency_scores, const long long* pos_idx, const long long* neg_idx, int n_pairs, int L,
                        const float* proj_queries, const float* proj_txt_mem, int T, int D,
                        float eos_coef, float temperature, float saliency_margin, const float* go, float* partials,
                        float* losses, float* d_logits, float* d_spans, float* d_saliency, float* d_proj_queries,
                        float* d_proj_txt_mem, uvtg_stream_t stream);

/* ---- inference glue: replaces main/inference_mr.py:109-160 + utils/temporal_nms.py -----------
 * windows[b,t] = clamp((timestamp + pred_spans) * duration[b], 0, duration[b]); scores masked to 0
 * on padded clips; rank by score (stable, descending); greedy hull-IoU NMS.
 * order [B,Lv] int32 = clip indices ranked; keep [B, max_after] int32 = ranked positions kept (-1 padded);
 * windows_out [B,Lv,3] fp64 rows (st, ed, score) in ranked order, each rounded to 4 decimals exactly like
 * float(f"{x:.4f}") (main/inference_mr.py:159), so they compare equal to the reference's Python floats. */
int uvtg_decode_rank_nms(const float* pred_logits, const float* pred_spans, const float* timestamp,
                         const float* timestamp_mask, const float* durations, int B, int Lv,
                         double nms_thd /* the reference's Python float: compared in double, so that a hull-IoU of exactly 7/10 is NOT > 0.7 */,
                         int max_before, int max_after,
                         double* windows_out, int* order, int* keep, int* n_keep, uvtg_stream_t stream);
/* The full per-batch tail of compute_mr_results + eval_epoch_post_processing (main/inference_mr.py:109-192,31-40): as above, plus
 *   clip_length > 0 : PostProcessorDETR's round_multiple (eval/postprocessing.py:46-51) on the 4-decimal rows BEFORE the NMS, as the
 *                     reference orders them -- fp32 torch.round(w / clip_length) * clip_length (half-to-even), so the windows come
 *                     out as exact integer multiples of clip_length; score re-rounded to 4 decimals (:35).  <= 0: off (--round_multiple -1)
 *   saliency_out    : (optional, [B,Lv] fp32) pred_saliency_scores = fp16(saliency) (+ pred_logits when eval_mode_add, the
 *                     reference's --eval_mode add), main/inference_mr.py:124-128; the caller truncates row b to its len_v (:133-136). */
int uvtg_postprocess_mr(const float* pred_logits, const float* pred_spans, const float* saliency /* [B,Lv] or NULL */,
                        const float* timestamp, const float* timestamp_mask, const float* durations, int B, int Lv,
                        float clip_length, int eval_mode_add, double nms_thd, int max_before, int max_after,
                        double* windows_out, int* order, int* keep, int* n_keep, float* saliency_out /* or NULL */,
                        uvtg_stream_t stream);

/* ---- training-step shell: replaces clip_grad_norm_ + AdamW.step (main/train_vlp_ddp.py:66-68,
 * main/config.py:349-350) over ONE flat fp32 buffer laid out by uvtg_param_offsets.
 * g' = grads * grad_scale (1/world after an all-reduce-sum), clipped to global norm max_norm (<=0: off);
 * torch.optim.AdamW semantics (decoupled decay, bias correction with `step` starting at 1).  scratch: UVTG_ADAMW_SCRATCH_FLOATS floats
 * (per-block partial sums of the squared norm, folded in a fixed order: the clipping coefficient -- hence the update -- is bit-identical
 * on every rank that holds the same reduced gradients, like the reference's clip_grad_norm_ on every DDP replica). */
#define UVTG_ADAMW_SCRATCH_FLOATS 1024
int uvtg_adamw_clip_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                         float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                         float max_norm, float grad_scale, float* scratch, uvtg_stream_t stream);
/* The same with the squared gradient norm already on the device: uvtg_backward accumulates it while it writes the gradients
 * (uvtg_backward_gradnorm2 returns its address inside the workspace), so a single-rank step needs no extra pass over the
 * gradient buffer.  NOT valid after a gradient all-reduce (the norm must be the reduced gradients'): use uvtg_adamw_clip_step there. */
int uvtg_adamw_clip_step_prenorm(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                                 float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                 float max_norm, float grad_scale, const float* sqnorm_dev, uvtg_stream_t stream);
const float* uvtg_backward_gradnorm2(const uvtg_dims* dm, void* workspace);


/* Data-parallel runs: leave k compute units out of every persistent GEMM grid (NT tiles and weight-gradient units are sized for
 * CUs - k), so that RCCL's all-reduce kernels on the communication stream always find free CUs while backward runs (the reference
 * gets this from DDP's bucket hooks running beside cuBLAS kernels that do not fill the chip, main/train_vlp_ddp.py:272-275).
 * k = 0 (default): whole chip.  Process-wide; returns the number of CUs the grids will use. */
int uvtg_set_reserved_cus(int k);
const char* uvtg_strerror(int code);
int uvtg_version(void);

#ifdef __cplusplus
}
#endif
#endif /* UVTG_H */
