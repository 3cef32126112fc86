/*
 * uvtg_dev.h -- developer side of libuvtg.so: measurement hooks (bench.py, tools/) and experiment / parity-test knobs of the GEMM
 * launch heuristics.  NOT part of the boundary a reference maintainer binds (that is include/uvtg.h: the model, the criterion, the
 * post-processing, the training-step shell); nothing here changes results beyond fp32 summation order, every knob is process-wide and
 * defaults to the shipped behaviour.  Same conventions as uvtg.h.
 */
#ifndef UVTG_DEV_H
#define UVTG_DEV_H
#include "uvtg.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- developer configuration (round 6) -------------------------------------------------------------------------------------------
 * The library's experiment switches (the UVTG_*_OFF / UVTG_* names quoted in DESIGN.md and profiles/) live in ONE process-wide name -> value
 * table that starts EMPTY: every switch at its shipped default.  No entry point of libuvtg.so reads the process environment on its own -- a
 * caller that binds include/uvtg.h cannot be steered by a stray variable.  The developer tools opt in:
 *   uvtg_dev_config_from_env()      copies every UVTG_* variable of the environment into the table (returns how many); the Python binding
 *                                   calls it at load time only when UVTG_DEV_ENV=1 is set (tools/ab5.sh and friends set it);
 *   uvtg_dev_config_set(name, v)    sets (v != NULL) or clears one switch.
 * A switch is looked up when its launch path first runs and most sites cache the answer: configure before the first launch.  The
 * uvtg_debug_* setters below are the test suite's run-time toggles for the switches that parity tests flip inside one process. */
int uvtg_dev_config_from_env(void);
int uvtg_dev_config_set(const char* name, const char* value);

/* ---- experiment / parity-test knobs of the persistent NT GEMM ----------------------------------------------------------------- */
/* uvtg_debug_nt_splitk(n): at most n K parts per tile from now on (0 / 1 = never split, default 4); uvtg_debug_nt_splitk_parts: host
 * arithmetic only, the parts an M x N x K launch (K in staged elements: 2 x real columns for split operands) would get on `cus` compute
 * units (0 = not split). */
/* Launches of at most one 128 x 256 tile per compute unit run a single-tile variant of the persistent kernel (three-stage staging ring: the K
 * loop of such a launch is a chain of memory round trips, not of MFMAs; results bit-identical).  uvtg_debug_nt_small(0) sends them through the
 * persistent two-stage kernel again (parity tests / A-B measurements), 1 restores the default. */
int uvtg_debug_nt_small(int on);
/* Persistent NT kernel, plain row mapping: bit (rows / 64 - 2) of `mask` = the staging pieces of tile height 128 / 192 / 256 are issued by one
 * wave per SIMD for both waves of that SIMD ("loader waves"; default 7; results bit-identical either way -- parity tests / A-B measurements). */
int uvtg_debug_nt_loader_waves(int mask);
int uvtg_debug_nt_splitk(int max_parts);
int uvtg_debug_nt_splitk_parts(int M, int N, int K, int groups, int cus);
int uvtg_debug_nt_small_tile(int M, int N, int K, int groups, int cus);      /* its tile: 128 = 128 x 128, 256 = 128 x 256, 0 = not a single-tile launch */

/* ---- measurement hooks (bench.py): HIP events around every launch of the GEMM kernels, recorded on the launch
 * stream.  index 0: gemm_nt bf16 (128-tile), 1: gemm_nt split-operand (fp16 hi / lo images; flops = algorithmic 2 M N K), 2: gemm_tn (wgrad), 3: gemm_nt256 bf16 (256-tile,
 * persistent), 4: attention forward, 5: attention backward (all its kernels; FLOPs counted on the padded S: 4 S^2 hd per
 * (sample, head) forward, 10 S^2 hd backward), 6: LayerNorm forward launches, 7: LayerNorm backward launches (for 6 / 7 the
 * "flops" entry carries the algorithmic BYTES of the row streams: input + every output + the position rows added into the +pos outputs).
 * host arrays [8]. */
int uvtg_profile_start(void);
int uvtg_profile_stop(double* total_ms, double* total_flops, long long* launches);
/* algorithmic bytes of the launches between uvtg_profile_start and _stop, per family (host array [8]; filled for family 3, the persistent
 * NT GEMM: A and W once, every output once, residual / pre-activation / position operands once) */
int uvtg_profile_bytes(double* bytes);
/* the empty-event-pair floor (ms) that uvtg_profile_stop measured on the launch stream and subtracted from every launch */
double uvtg_profile_event_floor_ms(void);

/* Section timing (its own pass, so that the per-launch events above do not sit inside the sections): one event pair on the launch
 * stream around 0: the E encoder layers of uvtg_forward, 1: their backward in uvtg_backward (LayerNorm / dgrad / attention /
 * weight-gradient kernels of the E layers), 2: the whole uvtg_forward, 3: the whole uvtg_backward.  host arrays [4]. */
int uvtg_profile_sections_start(void);
int uvtg_profile_sections_stop(double* total_ms, long long* counts);

/* Test knob: force the NT GEMM tile size (0 = automatic choice, 128, 256) so that both kernels can be compared on
 * identical inputs.  Process-wide. */
int uvtg_debug_force_nt_tile(int tile);
/* ... and the tile HEIGHT of the persistent 256-wide kernel (0 = automatic per launch, 128, 192, 256, 320; 320 applies to the launches
 * with the plain row mapping, the gather launches run 256 rows then) */
int uvtg_debug_force_nt_bm(int bm);
/* Host arithmetic only (no device needed): the tile height the persistent NT GEMM picks for an M x N launch of `groups` groups on `cus`
 * compute units, gather != 0 for launches with row gather / scatter / conv taps / row tables.  Returns 128, 192, 256 or 320. */
int uvtg_debug_nt_tile_rows(int M, int N, int groups, int gather, int cus);
/* Host arithmetic only: the launch plan of the persistent NT GEMM for an M x N x K launch -- out3 = {tile rows of the head (or only)
 * launch, rows the head covers (0 = a single launch), tile rows of the tail launch}.  A launch whose last CU round would be sparsely
 * filled is cut into whole rounds of tall tiles + a tail of short ones. */
int uvtg_debug_nt_plan(int M, int N, int K, int groups, int gather, int cus, int* out3);
/* Experiment knob: the persistent GEMM launches that follow size their grids for at most n CUs (0 = the whole chip), so that two
 * launches on different streams can run side by side. */
int uvtg_debug_gemm_cus(int n);
/* LayerNorm (round 5): kernel-level entry of the encoder's bf16 launches -- x bf16 [rows, D] -> y (bf16, optional), y + pos (bf16, optional: row s < Lv
 * of each S-row sample adds the fp32 row pos[b * Lv + s]; S = 0 / pos = NULL: none), mean / rstd [rows] (optional).  uvtg_debug_ln_fwd_lean(0) sends
 * these launches through the generic row kernel again (1 = default: the lean kernel with the next row in flight). */
int uvtg_debug_layernorm_fwd_bf16(const void* xB, const float* gamma, const float* beta, void* yB, void* yU, const float* pos, int S, int Lv,
                                  float* mean, float* rstd, int rows, int D, uvtg_stream_t stream);
int uvtg_debug_ln_fwd_lean(int on);
/* Attention backward's delta = rowsum_head(dO * O) is produced by the out-projection dgrad GEMM's epilogue where that launch can carry it (round 5);
 * uvtg_debug_delta_fuse(0) restores attn_delta_kernel's own pass (parity tests / A-B), 1 = default. */
int uvtg_debug_delta_fuse(int on);
/* Long-sequence attention backward at head_dim 128 (round 5): dK / dV by the role-split kernel (score waves + product waves, two per SIMD; default)
 * or, 0, by the one-wave-per-SIMD kernel it replaces (parity tests / A-B). */
int uvtg_debug_attn_ws(int on);
/* ... and the head_dim-128 bf16 FORWARD: K / V tiles by LDS-DMA into a double buffer (default) or, 0, the register-staged kernel (bit-identical). */
int uvtg_debug_attn_fwd_dma(int on);
/* The last encoder layer's FFN half on the B Lv clip rows only (default, bf16 mode, unpacked stream, d = 512 / 1024: the text rows of the encoder
 * output are read by nobody, model/univtg.py:127) or, 0, on every row.  uvtg_forward and the uvtg_backward that follows it must see the same setting. */
int uvtg_debug_last_layer_clip(int on);
/* The four conv-head weight gradients inside the encoder's deferred weight-gradient launch (default: conv taps and the (d, d, 3) weight layout in the
 * hybrid kernel's epilogue, no slab + reduce pass) or, 0, as their own slab + reduce launch. */
int uvtg_debug_tn_conv_defer(int on);
/* Host arithmetic only: uvtg_debug_nt_plan with the launch's epilogue class (eop != 0: the launch reads a bf16 residual / pre-activation operand in
 * its epilogue; uvtg_debug_nt_plan assumes it does) and whether the caller hands the launch a split-K workspace (have_ws). */
int uvtg_debug_nt_plan2(int M, int N, int K, int groups, int gather, int cus, int eop, int have_ws, int* out3);
/* Experiment knob (round 5): force the launch plan of the plain-row persistent NT launches of ONE shape -- head of tm1_rows-row tiles over rows
 * [0, rows1) + tail of tm2_rows-row tiles (rows1 = 0: a single launch at tm1_rows); M <= 0 clears every override.  In-box A/B of the cost model. */
int uvtg_debug_nt_plan_override(int M, int N, int tm1_rows, int rows1, int tm2_rows);
/* Experiment knob (round 5): tile order of the wide plain-row launches (>= 8 column tiles): column groups of `tiles_per_group` tiles,
 * row-block-major inside a group (0 = row-block-major over the whole width).  Same tiles, same K order: bit-identical results. */
int uvtg_debug_nt_cgw(int tiles_per_group);

#ifdef __cplusplus
}
#endif
#endif /* UVTG_DEV_H */
