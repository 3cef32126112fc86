// Training-step shell on device: global-norm gradient clipping + AdamW over ONE flat fp32 parameter buffer
// (replaces nn.utils.clip_grad_norm_ + torch.optim.AdamW of main/train_vlp_ddp.py:66-68, main/config.py:349-350:
// ~100 small launches -> 2 kernels), plus the per-kernel timing hooks bench.py uses for its roofline line.
#include "uvtg_kernels.h"
#include "../../include/uvtg.h"
#include <vector>

namespace {

// Squared-norm partials, one per block, summed later in a FIXED order: every rank of a data-parallel run must derive the bit-identical
// clipping coefficient from the bit-identical reduced gradients, or the replicas drift apart (an atomicAdd total depends on arrival order;
// found by tests/test_gpu_parity_full.py::test_two_rank_native_trainstep_is_a_data_parallel_step)
constexpr int SQN_BLOCKS = 1024;
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* g, long long n, float* out) {
  __shared__ float red[4];
  float acc = 0.f;
  const long long n4 = n / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 v = ((const f32x4*)g)[i];
    acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0) for (long long i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) acc += g[i] * g[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, long long n, float lr, float b1,
                                                    float b2, float eps, float wd, float bc1, float bc2s, float max_norm,
                                                    float grad_scale, const float* sqn, int n_partials) {
  // squared norm: one value, or n_partials (a multiple of 256) per-block partials folded here in a fixed order -- every block, and every
  // rank of a data-parallel run, computes the identical sum
  float sq = *sqn;
  if (n_partials > 0) {
    __shared__ float red[4];
    float t = 0.f;
    for (int i = threadIdx.x; i < n_partials; i += 256) t += sqn[i];
    t = wave_sum(t);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
    __syncthreads();
    sq = (red[0] + red[1]) + (red[2] + red[3]);
  }
  // torch.nn.utils.clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1
  const float total = sqrtf(sq) * grad_scale;
  const float coef = grad_scale * (max_norm > 0.f ? fminf(1.f, max_norm / (total + 1e-6f)) : 1.f);
  const long long n4 = n / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 pv = ((f32x4*)p)[i], mv = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
    const f32x4 gv = ((const f32x4*)g)[i];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float gr = gv[e] * coef;
      pv[e] *= 1.f - lr * wd;                                   // decoupled weight decay
      mv[e] = b1 * mv[e] + (1.f - b1) * gr;
      vv[e] = b2 * vv[e] + (1.f - b2) * gr * gr;
      pv[e] -= (lr / bc1) * mv[e] / (sqrtf(vv[e]) / bc2s + eps);
    }
    ((f32x4*)p)[i] = pv; ((f32x4*)m)[i] = mv; ((f32x4*)v)[i] = vv;
  }
}

struct Prof {
  bool on = false;
  hipStream_t last = nullptr;
  double floor_ms = 0;
  std::vector<hipEvent_t> ev[8];
  size_t used[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double flops[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double bytes[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // algorithmic operand + output bytes of the family's launches (GEMM families)
  // section timing (its own pass: the per-launch events above would sit inside the sections)
  bool sec_on = false;
  std::vector<hipEvent_t> sev[4];
  size_t sused[4] = {0, 0, 0, 0};
} g_prof;

}  // namespace

extern "C" int uvtg_adamw_clip_step(float* params, const float* grads, float* m, float* v, long long n, float lr, float beta1,
                                    float beta2, float eps, float wd, int step, float max_norm, float grad_scale, float* scratch,
                                    uvtg_stream_t stream) {
  if (!params || !grads || !m || !v || !scratch) return -20;
  if (n <= 0 || n % 4 || step <= 0) return -11;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(SQN_BLOCKS), dim3(256), 0, s, grads, n, scratch);
  const float bc1 = 1.f - powf(beta1, (float)step), bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adamw_kernel, dim3(2048), dim3(256), 0, s, params, grads, m, v, n, lr, beta1, beta2, eps, wd, bc1, bc2s,
                     max_norm, grad_scale, scratch, SQN_BLOCKS);
  UVTG_CHECK_LAUNCH();
  return 0;
}

extern "C" int uvtg_adamw_clip_step_prenorm(float* params, const float* grads, float* m, float* v, long long n, float lr, float beta1,
                                            float beta2, float eps, float wd, int step, float max_norm, float grad_scale,
                                            const float* sqnorm_dev, uvtg_stream_t stream) {
  if (!params || !grads || !m || !v || !sqnorm_dev) return -20;
  if (n <= 0 || n % 4 || step <= 0) return -11;
  hipStream_t s = (hipStream_t)stream;
  const float bc1 = 1.f - powf(beta1, (float)step), bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adamw_kernel, dim3(2048), dim3(256), 0, s, params, grads, m, v, n, lr, beta1, beta2, eps, wd, bc1, bc2s,
                     max_norm, grad_scale, sqnorm_dev, 0);
  UVTG_CHECK_LAUNCH();
  return 0;
}

// ---- timing hooks: HIP events around every launch of one kernel family, on the launch stream ----
void uvtg_prof_begin_launch(int family, double flops, hipStream_t s) {
  if (!g_prof.on) return;
  auto& ev = g_prof.ev[family];
  size_t& u = g_prof.used[family];
  while (ev.size() < u + 2) { hipEvent_t e; hipEventCreate(&e); ev.push_back(e); }
  hipEventRecord(ev[u], s);
  g_prof.flops[family] += flops;
  g_prof.last = s;
}
void uvtg_prof_end_launch(int family, hipStream_t s) {
  if (!g_prof.on) return;
  size_t& u = g_prof.used[family];
  hipEventRecord(g_prof.ev[family][u + 1], s);
  u += 2;
}
extern "C" int uvtg_profile_start(void) {
  for (int f = 0; f < 8; f++) { g_prof.used[f] = 0; g_prof.flops[f] = 0; g_prof.bytes[f] = 0; }
  g_prof.on = true;
  return 0;
}
extern "C" int uvtg_profile_stop(double* ms, double* flops, long long* launches) {
  g_prof.on = false;
  if (!ms || !flops || !launches) return -20;
  // An event pair brackets more than the kernel (the two event packets themselves): measure that floor with empty pairs on the
  // same stream and take it off every launch, so that the per-launch durations agree with a rocprofv3 kernel trace.
  constexpr int NCAL = 32;
  hipEvent_t ca[NCAL], cb[NCAL];
  for (int i = 0; i < NCAL; i++) { hipEventCreate(&ca[i]); hipEventCreate(&cb[i]); }
  for (int i = 0; i < NCAL; i++) { hipEventRecord(ca[i], g_prof.last); hipEventRecord(cb[i], g_prof.last); }
  if (hipError_t e = hipDeviceSynchronize()) return (int)e;
  float floor_ms = 1e9f;
  for (int i = 0; i < NCAL; i++) {
    float t = 0;
    hipEventElapsedTime(&t, ca[i], cb[i]);
    if (t < floor_ms) floor_ms = t;
    hipEventDestroy(ca[i]); hipEventDestroy(cb[i]);
  }
  for (int f = 0; f < 8; f++) {
    double tot = 0;
    for (size_t i = 0; i + 1 < g_prof.used[f]; i += 2) {
      float t = 0;
      hipEventElapsedTime(&t, g_prof.ev[f][i], g_prof.ev[f][i + 1]);
      tot += t > floor_ms ? t - floor_ms : 0.0;
    }
    ms[f] = tot; flops[f] = g_prof.flops[f]; launches[f] = (long long)(g_prof.used[f] / 2);
  }
  g_prof.floor_ms = floor_ms;
  return 0;
}
extern "C" double uvtg_profile_event_floor_ms(void) { return g_prof.floor_ms; }
void uvtg_prof_add_bytes(int family, double bytes) { if (g_prof.on) g_prof.bytes[family] += bytes; }
extern "C" int uvtg_profile_bytes(double* bytes) {
  if (!bytes) return -20;
  for (int f = 0; f < 8; f++) bytes[f] = g_prof.bytes[f];
  return 0;
}

// ---- section timing: one event pair around a whole section of uvtg_forward / uvtg_backward, on the launch stream ----
// sections: 0 encoder forward (the E layers), 1 encoder backward (LayerNorm / dgrad / attention / weight gradients of the E layers),
// 2 whole uvtg_forward, 3 whole uvtg_backward
void uvtg_prof_section(int section, int end, hipStream_t s) {
  if (!g_prof.sec_on) return;
  auto& ev = g_prof.sev[section];
  size_t& u = g_prof.sused[section];
  if (!end) {
    while (ev.size() < u + 2) { hipEvent_t e; hipEventCreate(&e); ev.push_back(e); }
    hipEventRecord(ev[u], s);
  } else {
    hipEventRecord(ev[u + 1], s);
    u += 2;
  }
}
extern "C" int uvtg_profile_sections_start(void) {
  for (int f = 0; f < 4; f++) g_prof.sused[f] = 0;
  g_prof.sec_on = true;
  return 0;
}
extern "C" int uvtg_profile_sections_stop(double* ms, long long* counts) {
  g_prof.sec_on = false;
  if (!ms || !counts) return -20;
  if (hipError_t e = hipDeviceSynchronize()) return (int)e;
  for (int f = 0; f < 4; f++) {
    double tot = 0;
    for (size_t i = 0; i + 1 < g_prof.sused[f]; i += 2) {
      float t = 0;
      hipEventElapsedTime(&t, g_prof.sev[f][i], g_prof.sev[f][i + 1]);
      tot += t;
    }
    ms[f] = tot; counts[f] = (long long)(g_prof.sused[f] / 2);
  }
  return 0;
}
