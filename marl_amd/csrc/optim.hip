// clip_grad_norm_ + RMSprop / Adam over ONE flat parameter buffer (reference
// algorithm/q_learner.py:42-47,170-173; torch defaults).  The gradient arrives un-normalised
// (loss numerator); 1/sum(mask) and the clip coefficient are folded into the update so the whole
// optimizer is two launches and, on several GPUs, follows a single all-reduce.
#include "common.h"
#include <string.h>
#include "../../include/marl_hip.h"

namespace {
constexpr int TPB = 256;
constexpr int MAXB = 1024;

__global__ void sumsq_kernel(const float* g, long n, float* ws) {
  __shared__ float sh[TPB / 64];
  float s = 0.f;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) ws[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ void sumsq_finish_kernel(const float* ws, int nb, float* out) {
  __shared__ float sh[TPB];
  float s = 0.f;
  for (int b = threadIdx.x; b < nb; b += TPB) s += ws[b];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = TPB / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0];
}

__device__ __forceinline__ float grad_scale(const float* sumsq, const float* den, float clip) {
  const float inv = den ? 1.f / den[0] : 1.f;
  const float norm = sqrtf(sumsq[0]) * inv;                 // ||g / sum(mask)||_2
  float coef = clip / (norm + 1e-6f);                       // torch clip_grad_norm_
  coef = coef < 1.f ? coef : 1.f;
  return inv * coef;
}

__global__ void rmsprop_kernel(float* p, const float* g, float* sq, long n, float lr, float alpha, float eps,
                               float clip, const float* sumsq, const float* den) {
  const float sc = grad_scale(sumsq, den, clip);
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) {
    const float gi = g[i] * sc;
    const float s = alpha * sq[i] + (1.f - alpha) * gi * gi;
    sq[i] = s;
    p[i] -= lr * gi / (sqrtf(s) + eps);
  }
}

__global__ void adam_kernel(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2,
                            float eps, float bc1, float bc2s, float clip, const float* sumsq, const float* den) {
  const float sc = grad_scale(sumsq, den, clip);
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) {
    const float gi = g[i] * sc;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2s + eps);
  }
}

inline int blocks_for(long n) {
  long b = (n + TPB - 1) / TPB;
  if (b > MAXB) b = MAXB;
  if (b < 1) b = 1;
  return (int)b;
}
}  // namespace

extern "C" size_t marl_sumsq_workspace(long n) { return MAXB * sizeof(float); }

extern "C" int marl_grad_sumsq(const float* g, long n, float* sumsq, float* ws, void* stream) {
  const int nb = blocks_for(n);
  hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(TPB), 0, (hipStream_t)stream, g, n, ws);
  MARL_CHECK_LAUNCH();
  hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(TPB), 0, (hipStream_t)stream, (const float*)ws, nb, sumsq);
  MARL_CHECK_LAUNCH();
  return 0;
}

extern "C" int marl_rmsprop_step(float* p, const float* g, float* sq, long n, float lr, float alpha, float eps,
                                 float clip, const float* sumsq, const float* den, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(rmsprop_kernel, dim3(blocks_for(n)), dim3(TPB), 0, (hipStream_t)stream, p, g, sq, n, lr, alpha,
                     eps, clip, sumsq, den);
  MARL_CHECK_LAUNCH();
  return 0;
}

extern "C" int marl_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1,
                              float beta2, float eps, float bc1, float bc2_sqrt, float clip, const float* sumsq,
                              const float* den, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(n)), dim3(TPB), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                     beta2, eps, bc1, bc2_sqrt, clip, sumsq, den);
  MARL_CHECK_LAUNCH();
  return 0;
}

// ---- experiment switches (common.h: MarlSwitches) ----------------------------------------------------------------------------
static MarlSwitches g_switches = {1, 0, 1, 4, 1, 1, 0, 0, 1};
extern "C" const MarlSwitches* marl_switches(void) { return &g_switches; }
static int* switch_slot(const char* name) {
  if (!name) return nullptr;
  struct { const char* n; int* p; } tab[] = {{"fwd_xs", &g_switches.fwd_xs}, {"fwd_dma", &g_switches.fwd_dma}, {"fwd_w2l", &g_switches.fwd_w2l},
                                             {"bwd_pipe_max_rt", &g_switches.bwd_pipe_max_rt}, {"wgrad_tall", &g_switches.wgrad_tall},
                                             {"wide_res", &g_switches.wide_res}, {"wide_res32", &g_switches.wide_res32}, {"rollout_v1", &g_switches.rollout_v1}, {"unroll_r6", &g_switches.unroll_r6}};
  for (auto& t : tab)
    if (!strcmp(t.n, name)) return t.p;
  return nullptr;
}
// set / read one switch by name; unknown names return -1 (set) / INT_MIN (get).  Not thread-safe against concurrent launches:
// meant for process start-up and single-threaded tests.
extern "C" int marl_experiment_set(const char* name, int value) {
  int* p = switch_slot(name);
  if (!p) return -1;
  *p = value;
  return 0;
}
extern "C" int marl_experiment_get(const char* name) {
  int* p = switch_slot(name);
  return p ? *p : (-2147483647 - 1);
}

#ifndef MARL_SRC_HASH
#define MARL_SRC_HASH "unknown"
#endif
// library version + a hash of the kernel sources it was built from (Makefile): the PMC evidence under profiles/ records it
extern "C" const char* marl_hip_version(void) { return "marl_hip 0.4 (gfx950) src " MARL_SRC_HASH; }
