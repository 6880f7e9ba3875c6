// Rollout-side kernels: batched epsilon-greedy action choice (reference
// controller/share_params.py:66-70, vectorised over envs x agents) and the synthetic SMAC-shaped
// environment that stands in for StarCraft II (main.py:16-20 is not vendored).  Everything random
// is a pure function of (seed, stream, env, step, index) through a 32-bit hash, so the numpy
// restatement in oracle/rollout.py reproduces it bit for bit.
#include "common.h"
#include "synth_hash.h"
#include "../../include/marl_hip.h"

namespace {
constexpr int TPB = 256;

__global__ void select_kernel(const float* q, const float* avail, long avail_es, const int* alive, float eps,
                              unsigned rseed, int env0, const int* tg, int tg0, int* act_out, long act_es, int E,
                              int N, int A) {
  const long total = (long)E * N;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const int e = (int)(i / N), n = (int)(i - (long)e * N);
    int* out = act_out + e * act_es + n;
    if (alive && !alive[e]) { *out = -1; continue; }
    const float* qa = q + i * A;
    const float* av = avail + e * avail_es + (long)n * A;
    float best = 0.f; int arg = -1, navail = 0;
    for (int a = 0; a < A; ++a) {
      if (av[a] == 0.f) continue;
      ++navail;
      if (arg < 0 || qa[a] > best) { best = qa[a]; arg = a; }
    }
    if (arg < 0) arg = 0;     // no action available: cannot happen for a live agent; mirror argmax of all -inf
    const unsigned tgl = (unsigned)(tg ? tg[e] : tg0);
    const bool explore = u01(hkey(rseed, ST_EXPLORE, (unsigned)(env0 + e), tgl, (unsigned)n)) < eps;
    if (explore && navail > 0) {
      int k = (int)floorf(u01(hkey(rseed, ST_PICK, (unsigned)(env0 + e), tgl, (unsigned)n)) * (float)navail);
      if (k > navail - 1) k = navail - 1;
      int c = 0;
      for (int a = 0; a < A; ++a) {
        if (av[a] == 0.f) continue;
        if (c == k) { arg = a; break; }
        ++c;
      }
    }
    *out = arg;
  }
}

__global__ void synth_lengths_kernel(unsigned seed, int env0, int episode, int* len, int* won, int E, int T) {
  const int e = blockIdx.x * TPB + threadIdx.x;
  if (e >= E) return;
  const int lmin = T / 2 > 1 ? T / 2 : 1;
  len[e] = lmin + (int)(hkey(seed, ST_LEN, (unsigned)(env0 + e), (unsigned)episode, 0u) % (unsigned)(T - lmin + 1));
  if (won) won[e] = (int)(hkey(seed, ST_WON, (unsigned)(env0 + e), (unsigned)episode, 0u) & 1u);
}

// one block row per env slot; slot t of the (T+1)-slot storage.  Slots past the episode end are zero
// (rollout.py:122-133); slot len[e] is the final observation (o_next of the last step).
__global__ void synth_observe_kernel(unsigned seed, int env0, int episode, int t, const int* len, float* obs,
                                     float* state, long SL, float* avail, int E, int T, int N, int O, int S, int A) {
  const int e = blockIdx.x;
  const bool live = t <= len[e];
  const unsigned env = (unsigned)(env0 + e), tg = (unsigned)(episode * (T + 1) + t);
  float* o = obs + ((long)e * (T + 1) + t) * N * O;
  for (int i = threadIdx.x; i < N * O; i += TPB) o[i] = live ? 2.0f * u01(hkey(seed, ST_OBS, env, tg, (unsigned)i)) - 1.0f : 0.f;
  float* s = state + ((long)e * (T + 1) + t) * SL;       // SL = row stride of the state storage (>= S)
  for (int i = threadIdx.x; i < S; i += TPB) s[i] = live ? 2.0f * u01(hkey(seed, ST_STATE, env, tg, (unsigned)i)) - 1.0f : 0.f;
  float* a = avail + ((long)e * (T + 1) + t) * N * A;
  for (int i = threadIdx.x; i < N * A; i += TPB) {
    float v = 0.f;
    if (live) v = (i % A == 0 || u01(hkey(seed, ST_AVAIL, env, tg, (unsigned)i)) < 0.7f) ? 1.f : 0.f;
    a[i] = v;
  }
}

__global__ void synth_step_kernel(unsigned seed, int env0, int episode, int t, const int* len, const int* act, int* u,
                                  float* r, float* term, float* padded, int* alive_next, int E, int T, int N, int A) {
  const int e = blockIdx.x * TPB + threadIdx.x;
  if (e >= E) return;
  const int L = len[e];
  const bool live = t < L;
  const unsigned env = (unsigned)(env0 + e), tg = (unsigned)(episode * (T + 1) + t);
  float acc = 0.f;
  for (int n = 0; n < N; ++n) {
    const int a = live ? act[(long)e * N + n] : -1;
    u[((long)e * T + t) * N + n] = a;
    if (live) acc = acc + (u01(hkey(seed, ST_REWARD, env, tg, (unsigned)(n * A + a))) - 0.5f);
  }
  r[(long)e * T + t] = live ? acc * (1.0f / (float)N) : 0.f;
  term[(long)e * T + t] = live ? (t + 1 >= L ? 1.f : 0.f) : 1.f;
  padded[(long)e * T + t] = live ? 0.f : 1.f;
  if (alive_next) alive_next[e] = (t + 1 < L) ? 1 : 0;
}
// One launch per lock-step for the synthetic env: epsilon-greedy choice (same rule as select_kernel),
// env step (reward / terminated / padded / u) and the observation of slot t+1.  One block per env.
__global__ void synth_fused_step_kernel(unsigned seed, unsigned rseed, int env0, int episode, int t, float eps,
                                        const int* len, const float* q, float* obs, float* state, long SL,
                                        float* avail, int* u, float* r, float* term, float* padded, int E, int T, int N,
                                        int O, int S, int A) {
  __shared__ int act[64];
  const int e = blockIdx.x;
  const int L = len[e];
  const bool live = t < L;
  const unsigned env = (unsigned)(env0 + e), tg = (unsigned)(episode * (T + 1) + t);
  if (threadIdx.x < N) {
    const int n = threadIdx.x;
    int arg = -1;
    if (live) {
      const float* qa = q + ((long)e * N + n) * A;
      const float* av = avail + (((long)e * (T + 1) + t) * N + n) * A;
      float best = 0.f; int navail = 0;
      for (int a = 0; a < A; ++a) {
        if (av[a] == 0.f) continue;
        ++navail;
        if (arg < 0 || qa[a] > best) { best = qa[a]; arg = a; }
      }
      if (arg < 0) arg = 0;
      const bool explore = u01(hkey(rseed, ST_EXPLORE, env, tg, (unsigned)n)) < eps;
      if (explore && navail > 0) {
        int k = (int)floorf(u01(hkey(rseed, ST_PICK, env, tg, (unsigned)n)) * (float)navail);
        if (k > navail - 1) k = navail - 1;
        int c = 0;
        for (int a = 0; a < A; ++a) {
          if (av[a] == 0.f) continue;
          if (c == k) { arg = a; break; }
          ++c;
        }
      }
    }
    act[n] = arg;
    u[((long)e * T + t) * N + n] = arg;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float acc = 0.f;
    if (live)
      for (int n = 0; n < N; ++n) acc = acc + (u01(hkey(seed, ST_REWARD, env, tg, (unsigned)(n * A + act[n]))) - 0.5f);
    r[(long)e * T + t] = live ? acc * (1.0f / (float)N) : 0.f;
    term[(long)e * T + t] = live ? (t + 1 >= L ? 1.f : 0.f) : 1.f;
    padded[(long)e * T + t] = live ? 0.f : 1.f;
  }
  // observation of slot t+1 (zeros once the episode is over; slot L is the final observation)
  const int t1 = t + 1;
  const bool live1 = t1 <= L;
  const unsigned tg1 = tg + 1u;
  float* o = obs + ((long)e * (T + 1) + t1) * N * O;
  for (int i = threadIdx.x; i < N * O; i += TPB) o[i] = live1 ? 2.0f * u01(hkey(seed, ST_OBS, env, tg1, (unsigned)i)) - 1.0f : 0.f;
  float* sp = state + ((long)e * (T + 1) + t1) * SL;
  for (int i = threadIdx.x; i < S; i += TPB) sp[i] = live1 ? 2.0f * u01(hkey(seed, ST_STATE, env, tg1, (unsigned)i)) - 1.0f : 0.f;
  float* ap = avail + ((long)e * (T + 1) + t1) * N * A;
  for (int i = threadIdx.x; i < N * A; i += TPB) {
    float v = 0.f;
    if (live1) v = (i % A == 0 || u01(hkey(seed, ST_AVAIL, env, tg1, (unsigned)i)) < 0.7f) ? 1.f : 0.f;
    ap[i] = v;
  }
}
}  // namespace

extern "C" int marl_synth_fused_step(unsigned seed, unsigned rseed, int env0, int episode, int t, float eps,
                                     const int* len, const float* q, float* obs, float* state, long state_ld,
                                     float* avail, int* u, float* r, float* term, float* padded, int E, int T, int N,
                                     int O, int S, int A, void* stream) {
  if (E <= 0) return 0;
  if (N > 64 || state_ld < S) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(synth_fused_step_kernel, dim3(E), dim3(TPB), 0, (hipStream_t)stream, seed, rseed, env0, episode,
                     t, eps, len, q, obs, state, state_ld, avail, u, r, term, padded, E, T, N, O, S, A);
  MARL_CHECK_LAUNCH();
  return 0;
}

extern "C" int marl_select_actions(const float* q, const float* avail, long avail_es, const int* alive, float eps,
                                   unsigned rseed, int env0, const int* tg, int tg0, int* act_out, long act_es,
                                   int E, int N, int A, void* stream) {
  if (E <= 0) return 0;
  const long total = (long)E * N;
  hipLaunchKernelGGL(select_kernel, dim3((unsigned)((total + TPB - 1) / TPB)), dim3(TPB), 0, (hipStream_t)stream, q,
                     avail, avail_es, alive, eps, rseed, env0, tg, tg0, act_out, act_es, E, N, A);
  MARL_CHECK_LAUNCH();
  return 0;
}

extern "C" int marl_synth_lengths(unsigned seed, int env0, int episode, int* len, int* won, int E, int T,
                                  void* stream) {
  if (E <= 0) return 0;
  hipLaunchKernelGGL(synth_lengths_kernel, dim3((E + TPB - 1) / TPB), dim3(TPB), 0, (hipStream_t)stream, seed, env0,
                     episode, len, won, E, T);
  MARL_CHECK_LAUNCH();
  return 0;
}

extern "C" int marl_synth_observe(unsigned seed, int env0, int episode, int t, const int* len, float* obs,
                                  float* state, long state_ld, float* avail, int E, int T, int N, int O, int S, int A,
                                  void* stream) {
  if (E <= 0) return 0;
  if (state_ld < S) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(synth_observe_kernel, dim3(E), dim3(TPB), 0, (hipStream_t)stream, seed, env0, episode, t, len,
                     obs, state, state_ld, avail, E, T, N, O, S, A);
  MARL_CHECK_LAUNCH();
  return 0;
}

extern "C" int marl_synth_step(unsigned seed, int env0, int episode, int t, const int* len, const int* act, int* u,
                               float* r, float* term, float* padded, int* alive_next, int E, int T, int N, int A,
                               void* stream) {
  if (E <= 0) return 0;
  hipLaunchKernelGGL(synth_step_kernel, dim3((E + TPB - 1) / TPB), dim3(TPB), 0, (hipStream_t)stream, seed, env0,
                     episode, t, len, act, u, r, term, padded, alive_next, E, T, N, A);
  MARL_CHECK_LAUNCH();
  return 0;
}
