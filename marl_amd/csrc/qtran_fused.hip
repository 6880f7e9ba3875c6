// Fused QTRAN-base heads for gfx950: the joint action-value network QtranQBase (reference network/mixer.py:355-388)
// and the state-value network QtranV (:392-418), forward and backward.
//
//   enc_i  = W_e2 relu(W_e1 [h_i | onehot(u_i)] + b_e1) + b_e2          per agent            (:378-383, :411-413)
//   esum   = sum_i enc_i                                                 per (episode, step)  (:384, :414)
//   out    = W_q3 relu(W_q2 relu(W_q1 [s | esum] + b_q1) + b_q2) + b_q3                       (:386-387, :416-417)
//
// Composed from generic GEMMs the 78-wide encoder activations of all B*T*N agent rows cross HBM between every layer
// (6 x 192 us + 4 x 187 us of a 7.3 ms update at 3s5z / 512 envs).  Here:
//   * the second encoder layer is LINEAR, so the agent sum is taken BEFORE it:  esum = W_e2 (sum_i e1_i) + N b_e2  -
//     1/N of the multiply-adds, and the sum over agents becomes an in-register accumulation: a wave owns a tile of 16
//     (episode, step) rows and walks the agents, so rows m of every MFMA tile are 16 different (episode, step) pairs
//     of the SAME agent and s1 += relu(e1) needs no cross-lane traffic, for any N;
//   * the one-hot columns of W_e1 are a table lookup (bias + column u of W_e1, one 16-byte LDS read per tile);
//   * W_q1 [s | esum] = W_q1s s + W_q1e esum: the state part "sp" is shared by the joint-Q evaluations of one network
//     (taken actions / greedy actions, qtran_learner.py:116,133) and comes from one marl_linear call; the rest of the
//     head is chained in registers in the transposed formulation of mlp3_fused.hip (the accumulator tile of one
//     layer is the B fragment of the next);
//   * backward: B1 walks the head chain back per (episode, step) row (dy2, dy1, d esum, d s1 -> HBM: small, B*T
//     rows), B2 walks the agents again: recomputes e1 for the relu mask, forms dh = W_e1h^T de1 and accumulates
//     dW_e1 += de1^T [h | onehot] in registers (the two operands change from the transposed to the row layout
//     through a wave-private LDS tile; no workgroup barrier in the loop).  The weight gradients of the row-level
//     layers (W_q1..3, W_e2) are reductions over B*T rows of tensors B1 writes anyway and use marl_linear_wgrad.
// All arithmetic fp32 on v_mfma_f32_16x16x4_f32.
#include "common.h"
#include "../../include/marl_hip.h"

namespace {

constexpr int HD = 64;            // rnn_hidden_dim = qtran_hidden_dim
constexpr int NW = 8;             // waves per workgroup (two per SIMD)
constexpr int XS = HD + 4;        // row stride of the wave-private h tile
constexpr int KW = 84;            // slab row of dW_e1: 64 h columns | 16 one-hot columns | bias | pad

struct QtArgs {
  // weights
  const float *We1, *be1, *We2, *be2;       // (AE,AE), (AE)
  const float* Wq1; long ldq1; int S;       // (64, S + AE): the encoder part starts at column S
  const float *Wq2, *bq2, *wq3, *bq3;
  // per-row inputs
  const float* hidden;                      // (BT*N, 64)
  const int* u;                             // (BT*N) action index (< 0: none) or null
  const int* u2; float* out2;               // forward, DUAL: a second set of actions on the same rows -> out2 (BT)
  const float* sp;                          // (BT, 64) = W_q1[:, :S] s + b_q1
  const float* d_out;                       // (BT)            backward
  // outputs / saved activations
  float* out;                               // (BT)
  float *s1, *e2;                           // (BT, AEP)
  float *y1, *y2;                           // (BT, 64)
  float *dy1, *dy2, *de2, *ds1;             // backward, same shapes
  float* dhidden; int accumulate;           // (BT*N, 64)
  float *slab1, *slab2;                     // [grid][AEP] (colsum of de2), [grid][AEP][KW]
  long BT; int N, A, AE;
};

// fragment-major weight tiles: dst[(t * KCn + c) * 64 + lane] (f32x4) = W[16t + m][col0 + 16c + 4q + 0..3]
__device__ __forceinline__ void stage_frag(float* dst, const float* W, long ldw, int col0, int rows_valid, int K,
                                           int NTn, int KCn, int nthreads) {
  for (int e = threadIdx.x; e < NTn * KCn * 64; e += nthreads) {
    const int l = e & 63, tc = e >> 6, t = tc / KCn, c = tc - t * KCn;
    const int n = 16 * t + (l & 15), k0 = 16 * c + 4 * (l >> 4);
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (n < rows_valid && k0 + i < K) ? W[(long)n * ldw + col0 + k0 + i] : 0.f;
    *reinterpret_cast<f32x4*>(dst + (long)e * 4) = v;
  }
}
// transposed tiles: dst[(t * KCn + c) * 64 + lane] (f32x4) = W[16c + 4q + 0..3][col0 + 16t + m]  (A operand of W^T g)
__device__ __forceinline__ void stage_fragT(float* dst, const float* W, long ldw, int col0, int rows_valid, int cols_valid,
                                            int NTn, int KCn, int nthreads) {
  for (int e = threadIdx.x; e < NTn * KCn * 64; e += nthreads) {
    const int l = e & 63, tc = e >> 6, t = tc / KCn, c = tc - t * KCn;
    const int col = 16 * t + (l & 15), r0 = 16 * c + 4 * (l >> 4);
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (r0 + i < rows_valid && col < cols_valid) ? W[(long)(r0 + i) * ldw + col0 + col] : 0.f;
    *reinterpret_cast<f32x4*>(dst + (long)e * 4) = v;
  }
}
// bias + one-hot column table of encoder layer 1: tab[a][f] = b_e1[f] + W_e1[f][64 + a]  (a < A), row 16 = b_e1 only
template <int AEP, bool HOT>
__device__ __forceinline__ void stage_tab(float* tab, const QtArgs& a, int nthreads) {
  constexpr int TS = AEP + 4, ROWS = HOT ? 17 : 1;
  for (int e = threadIdx.x; e < ROWS * AEP; e += nthreads) {
    const int r = e / AEP, f = e - r * AEP;
    float v = 0.f;
    if (f < a.AE) {
      v = a.be1[f];
      if (HOT && r < a.A) v += a.We1[(long)f * a.AE + HD + r];
    }
    tab[r * TS + f] = v;
  }
}

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  return (f32x4){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
}
__device__ __forceinline__ const f32x4& frag(const float* base, int idx, int lane) {
  return *reinterpret_cast<const f32x4*>(base + ((long)idx * 64 + lane) * 4);
}

// acc[t] += W[t][c] * b[c] over the KC k-chunks of one layer (transposed formulation: weights are the A operand).  The
// fragments of chunk c+1 are read while chunk c multiplies; the scheduling barrier keeps the compiler from hoisting
// EVERY LDS read of the layer to the top (hundreds of registers -> spills, as in mlp3_fused.hip).
template <int NT, int KC, bool PF = true>
__device__ __forceinline__ void layer(f32x4 (&acc)[NT], const float* Wf, const f32x4 (&b)[KC], int lane) {
  if (!PF) {          // register-tight callers: fragments of one chunk at a time, the SIMD partner covers the LDS latency
#pragma unroll
    for (int c = 0; c < KC; ++c) {
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = mfma16x4(frag(Wf, t * KC + c, lane), b[c], acc[t]);
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
  f32x4 wn[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) wn[t] = frag(Wf, t * KC, lane);
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    f32x4 wc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wc[t] = wn[t];
    if (c + 1 < KC) {
#pragma unroll
      for (int t = 0; t < NT; ++t) wn[t] = frag(Wf, t * KC + c + 1, lane);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = mfma16x4(wc[t], b[c], acc[t]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// tile -> (workgroup, wave): tile = (k * NW + wave) * grid + block, so a remainder of tiles lands on the first waves of
// EVERY workgroup (one more tile on some SIMDs of all CUs) instead of on all waves of the first workgroups
__device__ __forceinline__ long first_tile(int wave) { return (long)wave * gridDim.x + blockIdx.x; }
__device__ __forceinline__ long tile_step() { return (long)NW * gridDim.x; }

// ------------------------------------------------------------------------------------------------- forward
// DUAL: the joint-Q head on the same states and hidden states for TWO action sets (the taken actions, whose activations are
// saved for the backward pass, and the greedy ones: qtran_learner.py:116 / :133) - the encoder's first-layer product W_e1h h,
// 72 % of the multiply-adds of one evaluation at 8 agents, is shared; the second set's pre-activation is the first's plus the
// difference of the two one-hot table rows.
template <int FT, bool HOT, bool DUAL = false>
__global__ __launch_bounds__(64 * NW, 2) void qtran_fwd_kernel(QtArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int AEP = 16 * FT, TS = AEP + 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, m = lane & 15;
  float* W1s = smem;                               // [FT][4]   W_e1[:, :64]
  float* W2s = W1s + FT * 4 * 256;                 // [FT][FT]  W_e2
  float* Q1s = W2s + FT * FT * 256;                // [4][FT]   W_q1[:, S:]
  float* Q2s = Q1s + 4 * FT * 256;                 // [4][4]    W_q2
  float* tab = Q2s + 16 * 256;                     // [17 | 1][TS]
  float* nb2 = tab + (HOT ? 17 : 1) * TS;          // [AEP]  N * b_e2
  float* bq2s = nb2 + AEP;                         // [64]
  float* w3s = bq2s + HD;                          // [64]
  stage_frag(W1s, a.We1, a.AE, 0, a.AE, HD, FT, 4, 64 * NW);
  stage_frag(W2s, a.We2, a.AE, 0, a.AE, a.AE, FT, FT, 64 * NW);
  stage_frag(Q1s, a.Wq1, a.ldq1, a.S, HD, a.AE, 4, FT, 64 * NW);
  stage_frag(Q2s, a.Wq2, HD, 0, HD, HD, 4, 4, 64 * NW);
  stage_tab<AEP, HOT>(tab, a, 64 * NW);
  for (int e = tid; e < AEP; e += 64 * NW) nb2[e] = e < a.AE ? (float)a.N * a.be2[e] : 0.f;
  for (int e = tid; e < HD; e += 64 * NW) { bq2s[e] = a.bq2[e]; w3s[e] = a.wq3[e]; }
  __syncthreads();
  const float bq3 = a.bq3[0];
  const long tiles = (a.BT + 15) / 16;
  const int N = a.N;

  for (long tile = first_tile(wave); tile < tiles; tile += tile_step()) {
    const bool live = tile * 16 + m < a.BT;
    const long bt = live ? tile * 16 + m : a.BT - 1;
    const float* hrow = a.hidden + bt * N * HD + 4 * q;
    const int* urow = HOT ? a.u + bt * N : nullptr;
    const int* urow2 = DUAL ? a.u2 + bt * N : nullptr;
    f32x4 spv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) spv[t] = *reinterpret_cast<const f32x4*>(a.sp + bt * HD + 16 * t + 4 * q);
    f32x4 s1[FT], s1d[DUAL ? FT : 1];
#pragma unroll
    for (int t = 0; t < FT; ++t) s1[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < (DUAL ? FT : 1); ++t) s1d[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // one agent: e1 = W_e1h h + (b_e1 + W_e1[:, 64 + u]);  s1 += relu(e1)
    auto step = [&](const f32x4 (&xv)[4], int uu, int uu2) __attribute__((always_inline)) {
      f32x4 acc[FT];
      const int tr = HOT ? ((uu >= 0 && uu < 16) ? uu : 16) : 0;
      const float* tp = tab + tr * TS + 4 * q;
#pragma unroll
      for (int t = 0; t < FT; ++t) acc[t] = *reinterpret_cast<const f32x4*>(tp + 16 * t);
      layer<FT, 4>(acc, W1s, xv, lane);
#pragma unroll
      for (int t = 0; t < FT; ++t) s1[t] += relu4(acc[t]);
      if constexpr (DUAL) {
        const float* tp2 = tab + ((uu2 >= 0 && uu2 < 16) ? uu2 : 16) * TS + 4 * q;
#pragma unroll
        for (int t = 0; t < FT; ++t) {
          const f32x4 d = *reinterpret_cast<const f32x4*>(tp2 + 16 * t) - *reinterpret_cast<const f32x4*>(tp + 16 * t);
          s1d[t] += relu4(acc[t] + d);
        }
      }
    };
    auto load = [&](f32x4 (&xv)[4], int& uu, int& uu2, int n) __attribute__((always_inline)) {
#pragma unroll
      for (int c = 0; c < 4; ++c) xv[c] = *reinterpret_cast<const f32x4*>(hrow + (long)n * HD + 16 * c);
      uu = HOT ? urow[n] : -1;
      uu2 = DUAL ? urow2[n] : -1;
    };
    // two named register sets (no set-to-set copies: the next agent's loads are in flight while this one computes)
    f32x4 xA[4], xB[4];
    int uA, uB, vA, vB;
    load(xA, uA, vA, 0);
    for (int n = 0; n < N; n += 2) {
      load(xB, uB, vB, n + 1 < N ? n + 1 : N - 1);
      step(xA, uA, vA);
      if (n + 1 < N) {
        load(xA, uA, vA, n + 2 < N ? n + 2 : N - 1);
        step(xB, uB, vB);
      }
    }
    if constexpr (DUAL) {            // the tail for the second action set (nothing saved)
      f32x4 e2d[FT], y1d[4], y2d[4];
#pragma unroll
      for (int t = 0; t < FT; ++t) e2d[t] = *reinterpret_cast<const f32x4*>(nb2 + 16 * t + 4 * q);
      layer<FT, FT>(e2d, W2s, s1d, lane);
#pragma unroll
      for (int t = 0; t < 4; ++t) y1d[t] = spv[t];
      layer<4, FT>(y1d, Q1s, e2d, lane);
#pragma unroll
      for (int t = 0; t < 4; ++t) { y1d[t] = relu4(y1d[t]); y2d[t] = *reinterpret_cast<const f32x4*>(bq2s + 16 * t + 4 * q); }
      layer<4, 4>(y2d, Q2s, y1d, lane);
      float o = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        y2d[t] = relu4(y2d[t]);
        const f32x4 w = *reinterpret_cast<const f32x4*>(w3s + 16 * t + 4 * q);
        o += (w[0] * y2d[t][0] + w[1] * y2d[t][1]) + (w[2] * y2d[t][2] + w[3] * y2d[t][3]);
      }
      o += __shfl_xor(o, 16, 64);
      o += __shfl_xor(o, 32, 64);
      if (live && q == 0) a.out2[bt] = o + bq3;
    }
    // esum = W_e2 s1 + N b_e2
    f32x4 e2[FT];
#pragma unroll
    for (int t = 0; t < FT; ++t) e2[t] = *reinterpret_cast<const f32x4*>(nb2 + 16 * t + 4 * q);
    layer<FT, FT>(e2, W2s, s1, lane);
    // y1 = relu(sp + W_q1e esum), y2 = relu(W_q2 y1 + b_q2), out = w_q3 . y2 + b_q3
    f32x4 y1[4], y2[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) y1[t] = spv[t];
    layer<4, FT>(y1, Q1s, e2, lane);
#pragma unroll
    for (int t = 0; t < 4; ++t) { y1[t] = relu4(y1[t]); y2[t] = *reinterpret_cast<const f32x4*>(bq2s + 16 * t + 4 * q); }
    layer<4, 4>(y2, Q2s, y1, lane);
    float o = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      y2[t] = relu4(y2[t]);
      const f32x4 w = *reinterpret_cast<const f32x4*>(w3s + 16 * t + 4 * q);
      o += (w[0] * y2[t][0] + w[1] * y2[t][1]) + (w[2] * y2[t][2] + w[3] * y2[t][3]);
    }
    o += __shfl_xor(o, 16, 64);
    o += __shfl_xor(o, 32, 64);
    if (live) {
      if (q == 0) a.out[bt] = o + bq3;
      if (a.s1) {
#pragma unroll
        for (int t = 0; t < FT; ++t) {
          *reinterpret_cast<f32x4*>(a.s1 + bt * AEP + 16 * t + 4 * q) = s1[t];
          *reinterpret_cast<f32x4*>(a.e2 + bt * AEP + 16 * t + 4 * q) = e2[t];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          *reinterpret_cast<f32x4*>(a.y1 + bt * HD + 16 * t + 4 * q) = y1[t];
          *reinterpret_cast<f32x4*>(a.y2 + bt * HD + 16 * t + 4 * q) = y2[t];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------- backward, row level
// dy2 = d_out w_q3 (y2 > 0); dy1 = W_q2^T dy2 (y1 > 0); de2 = W_q1e^T dy1; ds1 = W_e2^T de2; colsum(de2) -> slab1
template <int FT>
__global__ __launch_bounds__(64 * NW, 2) void qtran_bwd_rows_kernel(QtArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int AEP = 16 * FT;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, m = lane & 15;
  float* Q2T = smem;                               // [4][4]    W_q2^T
  float* Q1T = Q2T + 16 * 256;                     // [FT][4]   W_q1[:, S:]^T
  float* W2T = Q1T + FT * 4 * 256;                 // [FT][FT]  W_e2^T
  float* w3s = W2T + FT * FT * 256;                // [64]
  float* red = w3s + HD;                           // [NW][AEP]
  stage_fragT(Q2T, a.Wq2, HD, 0, HD, HD, 4, 4, 64 * NW);
  stage_fragT(Q1T, a.Wq1, a.ldq1, a.S, HD, a.AE, FT, 4, 64 * NW);
  stage_fragT(W2T, a.We2, a.AE, 0, a.AE, a.AE, FT, FT, 64 * NW);
  for (int e = tid; e < HD; e += 64 * NW) w3s[e] = a.wq3[e];
  __syncthreads();
  const long tiles = (a.BT + 15) / 16;
  f32x4 bsum[FT];
#pragma unroll
  for (int t = 0; t < FT; ++t) bsum[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (long tile = first_tile(wave); tile < tiles; tile += tile_step()) {
    const bool live = tile * 16 + m < a.BT;
    const long bt = live ? tile * 16 + m : a.BT - 1;
    const float d = live ? a.d_out[bt] : 0.f;      // rows past the end contribute exact zeros everywhere below
    f32x4 dy2[4], dy1[4], de2[FT], ds1[FT], y1v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 y2v = *reinterpret_cast<const f32x4*>(a.y2 + bt * HD + 16 * t + 4 * q);
      y1v[t] = *reinterpret_cast<const f32x4*>(a.y1 + bt * HD + 16 * t + 4 * q);
      const f32x4 w = *reinterpret_cast<const f32x4*>(w3s + 16 * t + 4 * q);
#pragma unroll
      for (int i = 0; i < 4; ++i) dy2[t][i] = y2v[i] > 0.f ? d * w[i] : 0.f;
      dy1[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    layer<4, 4>(dy1, Q2T, dy2, lane);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) dy1[t][i] = y1v[t][i] > 0.f ? dy1[t][i] : 0.f;
#pragma unroll
    for (int t = 0; t < FT; ++t) { de2[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; ds1[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    layer<FT, 4>(de2, Q1T, dy1, lane);
    layer<FT, FT>(ds1, W2T, de2, lane);
#pragma unroll
    for (int t = 0; t < FT; ++t) bsum[t] += de2[t];
    if (live) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        *reinterpret_cast<f32x4*>(a.dy2 + bt * HD + 16 * t + 4 * q) = dy2[t];
        *reinterpret_cast<f32x4*>(a.dy1 + bt * HD + 16 * t + 4 * q) = dy1[t];
      }
#pragma unroll
      for (int t = 0; t < FT; ++t) {
        *reinterpret_cast<f32x4*>(a.de2 + bt * AEP + 16 * t + 4 * q) = de2[t];
        *reinterpret_cast<f32x4*>(a.ds1 + bt * AEP + 16 * t + 4 * q) = ds1[t];
      }
    }
  }
  // column sums of de2 (the b_e2 gradient is N times this): rows m of the wave, then the waves in fixed order
#pragma unroll
  for (int t = 0; t < FT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = bsum[t][i];
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
      if (m == 0) red[wave * AEP + 16 * t + 4 * q + i] = v;
    }
  __syncthreads();
  for (int f = tid; f < AEP; f += 64 * NW) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w * AEP + f];
    a.slab1[(long)blockIdx.x * AEP + f] = s;
  }
}

// ------------------------------------------------------------------------------------------------- backward, agent level
// per agent row: e1 (recomputed) -> de1 = ds1 (e1 > 0); dh = W_e1h^T de1; dW_e1 += de1^T [h | onehot(u)], db_e1 += de1
template <int FT, bool HOT>
__global__ __launch_bounds__(64 * NW, 2) void qtran_bwd_agents_kernel(QtArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int AEP = 16 * FT, TS = AEP + 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, m = lane & 15;
  float* W1s = smem;                               // [FT][4]  W_e1[:, :64]            (recompute of e1)
  float* W1T = W1s + FT * 4 * 256;                 // [4][FT]  W_e1[:, :64]^T          (dh)
  float* tab = W1T + 4 * FT * 256;                 // [17 | 1][TS]
  float* stg = tab + (HOT ? 17 : 1) * TS;          // [NW][16][TS]  de1 of the wave's current (tile, agent), row major
  float* xst = stg + NW * 16 * TS;                 // [NW][16][XS]  h rows of the same
  int* ust = reinterpret_cast<int*>(xst + NW * 16 * XS);   // [NW][16]
  stage_frag(W1s, a.We1, a.AE, 0, a.AE, HD, FT, 4, 64 * NW);
  stage_fragT(W1T, a.We1, a.AE, 0, a.AE, HD, 4, FT, 64 * NW);
  stage_tab<AEP, HOT>(tab, a, 64 * NW);
  __syncthreads();
  float* st = stg + wave * 16 * TS;
  float* xs = xst + wave * 16 * XS;
  int* us = ust + wave * 16;
  const long tiles = (a.BT + 15) / 16;
  const int N = a.N;
  f32x4 accW[FT][4], accU[FT];
  float sb[FT];
#pragma unroll
  for (int t = 0; t < FT; ++t) {
    accU[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    sb[t] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) accW[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  for (long tile = first_tile(wave); tile < tiles; tile += tile_step()) {
    const bool live = tile * 16 + m < a.BT;
    const long bt = live ? tile * 16 + m : a.BT - 1;
    const float* hrow = a.hidden + bt * N * HD + 4 * q;
    float* dhrow = a.dhidden + bt * N * HD + 4 * q;
    const int* urow = HOT ? a.u + bt * N : nullptr;
    f32x4 ds1[FT];
#pragma unroll
    for (int t = 0; t < FT; ++t) {
      ds1[t] = *reinterpret_cast<const f32x4*>(a.ds1 + bt * AEP + 16 * t + 4 * q);
      if (!live) ds1[t] = (f32x4){0.f, 0.f, 0.f, 0.f};      // rows past the end: zero gradient, nothing stored
    }
    auto load = [&](f32x4 (&xv)[4], int& uu, int n) __attribute__((always_inline)) {
#pragma unroll
      for (int c = 0; c < 4; ++c) xv[c] = *reinterpret_cast<const f32x4*>(hrow + (long)n * HD + 16 * c);
      uu = HOT ? urow[n] : -1;
    };
    auto step = [&](const f32x4 (&xv)[4], int uu, int n) __attribute__((always_inline)) {
      f32x4 old[4];
      if (a.accumulate) {
#pragma unroll
        for (int tk = 0; tk < 4; ++tk) old[tk] = *reinterpret_cast<const f32x4*>(dhrow + (long)n * HD + 16 * tk);
      }
      f32x4 acc[FT];
      const int tr = HOT ? ((uu >= 0 && uu < 16) ? uu : 16) : 0;
      const float* tp = tab + tr * TS + 4 * q;
#pragma unroll
      for (int t = 0; t < FT; ++t) acc[t] = *reinterpret_cast<const f32x4*>(tp + 16 * t);
      layer<FT, 4, false>(acc, W1s, xv, lane);
      // de1 (transposed layout: lane (q, m) = features 16t + 4q + i of row m) -> the wave's LDS tile [row][feature]
#pragma unroll
      for (int t = 0; t < FT; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = acc[t][i] > 0.f ? ds1[t][i] : 0.f;
        *reinterpret_cast<f32x4*>(st + m * TS + 16 * t + 4 * q) = acc[t];
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) *reinterpret_cast<f32x4*>(xs + m * XS + 16 * c + 4 * q) = xv[c];
      if (HOT && q == 0) us[m] = uu;
      // dh = W_e1h^T de1
      f32x4 dh[4];
#pragma unroll
      for (int tk = 0; tk < 4; ++tk) dh[tk] = (f32x4){0.f, 0.f, 0.f, 0.f};
      layer<4, FT, false>(dh, W1T, acc, lane);
      if (live) {
#pragma unroll
        for (int tk = 0; tk < 4; ++tk) {
          if (a.accumulate) dh[tk] += old[tk];
          *reinterpret_cast<f32x4*>(dhrow + (long)n * HD + 16 * tk) = dh[tk];
        }
      }
      // dW_e1 += de1^T [h | onehot]: both operands in the ROW layout (lane (q, m), register r = row 4q + r, column m
      // of the tile) - two accumulator-layout tiles over the same 16 rows are the (A^T, B) pair of the MFMA
      f32x4 xr[4], hot;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) xr[c][r] = xs[(4 * q + r) * XS + 16 * c + m];
      if (HOT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) hot[r] = us[4 * q + r] == m ? 1.f : 0.f;
      }
#pragma unroll
      for (int t = 0; t < FT; ++t) {
        f32x4 g;
#pragma unroll
        for (int r = 0; r < 4; ++r) g[r] = st[(4 * q + r) * TS + 16 * t + m];
        sb[t] += (g[0] + g[1]) + (g[2] + g[3]);
#pragma unroll
        for (int c = 0; c < 4; ++c) accW[t][c] = mfma16x4(g, xr[c], accW[t][c]);
        if (HOT) accU[t] = mfma16x4(g, hot, accU[t]);
      }
    };
    f32x4 xA[4], xB[4];
    int uA, uB;
    load(xA, uA, 0);
    for (int n = 0; n < N; n += 2) {
      load(xB, uB, n + 1 < N ? n + 1 : N - 1);
      step(xA, uA, n);
      if (n + 1 < N) {
        load(xA, uA, n + 2 < N ? n + 2 : N - 1);
        step(xB, uB, n + 1);
      }
    }
  }
  // ---- the waves' partial dW_e1 meet in LDS in wave order (deterministic), one slab per workgroup
  __syncthreads();
  float* red = smem;                                // [AEP][KW] over the weight tiles (no longer needed)
#pragma unroll
  for (int t = 0; t < FT; ++t) {
    sb[t] += __shfl_xor(sb[t], 16, 64);
    sb[t] += __shfl_xor(sb[t], 32, 64);
  }
  for (int w = 0; w < NW; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < FT; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float* rr = red + (16 * t + 4 * q + i) * KW;
#pragma unroll
          for (int c = 0; c < 4; ++c) rr[16 * c + m] = (w == 0 ? 0.f : rr[16 * c + m]) + accW[t][c][i];
          rr[HD + m] = (w == 0 ? 0.f : rr[HD + m]) + (HOT ? accU[t][i] : 0.f);
        }
        if (q == 0) {
          float* rb = red + (16 * t + m) * KW + HD + 16;
          *rb = (w == 0 ? 0.f : *rb) + sb[t];
        }
      }
    }
    __syncthreads();
  }
  float* slab = a.slab2 + (long)blockIdx.x * AEP * KW;
  for (int e = tid; e < AEP * KW; e += 64 * NW) slab[e] = (e % KW) <= HD + 16 ? red[e] : 0.f;
}

// ------------------------------------------------------------------------------------------------- state parts
// sp_k = W_k[:, :S] s + b_k  (k < NSETS), (BT, 64) each: the state columns of the heads' first layers (:386, :416).
// The joint-Q head and the V head of one update read the SAME states, so NSETS = 2 serves both from one pass over s.
// A workgroup stages 64 contiguous rows of s (one contiguous 64*S*4-byte block, 16-byte loads) in LDS; wave (t, g) keeps
// the fragments of feature tile t of its weight set in registers (KC chunks of 16 columns) and multiplies them with the
// row tiles of the block: NSETS = 2: set g, all four row tiles; NSETS = 1: row tiles 2g, 2g + 1.  Two workgroups per CU
// cover each other's loads.  fp32 on v_mfma_f32_16x16x4_f32, the accumulators of the row tiles interleaved.
struct SpArgs {
  ConcatSrc s; long BT; int S, kc;
  // per (set, 16-feature tile): first weight row (S columns used, row stride ldw), bias, first output column, relu?
  const float* W[2][4]; long ldw[2]; const float* b[2][4]; float* out[2][4]; long ldo[2]; int relu[2][4];
};
constexpr int SPR = 64;           // rows per block

// row `row` of the states (remap + episode map of ConcatSrc) as an element offset into p0, in two steps so that callers can
// issue the episode-map reads of several rows together before the first dependent row read: (e, w) first, then the offset
__device__ __forceinline__ void state_row_ew(const ConcatSrc& s, long row, unsigned& e, long& w) {
  if (s.rpe0 == 0) { e = 0; w = row; return; }
  e = fastdiv((unsigned)row, s.fd0);
  w = (long)((unsigned)row - e * (unsigned)s.rpe0) + s.off0;
}
__device__ __forceinline__ long state_row_off(const ConcatSrc& s, unsigned e, long w, int ev) {
  return (s.rpe0 == 0 ? w : (long)(s.emap0 ? ev : (int)e) * s.bs0 + w) * s.ld0;
}

template <int KC, int NSETS>
__global__ __launch_bounds__(64 * NW, 1) void qtran_state_parts_kernel(SpArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int RT = NSETS == 2 ? 4 : 2;          // row tiles per wave and block
  constexpr int CW = KC <= 16 ? 1 : 2;            // 16-byte column slots per lane and row (S <= 256: one)
  constexpr int RW = SPR / NW;                    // rows a wave stages per block
  constexpr int XR = 16 * KC + 4;                 // LDS row stride; the columns S..16 KC stay zero
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, m = lane & 15;
  const int t = wave & 3, g = wave >> 2;
  const int set = NSETS == 2 ? g : 0, rt0 = NSETS == 2 ? 0 : 2 * g;
  const int S = a.S;
  const float* W = a.W[set][t] + (long)m * a.ldw[set];
  f32x4 wf[KC];
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    const int k0 = 16 * c + 4 * q;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float v = W[k0 + i < S ? k0 + i : S - 1];
      wf[c][i] = k0 + i < S ? v : 0.f;
    }
  }
  const f32x4 bv = *reinterpret_cast<const f32x4*>(a.b[set][t] + 4 * q);
  float* sp = a.out[set][t];
  const long ldo = a.ldo[set];
  const bool relu = a.relu[set][t] != 0;
  for (int e = tid; e < SPR * (XR - S); e += 64 * NW) {
    const int r = e / (XR - S), c = e - r * (XR - S);
    smem[r * XR + S + c] = 0.f;
  }
  const int S4 = (S + 3) >> 2;         // (S % 4 != 0: the row stride covers the last 16 bytes; the elements past S are zeroed at the LDS store)
  // the next block's rows travel in registers while this one multiplies: wave w stages rows w, w + 8, .. (one contiguous
  // 4*S-byte read per row; the row's place in s - remap, episode map - is wave-uniform).  Rows past BT re-read row BT - 1
  // (their results are not stored), lanes past the row its last 16 bytes (not staged): no branches around the loads.
  f32x4 ps[RW][CW];
  auto fetch = [&](long blk) {
    const long row0 = blk * SPR;
    unsigned e[RW]; long w[RW]; int ev[RW];
#pragma unroll
    for (int j = 0; j < RW; ++j) {
      long row = row0 + wave + NW * j;
      row = row < a.BT ? row : a.BT - 1;
      state_row_ew(a.s, row, e[j], w[j]);
      ev[j] = a.s.emap0 ? a.s.emap0[e[j]] : 0;
    }
#pragma unroll
    for (int j = 0; j < RW; ++j) {
      const bool ok = w[j] >= 0 || a.s.rpe0 == 0;
      const float* src = a.s.p0 + state_row_off(a.s, e[j], ok ? w[j] : 0, ev[j]);
#pragma unroll
      for (int k = 0; k < CW; ++k) {
        int c4 = lane + 64 * k;
        c4 = c4 < S4 ? c4 : S4 - 1;
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * c4);
        ps[j][k] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  const long nblk = (a.BT + SPR - 1) / SPR;
  long blk = blockIdx.x;
  if (blk < nblk) fetch(blk);
  for (; blk < nblk; blk += gridDim.x) {
    const long row0 = blk * SPR;
#pragma unroll
    for (int j = 0; j < RW; ++j)
#pragma unroll
      for (int k = 0; k < CW; ++k) {
        const int c4 = lane + 64 * k;
        f32x4 v = ps[j][k];
        if ((S & 3) && c4 == S4 - 1) {      // the row's last 16 bytes reach past S: whatever the neighbour holds there (a
#pragma unroll                            // column slice of a wider buffer) must not meet the zero weights as NaN / Inf
          for (int i = 1; i < 4; ++i)
            if (i >= (S & 3)) v[i] = 0.f;
        }
        if (c4 < S4) *reinterpret_cast<f32x4*>(smem + (wave + NW * j) * XR + 4 * c4) = v;
      }
    __syncthreads();
    if (blk + gridDim.x < nblk) fetch(blk + gridDim.x);
    f32x4 acc[RT];
    const float* xb = smem + (16 * rt0 + m) * XR + 4 * q;
#pragma unroll
    for (int i = 0; i < RT; ++i) acc[i] = bv;
    f32x4 xn[RT];                                          // fragments of chunk c + 1 are read while chunk c multiplies
#pragma unroll
    for (int i = 0; i < RT; ++i) xn[i] = *reinterpret_cast<const f32x4*>(xb + i * 16 * XR);
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      f32x4 x[RT];
#pragma unroll
      for (int i = 0; i < RT; ++i) x[i] = xn[i];
      if (c + 1 < KC) {
#pragma unroll
        for (int i = 0; i < RT; ++i) xn[i] = *reinterpret_cast<const f32x4*>(xb + i * 16 * XR + 16 * (c + 1));
      }
      if constexpr (RT == 4) mfma16x4_il4(wf[c], x[0], acc[0], wf[c], x[1], acc[1], wf[c], x[2], acc[2], wf[c], x[3], acc[3]);
      else mfma16x4_il2(wf[c], x[0], acc[0], wf[c], x[1], acc[1]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const long row = row0 + 16 * (rt0 + i) + m;
      if (row < a.BT) *reinterpret_cast<f32x4*>(sp + row * ldo + 4 * q) = relu ? relu4(acc[i]) : acc[i];
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------- row-level weight gradients
// One pass over the B*T rows for every weight gradient that is a reduction over rows (:386-387, :416-417, the second
// encoder layer :381, :412):
//   dW_q1 += dy1^T [s | e2]   db_q1 += colsum(dy1)      dW_q2 += dy2^T y1   db_q2 += colsum(dy2)
//   dw_q3 += d_out^T y2       db_q3 += sum(d_out)       dW_e2 += de2^T s1
// (five marl_linear_wgrad launches and their reduces before).  A workgroup stages WR = 32 rows of all nine tensors side by
// side in ONE LDS matrix Z[32][ZW]: a single compile-time row stride = 4 mod 16, so the column-wise fragment reads below
// are conflict-free and their row offsets are instruction immediates; the next block's rows travel in registers while
// this one multiplies.  Every product is out[j][k] = sum_r Z[r][acol + j] Z[r][bcol + k]: 16x16 output tiles dealt
// round-robin to the 8 waves (tile ids: dy1^T s k-tile-major, dy1^T e2, dy2^T y1, de2^T s1), accumulated in registers
// over the workgroup's blocks, written to a slab per workgroup and summed in a fixed order by
// qtran_wgrad_reduce_kernel: bitwise reproducible.
constexpr int WR = 32;

struct WgArgs {
  ConcatSrc s;
  const float *e2, *y1, *s1, *dy1, *dy2, *de2, *y2, *d_out;
  float* slab;                    // [grid][slab_elems]
  long BT; int S, AE, TS, ntiles;
};
struct WgRedArgs {
  const float* slab;
  float *dWq1, *dbq1, *dWq2, *dbq2, *dwq3, *dbq3, *dWe2;
  long ldq1; int nwg, S, AE, AEP, TS;
};
// slab of one workgroup: [64][16 TS] dy1^T s | [64][AEP] dy1^T e2 | [64][64] dy2^T y1 | [AEP][AEP] de2^T s1 | 256 vector sums
__host__ __device__ inline int wg_slab_elems(int TS, int AEP) { return 64 * 16 * TS + 64 * AEP + 64 * 64 + AEP * AEP + 256; }

template <int W4>
__device__ __forceinline__ void wg_slot(int e, int& r, int& c4) { r = e / W4; c4 = e - r * W4; }

// FT: feature tiles of the encoder (5: joint-Q, 4: V); SMAX: columns reserved for s (224: 4 register slots, 16 tiles per
// wave; 384: 6 slots, 20 tiles)
template <int FT, int SMAX>
__global__ __launch_bounds__(64 * NW, 1) void qtran_wgrad_rows_kernel(WgArgs a) {
  extern __shared__ __attribute__((aligned(16))) float Z[];
  constexpr int AEP = 16 * FT, E4 = AEP / 4, ESL = (WR * E4 + 64 * NW - 1) / (64 * NW);
  constexpr int SSL = (WR * (SMAX / 4) + 64 * NW - 1) / (64 * NW), MAXT = SMAX <= 224 ? 16 : 20;
  constexpr int cE2 = 0, cY1 = AEP, cS1 = cY1 + 64, cD1 = cS1 + AEP, cD2 = cD1 + 64, cDE = cD2 + 64, cY2 = cDE + AEP, cDO = cY2 + 64,
                cS = cDO + 4, ZW = cS + SMAX;
  static_assert(ZW % 16 == 4 || ZW % 16 == 12, "row stride");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, m = lane & 15;
  const int S = a.S, S4 = S >> 2, TS = a.TS;
  const int NS = 4 * TS, XW = 16 * TS;
  const int OFF1 = 64 * XW, OFF2 = OFF1 + 64 * AEP, OFF3 = OFF2 + 64 * 64, OFFB = OFF3 + AEP * AEP;

  // tile id -> fragment columns in Z (ac, bc) and the slab position (so, ld)
  auto tile_of = [&](int id, int& ac, int& bc, int& so, int& ld) {
    ac = cD1; bc = cY1; so = -1; ld = 0;
    if (id < NS) { const int tj = id & 3, tk = id >> 2; ac = cD1 + 16 * tj; bc = cS + 16 * tk; so = 16 * tj * XW + 16 * tk; ld = XW; return; }
    id -= NS;
    if (id < 4 * FT) { const int tj = id & 3, tk = id >> 2; ac = cD1 + 16 * tj; bc = cE2 + 16 * tk; so = OFF1 + 16 * tj * AEP + 16 * tk; ld = AEP; return; }
    id -= 4 * FT;
    if (id < 16) { const int tj = id & 3, tk = id >> 2; ac = cD2 + 16 * tj; bc = cY1 + 16 * tk; so = OFF2 + 16 * tj * 64 + 16 * tk; ld = 64; return; }
    id -= 16;
    if (id < FT * FT) { const int tk = id / FT, tj = id - tk * FT; ac = cDE + 16 * tj; bc = cS1 + 16 * tk; so = OFF3 + 16 * tj * AEP + 16 * tk; ld = AEP; }
  };
  const float* zl = Z + 4 * q * ZW + m;
  f32x4 acc[MAXT];
#pragma unroll
  for (int i = 0; i < MAXT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float vsum = 0.f;               // threads 0..63: db_q1[tid]; 64..127: db_q2; 128..191: dw_q3; 192: db_q3

  // the columns of the last s tile past S are read by its products (their results are never used): keep them finite
  for (int e = tid; e < WR * (XW - S); e += 64 * NW) {
    const int r = e / (XW - S), c = e - r * (XW - S);
    if (S + c < SMAX) Z[r * ZW + cS + S + c] = 0.f;
  }
  // staging slots (block-invariant): slot j of a segment is its float4 number j*512 + tid -> (row, column/4)
  int srow[SSL], sc4[SSL];
#pragma unroll
  for (int j = 0; j < SSL; ++j) { const int e = j * 64 * NW + tid; srow[j] = e / S4; sc4[j] = e - srow[j] * S4; }
  f32x4 ps[SSL], pe2[ESL], ps1[ESL], pde[ESL], py1, pd1, pd2, py2;
  float pdo = 0.f;
  const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto fetch = [&](long blk) {
    const long row0 = blk * WR;
    const int nrows = (int)(a.BT - row0 < WR ? a.BT - row0 : WR);
    unsigned se[SSL]; long sw[SSL]; int sev[SSL];
#pragma unroll
    for (int j = 0; j < SSL; ++j) {                     // (rows past the block re-read its first row; selected away below)
      state_row_ew(a.s, row0 + (srow[j] < nrows ? srow[j] : 0), se[j], sw[j]);
      sev[j] = a.s.emap0 ? a.s.emap0[se[j]] : 0;
    }
#pragma unroll
    for (int j = 0; j < SSL; ++j) {
      const bool ok = srow[j] < nrows && (sw[j] >= 0 || a.s.rpe0 == 0);
      const f32x4 v = *reinterpret_cast<const f32x4*>(a.s.p0 + state_row_off(a.s, se[j], sw[j] >= 0 ? sw[j] : 0, sev[j]) + 4 * sc4[j]);
      ps[j] = ok ? v : zero4;
    }
#pragma unroll
    for (int j = 0; j < ESL; ++j) {
      int r, c4; wg_slot<E4>(j * 64 * NW + tid, r, c4);
      const bool ok = r < nrows;
      const long o = (row0 + r) * AEP + 4 * c4;
      pe2[j] = ok ? *reinterpret_cast<const f32x4*>(a.e2 + o) : zero4;
      ps1[j] = ok ? *reinterpret_cast<const f32x4*>(a.s1 + o) : zero4;
      pde[j] = ok ? *reinterpret_cast<const f32x4*>(a.de2 + o) : zero4;
    }
    {
      int r, c4; wg_slot<16>(tid, r, c4);
      const bool ok = r < nrows;
      const long o = (row0 + r) * 64 + 4 * c4;
      py1 = ok ? *reinterpret_cast<const f32x4*>(a.y1 + o) : zero4;
      pd1 = ok ? *reinterpret_cast<const f32x4*>(a.dy1 + o) : zero4;
      pd2 = ok ? *reinterpret_cast<const f32x4*>(a.dy2 + o) : zero4;
      py2 = ok ? *reinterpret_cast<const f32x4*>(a.y2 + o) : zero4;
    }
    if (tid < WR) pdo = tid < nrows ? a.d_out[row0 + tid] : 0.f;
  };
  auto stage = [&]() {
#pragma unroll
    for (int j = 0; j < SSL; ++j)
      if (srow[j] < WR) *reinterpret_cast<f32x4*>(Z + srow[j] * ZW + cS + 4 * sc4[j]) = ps[j];
#pragma unroll
    for (int j = 0; j < ESL; ++j) {
      int r, c4; wg_slot<E4>(j * 64 * NW + tid, r, c4);
      if (r < WR) {
        *reinterpret_cast<f32x4*>(Z + r * ZW + cE2 + 4 * c4) = pe2[j];
        *reinterpret_cast<f32x4*>(Z + r * ZW + cS1 + 4 * c4) = ps1[j];
        *reinterpret_cast<f32x4*>(Z + r * ZW + cDE + 4 * c4) = pde[j];
      }
    }
    {
      int r, c4; wg_slot<16>(tid, r, c4);
      *reinterpret_cast<f32x4*>(Z + r * ZW + cY1 + 4 * c4) = py1;
      *reinterpret_cast<f32x4*>(Z + r * ZW + cD1 + 4 * c4) = pd1;
      *reinterpret_cast<f32x4*>(Z + r * ZW + cD2 + 4 * c4) = pd2;
      *reinterpret_cast<f32x4*>(Z + r * ZW + cY2 + 4 * c4) = py2;
    }
    if (tid < WR) Z[tid * ZW + cDO] = pdo;
  };

  const long nblk = (a.BT + WR - 1) / WR;
  long blk = blockIdx.x;
  if (blk < nblk) fetch(blk);
  for (; blk < nblk; blk += gridDim.x) {
    stage();
    __syncthreads();
    if (blk + gridDim.x < nblk) fetch(blk + gridDim.x);
    // tile pairs (i, i + 1) x two 16-row chunks, software-pipelined: the 16 fragment reads of the next step are issued
    // before the 8 MFMAs of this one (tiles past ntiles read valid columns; their products are skipped)
    auto frags = [&](int i, int c, f32x4& fa0, f32x4& fb0, f32x4& fa1, f32x4& fb1) {
      int ac0, bc0, ac1, bc1, so, ld;
      tile_of(wave + NW * i, ac0, bc0, so, ld);
      tile_of(wave + NW * (i + 1), ac1, bc1, so, ld);
      const float *pa0 = zl + ac0, *pb0 = zl + bc0, *pa1 = zl + ac1, *pb1 = zl + bc1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        fa0[k] = pa0[(16 * c + k) * ZW]; fb0[k] = pb0[(16 * c + k) * ZW];
        fa1[k] = pa1[(16 * c + k) * ZW]; fb1[k] = pb1[(16 * c + k) * ZW];
      }
    };
    f32x4 na0, nb0, na1, nb1;
    frags(0, 0, na0, nb0, na1, nb1);
#pragma unroll
    for (int i = 0; i < MAXT; i += 2) {
#pragma unroll
      for (int c = 0; c < WR / 16; ++c) {
        const f32x4 fa0 = na0, fb0 = nb0, fa1 = na1, fb1 = nb1;
        if (c + 1 < WR / 16) frags(i, c + 1, na0, nb0, na1, nb1);
        else if (i + 2 < MAXT) frags(i + 2, 0, na0, nb0, na1, nb1);
        if (wave + NW * i < a.ntiles) mfma16x4_il2(fa0, fb0, acc[i], fa1, fb1, acc[i + 1]);      // wave-uniform
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // bias / last-layer sums on the vector unit (4 waves, one column each)
    if (tid < 193) {
      const int col = tid < 64 ? cD1 + tid : tid < 128 ? cD2 + tid - 64 : tid < 192 ? cY2 + tid - 128 : cDO;
      const bool wd = tid >= 128 && tid < 192;
      float v = 0.f;
#pragma unroll 8
      for (int r = 0; r < WR; ++r) v += wd ? Z[r * ZW + col] * Z[r * ZW + cDO] : Z[r * ZW + col];
      vsum += v;
    }
    __syncthreads();
  }
  float* slab = a.slab + (long)blockIdx.x * wg_slab_elems(TS, AEP);
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    int ac, bc, so, ld;
    tile_of(wave + NW * i, ac, bc, so, ld);
    if (so >= 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) slab[so + (4 * q + k) * ld + m] = acc[i][k];
    }
  }
  if (tid < 256) slab[OFFB + tid] = tid < 193 ? vsum : 0.f;
}

// gradients += slabs (fixed order: 4 interleaved partial sums per element, as qtran_reduce_kernel)
__global__ __launch_bounds__(256) void qtran_wgrad_reduce_kernel(WgRedArgs a) {
  __shared__ float part[4][64];
  const int el = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;
  const int XW = 16 * a.TS, OFF1 = 64 * XW, OFF2 = OFF1 + 64 * a.AEP, OFF3 = OFF2 + 64 * 64, OFFB = OFF3 + a.AEP * a.AEP, n = OFFB + 256;
  // which destination (if any) this element feeds: the pad elements are skipped without being read
  float* dst = nullptr;
  if (e < OFF1) { const int j = e / XW, k = e - j * XW; if (k < a.S) dst = a.dWq1 + (long)j * a.ldq1 + k; }
  else if (e < OFF2) { const int i1 = e - OFF1, j = i1 / a.AEP, k = i1 - j * a.AEP; if (k < a.AE) dst = a.dWq1 + (long)j * a.ldq1 + a.S + k; }
  else if (e < OFF3) dst = a.dWq2 + (e - OFF2);
  else if (e < OFFB) { const int i3 = e - OFF3, j = i3 / a.AEP, k = i3 - j * a.AEP; if (j < a.AE && k < a.AE) dst = a.dWe2 + (long)j * a.AE + k; }
  else if (e < n) { const int b = e - OFFB; dst = b < 64 ? a.dbq1 + b : b < 128 ? a.dbq2 + (b - 64) : b < 192 ? a.dwq3 + (b - 128) : b == 192 ? a.dbq3 : nullptr; }
  float s = 0.f;
  if (dst) s = slab_sum(a.slab + e, n, sg, 4, a.nwg);
  part[sg][el] = s;
  __syncthreads();
  if (sg != 0 || !dst) return;
  *dst += ((part[0][el] + part[1][el]) + part[2][el]) + part[3][el];
}

struct QtRedArgs {
  const float *slab1, *slab2;
  float *dWe1, *dbe1, *dbe2;
  int nwg, AE, AEP, A, N;
};
// gradients += slabs: 64 elements per block, 4 slab groups per element (thread (e, sg) sums slabs sg, sg+4, .. in order,
// the 4 partial sums are added in a fixed order) -> deterministic, and 4x shorter dependent chains
__global__ __launch_bounds__(256) void qtran_reduce_kernel(QtRedArgs a) {
  __shared__ float part[4][64];
  const int el = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;
  const int n2 = a.AEP * KW;
  float s = 0.f;
  if (e < n2) s = slab_sum(a.slab2 + e, n2, sg, 4, a.nwg);
  else if (e < n2 + a.AEP) s = slab_sum(a.slab1 + (e - n2), a.AEP, sg, 4, a.nwg);
  part[sg][el] = s;
  __syncthreads();
  if (sg != 0) return;
  s = ((part[0][el] + part[1][el]) + part[2][el]) + part[3][el];
  if (e < n2) {
    const int f = e / KW, k = e - f * KW;
    if (f >= a.AE || k > HD + 16) return;
    if (k < HD) a.dWe1[(long)f * a.AE + k] += s;
    else if (k < HD + 16) { if (k - HD < a.A) a.dWe1[(long)f * a.AE + k] += s; }
    else a.dbe1[f] += s;
  } else if (e < n2 + a.AEP) {
    const int f = e - n2;
    if (f < a.AE) a.dbe2[f] += (float)a.N * s;
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// the states as the callers hold them: dense (BT, S) or rows of the (T+1)-slot episode storage through a row remap and an
// optional episode map (replay samples read in place); one dense segment, rows 16-byte aligned
inline bool state_src_ok(const marl_src_t* s, int S) {
  return s && s->p0 && s->k0 == S && !s->k1 && !s->nhot && !s->nid && !s->m0 && s->ld0 % 4 == 0 && aligned16(s->p0);
}
inline ConcatSrc state_src(const marl_src_t* s) {
  ConcatSrc c = {};
  c.p0 = s->p0; c.ld0 = s->ld0; c.k0 = s->k0;
  c.rpe0 = s->rpe0; c.bs0 = s->bs0; c.off0 = s->off0;
  c.fd0 = make_fastdiv((unsigned)(s->rpe0 > 0 ? s->rpe0 : 1));
  c.fdi = make_fastdiv(1); c.fdn = make_fastdiv(1);
  c.emap0 = s->emap0;
  return c;
}
inline int ft_of(int AE) { return (AE + 15) / 16; }
inline size_t fwd_lds(int FT, bool hot) {
  const int AEP = 16 * FT;
  return (size_t)((FT * 4 + FT * FT + 4 * FT + 16) * 256 + (hot ? 17 : 1) * (AEP + 4) + AEP + 2 * HD) * 4;
}
inline size_t rows_lds(int FT) { return (size_t)((16 + FT * 4 + FT * FT) * 256 + HD + NW * 16 * FT) * 4; }
inline size_t agents_lds(int FT, bool hot) {
  const int AEP = 16 * FT;
  size_t w = (size_t)((FT * 4 + 4 * FT) * 256 + (hot ? 17 : 1) * (AEP + 4) + NW * 16 * (AEP + 4) + NW * 16 * XS + NW * 16) * 4;
  const size_t r = (size_t)AEP * KW * 4;
  return w > r ? w : r;
}
inline unsigned grid_for(long BT) {
  const long tiles = (BT + 15) / 16;
  long g = (tiles + NW - 1) / NW;
  if (g > 256) g = 256;
  return (unsigned)(g < 1 ? 1 : g);
}
// A = 0: no one-hot block (QtranV, AE = 64 -> 4 feature tiles); 1 <= A <= 16: joint-Q (AE = 64 + A -> 5 feature tiles)
inline bool supported(int N, int A, int AE) { return N >= 1 && A >= 0 && A <= 16 && AE == HD + A; }
inline void fill_w(QtArgs& a, const marl_qtran_weights_t* w) {
  a.We1 = w->enc0_w; a.be1 = w->enc0_b; a.We2 = w->enc2_w; a.be2 = w->enc2_b;
  a.Wq1 = w->q0_w; a.ldq1 = w->q0_ld; a.S = w->q0_s;
  a.Wq2 = w->q2_w; a.bq2 = w->q2_b; a.wq3 = w->q4_w; a.bq3 = w->q4_b;
}

template <typename K>
inline int launch(K fn, const QtArgs& a, unsigned grid, size_t lds, hipStream_t s) {
  hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  void* kargs[] = {(void*)&a};
  e = hipLaunchKernel((const void*)fn, dim3(grid), dim3(64 * NW), kargs, lds, s);
  if (e != hipSuccess) return (int)e;
  MARL_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" int marl_qtran_supported(int N, int A, int AE) { return supported(N, A, AE) ? 1 : 0; }

extern "C" int marl_qtran_head_fwd(const marl_qtran_weights_t* w, const float* hidden, const int* u, const float* sp,
                                   float* out, float* s1, float* e2, float* y1, float* y2, long BT, int N, int A, int AE,
                                   void* stream) {
  if (BT <= 0) return 0;
  if (!supported(N, A, AE) || (A > 0) != (u != nullptr)) return (int)hipErrorInvalidValue;
  if (!aligned16(hidden) || !aligned16(sp)) return (int)hipErrorInvalidValue;
  const bool save = s1 != nullptr;
  if (save && (!e2 || !y1 || !y2 || !aligned16(s1) || !aligned16(e2) || !aligned16(y1) || !aligned16(y2))) return (int)hipErrorInvalidValue;
  QtArgs a = {};
  fill_w(a, w);
  a.hidden = hidden; a.u = u; a.sp = sp; a.out = out; a.s1 = s1; a.e2 = e2; a.y1 = y1; a.y2 = y2;
  a.BT = BT; a.N = N; a.A = A; a.AE = AE;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = grid_for(BT);
  if (A > 0) return launch(qtran_fwd_kernel<5, true>, a, grid, fwd_lds(5, true), s);
  return launch(qtran_fwd_kernel<4, false>, a, grid, fwd_lds(4, false), s);
}

extern "C" int marl_qtran_head_fwd2(const marl_qtran_weights_t* w, const float* hidden, const int* u, const int* u2, const float* sp,
                                    float* out, float* out2, float* s1, float* e2, float* y1, float* y2, long BT, int N, int A,
                                    int AE, void* stream) {
  if (BT <= 0) return 0;
  if (!supported(N, A, AE) || A <= 0 || !u || !u2 || !out2) return (int)hipErrorInvalidValue;
  if (!aligned16(hidden) || !aligned16(sp)) return (int)hipErrorInvalidValue;
  const bool save = s1 != nullptr;
  if (save && (!e2 || !y1 || !y2 || !aligned16(s1) || !aligned16(e2) || !aligned16(y1) || !aligned16(y2))) return (int)hipErrorInvalidValue;
  QtArgs a = {};
  fill_w(a, w);
  a.hidden = hidden; a.u = u; a.u2 = u2; a.sp = sp; a.out = out; a.out2 = out2; a.s1 = s1; a.e2 = e2; a.y1 = y1; a.y2 = y2;
  a.BT = BT; a.N = N; a.A = A; a.AE = AE;
  return launch(qtran_fwd_kernel<5, true, true>, a, grid_for(BT), fwd_lds(5, true), (hipStream_t)stream);
}

namespace {
// launch of the row GEMM [BT, S] x [S, 64 per set] -> 16-feature tiles written where the descriptors say
int sp_launch(SpArgs& a, int nsets, hipStream_t stream) {
  a.kc = (a.S + 15) / 16;
  const long nblk = (a.BT + SPR - 1) / SPR;
  static const int buckets[] = {4, 8, 12, 14, 16, 20, 24};
  int KCT = 24;
  for (int b : buckets) if (a.kc <= b) { KCT = b; break; }
  const size_t lds = (size_t)SPR * (16 * KCT + 4) * sizeof(float);
  const long cap = 256;                                   // workgroups resident on the chip
  // equal shares: every workgroup walks ceil(nblk / cap) blocks (a grid of cap would leave a partial last round)
  const long rounds = (nblk + cap - 1) / cap;
  const unsigned grid = (unsigned)((nblk + rounds - 1) / rounds);
  const void* fn = nullptr;
#define SP_CASE(K) case K: fn = nsets == 2 ? (const void*)qtran_state_parts_kernel<K, 2> : (const void*)qtran_state_parts_kernel<K, 1>; break;
  switch (KCT) { SP_CASE(4) SP_CASE(8) SP_CASE(12) SP_CASE(14) SP_CASE(16) SP_CASE(20) SP_CASE(24) }
#undef SP_CASE
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  void* kargs[] = {(void*)&a};
  e = hipLaunchKernel(fn, dim3(grid), dim3(64 * NW), kargs, lds, stream);
  if (e != hipSuccess) return (int)e;
  MARL_CHECK_LAUNCH();
  return 0;
}
}  // namespace

// (S % 4 != 0 is covered when the rows are padded to 16 bytes - ld0 % 4 == 0, as the episode storage is - and the pad is finite)
extern "C" int marl_qtran_state_parts_supported(int S) { return (S >= 1 && S <= 384) ? 1 : 0; }

extern "C" int marl_qtran_state_parts(const marl_src_t* s, long BT, int S, int nsets, const float* W0, long ldw0, const float* b0,
                                      float* sp0, const float* W1, long ldw1, const float* b1, float* sp1, void* stream) {
  if (BT <= 0) return 0;
  if (!marl_qtran_state_parts_supported(S) || nsets < 1 || nsets > 2 || !s || !W0 || !b0 || !sp0) return (int)hipErrorInvalidValue;
  if (nsets == 2 && (!W1 || !b1 || !sp1)) return (int)hipErrorInvalidValue;
  if (!state_src_ok(s, S) || !aligned16(sp0) || !aligned16(b0) || (nsets == 2 && (!aligned16(sp1) || !aligned16(b1)))) return (int)hipErrorInvalidValue;
  SpArgs a = {};
  a.s = state_src(s); a.BT = BT; a.S = S;
  const float* W[2] = {W0, W1}; const long ldw[2] = {ldw0, ldw1}; const float* b[2] = {b0, b1}; float* sp[2] = {sp0, sp1};
  for (int k = 0; k < nsets; ++k) {
    a.ldw[k] = ldw[k]; a.ldo[k] = HD;
    for (int t = 0; t < 4; ++t) { a.W[k][t] = W[k] + 16 * t * ldw[k]; a.b[k][t] = b[k] + 16 * t; a.out[k][t] = sp[k] + 16 * t; a.relu[k][t] = 0; }
  }
  return sp_launch(a, nsets, (hipStream_t)stream);
}

// The state-conditioned bias path of QMixMixer (network/mixer.py:44-47, :69-77) in one pass over s, written straight into the
// hypernet output matrix hy (rows, ldhy) the mixing kernels read:  hy[:, c_b1 : c_b1 + 32] = hyper_b1(s),
// hy[:, c_h : c_h + 32] = relu(hyper_b2.0(s)).  E = qmix_hidden_dim = 32 (two feature tiles each).
extern "C" int marl_qmix_tail_fwd(const marl_src_t* s, long rows, int S, const float* Wb1, long ldb1, const float* bb1,
                                  const float* Wh, long ldwh, const float* bh, float* hy, long ldhy, int c_b1, int c_h,
                                  void* stream) {
  if (rows <= 0) return 0;
  if (!marl_qtran_state_parts_supported(S) || !state_src_ok(s, S) || !Wb1 || !bb1 || !Wh || !bh || !hy) return (int)hipErrorInvalidValue;
  if (ldb1 != ldwh || !aligned16(bb1) || !aligned16(bh) || !aligned16(hy) || ldhy % 4 || c_b1 % 4 || c_h % 4) return (int)hipErrorInvalidValue;
  SpArgs a = {};
  a.s = state_src(s); a.BT = rows; a.S = S;
  a.ldw[0] = ldb1; a.ldo[0] = ldhy;
  for (int t = 0; t < 2; ++t) {
    a.W[0][t] = Wb1 + 16 * t * ldb1; a.b[0][t] = bb1 + 16 * t; a.out[0][t] = hy + c_b1 + 16 * t; a.relu[0][t] = 0;
    a.W[0][2 + t] = Wh + 16 * t * ldwh; a.b[0][2 + t] = bh + 16 * t; a.out[0][2 + t] = hy + c_h + 16 * t; a.relu[0][2 + t] = 1;
  }
  return sp_launch(a, 1, (hipStream_t)stream);
}

namespace {
inline int wg_ts(int S) { return (S + 15) / 16; }
inline int wg_zw(int FT, int smax) { return 3 * 16 * FT + 4 * 64 + 4 + smax; }
}  // namespace

extern "C" int marl_qtran_wgrad_rows_supported(int S, int AE) {
  const int FT = ft_of(AE);
  return (S >= 4 && S % 4 == 0 && S <= 384 && (FT == 4 || FT == 5)) ? 1 : 0;
}
extern "C" size_t marl_qtran_wgrad_rows_workspace(int S, int AE) {
  return (size_t)256 * wg_slab_elems(wg_ts(S), 16 * ft_of(AE)) * sizeof(float);
}

extern "C" int marl_qtran_wgrad_rows(const marl_src_t* s, const float* s1, const float* e2, const float* y1, const float* y2,
                                     const float* d_out, const float* dy1, const float* dy2, const float* de2,
                                     float* d_q0_w, long ld_q0, float* d_q0_b, float* d_q2_w, float* d_q2_b, float* d_q4_w,
                                     float* d_q4_b, float* d_enc2_w, float* ws, size_t ws_bytes, long BT, int S, int AE,
                                     void* stream) {
  if (BT <= 0) return 0;
  if (!marl_qtran_wgrad_rows_supported(S, AE) || ws_bytes < marl_qtran_wgrad_rows_workspace(S, AE)) return (int)hipErrorInvalidValue;
  if (!state_src_ok(s, S) || !aligned16(s1) || !aligned16(e2) || !aligned16(y1) || !aligned16(y2) || !aligned16(dy1) || !aligned16(dy2) ||
      !aligned16(de2) || !aligned16(ws)) return (int)hipErrorInvalidValue;
  const int FT = ft_of(AE), AEP = 16 * FT;
  WgArgs a = {};
  a.s = state_src(s); a.e2 = e2; a.y1 = y1; a.s1 = s1; a.dy1 = dy1; a.dy2 = dy2; a.de2 = de2; a.y2 = y2; a.d_out = d_out;
  a.slab = ws; a.BT = BT; a.S = S; a.AE = AE; a.TS = wg_ts(S);
  a.ntiles = 4 * a.TS + 4 * FT + 16 + FT * FT;
  const long nblk = (BT + WR - 1) / WR;
  const unsigned grid = (unsigned)(nblk < 256 ? nblk : 256);
  const bool big = S > 224;
  const size_t lds = (size_t)WR * wg_zw(FT, big ? 384 : 224) * sizeof(float);
  const void* fn = FT == 5 ? (big ? (const void*)qtran_wgrad_rows_kernel<5, 384> : (const void*)qtran_wgrad_rows_kernel<5, 224>)
                           : (big ? (const void*)qtran_wgrad_rows_kernel<4, 384> : (const void*)qtran_wgrad_rows_kernel<4, 224>);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  void* kargs[] = {(void*)&a};
  e = hipLaunchKernel(fn, dim3(grid), dim3(64 * NW), kargs, lds, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  MARL_CHECK_LAUNCH();
  WgRedArgs r = {};
  r.slab = ws; r.dWq1 = d_q0_w; r.ldq1 = ld_q0; r.dbq1 = d_q0_b; r.dWq2 = d_q2_w; r.dbq2 = d_q2_b; r.dwq3 = d_q4_w; r.dbq3 = d_q4_b;
  r.dWe2 = d_enc2_w; r.nwg = (int)grid; r.S = S; r.AE = AE; r.AEP = AEP; r.TS = a.TS;
  const int total = wg_slab_elems(a.TS, AEP);
  hipLaunchKernelGGL(qtran_wgrad_reduce_kernel, dim3((total + 63) / 64), dim3(256), 0, (hipStream_t)stream, r);
  MARL_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t marl_qtran_bwd_workspace(long BT, int AE) {
  const int AEP = 16 * ft_of(AE);
  return ((size_t)BT * AEP + (size_t)256 * AEP + (size_t)256 * AEP * KW) * sizeof(float);
}

extern "C" int marl_qtran_head_bwd(const marl_qtran_weights_t* w, const float* hidden, const int* u, const float* d_out,
                                   const float* y1, const float* y2, float* dy1, float* dy2, float* de2, float* dhidden,
                                   int accumulate, float* d_enc0_w, float* d_enc0_b, float* d_enc2_b, float* ws,
                                   size_t ws_bytes, long BT, int N, int A, int AE, void* stream) {
  if (BT <= 0) return 0;
  if (!supported(N, A, AE) || (A > 0) != (u != nullptr)) return (int)hipErrorInvalidValue;
  if (ws_bytes < marl_qtran_bwd_workspace(BT, AE)) return (int)hipErrorInvalidValue;
  if (!aligned16(hidden) || !aligned16(dhidden) || !aligned16(y1) || !aligned16(y2) || !aligned16(dy1) || !aligned16(dy2) ||
      !aligned16(de2) || !aligned16(ws)) return (int)hipErrorInvalidValue;
  QtArgs a = {};
  fill_w(a, w);
  const int FT = ft_of(AE), AEP = 16 * FT;
  a.hidden = hidden; a.u = u; a.d_out = d_out; a.y1 = const_cast<float*>(y1); a.y2 = const_cast<float*>(y2);
  a.dy1 = dy1; a.dy2 = dy2; a.de2 = de2; a.dhidden = dhidden; a.accumulate = accumulate;
  a.ds1 = ws; a.slab1 = ws + (size_t)BT * AEP; a.slab2 = a.slab1 + (size_t)256 * AEP;
  a.BT = BT; a.N = N; a.A = A; a.AE = AE;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = grid_for(BT);
  int rc = FT == 5 ? launch(qtran_bwd_rows_kernel<5>, a, grid, rows_lds(5), s) : launch(qtran_bwd_rows_kernel<4>, a, grid, rows_lds(4), s);
  if (rc) return rc;
  if (A > 0) rc = launch(qtran_bwd_agents_kernel<5, true>, a, grid, agents_lds(5, true), s);
  else rc = launch(qtran_bwd_agents_kernel<4, false>, a, grid, agents_lds(4, false), s);
  if (rc) return rc;
  QtRedArgs r;
  r.slab1 = a.slab1; r.slab2 = a.slab2; r.dWe1 = d_enc0_w; r.dbe1 = d_enc0_b; r.dbe2 = d_enc2_b;
  r.nwg = (int)grid; r.AE = AE; r.AEP = AEP; r.A = A; r.N = N;
  const int total = AEP * KW + AEP;
  hipLaunchKernelGGL(qtran_reduce_kernel, dim3((total + 63) / 64), dim3(256), 0, s, r);
  MARL_CHECK_LAUNCH();
  return 0;
}
