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
template <int FT, bool HOT>
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
    f32x4 spv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) spv[t] = *reinterpret_cast<const f32x4*>(a.sp + bt * HD + 16 * t + 4 * q);
    f32x4 s1[FT];
#pragma unroll
    for (int t = 0; t < FT; ++t) s1[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // one agent: e1 = W_e1h h + (b_e1 + W_e1[:, 64 + u]);  s1 += relu(e1)
    auto step = [&](const f32x4 (&xv)[4], int uu) __attribute__((always_inline)) {
      f32x4 acc[FT];
      const int tr = HOT ? ((uu >= 0 && uu < 16) ? uu : 16) : 0;
      const float* tp = tab + tr * TS + 4 * q;
#pragma unroll
      for (int t = 0; t < FT; ++t) acc[t] = *reinterpret_cast<const f32x4*>(tp + 16 * t);
      layer<FT, 4>(acc, W1s, xv, lane);
#pragma unroll
      for (int t = 0; t < FT; ++t) s1[t] += relu4(acc[t]);
    };
    auto load = [&](f32x4 (&xv)[4], int& uu, int n) __attribute__((always_inline)) {
#pragma unroll
      for (int c = 0; c < 4; ++c) xv[c] = *reinterpret_cast<const f32x4*>(hrow + (long)n * HD + 16 * c);
      uu = HOT ? urow[n] : -1;
    };
    // two named register sets (no set-to-set copies: the next agent's loads are in flight while this one computes)
    f32x4 xA[4], xB[4];
    int uA, uB;
    load(xA, uA, 0);
    for (int n = 0; n < N; n += 2) {
      load(xB, uB, n + 1 < N ? n + 1 : N - 1);
      step(xA, uA);
      if (n + 1 < N) {
        load(xA, uA, n + 2 < N ? n + 2 : N - 1);
        step(xB, uB);
      }
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
  if (e < n2) {
    for (int w = sg; w < a.nwg; w += 4) s += a.slab2[(long)w * n2 + e];
  } else if (e < n2 + a.AEP) {
    for (int w = sg; w < a.nwg; w += 4) s += a.slab1[(long)w * a.AEP + (e - n2)];
  }
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
