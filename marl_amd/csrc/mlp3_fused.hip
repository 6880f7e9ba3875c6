// Fused three-layer heads  y = W3 relu(W2 relu(W1 x + b1) + b2) + b3  (hidden width 64) for gfx950, forward and
// backward, batched over `groups` equally shaped heads: the key / agents / action extractors of QPLEX's
// lambda-net (reference network/mixer.py:117-145, evaluated at :155-169) - 30 such heads per mixer call.
// Composed from marl_linear / marl_linear_wgrad the two 64-wide hidden activations of every head travel
// through HBM four times per update (3.8 GB per layer at 4096 envs); here they never leave the CU.
//
// "Transposed" formulation: every layer is computed as  out^T[feature][row] = W[feature][k] * in^T[k][row],
// i.e. the WEIGHTS are the MFMA A operand (fragments staged once per workgroup in LDS) and the activations the
// B operand.  With the K-permutation of common.h the accumulator tile c of one layer (lane (q,m), register j:
// feature 16c+4q+j of row m) IS the B fragment of k-chunk c of the next layer, so a wave chains the three
// layers of a 16-row tile in registers: no LDS round trip, no shuffles, no barriers in the forward kernel.
//
//   forward : 8 waves per workgroup, each walks its own 16-row tiles; x is prefetched one tile ahead in registers.
//   backward: 4 waves; per 64-row iteration each wave recomputes h1, h2 of its tile and forms dh2, dh1 (same
//             chaining with W3^T, W2^T fragments); the weight gradients  dW_l = dh_l^T a_{l-1}  reduce over ROWS,
//             so the operands are exchanged through a [feature][row] LDS stage tile (two stages sharing one
//             buffer) and wave w accumulates rows [16w,16w+16) of dW1 / dW2 (+ a quarter of dW3) in registers
//             for the whole stripe; one slab per workgroup, fixed-order reduce (bitwise reproducible).
// x is a virtual concat [dense0 | dense1 | one-hot blocks] (row remap / episode map allowed): chunks inside the
// 16-byte aligned part of dense0 are one 16-byte load per lane, the rest (segment tails, one-hot columns) are one
// raw 32-bit load per element from a per-lane selected address described by an LDS table built once.
#include "common.h"
#include "../../include/marl_hip.h"

#include "mlp3_common.h"

namespace {

// layer 1 of one tile in registers (transposed form); h1 comes out post-relu
template <int KC>
__device__ __forceinline__ void fwd1(const f32x4 (&xv)[KC], const float* W1s, const f32x4 (&b1v)[4], f32x4 (&h1)[4], int lane) {
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = b1v[t];
  // weight fragments one chunk ahead in two named sets that alternate (no set-to-set copies: 16 moves per chunk on a SIMD
  // where every vector instruction is matrix time lost); the scheduling barrier keeps the compiler from hoisting every
  // LDS read of the layer to the top (that cost > 256 registers and spilled)
  f32x4 wA[4], wB[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) wA[t] = *reinterpret_cast<const f32x4*>(W1s + ((t * KC) * 64 + lane) * 4);
#pragma unroll
  for (int c = 0; c < KC; c += 2) {
    if (c + 1 < KC) {
#pragma unroll
      for (int t = 0; t < 4; ++t) wB[t] = *reinterpret_cast<const f32x4*>(W1s + ((t * KC + c + 1) * 64 + lane) * 4);
    }
    mfma16x4_il4(wA[0], xv[c], acc[0], wA[1], xv[c], acc[1], wA[2], xv[c], acc[2], wA[3], xv[c], acc[3]);
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 < KC) {
      if (c + 2 < KC) {
#pragma unroll
        for (int t = 0; t < 4; ++t) wA[t] = *reinterpret_cast<const f32x4*>(W1s + ((t * KC + c + 2) * 64 + lane) * 4);
      }
      mfma16x4_il4(wB[0], xv[c + 1], acc[0], wB[1], xv[c + 1], acc[1], wB[2], xv[c + 1], acc[2], wB[3], xv[c + 1], acc[3]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) h1[t] = relu4(acc[t]);
}
// layer 2: the accumulator tiles of layer 1 are the B fragments
__device__ __forceinline__ void fwd2(const f32x4 (&h1)[4], const float* W2s, const f32x4 (&b2v)[4], f32x4 (&h2)[4], int lane) {
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = b2v[t];
  f32x4 wA[4], wB[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) wA[t] = *reinterpret_cast<const f32x4*>(W2s + ((t * 4) * 64 + lane) * 4);
#pragma unroll
  for (int c = 0; c < 4; c += 2) {
#pragma unroll
    for (int t = 0; t < 4; ++t) wB[t] = *reinterpret_cast<const f32x4*>(W2s + ((t * 4 + c + 1) * 64 + lane) * 4);
    mfma16x4_il4(wA[0], h1[c], acc[0], wA[1], h1[c], acc[1], wA[2], h1[c], acc[2], wA[3], h1[c], acc[3]);
    __builtin_amdgcn_sched_barrier(0);
    if (c + 2 < 4) {
#pragma unroll
      for (int t = 0; t < 4; ++t) wA[t] = *reinterpret_cast<const f32x4*>(W2s + ((t * 4 + c + 2) * 64 + lane) * 4);
    }
    mfma16x4_il4(wB[0], h1[c + 1], acc[0], wB[1], h1[c + 1], acc[1], wB[2], h1[c + 1], acc[2], wB[3], h1[c + 1], acc[3]);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) h2[t] = relu4(acc[t]);
}

// ------------------------------------------------------------------------------------------------- forward
// THREE = false: two-layer heads  y = W3 relu(W1 x + b1) + b3  (W2 == NULL; QPLEX transformation nets)
// KC > 12 (K1 up to 512: QPLEX on MMM2-sized maps): x of a tile is 64-128 registers and W1 64-128 KB of LDS - one workgroup per CU
// WIDE: N3 up to 160 outputs in tiles of 16 (two-layer hypernet heads S -> 64 -> N*E of QMixMixer, mixer.py:36-43)
template <int KC, bool THREE, int CFT = -1, bool WIDE = false>
__global__ __launch_bounds__(64 * FNW, (KC > 12 || WIDE) ? 1 : 2) void mlp3_fwd_kernel(Mlp3Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int stripe, g;
  if (!wg_map(a.groups, a.nst, stripe, g)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, m = lane & 15;
  float* W1s = smem;                              // [4][KC][64] f32x4
  float* W2s = W1s + 4 * KC * 256;                // [4][4][64] f32x4
  float* W3s = W2s + 16 * 256;                    // [NT][4][64] f32x4 (rows >= N3 zero)
  float* b3s = W3s + (WIDE ? NTW : 1) * 4 * 256;  // WIDE: [16 * NTW] output biases
  int* tab = reinterpret_cast<int*>(b3s + (WIDE ? 16 * NTW : 0));
  const float* W1 = a.W1 + g * a.gs_w1;
  const float* W2 = a.W2 + g * a.gs_w2;
  const float* W3 = a.W3 + g * a.gs_w3;
  stage_w<4 * KC * 64, 64 * FNW>(W1s, W1, a.K1, HD, a.K1, KC, a.x.k0, a.kpad);
  if (THREE) stage_w<16 * 64, 64 * FNW>(W2s, W2, HD, HD, HD, 4);
  stage_w<(WIDE ? NTW : 1) * 4 * 64, 64 * FNW>(W3s, W3, HD, a.N3, HD, 4);
  if (WIDE)
    for (int e = tid; e < 16 * NTW; e += 64 * FNW) b3s[e] = e < a.N3 ? a.b3[g * a.gs_b3 + e] : 0.f;
  build_tab(tab, a.x, a.KV, a.kpad, a.CF, KC, 64 * FNW);
  f32x4 b1v[4], b2v[4], b3v;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    b1v[t] = *reinterpret_cast<const f32x4*>(a.b1 + g * a.gs_b1 + 16 * t + 4 * q);
    if (THREE) b2v[t] = *reinterpret_cast<const f32x4*>(a.b2 + g * a.gs_b2 + 16 * t + 4 * q);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) b3v[i] = 4 * q + i < a.N3 ? a.b3[g * a.gs_b3 + 4 * q + i] : 0.f;
  __syncthreads();

  const long tiles = (a.M + 15) / 16;
  const long per = (tiles + a.nst - 1) / a.nst;
  const long t_begin = (long)stripe * per;
  long t_end = t_begin + per; if (t_end > tiles) t_end = tiles;
  float* Y = a.Y + g * a.gs_y;

  // one register set for x: the next tile's loads are issued as soon as layer 1 has consumed the current one
  // (unconditionally - the last iteration re-reads its own tile) and fly during layers 2 / 3 and the partner
  // wave's math
  f32x4 xv[KC];
  long tile = t_begin + wave;
  if (tile >= t_end) return;
  XRow xr = x_row(a.x, tile * 16 + m, a.M);
  x_issue<KC, CFT>(xv, a.x, xr, tab, a.CF, lane);
  for (; tile < t_end; tile += FNW) {
    x_finish<KC, CFT>(xv, xr, tab, a.CF, lane);
    const bool live = (xr.flags & 1) != 0;
    float* y = Y + xr.rowc * a.ldy + 4 * q;
    f32x4 h1[4], h2[4];
    fwd1<KC>(xv, W1s, b1v, h1, lane);
    {
      const long nt = tile + FNW < t_end ? tile + FNW : tile;
      xr = x_row(a.x, nt * 16 + m, a.M);
      x_issue<KC, CFT>(xv, a.x, xr, tab, a.CF, lane);
    }
    if (THREE) fwd2(h1, W2s, b2v, h2, lane);
    else {
#pragma unroll
      for (int t = 0; t < 4; ++t) h2[t] = h1[t];
    }
    if (a.hs) {
      // kept for the backward as the fragments it consumes (1 KB per store instruction; streaming - x lives in this XCD's L2 for
      // the other heads of the stripe and these 8 KB per tile and head must not push it out)
      constexpr int NP = THREE ? 8 : 4;
      f32x4* hp = reinterpret_cast<f32x4*>(a.hs) + (((long)g * tiles + tile) * NP) * 64 + lane;
#pragma unroll
      for (int c = 0; c < 4; ++c) KEEP_ST(h1[c], hp + c * 64);
      if (THREE) {
#pragma unroll
        for (int c = 0; c < 4; ++c) KEEP_ST(h2[c], hp + (4 + c) * 64);
      }
    }
    if (WIDE) {
      // output tiles two at a time (independent accumulators); a tile's four outputs of a row are one 16-byte store
      for (int nt = 0; nt < a.n3t; nt += 2) {
        const bool two = nt + 1 < a.n3t;
        const int n1 = two ? nt + 1 : nt;
        f32x4 acc0 = *reinterpret_cast<const f32x4*>(b3s + 16 * nt + 4 * q);
        f32x4 acc1 = *reinterpret_cast<const f32x4*>(b3s + 16 * n1 + 4 * q);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(W3s + ((nt * 4 + c) * 64 + lane) * 4);
          const f32x4 w1 = *reinterpret_cast<const f32x4*>(W3s + ((n1 * 4 + c) * 64 + lane) * 4);
          mfma16x4_il2(w0, h2[c], acc0, w1, h2[c], acc1);
        }
        if (live) {
          if (a.yvec) {
            if (16 * nt + 4 * q < a.N3) *reinterpret_cast<f32x4*>(y + 16 * nt) = acc0;
            if (two && 16 * n1 + 4 * q < a.N3) *reinterpret_cast<f32x4*>(y + 16 * n1) = acc1;
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if (16 * nt + 4 * q + i < a.N3) y[16 * nt + i] = acc0[i];
              if (two && 16 * n1 + 4 * q + i < a.N3) y[16 * n1 + i] = acc1[i];
            }
          }
        }
      }
    } else {
      f32x4 acc = b3v;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(W3s + (c * 64 + lane) * 4);
        acc = mfma16x4(w, h2[c], acc);
      }
      if (live) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (4 * q + i < a.N3) y[i] = acc[i];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------- backward
// T-layout register tile (lane (q,m), tile t, register i = feature 16t+4q+i of row m) -> stage[feature][16*wave + m]
__device__ __forceinline__ void stash4(float* st, const f32x4 (&v)[4], int wave, int q, int m) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) st[(16 * t + 4 * q + i) * RS + 16 * wave + m] = v[t][i];
}

// LOAD: h1 / h2 come from the forward's `hs` planes (one iteration ahead in a second register set) instead of being recomputed
// WIDE: N3 up to 160 (kept activations only); dY / dW3 in tiles of 16 outputs
template <int KC, bool THREE, int CFT = -1, bool LOAD = false, bool WIDE = false>
__global__ __launch_bounds__(64 * BNW, (LOAD && !WIDE && KC <= MLP3_BWD2_KC) ? 2 : 1) void mlp3_bwd_kernel(Mlp3Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int stripe, g;
  if (!wg_map(a.groups, a.nst, stripe, g)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, m = lane & 15;
  // x^T is staged KH chunks at a time (KC > 24: two passes of 16 - 32 chunks of 64 rows are 139 KB on their own)
  // (two workgroups per CU: 80 KB each - more than 8 chunks go through the stage 6 at a time)
  constexpr bool TWO = LOAD && !WIDE && KC <= MLP3_BWD2_KC;
  constexpr int KH = KC > 24 ? 16 : (TWO && KC > 8) ? 6 : KC, NH = (KC + KH - 1) / KH;
  static_assert(LOAD || KC <= 12, "K1 > 192 only with kept activations (W1 does not fit beside the stage)");
  static_assert(!WIDE || LOAD, "wide outputs only with kept activations");
  constexpr int NT = WIDE ? NTW : 1;
  constexpr int S2H = THREE ? 192 : 64;                               // stage 2: [h1 | dh2 |] h2, then dY^T
  constexpr int SF2 = S2H + 16 * NT;
  constexpr int SF = (16 * KH + HD) > SF2 ? (16 * KH + HD) : SF2;    // stage features: max(x + dh1, h1 + dh2 + h2 + dY)
  float* W1s = smem;                              // [4][KC][64] f32x4      (recomputing variant only)
  float* W2s = W1s + (LOAD ? 0 : 4 * KC * 256);   // [4][4][64] f32x4       (recomputing variant only)
  float* W2Ts = W2s + (LOAD ? 0 : 16 * 256);      // [4][4][64] f32x4
  float* W3Ts = W2Ts + (THREE || !WIDE ? 16 * 256 : 0);   // [4 t][4 j][64]: W3[4j + q][16t + m]; WIDE: [NT][4 t][64] f32x4 = W3[16nt + 4q + i][16t + m]
  float* stage = W3Ts + (WIDE ? NT * 4 * 256 : 16 * 64);  // [SF][RS]
  int* tab = reinterpret_cast<int*>(stage + SF * RS);
  const float* W1 = a.W1 + g * a.gs_w1;
  const float* W2 = a.W2 + g * a.gs_w2;
  const float* W3 = a.W3 + g * a.gs_w3;
  if (!LOAD) stage_w<4 * KC * 64, 64 * BNW>(W1s, W1, a.K1, HD, a.K1, KC, a.x.k0, a.kpad);
  if (THREE) {
    if (!LOAD) stage_w<16 * 64, 64 * BNW>(W2s, W2, HD, HD, HD, 4);
    stage_wT<64 * BNW>(W2Ts, W2, HD);
  }
  if (WIDE) {
    for (int e = tid; e < a.n3t * 4 * 64; e += 64 * BNW) {
      const int l = e & 63, t = (e >> 6) & 3, nt = e >> 8;
      f32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n3 = 16 * nt + 4 * (l >> 4) + i;
        v[i] = n3 < a.N3 ? W3[(long)n3 * HD + 16 * t + (l & 15)] : 0.f;
      }
      *reinterpret_cast<f32x4*>(W3Ts + (long)e * 4) = v;
    }
  } else {
    for (int e = tid; e < 16 * 64; e += 64 * BNW) {
      const int l = e & 63, tj = e >> 6, t = tj >> 2, j = tj & 3;
      const int n3 = 4 * j + (l >> 4);
      W3Ts[e] = n3 < a.N3 ? W3[(long)n3 * HD + 16 * t + (l & 15)] : 0.f;
    }
  }
  build_tab(tab, a.x, a.KV, a.kpad, a.CF, KC, 64 * BNW);
  f32x4 b1v[4], b2v[4];
  if (!LOAD) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      b1v[t] = *reinterpret_cast<const f32x4*>(a.b1 + g * a.gs_b1 + 16 * t + 4 * q);
      if (THREE) b2v[t] = *reinterpret_cast<const f32x4*>(a.b2 + g * a.gs_b2 + 16 * t + 4 * q);
    }
  }
  __syncthreads();

  const long its = (a.M + 63) / 64;
  const long per = (its + a.nst - 1) / a.nst;
  const long i_begin = (long)stripe * per;
  long i_end = i_begin + per; if (i_end > its) i_end = its;
  const float* dY = a.Y + g * a.gs_y;

  f32x4 dW1[KC], dW2[4], dW3 = {0.f, 0.f, 0.f, 0.f};
  float bs1 = 0.f, bs2 = 0.f, bs3 = 0.f;
  f32x4 dW3w[NT];                        // WIDE: [16 outputs of tile nt] x [features 16w .. 16w+15]
  float bs3w[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { dW3w[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; bs3w[nt] = 0.f; }
#pragma unroll
  for (int c = 0; c < KC; ++c) dW1[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 4; ++c) dW2[c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float* xT = stage;                     // stage 1: [16*KH][RS] x^T (KH chunks at a time), then [64][RS] dh1^T
  float* dh1T = stage + 16 * KH * RS;
  float* h1T = stage;                    // stage 2: h1^T, dh2^T, h2^T [64][RS] each, dY^T [16][RS]
  float* dh2T = stage + 64 * RS;
  float* h2T = stage + (S2H - 64) * RS;
  float* dYT = stage + S2H * RS;

  f32x4 xv[KC];
  XRow xr;
  f32x4 hn1[4], hn2[4];                  // LOAD: the next iteration's kept activations and dY elements
  float dyn[4];
  f32x4 dynw[NT];                        // WIDE: dY[row m][16nt + 4q .. +3]
  const long tiles = (a.M + 15) / 16;
  constexpr int NP = THREE ? 8 : 4;
  auto issue_kept = [&](long it_, const XRow& xk) __attribute__((always_inline)) {
    const bool lv = (xk.flags & 1) != 0;
    if (WIDE) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        dynw[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (nt < a.n3t && lv && 16 * nt + 4 * q < a.N3) dynw[nt] = *reinterpret_cast<const f32x4*>(dY + xk.rowc * a.ldy + 16 * nt + 4 * q);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n3 = 4 * j + q;
        dyn[j] = (lv && n3 < a.N3) ? dY[xk.rowc * a.ldy + n3] : 0.f;
      }
    }
    long tl = it_ * BNW + wave; if (tl > tiles - 1) tl = tiles - 1;      // (a tile past the end multiplies zero gradients)
    const f32x4* hp = reinterpret_cast<const f32x4*>(a.hs) + (((long)g * tiles + tl) * NP) * 64 + lane;
#pragma unroll
    for (int c = 0; c < 4; ++c) hn1[c] = KEEP_LD(hp + c * 64);
    if (THREE) {
#pragma unroll
      for (int c = 0; c < 4; ++c) hn2[c] = KEEP_LD(hp + (4 + c) * 64);
    }
  };
  if (i_begin < i_end) {
    xr = x_row(a.x, (i_begin * BNW + wave) * 16 + m, a.M);
    x_issue<KC, CFT>(xv, a.x, xr, tab, a.CF, lane);
    if (LOAD) issue_kept(i_begin, xr);
  }
  ST_DECL(12);
  for (long it = i_begin; it < i_end; ++it) {
    // ---------------- phase A: this wave's 16-row tile, all in registers
    x_finish<KC, CFT>(xv, xr, tab, a.CF, lane);
    const bool live = (xr.flags & 1) != 0;
    const long rowc = xr.rowc;
    float dy[4];                                   // dY[row m][4j + q]
    f32x4 dyw[NT];
    if (WIDE) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) dyw[nt] = dynw[nt];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n3 = 4 * j + q;
        if (LOAD) dy[j] = dyn[j];
        else dy[j] = (live && n3 < a.N3) ? dY[rowc * a.ldy + n3] : 0.f;
      }
    }
#pragma unroll
    for (int c = 0; c < KH; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) xT[(16 * c + 4 * q + i) * RS + 16 * wave + m] = xv[c][i];
    f32x4 h1[4], h2[4], dh2[4], dh1[4];
    ST_MARK(0);
    if (LOAD) {
#pragma unroll
      for (int t = 0; t < 4; ++t) { h1[t] = hn1[t]; h2[t] = THREE ? hn2[t] : hn1[t]; }
    } else {
      fwd1<KC>(xv, W1s, b1v, h1, lane);
    }
    // x is consumed (NH == 1): start the loads of the next iteration's tile (unconditional; the last one re-reads its own)
    {
      const long ni = it + 1 < i_end ? it + 1 : it;
      xr = x_row(a.x, (ni * BNW + wave) * 16 + m, a.M);
      if (NH == 1) x_issue<KC, CFT>(xv, a.x, xr, tab, a.CF, lane);
      if (LOAD) issue_kept(ni, xr);
    }
    ST_MARK(1);
    if (!LOAD) {
      if (THREE) fwd2(h1, W2s, b2v, h2, lane);
      else {
#pragma unroll
        for (int t = 0; t < 4; ++t) h2[t] = h1[t];
      }
    }
    ST_MARK(2);
    {
      f32x4 acc[4];                      // the four feature tiles round robin (independent accumulators, see common.h)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (WIDE) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          if (nt < a.n3t) {
            f32x4 w[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) w[t] = *reinterpret_cast<const f32x4*>(W3Ts + ((nt * 4 + t) * 64 + lane) * 4);
            mfma16x4_il4(w[0], dyw[nt], acc[0], w[1], dyw[nt], acc[1], w[2], dyw[nt], acc[2], w[3], dyw[nt], acc[3]);
            __builtin_amdgcn_sched_barrier(0);
          }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (4 * j < a.N3) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma16(W3Ts[(t * 4 + j) * 64 + lane], dy[j], acc[t]);      // heads n3 = 4j + q
          }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) dh2[t][i] = h2[t][i] > 0.f ? acc[t][i] : 0.f;
    }
    ST_MARK(3);
    if (THREE) {
      f32x4 acc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        f32x4 w[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) w[t] = *reinterpret_cast<const f32x4*>(W2Ts + ((t * 4 + c) * 64 + lane) * 4);
        mfma16x4_il4(w[0], dh2[c], acc[0], w[1], dh2[c], acc[1], w[2], dh2[c], acc[2], w[3], dh2[c], acc[3]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) dh1[t][i] = h1[t][i] > 0.f ? acc[t][i] : 0.f;
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) dh1[t] = dh2[t];
    }
    stash4(dh1T, dh1, wave, q, m);
    ST_MARK(4);
    WG_BARRIER();
    ST_MARK(5);
    // ---------------- phase B1: dW1 rows [16w, 16w+16) over the 64 rows of the iteration
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {
      if (hf > 0) {          // next KH chunks of x^T into the same stage rows (dh1^T stays where it is)
        WG_BARRIER();
#pragma unroll
        for (int c = 0; c < KH; ++c)
          if (hf * KH + c < KC) {
#pragma unroll
            for (int i = 0; i < 4; ++i) xT[(16 * c + 4 * q + i) * RS + 16 * wave + m] = xv[hf * KH + c][i];
          }
        if (hf == NH - 1) x_issue<KC, CFT>(xv, a.x, xr, tab, a.CF, lane);      // x is consumed: the next iteration's tile
        WG_BARRIER();
      }
#pragma unroll
      for (int rt = 0; rt < BNW; ++rt) {
        const f32x4 af = *reinterpret_cast<const f32x4*>(dh1T + (16 * wave + m) * RS + 16 * rt + 4 * q);
        if (hf == 0) bs1 += (af[0] + af[1]) + (af[2] + af[3]);
#pragma unroll
        for (int c = 0; c < KH; c += 4) {        // up to four k tiles round robin (independent accumulators)
          const int cg = hf * KH + c;
          const int lim = KC - hf * KH < KH ? KC - hf * KH : KH;           // chunks of this pass (a constant once unrolled)
          if (c >= lim) continue;
          f32x4 bf[4];
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (c + u < lim) bf[u] = *reinterpret_cast<const f32x4*>(xT + (16 * (c + u) + m) * RS + 16 * rt + 4 * q);
          if (c + 3 < lim) mfma16x4_il4(af, bf[0], dW1[cg], af, bf[1], dW1[cg + 1], af, bf[2], dW1[cg + 2], af, bf[3], dW1[cg + 3]);
          else if (c + 2 < lim) mfma16x4_il3(af, bf[0], dW1[cg], af, bf[1], dW1[cg + 1], af, bf[2], dW1[cg + 2]);
          else if (c + 1 < lim) mfma16x4_il2(af, bf[0], dW1[cg], af, bf[1], dW1[cg + 1]);
          else dW1[cg] = mfma16x4(af, bf[0], dW1[cg]);
        }
      }
    }
    ST_MARK(6);
    WG_BARRIER();
    ST_MARK(7);
    // ---------------- stage 2 operands
    if (THREE) {
      stash4(h1T, h1, wave, q, m);
      stash4(dh2T, dh2, wave, q, m);
    }
    stash4(h2T, h2, wave, q, m);
    if (WIDE) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        if (nt < a.n3t) {
#pragma unroll
          for (int i = 0; i < 4; ++i) dYT[(16 * nt + 4 * q + i) * RS + 16 * wave + m] = dyw[nt][i];
        }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) dYT[(4 * j + q) * RS + 16 * wave + m] = dy[j];
    }
    ST_MARK(8);
    WG_BARRIER();
    ST_MARK(9);
    // ---------------- phase B2: dW2 rows [16w, 16w+16), dW3 columns [16w, 16w+16)
#pragma unroll
    for (int rt = 0; rt < BNW; ++rt) {
      if (THREE) {
        const f32x4 af = *reinterpret_cast<const f32x4*>(dh2T + (16 * wave + m) * RS + 16 * rt + 4 * q);
        bs2 += (af[0] + af[1]) + (af[2] + af[3]);
        f32x4 bf[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) bf[c] = *reinterpret_cast<const f32x4*>(h1T + (16 * c + m) * RS + 16 * rt + 4 * q);
        mfma16x4_il4(af, bf[0], dW2[0], af, bf[1], dW2[1], af, bf[2], dW2[2], af, bf[3], dW2[3]);
      }
      const f32x4 bh = *reinterpret_cast<const f32x4*>(h2T + (16 * wave + m) * RS + 16 * rt + 4 * q);
      if (WIDE) {
#pragma unroll
        for (int nt = 0; nt < NT; nt += 2)
          if (nt < a.n3t) {
            const int n1 = nt + 1 < a.n3t ? nt + 1 : nt;
            const f32x4 ay0 = *reinterpret_cast<const f32x4*>(dYT + (16 * nt + m) * RS + 16 * rt + 4 * q);
            const f32x4 ay1 = *reinterpret_cast<const f32x4*>(dYT + (16 * n1 + m) * RS + 16 * rt + 4 * q);
            bs3w[nt] += (ay0[0] + ay0[1]) + (ay0[2] + ay0[3]);
            if (nt + 1 < NT) {
              if (nt + 1 < a.n3t) {
                bs3w[nt + 1] += (ay1[0] + ay1[1]) + (ay1[2] + ay1[3]);
                mfma16x4_il2(ay0, bh, dW3w[nt], ay1, bh, dW3w[nt + 1]);
              } else {
                dW3w[nt] = mfma16x4(ay0, bh, dW3w[nt]);
              }
            } else {
              dW3w[nt] = mfma16x4(ay0, bh, dW3w[nt]);
            }
          }
      } else {
        const f32x4 ay = *reinterpret_cast<const f32x4*>(dYT + m * RS + 16 * rt + 4 * q);
        bs3 += (ay[0] + ay[1]) + (ay[2] + ay[3]);
        dW3 = mfma16x4(ay, bh, dW3);
      }
    }
    ST_MARK(10);
    WG_BARRIER();
    ST_MARK(11);
  }
  if (KC == 8 && THREE) { ST_DUMP(12); }      // (diagnostic build: the key / agents extractor launches)

  // ---------------- slab: [dW1 64 x (K1+1) | dW2 64 x 65 | dW3 16 x 65], bias gradient in the last column
  bs1 += __shfl_xor(bs1, 16, 64); bs1 += __shfl_xor(bs1, 32, 64);
  bs2 += __shfl_xor(bs2, 16, 64); bs2 += __shfl_xor(bs2, 32, 64);
  bs3 += __shfl_xor(bs3, 16, 64); bs3 += __shfl_xor(bs3, 32, 64);
  const int K1x = a.K1 + 1;
  float* s1 = a.ws + ((long)stripe * a.groups + g) * mlp3_slab_floats(a.K1, a.N3);
  float* s2 = s1 + (long)HD * K1x;
  float* s3 = s2 + (long)HD * (HD + 1);
#pragma unroll
  for (int c = 0; c < KC; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = 16 * wave + 4 * q + i, k = vcol(16 * c + m, a.x.k0, a.kpad);      // pad columns of the K axis hold nothing
      if (k >= 0 && k < a.K1) s1[(long)n * K1x + k] = dW1[c][i];
    }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) s2[(16 * wave + 4 * q + i) * (HD + 1) + 16 * c + m] = dW2[c][i];
  if (WIDE) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      if (nt < a.n3t) {
        float b = bs3w[nt];
        b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) s3[(16 * nt + 4 * q + i) * (HD + 1) + 16 * wave + m] = dW3w[nt][i];
        if (q == 0 && wave == 0) s3[(16 * nt + m) * (HD + 1) + HD] = b;
      }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) s3[(4 * q + i) * (HD + 1) + 16 * wave + m] = dW3[i];
  }
  if (q == 0) {
    s1[(long)(16 * wave + m) * K1x + a.K1] = bs1;
    s2[(16 * wave + m) * (HD + 1) + HD] = bs2;
    if (wave == 0 && !WIDE) s3[m * (HD + 1) + HD] = bs3;
  }
}

}  // namespace

ST_DEFINE_SETTER(marl_debug_stamps_mlp3)

extern "C" int marl_mlp3_supported(const marl_src_t* x, int K1, int H1, int H2, int N3, int groups) {
  if (H1 != HD || (H2 != HD && H2 != 0) || N3 < 1 || N3 > 16 * NTW || groups < 1 || K1 < 1) return 0;
  const bool wide = N3 > 16;              // two-layer heads with up to 160 outputs in tiles of 16 (kept activations only)
  if (wide && (H2 != 0 || N3 % 4)) return 0;
  if (!x->p0 || x->k0 < 4 || x->m0 || x->nid) return 0;
  if (x->nhot && (x->hot_w < 1 || x->hot_w >= 16384 || x->nhot >= 8192)) return 0;
  if (x->k0 >= 16384 || x->k1 >= 16384) return 0;
  if (x->k1 % 4) return 0;               // a lane's four consecutive columns come from one segment (dense0 is padded by the kernels)
  const int KC = kc_bucket(K1 + kpad_of(x));
  if (KC > 32) return 0;
  if (wide && KC != 8 && KC != 16 && KC != 24) return 0;      // instantiated: state widths of 2s3z / 3s5z / MMM2-sized maps
  const int CF = lead_chunks(x);
  return bwd_lds(KC, CF, KC > 12 || wide, wide, H2 != 0) <= 160 * 1024 && fwd_lds(KC, CF, wide) <= 160 * 1024;
}

extern "C" int marl_mlp3_needs_kept(const marl_src_t* x, int K1, int N3) { return kc_bucket(K1 + kpad_of(x)) > 12 || N3 > 16; }

extern "C" size_t marl_mlp3_save_floats(long M, int three, int groups) {
  return M <= 0 ? 0 : (size_t)groups * (size_t)((M + 15) / 16) * (three ? 8 : 4) * 256;
}

extern "C" int marl_mlp3_fwd_save(const marl_mlp3_weights_t* w, const marl_src_t* x, float* Y, long ldy, long gs_y,
                                  float* hsave, size_t hsave_floats, long M, int K1, int N3, int groups, void* stream) {
  if (M <= 0) return 0;
  const bool three = w->w2 != nullptr;
  if (hsave && (hsave_floats < marl_mlp3_save_floats(M, three, groups) || !aligned16(hsave))) return (int)hipErrorInvalidValue;
  if (!marl_mlp3_supported(x, K1, HD, three ? HD : 0, N3, groups)) return (int)hipErrorInvalidValue;
  // bias rows are read with 16-byte loads
  if (!aligned16(w->b1) || w->gs_b1 % 4 || (three && (!aligned16(w->b2) || w->gs_b2 % 4))) return (int)hipErrorInvalidValue;
  Mlp3Args a;
  if (!fill_args(a, w, x, M, K1, N3, groups)) return (int)hipErrorInvalidValue;
  a.Y = Y; a.ldy = ldy; a.gs_y = gs_y; a.ws = nullptr; a.hs = hsave;
  const long tiles = (M + 15) / 16;
  a.nst = stripes((tiles + FNW - 1) / FNW, groups);
  const int KC = kc_bucket(a.KV);
  const bool wide = N3 > 16;
  a.yvec = (ldy % 4 == 0) && (gs_y % 4 == 0) && aligned16(Y) && (N3 % 4 == 0);
  const size_t lds = fwd_lds(KC, a.CF, wide);
// (seven leading full chunks = a 120-wide dense segment 0: the QPLEX heads on 2s3z-sized maps get the compile-time variants)
#define MLP3_PICK(K, ...) (KC == 4 ? (const void*)K<4, __VA_ARGS__> : KC == 8 ? (a.CF == 7 ? (const void*)K<8, MLP3_CF7(__VA_ARGS__)> : (const void*)K<8, __VA_ARGS__>) \
                           : KC == 11 ? (a.CF == 7 ? (const void*)K<11, MLP3_CF7(__VA_ARGS__)> : (const void*)K<11, __VA_ARGS__>) : (const void*)K<12, __VA_ARGS__>)
#define MLP3_PICK_BIG(K, ...) (KC == 16 ? (const void*)K<16, __VA_ARGS__> : KC == 24 ? (const void*)K<24, __VA_ARGS__> : (const void*)K<32, __VA_ARGS__>)
#define MLP3_PICK_WIDE(K, ...) (KC == 8 ? (const void*)K<8, __VA_ARGS__> : KC == 16 ? (const void*)K<16, __VA_ARGS__> : (const void*)K<24, __VA_ARGS__>)
  const void* fn = wide ? MLP3_PICK_WIDE(mlp3_fwd_kernel, false, -1, true)
                   : KC > 12 ? (three ? MLP3_PICK_BIG(mlp3_fwd_kernel, true, -1) : MLP3_PICK_BIG(mlp3_fwd_kernel, false, -1))
                             : (three ? MLP3_PICK(mlp3_fwd_kernel, true, -1) : MLP3_PICK(mlp3_fwd_kernel, false, -1));
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  dim3 grid((unsigned)((a.nst + 7) / 8 * 8 * groups)), block(64 * FNW);
  void* kargs[] = {(void*)&a};
  e = hipLaunchKernel(fn, grid, block, kargs, lds, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  MARL_CHECK_LAUNCH();
  return 0;
}

extern "C" int marl_mlp3_fwd(const marl_mlp3_weights_t* w, const marl_src_t* x, float* Y, long ldy, long gs_y,
                             long M, int K1, int N3, int groups, void* stream) {
  return marl_mlp3_fwd_save(w, x, Y, ldy, gs_y, nullptr, 0, M, K1, N3, groups, stream);
}

extern "C" size_t marl_mlp3_bwd_workspace(long M, int K1, int N3, int groups) {
  const int nst = stripes((M + 63) / 64, groups, 512);      // (the variants with two workgroups per CU write twice the slabs)
  return (size_t)nst * groups * mlp3_slab_floats(K1, N3) * sizeof(float);
}

extern "C" int marl_mlp3_bwd_saved(const marl_mlp3_weights_t* w, const marl_src_t* x, const float* dY, long lddy, long gs_dy,
                                   const marl_mlp3_weights_t* grads, float* ws, size_t ws_bytes, const float* hsave,
                                   size_t hsave_floats, long M, int K1, int N3, int groups, void* stream) {
  if (M <= 0) return 0;
  const bool three = w->w2 != nullptr;
  if (hsave && (hsave_floats < marl_mlp3_save_floats(M, three, groups) || !aligned16(hsave))) return (int)hipErrorInvalidValue;
  if (!marl_mlp3_supported(x, K1, HD, three ? HD : 0, N3, groups)) return (int)hipErrorInvalidValue;
  if (!aligned16(w->b1) || w->gs_b1 % 4 || (three && (!aligned16(w->b2) || w->gs_b2 % 4))) return (int)hipErrorInvalidValue;
  if (three != (grads->w2 != nullptr)) return (int)hipErrorInvalidValue;
  if (ws_bytes < marl_mlp3_bwd_workspace(M, K1, N3, groups)) return (int)hipErrorInvalidValue;
  Mlp3Args a;
  if (!fill_args(a, w, x, M, K1, N3, groups)) return (int)hipErrorInvalidValue;
  a.Y = const_cast<float*>(dY); a.ldy = lddy; a.gs_y = gs_dy; a.ws = ws; a.hs = const_cast<float*>(hsave);
  const int KC = kc_bucket(a.KV);
  const bool wide = N3 > 16;
  a.nst = stripes((M + 63) / 64, groups, (hsave && !wide && KC <= MLP3_BWD2_KC) ? 512 : 256);
  if ((KC > 12 || wide) && !hsave) return (int)hipErrorInvalidValue;      // marl_mlp3_needs_kept(): no recomputing backward for these
  if (wide && (lddy % 4 || gs_dy % 4 || !aligned16(dY))) return (int)hipErrorInvalidValue;      // dY tiles are 16-byte loads
  const size_t lds = bwd_lds(KC, a.CF, hsave != nullptr, wide, three);
  const void* fn = wide ? MLP3_PICK_WIDE(mlp3_bwd_kernel, false, -1, true, true) : KC > 12 ? (three ? MLP3_PICK_BIG(mlp3_bwd_kernel, true, -1, true) : MLP3_PICK_BIG(mlp3_bwd_kernel, false, -1, true))
                   : hsave ? (three ? MLP3_PICK(mlp3_bwd_kernel, true, -1, true) : MLP3_PICK(mlp3_bwd_kernel, false, -1, true))
                           : (three ? MLP3_PICK(mlp3_bwd_kernel, true, -1, false) : MLP3_PICK(mlp3_bwd_kernel, false, -1, false));
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  dim3 grid((unsigned)((a.nst + 7) / 8 * 8 * groups)), block(64 * BNW);
  void* kargs[] = {(void*)&a};
  hipStream_t s = (hipStream_t)stream;
  e = hipLaunchKernel(fn, grid, block, kargs, lds, s);
  if (e != hipSuccess) return (int)e;
  MARL_CHECK_LAUNCH();
  Mlp3RedArgs r;
  r.ws = ws; r.dW1 = const_cast<float*>(grads->w1); r.db1 = const_cast<float*>(grads->b1);
  r.dW2 = const_cast<float*>(grads->w2); r.db2 = const_cast<float*>(grads->b2);
  r.dW3 = const_cast<float*>(grads->w3); r.db3 = const_cast<float*>(grads->b3);
  r.gs_w1 = grads->gs_w1; r.gs_b1 = grads->gs_b1; r.gs_w2 = grads->gs_w2; r.gs_b2 = grads->gs_b2;
  r.gs_w3 = grads->gs_w3; r.gs_b3 = grads->gs_b3;
  r.K1 = K1; r.N3 = N3; r.groups = groups; r.nst = a.nst;
  const long total = mlp3_slab_floats(K1, N3) * groups;
  hipLaunchKernelGGL(mlp3_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, r);
  MARL_CHECK_LAUNCH();
  return 0;
}

extern "C" int marl_mlp3_bwd(const marl_mlp3_weights_t* w, const marl_src_t* x, const float* dY, long lddy, long gs_dy,
                             const marl_mlp3_weights_t* grads, float* ws, size_t ws_bytes, long M, int K1, int N3,
                             int groups, void* stream) {
  return marl_mlp3_bwd_saved(w, x, dY, lddy, gs_dy, grads, ws, ws_bytes, nullptr, 0, M, K1, N3, groups, stream);
}
