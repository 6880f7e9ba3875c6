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

namespace {

constexpr int HD = 64;            // hidden width
constexpr int RS = 68;            // row stride (floats) of the [feature][64 rows] stage tiles
constexpr int FNW = 8;            // forward: waves per workgroup
constexpr int BNW = 4;            // backward: waves per workgroup (= 16-row tiles per iteration)

#define WG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

struct Mlp3Args {
  ConcatSrc x;
  const float *W1, *b1, *W2, *b2, *W3, *b3;
  long gs_w1, gs_b1, gs_w2, gs_b2, gs_w3, gs_b3;     // element strides between heads
  float* Y; long ldy, gs_y;                          // forward: outputs; backward: dY (read only)
  float* ws;                                         // backward: [slab][group][slab_floats]
  long M;
  int K1, N3, groups, nst, CF;                       // nst stripes per group; CF leading 16-byte-loadable chunks
};

__host__ __device__ inline long mlp3_slab_floats(int K1) { return (long)HD * (K1 + 1) + (long)HD * (HD + 1) + 16L * (HD + 1); }

// workgroup -> (stripe, head): the `groups` heads of one stripe of rows run on the SAME XCD (blockIdx % 8) next to
// each other in time, so the stripe's x rows are fetched from HBM once and hit that XCD's L2 for the other heads
__device__ __forceinline__ bool wg_map(int groups, int nst, int& stripe, int& g) {
  const int L = blockIdx.x, xcd = L & 7, r = L >> 3;
  g = r % groups;
  stripe = (r / groups) * 8 + xcd;
  return stripe < nst;
}

// Table of the generic chunks (those not wholly inside the 16-byte aligned part of dense0), two int4 per (chunk, lane):
//   [0] byte offsets of the lane's four elements from the base of its group's source row
//   [1] {cmp0 | cmp1 << 16, cmp2 | cmp3 << 16, kind, -}   kind: 0 zero, 1 dense0, 2 one-hot index, 3 dense1
// Segment widths are multiples of 4 (checked on the host), so the four elements k = 16c+4q+0..3 of a lane come from
// ONE source: a lane selects one row base per chunk and the element loads are base + offset - no per-element branching.
__device__ __forceinline__ void build_tab(int* tab, const ConcatSrc& x, int K1, int CF, int KC, int nthreads) {
  for (int e = threadIdx.x; e < (KC - CF) * 64; e += nthreads) {
    const int gc = e >> 6, l = e & 63;
    const int kb = 16 * (CF + gc) + 4 * (l >> 4);
    int off[4], cmp[4], kind = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int k = kb + i;
      off[i] = 0; cmp[i] = 0xffff;
      if (k >= K1) continue;
      if (k < x.k0) { kind = 1; off[i] = 4 * k; }
      else if (k - x.k0 < x.k1) { kind = 3; off[i] = 4 * (k - x.k0); }
      else {
        k -= x.k0 + x.k1;
        const int j = k / x.hot_w;
        kind = 2; off[i] = 4 * j; cmp[i] = k - j * x.hot_w;
      }
    }
    int* t = tab + e * 8;
    t[0] = off[0]; t[1] = off[1]; t[2] = off[2]; t[3] = off[3];
    t[4] = cmp[0] | (cmp[1] << 16); t[5] = cmp[2] | (cmp[3] << 16); t[6] = kind; t[7] = 0;
  }
}

// weight fragments (A operands), fragment-major: [(t * KCn + c) * 64 + lane] f32x4 = W[16t + m][16c + 4q + 0..3]
__device__ __forceinline__ void stage_w(float* dst, const float* W, int ldw, int rows_valid, int K, int KCn, int NT_, int nthreads) {
  for (int e = threadIdx.x; e < NT_ * KCn * 64; e += nthreads) {
    const int l = e & 63, tc = e >> 6, t = tc / KCn, c = tc - t * KCn;
    const int n = 16 * t + (l & 15), k0 = 16 * c + 4 * (l >> 4);
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (n < rows_valid && k0 + i < K) ? W[(long)n * ldw + k0 + i] : 0.f;
    *reinterpret_cast<f32x4*>(dst + (long)e * 4) = v;
  }
}
// transposed fragments: [(t * 4 + c) * 64 + lane] f32x4 = W[16c + 4q + i][16t + m]   (A operand of dX^T = W^T dY^T)
__device__ __forceinline__ void stage_wT(float* dst, const float* W, int ldw, int nthreads) {
  for (int e = threadIdx.x; e < 16 * 64; e += nthreads) {
    const int l = e & 63, tc = e >> 6, t = tc >> 2, c = tc & 3;
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = W[(long)(16 * c + 4 * (l >> 4) + i) * ldw + 16 * t + (l & 15)];
    *reinterpret_cast<f32x4*>(dst + (long)e * 4) = v;
  }
}

struct XRow { long r0c, ric, rowc; int flags; };     // flags: 1 row < M, 2 dense0 row valid, 4 index row valid

__device__ __forceinline__ XRow x_row(const ConcatSrc& x, long row, long M) {
  XRow r;
  const bool live = row < M;
  r.rowc = live ? row : M - 1;
  const ConcatRow cr = concat_row(x, r.rowc);
  r.r0c = cr.ok0 ? cr.r0 : 0;
  r.ric = cr.oki ? cr.ri : 0;
  r.flags = (live ? 1 : 0) | (cr.ok0 ? 2 : 0) | (cr.oki ? 4 : 0);
  return r;
}

// issue the loads of one 16-row x tile (raw bits; nothing here consumes a loaded value)
template <int KC>
__device__ __forceinline__ void x_issue(f32x4 (&xv)[KC], const ConcatSrc& x, const XRow& r, const int* tab, int CF, int lane) {
  const int q = lane >> 4;
  const char* d0 = reinterpret_cast<const char*>(x.p0 + r.r0c * x.ld0);
  const char* d1 = x.p1 ? reinterpret_cast<const char*>(x.p1 + r.rowc * x.ld1) : d0;
  const char* di = x.idx ? reinterpret_cast<const char*>(x.idx + r.ric * x.nhot) : d0;
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    if (c < CF) {
      xv[c] = *reinterpret_cast<const f32x4*>(d0 + 64 * c + 16 * q);
    } else {
      const int* t = tab + ((c - CF) * 64 + lane) * 8;
      const uint4 off = *reinterpret_cast<const uint4*>(t);            // unsigned: no sign extension per address
      const int kind = t[6];
      const char* base = kind == 2 ? di : (kind == 3 ? d1 : d0);      // one row base per lane and chunk
      xv[c][0] = __int_as_float(*reinterpret_cast<const int*>(base + off.x));
      xv[c][1] = __int_as_float(*reinterpret_cast<const int*>(base + off.y));
      xv[c][2] = __int_as_float(*reinterpret_cast<const int*>(base + off.z));
      xv[c][3] = __int_as_float(*reinterpret_cast<const int*>(base + off.w));
    }
  }
}
// raw -> values of the virtual concat (selects only)
template <int KC>
__device__ __forceinline__ void x_finish(f32x4 (&xv)[KC], const XRow& r, const int* tab, int CF, int lane) {
  const bool ok0 = (r.flags & 2) != 0, oki = (r.flags & 4) != 0;
  // rows that read as zero (remap before the first slot) are rare: one wave-uniform test instead of 4 selects per chunk
  const bool any_bad0 = __builtin_amdgcn_ballot_w64(!ok0) != 0;
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    if (c < CF) {
      if (any_bad0 && !ok0) xv[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    } else {
      const int4 t1 = *reinterpret_cast<const int4*>(tab + ((c - CF) * 64 + lane) * 8 + 4);
      const int kind = t1.z;
      const bool dense = (kind == 1 && ok0) || kind == 3;
      const bool hot = kind == 2 && oki;
      const int cmp[4] = {t1.x & 0xffff, (int)((unsigned)t1.x >> 16), t1.y & 0xffff, (int)((unsigned)t1.y >> 16)};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float raw = xv[c][i];
        const float vd = dense ? raw : 0.f;
        xv[c][i] = (hot && __float_as_int(raw) == cmp[i]) ? 1.f : vd;
      }
    }
  }
}

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  return (f32x4){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
}

// layer 1 of one tile in registers (transposed form); h1 comes out post-relu
template <int KC>
__device__ __forceinline__ void fwd1(const f32x4 (&xv)[KC], const float* W1s, const f32x4 (&b1v)[4], f32x4 (&h1)[4], int lane) {
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = b1v[t];
  // weight fragments one chunk ahead; the scheduling barrier keeps the compiler from hoisting every LDS read of
  // the layer to the top (that cost > 256 registers and spilled)
  f32x4 wn[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) wn[t] = *reinterpret_cast<const f32x4*>(W1s + ((t * KC) * 64 + lane) * 4);
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    f32x4 wc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) wc[t] = wn[t];
    if (c + 1 < KC) {
#pragma unroll
      for (int t = 0; t < 4; ++t) wn[t] = *reinterpret_cast<const f32x4*>(W1s + ((t * KC + c + 1) * 64 + lane) * 4);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = mfma16x4(wc[t], xv[c], acc[t]);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) h1[t] = relu4(acc[t]);
}
// layer 2: the accumulator tiles of layer 1 are the B fragments
__device__ __forceinline__ void fwd2(const f32x4 (&h1)[4], const float* W2s, const f32x4 (&b2v)[4], f32x4 (&h2)[4], int lane) {
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = b2v[t];
  f32x4 wn[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) wn[t] = *reinterpret_cast<const f32x4*>(W2s + ((t * 4) * 64 + lane) * 4);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    f32x4 wc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) wc[t] = wn[t];
    if (c + 1 < 4) {
#pragma unroll
      for (int t = 0; t < 4; ++t) wn[t] = *reinterpret_cast<const f32x4*>(W2s + ((t * 4 + c + 1) * 64 + lane) * 4);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = mfma16x4(wc[t], h1[c], acc[t]);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) h2[t] = relu4(acc[t]);
}

// ------------------------------------------------------------------------------------------------- forward
// THREE = false: two-layer heads  y = W3 relu(W1 x + b1) + b3  (W2 == NULL; QPLEX transformation nets)
template <int KC, bool THREE>
__global__ __launch_bounds__(64 * FNW, 2) void mlp3_fwd_kernel(Mlp3Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int stripe, g;
  if (!wg_map(a.groups, a.nst, stripe, g)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, m = lane & 15;
  float* W1s = smem;                              // [4][KC][64] f32x4
  float* W2s = W1s + 4 * KC * 256;                // [4][4][64] f32x4
  float* W3s = W2s + 16 * 256;                    // [4][64] f32x4 (rows >= N3 zero)
  int* tab = reinterpret_cast<int*>(W3s + 4 * 256);
  const float* W1 = a.W1 + g * a.gs_w1;
  const float* W2 = a.W2 + g * a.gs_w2;
  const float* W3 = a.W3 + g * a.gs_w3;
  stage_w(W1s, W1, a.K1, HD, a.K1, KC, 4, 64 * FNW);
  if (THREE) stage_w(W2s, W2, HD, HD, HD, 4, 4, 64 * FNW);
  stage_w(W3s, W3, HD, a.N3, HD, 4, 1, 64 * FNW);
  build_tab(tab, a.x, a.K1, a.CF, KC, 64 * FNW);
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
  x_issue<KC>(xv, a.x, xr, tab, a.CF, lane);
  for (; tile < t_end; tile += FNW) {
    x_finish<KC>(xv, xr, tab, a.CF, lane);
    const bool live = (xr.flags & 1) != 0;
    float* y = Y + xr.rowc * a.ldy + 4 * q;
    f32x4 h1[4], h2[4];
    fwd1<KC>(xv, W1s, b1v, h1, lane);
    {
      const long nt = tile + FNW < t_end ? tile + FNW : tile;
      xr = x_row(a.x, nt * 16 + m, a.M);
      x_issue<KC>(xv, a.x, xr, tab, a.CF, lane);
    }
    if (THREE) fwd2(h1, W2s, b2v, h2, lane);
    else {
#pragma unroll
      for (int t = 0; t < 4; ++t) h2[t] = h1[t];
    }
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

// ------------------------------------------------------------------------------------------------- backward
// T-layout register tile (lane (q,m), tile t, register i = feature 16t+4q+i of row m) -> stage[feature][16*wave + m]
__device__ __forceinline__ void stash4(float* st, const f32x4 (&v)[4], int wave, int q, int m) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) st[(16 * t + 4 * q + i) * RS + 16 * wave + m] = v[t][i];
}

template <int KC, bool THREE>
__global__ __launch_bounds__(64 * BNW, 1) void mlp3_bwd_kernel(Mlp3Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int stripe, g;
  if (!wg_map(a.groups, a.nst, stripe, g)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, m = lane & 15;
  constexpr int SF = (16 * KC + HD) > 208 ? (16 * KC + HD) : 208;    // stage features: max(x + dh1, h1 + dh2 + h2 + dY)
  float* W1s = smem;                              // [4][KC][64] f32x4
  float* W2s = W1s + 4 * KC * 256;                // [4][4][64] f32x4
  float* W2Ts = W2s + 16 * 256;                   // [4][4][64] f32x4
  float* W3Ts = W2Ts + 16 * 256;                  // [4 t][4 j][64]: W3[4j + q][16t + m]
  float* stage = W3Ts + 16 * 64;                  // [SF][RS]
  int* tab = reinterpret_cast<int*>(stage + SF * RS);
  const float* W1 = a.W1 + g * a.gs_w1;
  const float* W2 = a.W2 + g * a.gs_w2;
  const float* W3 = a.W3 + g * a.gs_w3;
  stage_w(W1s, W1, a.K1, HD, a.K1, KC, 4, 64 * BNW);
  if (THREE) {
    stage_w(W2s, W2, HD, HD, HD, 4, 4, 64 * BNW);
    stage_wT(W2Ts, W2, HD, 64 * BNW);
  }
  for (int e = tid; e < 16 * 64; e += 64 * BNW) {
    const int l = e & 63, tj = e >> 6, t = tj >> 2, j = tj & 3;
    const int n3 = 4 * j + (l >> 4);
    W3Ts[e] = n3 < a.N3 ? W3[(long)n3 * HD + 16 * t + (l & 15)] : 0.f;
  }
  build_tab(tab, a.x, a.K1, a.CF, KC, 64 * BNW);
  f32x4 b1v[4], b2v[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    b1v[t] = *reinterpret_cast<const f32x4*>(a.b1 + g * a.gs_b1 + 16 * t + 4 * q);
    if (THREE) b2v[t] = *reinterpret_cast<const f32x4*>(a.b2 + g * a.gs_b2 + 16 * t + 4 * q);
  }
  __syncthreads();

  const long its = (a.M + 63) / 64;
  const long per = (its + a.nst - 1) / a.nst;
  const long i_begin = (long)stripe * per;
  long i_end = i_begin + per; if (i_end > its) i_end = its;
  const float* dY = a.Y + g * a.gs_y;

  f32x4 dW1[KC], dW2[4], dW3 = {0.f, 0.f, 0.f, 0.f};
  float bs1 = 0.f, bs2 = 0.f, bs3 = 0.f;
#pragma unroll
  for (int c = 0; c < KC; ++c) dW1[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 4; ++c) dW2[c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float* xT = stage;                     // stage 1: [16*KC][RS] x^T, then [64][RS] dh1^T
  float* dh1T = stage + 16 * KC * RS;
  float* h1T = stage;                    // stage 2: h1^T, dh2^T, h2^T [64][RS] each, dY^T [16][RS]
  float* dh2T = stage + 64 * RS;
  float* h2T = stage + 128 * RS;
  float* dYT = stage + 192 * RS;

  f32x4 xv[KC];
  XRow xr;
  if (i_begin < i_end) {
    xr = x_row(a.x, (i_begin * BNW + wave) * 16 + m, a.M);
    x_issue<KC>(xv, a.x, xr, tab, a.CF, lane);
  }
  for (long it = i_begin; it < i_end; ++it) {
    // ---------------- phase A: this wave's 16-row tile, all in registers
    x_finish<KC>(xv, xr, tab, a.CF, lane);
    const bool live = (xr.flags & 1) != 0;
    const long rowc = xr.rowc;
    float dy[4];                                   // dY[row m][4j + q]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n3 = 4 * j + q;
      dy[j] = (live && n3 < a.N3) ? dY[rowc * a.ldy + n3] : 0.f;
    }
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) xT[(16 * c + 4 * q + i) * RS + 16 * wave + m] = xv[c][i];
    f32x4 h1[4], h2[4], dh2[4], dh1[4];
    fwd1<KC>(xv, W1s, b1v, h1, lane);
    // x is consumed: start the loads of the next iteration's tile (unconditional; the last one re-reads its own)
    {
      const long ni = it + 1 < i_end ? it + 1 : it;
      xr = x_row(a.x, (ni * BNW + wave) * 16 + m, a.M);
      x_issue<KC>(xv, a.x, xr, tab, a.CF, lane);
    }
    if (THREE) fwd2(h1, W2s, b2v, h2, lane);
    else {
#pragma unroll
      for (int t = 0; t < 4; ++t) h2[t] = h1[t];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * j < a.N3) acc = mfma16(W3Ts[(t * 4 + j) * 64 + lane], dy[j], acc);      // heads n3 = 4j + q
#pragma unroll
      for (int i = 0; i < 4; ++i) dh2[t][i] = h2[t][i] > 0.f ? acc[i] : 0.f;
    }
    if (THREE) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(W2Ts + ((t * 4 + c) * 64 + lane) * 4);
          acc = mfma16x4(w, dh2[c], acc);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) dh1[t][i] = h1[t][i] > 0.f ? acc[i] : 0.f;
      }
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) dh1[t] = dh2[t];
    }
    stash4(dh1T, dh1, wave, q, m);
    WG_BARRIER();
    // ---------------- phase B1: dW1 rows [16w, 16w+16) over the 64 rows of the iteration
#pragma unroll
    for (int rt = 0; rt < BNW; ++rt) {
      const f32x4 af = *reinterpret_cast<const f32x4*>(dh1T + (16 * wave + m) * RS + 16 * rt + 4 * q);
      bs1 += (af[0] + af[1]) + (af[2] + af[3]);
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const f32x4 bf = *reinterpret_cast<const f32x4*>(xT + (16 * c + m) * RS + 16 * rt + 4 * q);
        dW1[c] = mfma16x4(af, bf, dW1[c]);
      }
    }
    WG_BARRIER();
    // ---------------- stage 2 operands
    if (THREE) {
      stash4(h1T, h1, wave, q, m);
      stash4(dh2T, dh2, wave, q, m);
    }
    stash4(h2T, h2, wave, q, m);
#pragma unroll
    for (int j = 0; j < 4; ++j) dYT[(4 * j + q) * RS + 16 * wave + m] = dy[j];
    WG_BARRIER();
    // ---------------- phase B2: dW2 rows [16w, 16w+16), dW3 columns [16w, 16w+16)
#pragma unroll
    for (int rt = 0; rt < BNW; ++rt) {
      if (THREE) {
        const f32x4 af = *reinterpret_cast<const f32x4*>(dh2T + (16 * wave + m) * RS + 16 * rt + 4 * q);
        bs2 += (af[0] + af[1]) + (af[2] + af[3]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 bf = *reinterpret_cast<const f32x4*>(h1T + (16 * c + m) * RS + 16 * rt + 4 * q);
          dW2[c] = mfma16x4(af, bf, dW2[c]);
        }
      }
      const f32x4 ay = *reinterpret_cast<const f32x4*>(dYT + m * RS + 16 * rt + 4 * q);
      const f32x4 bh = *reinterpret_cast<const f32x4*>(h2T + (16 * wave + m) * RS + 16 * rt + 4 * q);
      bs3 += (ay[0] + ay[1]) + (ay[2] + ay[3]);
      dW3 = mfma16x4(ay, bh, dW3);
    }
    WG_BARRIER();
  }

  // ---------------- slab: [dW1 64 x (K1+1) | dW2 64 x 65 | dW3 16 x 65], bias gradient in the last column
  bs1 += __shfl_xor(bs1, 16, 64); bs1 += __shfl_xor(bs1, 32, 64);
  bs2 += __shfl_xor(bs2, 16, 64); bs2 += __shfl_xor(bs2, 32, 64);
  bs3 += __shfl_xor(bs3, 16, 64); bs3 += __shfl_xor(bs3, 32, 64);
  const int K1x = a.K1 + 1;
  float* s1 = a.ws + ((long)stripe * a.groups + g) * mlp3_slab_floats(a.K1);
  float* s2 = s1 + (long)HD * K1x;
  float* s3 = s2 + (long)HD * (HD + 1);
#pragma unroll
  for (int c = 0; c < KC; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = 16 * wave + 4 * q + i, k = 16 * c + m;
      if (k < a.K1) s1[(long)n * K1x + k] = dW1[c][i];
    }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) s2[(16 * wave + 4 * q + i) * (HD + 1) + 16 * c + m] = dW2[c][i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s3[(4 * q + i) * (HD + 1) + 16 * wave + m] = dW3[i];
  if (q == 0) {
    s1[(long)(16 * wave + m) * K1x + a.K1] = bs1;
    s2[(16 * wave + m) * (HD + 1) + HD] = bs2;
    if (wave == 0) s3[m * (HD + 1) + HD] = bs3;
  }
}

struct Mlp3RedArgs {
  const float* ws;
  float *dW1, *db1, *dW2, *db2, *dW3, *db3;
  long gs_w1, gs_b1, gs_w2, gs_b2, gs_w3, gs_b3;
  int K1, N3, groups, nst;
};

// grads += sum over the stripes' slabs, in stripe order (deterministic)
__global__ __launch_bounds__(256) void mlp3_reduce_kernel(Mlp3RedArgs a) {
  const long SZ = mlp3_slab_floats(a.K1);
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= SZ * a.groups) return;
  const int g = (int)(e / SZ);
  long r = e - (long)g * SZ;
  float s = 0.f;
  for (int sl = 0; sl < a.nst; ++sl) s += a.ws[((long)sl * a.groups + g) * SZ + r];
  const int K1x = a.K1 + 1;
  if (r < (long)HD * K1x) {
    const int n = (int)(r / K1x), k = (int)(r - (long)n * K1x);
    if (k < a.K1) a.dW1[g * a.gs_w1 + (long)n * a.K1 + k] += s;
    else a.db1[g * a.gs_b1 + n] += s;
    return;
  }
  r -= (long)HD * K1x;
  if (r < HD * (HD + 1)) {
    const int n = (int)(r / (HD + 1)), k = (int)(r - n * (HD + 1));
    if (!a.dW2) return;                       // two-layer head
    if (k < HD) a.dW2[g * a.gs_w2 + n * HD + k] += s;
    else a.db2[g * a.gs_b2 + n] += s;
    return;
  }
  r -= HD * (HD + 1);
  const int n = (int)(r / (HD + 1)), k = (int)(r - n * (HD + 1));
  if (n >= a.N3) return;
  if (k < HD) a.dW3[g * a.gs_w3 + n * HD + k] += s;
  else a.db3[g * a.gs_b3 + n] += s;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline ConcatSrc to_src3(const marl_src_t* s) {
  ConcatSrc c;
  c.p0 = s->p0; c.ld0 = s->ld0; c.k0 = s->k0;
  c.p1 = s->p1; c.ld1 = s->ld1; c.k1 = s->k1;
  c.idx = s->idx; c.nhot = s->nhot; c.hot_w = s->hot_w > 0 ? s->hot_w : 1; c.nid = s->nid;
  c.m0 = s->m0; c.ldm0 = s->ldm0;
  c.rpe0 = s->rpe0; c.bs0 = s->bs0; c.off0 = s->off0;
  c.rpei = s->rpei; c.bsi = s->bsi; c.offi = s->offi;
  c.emap0 = s->emap0;
  c.fd0 = make_fastdiv((unsigned)(s->rpe0 > 0 ? s->rpe0 : 1));
  c.fdi = make_fastdiv((unsigned)(s->rpei > 0 ? s->rpei : 1));
  c.fdn = make_fastdiv(1u);
  return c;
}

// instantiated chunk counts: 4, 8, 11 (QPLEX [state 120 | one-hot 55] exactly), 12
inline int kc_bucket(int K1) { const int kc = (K1 + 15) / 16; return kc == 11 ? 11 : (kc + 3) / 4 * 4; }
inline size_t fwd_lds(int KC, int CF) { return (size_t)(4 * KC * 256 + 16 * 256 + 4 * 256) * 4 + (size_t)(KC - CF) * 512 * 4; }
inline size_t bwd_lds(int KC, int CF) {
  const int SF = (16 * KC + HD) > 208 ? (16 * KC + HD) : 208;
  return (size_t)(4 * KC * 256 + 2 * 16 * 256 + 16 * 64 + SF * RS) * 4 + (size_t)(KC - CF) * 512 * 4;
}
inline int lead_chunks(const marl_src_t* x) {
  const bool al = x->p0 && (x->ld0 % 4 == 0) && aligned16(x->p0);
  return al ? x->k0 / 16 : 0;
}
// stripes per group: a multiple of 8 (one per XCD and round, see wg_map) with stripes * groups <= 256 workgroups, so
// that every XCD gets the same number of workgroups and all of them are resident at once (one per CU): 26 stripes
// x 10 heads = 260 workgroups ran as two rounds and took twice as long as 24 x 10
inline int stripes(long units, int groups) {
  long n = 256 / groups / 8 * 8;
  if (n < 8) n = 8;
  if (n > units) n = units;
  return (int)(n < 1 ? 1 : n);
}

bool fill_args(Mlp3Args& a, const marl_mlp3_weights_t* w, const marl_src_t* x, long M, int K1, int N3, int groups) {
  a.x = to_src3(x);
  if (concat_width(a.x) != K1) return false;
  a.W1 = w->w1; a.b1 = w->b1; a.W2 = w->w2; a.b2 = w->b2; a.W3 = w->w3; a.b3 = w->b3;
  a.gs_w1 = w->gs_w1; a.gs_b1 = w->gs_b1; a.gs_w2 = w->gs_w2; a.gs_b2 = w->gs_b2; a.gs_w3 = w->gs_w3; a.gs_b3 = w->gs_b3;
  a.M = M; a.K1 = K1; a.N3 = N3; a.groups = groups;
  a.CF = lead_chunks(x);
  return true;
}

}  // namespace

extern "C" int marl_mlp3_supported(const marl_src_t* x, int K1, int H1, int H2, int N3, int groups) {
  if (H1 != HD || (H2 != HD && H2 != 0) || N3 < 1 || N3 > 16 || groups < 1 || K1 < 1) return 0;
  if (!x->p0 || x->k0 < 4 || x->m0 || x->nid) return 0;
  if (x->nhot && (x->hot_w < 1 || x->hot_w >= 16384 || x->nhot >= 8192)) return 0;
  if (x->k0 >= 16384 || x->k1 >= 16384) return 0;
  if (x->k0 % 4 || x->k1 % 4) return 0;               // a lane's four consecutive columns come from one segment
  const int KC = kc_bucket(K1);
  if (KC > 12) return 0;
  const int CF = lead_chunks(x);
  return bwd_lds(KC, CF) <= 160 * 1024 && fwd_lds(KC, CF) <= 160 * 1024;
}

extern "C" int marl_mlp3_fwd(const marl_mlp3_weights_t* w, const marl_src_t* x, float* Y, long ldy, long gs_y,
                             long M, int K1, int N3, int groups, void* stream) {
  if (M <= 0) return 0;
  const bool three = w->w2 != nullptr;
  if (!marl_mlp3_supported(x, K1, HD, three ? HD : 0, N3, groups)) return (int)hipErrorInvalidValue;
  // bias rows are read with 16-byte loads
  if (!aligned16(w->b1) || w->gs_b1 % 4 || (three && (!aligned16(w->b2) || w->gs_b2 % 4))) return (int)hipErrorInvalidValue;
  Mlp3Args a;
  if (!fill_args(a, w, x, M, K1, N3, groups)) return (int)hipErrorInvalidValue;
  a.Y = Y; a.ldy = ldy; a.gs_y = gs_y; a.ws = nullptr;
  const long tiles = (M + 15) / 16;
  a.nst = stripes((tiles + FNW - 1) / FNW, groups);
  const int KC = kc_bucket(K1);
  const size_t lds = fwd_lds(KC, a.CF);
#define MLP3_PICK(K, T3) (KC == 4 ? (const void*)K<4, T3> : KC == 8 ? (const void*)K<8, T3> : KC == 11 ? (const void*)K<11, T3> : (const void*)K<12, T3>)
  const void* fn = three ? MLP3_PICK(mlp3_fwd_kernel, true) : MLP3_PICK(mlp3_fwd_kernel, false);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  dim3 grid((unsigned)((a.nst + 7) / 8 * 8 * groups)), block(64 * FNW);
  void* kargs[] = {(void*)&a};
  e = hipLaunchKernel(fn, grid, block, kargs, lds, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  MARL_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t marl_mlp3_bwd_workspace(long M, int K1, int N3, int groups) {
  (void)N3;
  const int nst = stripes((M + 63) / 64, groups);
  return (size_t)nst * groups * mlp3_slab_floats(K1) * sizeof(float);
}

extern "C" int marl_mlp3_bwd(const marl_mlp3_weights_t* w, const marl_src_t* x, const float* dY, long lddy, long gs_dy,
                             const marl_mlp3_weights_t* grads, float* ws, size_t ws_bytes, long M, int K1, int N3,
                             int groups, void* stream) {
  if (M <= 0) return 0;
  const bool three = w->w2 != nullptr;
  if (!marl_mlp3_supported(x, K1, HD, three ? HD : 0, N3, groups)) return (int)hipErrorInvalidValue;
  if (!aligned16(w->b1) || w->gs_b1 % 4 || (three && (!aligned16(w->b2) || w->gs_b2 % 4))) return (int)hipErrorInvalidValue;
  if (three != (grads->w2 != nullptr)) return (int)hipErrorInvalidValue;
  if (ws_bytes < marl_mlp3_bwd_workspace(M, K1, N3, groups)) return (int)hipErrorInvalidValue;
  Mlp3Args a;
  if (!fill_args(a, w, x, M, K1, N3, groups)) return (int)hipErrorInvalidValue;
  a.Y = const_cast<float*>(dY); a.ldy = lddy; a.gs_y = gs_dy; a.ws = ws;
  a.nst = stripes((M + 63) / 64, groups);
  const int KC = kc_bucket(K1);
  const size_t lds = bwd_lds(KC, a.CF);
  const void* fn = three ? MLP3_PICK(mlp3_bwd_kernel, true) : MLP3_PICK(mlp3_bwd_kernel, false);
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
  const long total = mlp3_slab_floats(K1) * groups;
  hipLaunchKernelGGL(mlp3_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, r);
  MARL_CHECK_LAUNCH();
  return 0;
}
