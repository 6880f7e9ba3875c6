// Fused three-layer heads  y = W3 relu(W2 relu(W1 x + b1) + b2) + b3  (hidden width 64) with the multiplies done as an fp32-accurate
// SPLIT on the bf16 matrix cores ("bf16x6", opt-in: args.gemm_mode = "bf16x6"; the default path is mlp3_fused.hip on
// v_mfma_f32_16x16x4_f32).  Reference: the key / agents / action extractors of QPLEX's lambda-net, network/mixer.py:117-145,
// evaluated at :155-169.
//
// Every fp32 operand element is split EXACTLY into three bf16 terms, a = hi + mid + lo (round-to-nearest each: 8 + 8 + 8 significand
// bits), and a product a.b is the six bf16 products  mid.mid, hi.lo, lo.hi, hi.mid, mid.hi, hi.hi  accumulated in fp32 by
// v_mfma_f32_16x16x32_bf16 (smallest first); the dropped terms (mid.lo, lo.mid, lo.lo) are <= 2^-24 of |a.b|.  Measured against fp64:
// error / output scale 4.7e-7 at K = 64 where the fp32 MFMA has 6.8e-7 (profiles/archive/r03_bf16x3_probe.txt).  Six such MFMAs cover a
// 16 x 16 x 32 block in ~100 pipe cycles; the eight v_mfma_f32_16x16x4_f32 of the same block take 256.
//
// Layouts (checked with integer data by tools/probe/x6_layout_probe.hip, profiles/archive/r04_x6_layout_probe.txt):
//   * v_mfma_f32_16x16x32_bf16: lane (g = l >> 4, i = l & 15) holds A[i][slot j] and B[slot j][i], j = 0..7; D register r =
//     C[4g + r][i].  The slot -> k assignment is free as long as both operands use the same one.
//   * transposed formulation (as mlp3_fused.hip): out^T[feature][row] = W[feature][k] in^T[k][row]; the WEIGHTS are the A operand
//     (fragments pre-split once per workgroup, in LDS), the activations the B operand.  Two accumulator tiles of one layer ARE the
//     8 k-slots of the next layer's 32-chunk.  Which output feature an accumulator row holds is decided by the order in which the
//     weight ROWS are staged, so they are staged permuted: row 4g + r of tile t holds feature 32 (t >> 1) + 8g + 4 (t & 1) + r -
//     lane group g then owns the 8 CONSECUTIVE features 32c + 8g .. + 7 of chunk c (slot j <-> k = 32c + 8g + j, the natural
//     order): fragments of x are two adjacent 16-byte loads, an image write is one 16-byte store per plane.
//   * weight gradients reduce over ROWS: operands go through a [row][column] bf16 LDS image (each lane packs four consecutive
//     columns of its row: one 8-byte write per plane) and come back with ds_read_b64_tr_b16, which hands lane (g, i) the column i of
//     rows 8g .. 8g + 3 - the A / B fragment of a product that sums over the image's rows.  256-byte image rows, 16-byte chunks
//     XOR-swizzled by the row (cdna_hip_programming.md T10, image (b)).
#include "mlp3_common.h"
#include "x6.h"

namespace {

// weight fragments in LDS: item (tile t, chunk c2) = [3 planes][64 lanes] 16 bytes
__device__ __forceinline__ F3 lds_f3(const int* base, int item, int lane) {
  const i32x4* p = reinterpret_cast<const i32x4*>(base) + item * 192 + lane;
  F3 f;
  f.h = p[0]; f.m = p[64]; f.l = p[128];
  return f;
}
// feature held by accumulator row i = 4g + r of tile t (see the layout note at the top)
__host__ __device__ inline int permf(int t, int i) { return 32 * (t >> 1) + 8 * (i >> 2) + 4 * (t & 1) + (i & 3); }
// A fragments of W (TR = false: A[i][slot] = W[row of tile t, i][virtual column of the slot]) or of W^T (TR = true: A[i][slot] =
// W[row = slot][column of tile t, i]), split into planes.  Slot j of lane group g <-> 32 c2 + 8g + j.  PERM: the tile's rows / columns
// are the permuted features permf(t, i) (hidden layers); else 16t + i (the output layer).
template <bool TR, bool PERM, int NTHR>
__device__ __forceinline__ void stage6(int* dst, const float* W, int ldw, int rows_valid, int K, int T, int KC2, int k0, int kpad) {
  for (int e = threadIdx.x; e < T * KC2 * 64; e += NTHR) {
    const int l = e & 63, tc = e >> 6, t = tc / KC2, c2 = tc - t * KC2;
    const int n = PERM ? permf(t, l & 15) : 16 * t + (l & 15), qq = l >> 4;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int col = 32 * c2 + 8 * qq + j;
      float x = 0.f;
      if (TR) {
        if (col < rows_valid && n < K) x = W[(long)col * ldw + n];
      } else {
        const int kr = vcol(col, k0, kpad);
        if (n < rows_valid && kr >= 0 && kr < K) x = W[(long)n * ldw + kr];
      }
      v[j] = x;
    }
    const F3 f = split8((f32x4){v[0], v[1], v[2], v[3]}, (f32x4){v[4], v[5], v[6], v[7]});
    i32x4* p = reinterpret_cast<i32x4*>(dst) + tc * 192 + l;
    p[0] = f.h; p[64] = f.m; p[128] = f.l;
  }
}

// ---- the x tile in the natural slot order: chunk c of lane group q = columns 32 (c >> 1) + 8q + 4 (c & 1) + (0..3)
__device__ __forceinline__ int ncol(int c, int q) { return 32 * (c >> 1) + 8 * q + 4 * (c & 1); }
// (build_tab / x_issue of mlp3_common.h with that column map; x_finish is shared - it only reads the table)
__device__ __forceinline__ void build_tab_nat(int* tab, const ConcatSrc& x, int KV, int kpad, int CF, int KC, int nthreads) {
  for (int e = threadIdx.x; e < (KC - CF) * 4; e += nthreads) {
    const int gc = e >> 2, qq = e & 3;
    const int kb = ncol(CF + gc, qq);
    int off[4], cmp[4], kind = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int k = kb + i;
      off[i] = 0; cmp[i] = 0xffff;
      if (k >= KV) continue;
      if (k < x.k0) { kind = 1; off[i] = 4 * k; }
      else if (k < x.k0 + kpad) { kind = 1; continue; }      // pad column: reads column 0 (finite), meets a zero weight
      else if ((k -= kpad) - x.k0 < x.k1) { kind = 3; off[i] = 4 * (k - x.k0); }
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
template <int KC, int CFT>
__device__ __forceinline__ void x_issue_nat(f32x4 (&xv)[KC], const ConcatSrc& x, const XRow& r, const int* tab, int CFr, int lane) {
  const int CF = CFT >= 0 ? CFT : CFr;
  const int q = lane >> 4;
  const char* d0 = reinterpret_cast<const char*>(x.p0 + r.r0c * x.ld0);
  const char* d1 = x.p1 ? reinterpret_cast<const char*>(x.p1 + r.rowc * x.ld1) : d0;
  const char* di = x.idx ? reinterpret_cast<const char*>(x.idx + r.ric * x.nhot) : d0;
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    if (c < CF) {
      xv[c] = *reinterpret_cast<const f32x4*>(d0 + 4 * ncol(c, q));
    } else {
      const int* t = tab + ((c - CF) * 4 + q) * 8;
      const uint4 off = *reinterpret_cast<const uint4*>(t);
      const int kind = t[6];
      const char* base = kind == 2 ? di : (kind == 3 ? d1 : d0);
      xv[c][0] = __int_as_float(*reinterpret_cast<const int*>(base + off.x));
      xv[c][1] = __int_as_float(*reinterpret_cast<const int*>(base + off.y));
      xv[c][2] = __int_as_float(*reinterpret_cast<const int*>(base + off.z));
      xv[c][3] = __int_as_float(*reinterpret_cast<const int*>(base + off.w));
    }
  }
}
// chunks wholly inside the 16-byte aligned part of dense0, in chunk PAIRS (a pair = 32 columns)
inline int lead_chunks_nat(const marl_src_t* x) {
  const bool al = x->p0 && (x->ld0 % 4 == 0) && aligned16(x->p0);
  return al ? x->k0 / 32 * 2 : 0;
}
// chunk count of the split kernels: even (whole 32-column pairs), 4 / 8 / 12
inline int kc_bucket_x6(int KV) { return (KV + 63) / 64 * 4; }

// ------------------------------------------------------------------------------------------------- forward
// 8 waves, each walks its own 16-row tiles (x of the next tile in flight); hidden activations never leave the wave.  a.hs != NULL:
// relu(h1), relu(h2) are kept for the backward as fp32 accumulator fragments (the layout of the fp32 pair, 512 bytes per row and head).
template <int KC, bool THREE, int CFT>
__global__ __launch_bounds__(64 * FNW, 2) void mlp3x6_fwd_kernel(Mlp3Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  static_assert(KC % 4 == 0, "whole 64-column passes");
  constexpr int KC2 = KC / 2;
  int stripe, g;
  if (!wg_map(a.groups, a.nst, stripe, g)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, m = lane & 15;
  int* W1s = reinterpret_cast<int*>(smem);          // [4][KC2] items of 768 dwords
  int* W2s = W1s + 4 * KC2 * 768;                   // [4][2]
  int* W3s = W2s + (THREE ? 8 * 768 : 0);           // [1][2]  (rows >= N3 zero)
  int* tab = W3s + 2 * 768;
  stage6<false, true, 64 * FNW>(W1s, a.W1 + g * a.gs_w1, a.K1, HD, a.K1, 4, KC2, a.x.k0, a.kpad);
  if (THREE) stage6<false, true, 64 * FNW>(W2s, a.W2 + g * a.gs_w2, HD, HD, HD, 4, 2, 1 << 30, 0);
  stage6<false, false, 64 * FNW>(W3s, a.W3 + g * a.gs_w3, HD, a.N3, HD, 1, 2, 1 << 30, 0);
  build_tab_nat(tab, a.x, a.KV, a.kpad, a.CF, KC, 64 * FNW);
  f32x4 b1v[4], b2v[4], b3v;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    b1v[t] = *reinterpret_cast<const f32x4*>(a.b1 + g * a.gs_b1 + permf(t, 4 * q));
    if (THREE) b2v[t] = *reinterpret_cast<const f32x4*>(a.b2 + g * a.gs_b2 + permf(t, 4 * q));
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) b3v[i] = 4 * q + i < a.N3 ? a.b3[g * a.gs_b3 + 4 * q + i] : 0.f;
  __syncthreads();

  const long tiles = (a.M + 15) / 16;
  const long per = (tiles + a.nst - 1) / a.nst;
  const long t_begin = (long)stripe * per;
  long t_end = t_begin + per; if (t_end > tiles) t_end = tiles;
  float* Y = a.Y + g * a.gs_y;
  
  f32x4 xv[KC];
  long tile = t_begin + wave;
  if (tile >= t_end) return;
  XRow xr = x_row(a.x, tile * 16 + m, a.M);
  x_issue_nat<KC, CFT>(xv, a.x, xr, tab, a.CF, lane);
  for (; tile < t_end; tile += FNW) {
    x_finish<KC, CFT>(xv, xr, tab, a.CF, lane);
    const bool live = (xr.flags & 1) != 0;
    float* y = Y + xr.rowc * a.ldy + 4 * q;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = b1v[t];
#pragma unroll
    for (int c2 = 0; c2 < KC2; ++c2) {
      const F3 xb = split8(xv[2 * c2], xv[2 * c2 + 1]);
      F3 wa[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) wa[t] = lds_f3(W1s, t * KC2 + c2, lane);
      mm6x4(wa, xb, acc);
      __builtin_amdgcn_sched_barrier(0);
    }
    {   // x is consumed: the next tile's loads (unconditional - the last iteration re-reads its own tile)
      const long nt = tile + FNW < t_end ? tile + FNW : tile;
      xr = x_row(a.x, nt * 16 + m, a.M);
      x_issue_nat<KC, CFT>(xv, a.x, xr, tab, a.CF, lane);
    }
    // kept for the backward: relu(h1), relu(h2) as the fp32 accumulator fragments, in the layout of the fp32 pair
    // ([head][tile][plane 0..7][lane] f32x4 - the two pairs' kept buffers are interchangeable)
    f32x4* hp = a.hs ? reinterpret_cast<f32x4*>(a.hs) + (((long)g * tiles + tile) * (THREE ? 8 : 4)) * 64 + lane : nullptr;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = relu4(acc[t]);
    if (hp) {
#pragma unroll
      for (int c = 0; c < 4; ++c) KEEP_ST(acc[c], hp + c * 64);
    }
    F3 hb[2];
    hb[0] = split8(acc[0], acc[1]);
    hb[1] = split8(acc[2], acc[3]);
    if (THREE) {
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = b2v[t];
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        F3 wa[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) wa[t] = lds_f3(W2s, t * 2 + c2, lane);
        mm6x4(wa, hb[c2], acc);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = relu4(acc[t]);
      if (hp) {
#pragma unroll
        for (int c = 0; c < 4; ++c) KEEP_ST(acc[c], hp + (4 + c) * 64);
      }
      hb[0] = split8(acc[0], acc[1]);
      hb[1] = split8(acc[2], acc[3]);
    }
    f32x4 o = b3v;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) mm6(lds_f3(W3s, c2, lane), hb[c2], o);
    if (live) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (4 * q + i < a.N3) y[i] = o[i];
    }
  }
}

// ------------------------------------------------------------------------------------------------- backward (kept activations)
// TWO workgroups of 4 waves per CU (<= 80 KB of LDS and 256 registers each), 64 rows per iteration (wave w: the 16-row tile 4 it + w):
// the phases below alternate between matrix-pipe work and LDS / VALU work with a barrier after each - two independent workgroups are
// never in the same phase for long (one workgroup of 8 waves over 128 rows per iteration: 1.69 ms where this form takes ...).
//   phase A (registers, per wave): dh2^T = relu'(h2) (W3^T dy^T), dh1^T = relu'(h1) (W2^T dh2^T) - the transposed chain with
//            pre-split W3^T / W2^T fragments from LDS; relu' from the kept fp32 activations; bias sums of layers 2, 3.
//   phase B (weight gradients, reduce over the 128 rows): operands through the [plane][128 rows][256 B] image, two 64-column
//            blocks per row:  [x columns 64 pt .. 64 pt + 63 | dh1]  (one pass per 64 columns of x: dW1; the bias gradient of layer 1
//            comes out of the same products through a ones column placed at the first free virtual column of x),
//            [h1 | dh2] (dW2), [h2 | dy] (dW3).  Wave w accumulates feature tile w x the four column tiles of a block.
// Image addressing: byte(plane, row, 16-byte chunk ch, sub) = plane * 32768 + row * 256 + 16 * (ch ^ swz(row)) + sub with
// swz(row) = ((row & 3) << 2) | ((row >> 2) & 3).  A lane's row (writes) and its row mod 32 (transposed reads) never change, and
// the chunk of a column tile is even, so every address is  (a per-lane constant) ^ (16 * chunk)  + plane / k-chunk offsets: the
// per-lane parts are computed once, outside the row loop (computed per access they were most of the kernel's vector instructions).
constexpr int XR = 64;                                // rows per iteration
constexpr int XNW = 4;
constexpr int PLS = XR * 256;                         // bytes per plane
// a fragment's 8 slots of the lane's row -> columns cb + 8q + (0..7): one 16-byte chunk per plane   (pb: the lane's put base, cb % 32 == 0)
__device__ __forceinline__ void img_put8(char* st, int pb, const F3& f, int cb) {
  const int a0 = pb ^ (16 * (cb >> 3));
  *reinterpret_cast<i32x4*>(st + a0) = f.h;
  *reinterpret_cast<i32x4*>(st + a0 + PLS) = f.m;
  *reinterpret_cast<i32x4*>(st + a0 + 2 * PLS) = f.l;
}
// four slots (dy: outputs 4q .. 4q + 3) -> columns cb + 4q + (0..3)   (pb4: the lane's base for 8-byte pieces)
__device__ __forceinline__ void img_put4(char* st, int pb4, const F3h& f, int cb) {
  const int a0 = pb4 ^ (16 * (cb >> 3));
  *reinterpret_cast<i32x2*>(st + a0) = f.h;
  *reinterpret_cast<i32x2*>(st + a0 + PLS) = f.m;
  *reinterpret_cast<i32x2*>(st + a0 + 2 * PLS) = f.l;
}
// operand fragment of a product that sums over the image's rows 32 kc + 8g + (0..7): lane (g, i) gets column c0 + i.
// t0 / t1: the lane's read bases of this column tile for rows 8g + (0..3) / 8g + (4..7)  (tr_base below)
__device__ __forceinline__ i32x4 img_tr(const char* st, int t0, int t1, int off) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(st + t0 + off));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(st + t1 + off));
  return __builtin_bit_cast(i32x4, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ F3 img_tr3(const char* st, int t0, int t1, int kc) {
  F3 f;
  f.h = img_tr(st, t0, t1, kc * 8192); f.m = img_tr(st, t0, t1, kc * 8192 + PLS); f.l = img_tr(st, t0, t1, kc * 8192 + 2 * PLS);
  return f;
}
__device__ __forceinline__ int tr_base(int lane, int c0, int half) {
  const int g = lane >> 4, i = lane & 15, qq = i >> 2, p = i & 3;
  const int row = 8 * g + qq + 4 * half;
  return row * 256 + 16 * (((c0 >> 3) + (p >> 1)) ^ (((row & 3) << 2) | ((row >> 2) & 3))) + 8 * (p & 1);
}

template <int KC, bool THREE, int CFT>
__global__ __launch_bounds__(64 * XNW, 2) void mlp3x6_bwd_kernel(Mlp3Args a) {      // (2 waves per SIMD = two workgroups per CU)
  static_assert(THREE, "three-layer heads only");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NP = KC / 4;                        // passes of 64 x columns
  int stripe, g;
  if (!wg_map(a.groups, a.nst, stripe, g)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, m = lane & 15;
  int* W2Ts = reinterpret_cast<int*>(smem);         // [4][2] items of 768 dwords: A[i = f1][slot <-> f2] = W2[f2][f1]
  int* W3Ts = W2Ts + 8 * 768;                       // [4 t][3 planes][64 lanes] 8 bytes: A[i = f2 = 16t + m][k = n3 = 4q + j] = W3[n3][f2]
  char* st = reinterpret_cast<char*>(W3Ts + 4 * 3 * 128);      // the image: 3 * 128 * 256 bytes
  int* tab = reinterpret_cast<int*>(st + 3 * PLS);
  const float* W2 = a.W2 + g * a.gs_w2;
  const float* W3 = a.W3 + g * a.gs_w3;
  stage6<true, true, 64 * XNW>(W2Ts, W2, HD, HD, HD, 4, 2, 1 << 30, 0);
  for (int e = tid; e < 4 * 64; e += 64 * XNW) {
    const int l = e & 63, t = e >> 6;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n3 = 4 * (l >> 4) + j;
      v[j] = n3 < a.N3 ? W3[(long)n3 * HD + permf(t, l & 15)] : 0.f;
    }
    const F3h f = split4(v);
    i32x2* p = reinterpret_cast<i32x2*>(W3Ts) + t * 192 + l;
    p[0] = f.h; p[64] = f.m; p[128] = f.l;
  }
  build_tab_nat(tab, a.x, a.KV, a.kpad, a.CF, KC, 64 * XNW);
  __syncthreads();

  const long its = (a.M + XR - 1) / XR;
  const long per = (its + a.nst - 1) / a.nst;
  const long i_begin = (long)stripe * per;
  long i_end = i_begin + per; if (i_end > its) i_end = its;
  const float* dY = a.Y + g * a.gs_y;
  const long tiles = (a.M + 15) / 16;
  const f32x4* hs = reinterpret_cast<const f32x4*>(a.hs);

  const int tf = wave;                              // feature tile of this wave's weight-gradient accumulators
  f32x4 dW1[NP][4], dW2[4], dW3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int u = 0; u < 4; ++u) dW1[p][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 4; ++u) dW2[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // the ones column: first virtual column past the input (K1 + kpad), there is one because 16 KC > KV (marl_mlp3_x6_supported)
  const int kone = a.KV;
  // per-lane address parts (see the note above)
  const int row = 16 * wave + m;                   // this lane's row of the image
  const int pb = row * 256 + 16 * (q ^ (((row & 3) << 2) | ((row >> 2) & 3)));                        // chunk q of a 32-column group
  const int pb4 = row * 256 + 16 * ((q >> 1) ^ (((row & 3) << 2) | ((row >> 2) & 3))) + 8 * (q & 1);   // 8-byte piece q of a 16-column group
  const int tB0 = tr_base(lane, 0, 0), tB1 = tr_base(lane, 0, 1);      // column tile 0; tile ct: ^ (32 * ct)  (the chunk of a tile is even)
  const int tA0 = tB0 ^ (32 * (4 + tf)), tA1 = tB1 ^ (32 * (4 + tf));  // dh1 / dh2: feature tile tf of the right block
  const int tY0 = tB0 ^ (32 * 4), tY1 = tB1 ^ (32 * 4);                // dy
  const int tH0 = tB0 ^ (32 * wave), tH1 = tB1 ^ (32 * wave);          // h2 column tile of this wave
  const int one_off = (row * 256 + 16 * (((kone & 63) >> 3) ^ (((row & 3) << 2) | ((row >> 2) & 3))) + 2 * (kone & 7));

  f32x4 xv[KC];
  XRow xr;
  f32x4 h1f[4], h2f[4];            // kept relu(h1), relu(h2) of this wave's tile (accumulator layout), loaded an iteration ahead
  f32x4 dyn;                       // dY[row m][4q .. 4q + 3]
  auto issue_x = [&](long it_) __attribute__((always_inline)) {
    xr = x_row(a.x, (it_ * XNW + wave) * 16 + m, a.M);
    x_issue_nat<KC, CFT>(xv, a.x, xr, tab, a.CF, lane);
    const bool lv = (xr.flags & 1) != 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) dyn[j] = (lv && 4 * q + j < a.N3) ? dY[xr.rowc * a.ldy + 4 * q + j] : 0.f;
  };
  auto issue_h = [&](long it_) __attribute__((always_inline)) {
    long tl = it_ * XNW + wave; if (tl > tiles - 1) tl = tiles - 1;      // (a tile past the end multiplies zero gradients)
    const f32x4* hp = hs + (((long)g * tiles + tl) * 8) * 64 + lane;
#pragma unroll
    for (int c = 0; c < 4; ++c) { h1f[c] = KEEP_LD(hp + c * 64); h2f[c] = KEEP_LD(hp + (4 + c) * 64); }
  };
  if (i_begin < i_end) { issue_x(i_begin); issue_h(i_begin); }
  // bias gradients of layers 2 and 3 ride on the weight-gradient products: A fragment x a fragment of ones (bf16 1.0 in every slot:
  // only the three products against its hi plane are non-zero) = the row sum of the A operand, in every column of the tile
  const i32x4 ones = {0x3F803F80, 0x3F803F80, 0x3F803F80, 0x3F803F80};
  f32x4 bs2 = {0.f, 0.f, 0.f, 0.f}, bs3 = {0.f, 0.f, 0.f, 0.f};
  // two accumulators of one A fragment against two B fragments, round robin; with ONES also the A operand's row sums
#define X6_MM2(AF, B0, B1, C0, C1)                                                                          \
  C0 = mm(AF.m, B0.m, C0); C1 = mm(AF.m, B1.m, C1); C0 = mm(AF.h, B0.l, C0); C1 = mm(AF.h, B1.l, C1);      \
  C0 = mm(AF.l, B0.h, C0); C1 = mm(AF.l, B1.h, C1); C0 = mm(AF.h, B0.m, C0); C1 = mm(AF.h, B1.m, C1);      \
  C0 = mm(AF.m, B0.h, C0); C1 = mm(AF.m, B1.h, C1); C0 = mm(AF.h, B0.h, C0); C1 = mm(AF.h, B1.h, C1);
  ST_DECL(15);
  for (long it = i_begin; it < i_end; ++it) {
    // ---------------- phase A
    x_finish<KC, CFT>(xv, xr, tab, a.CF, lane);
    const F3h dy3 = split4(dyn);
    f32x4 dh[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const i32x2* wp = reinterpret_cast<const i32x2*>(W3Ts) + t * 192 + lane;
      F3h w;
      w.h = wp[0]; w.m = wp[64]; w.l = wp[128];
      dh[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      mm6h(w, dy3, dh[t]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) dh[t][r] = h2f[t][r] > 0.f ? dh[t][r] : 0.f;
    F3 dh2f[2];
    dh2f[0] = split8(dh[0], dh[1]);
    dh2f[1] = split8(dh[2], dh[3]);
#pragma unroll
    for (int tp = 0; tp < 2; ++tp) {               // feature tiles two at a time (two accumulators round robin)
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        const F3 w0 = lds_f3(W2Ts, (2 * tp) * 2 + c2, lane), w1 = lds_f3(W2Ts, (2 * tp + 1) * 2 + c2, lane);
        const F3& b = dh2f[c2];
        a0 = mm(w0.m, b.m, a0); a1 = mm(w1.m, b.m, a1); a0 = mm(w0.h, b.l, a0); a1 = mm(w1.h, b.l, a1);
        a0 = mm(w0.l, b.h, a0); a1 = mm(w1.l, b.h, a1); a0 = mm(w0.h, b.m, a0); a1 = mm(w1.h, b.m, a1);
        a0 = mm(w0.m, b.h, a0); a1 = mm(w1.m, b.h, a1); a0 = mm(w0.h, b.h, a0); a1 = mm(w1.h, b.h, a1);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dh[2 * tp][r] = h1f[2 * tp][r] > 0.f ? a0[r] : 0.f;
        dh[2 * tp + 1][r] = h1f[2 * tp + 1][r] > 0.f ? a1[r] : 0.f;
      }
    }
    ST_MARK(0);
    // ---------------- image [h1 | dh2]: dW2 (and the layer-2 bias gradient)
    img_put8(st, pb, split8(h1f[0], h1f[1]), 0);
    img_put8(st, pb, split8(h1f[2], h1f[3]), 32);
    img_put8(st, pb, dh2f[0], 64);
    img_put8(st, pb, dh2f[1], 96);
    ST_MARK(1);
    WG_BARRIER();
    ST_MARK(2);
#pragma unroll
    for (int kc = 0; kc < XR / 32; ++kc) {
      const F3 af = img_tr3(st, tA0, tA1, kc);
      bs2 = mm(af.l, ones, bs2); bs2 = mm(af.m, ones, bs2); bs2 = mm(af.h, ones, bs2);
#pragma unroll
      for (int cp = 0; cp < 2; ++cp) {
        const F3 b0 = img_tr3(st, tB0 ^ (32 * (2 * cp)), tB1 ^ (32 * (2 * cp)), kc);
        const F3 b1 = img_tr3(st, tB0 ^ (32 * (2 * cp + 1)), tB1 ^ (32 * (2 * cp + 1)), kc);
        X6_MM2(af, b0, b1, dW2[2 * cp], dW2[2 * cp + 1])
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    ST_MARK(3);
    WG_BARRIER();
    ST_MARK(4);
    // ---------------- image [h2 | dy]: dW3 (and the layer-3 bias gradient)
    img_put8(st, pb, split8(h2f[0], h2f[1]), 0);
    img_put8(st, pb, split8(h2f[2], h2f[3]), 32);
    img_put4(st, pb4, dy3, 64);
#ifndef X6_EXP_NOH
    issue_h(it + 1 < i_end ? it + 1 : it);          // kept activations of the next iteration's tile (land during the dW1 passes)
#endif
    ST_MARK(5);
    WG_BARRIER();
    ST_MARK(6);
#pragma unroll
    for (int kc = 0; kc < XR / 32; ++kc) {
      const F3 af = img_tr3(st, tY0, tY1, kc);
      const F3 bf = img_tr3(st, tH0, tH1, kc);
      mm6(af, bf, dW3);
      bs3 = mm(af.l, ones, bs3); bs3 = mm(af.m, ones, bs3); bs3 = mm(af.h, ones, bs3);
    }
    ST_MARK(7);
    WG_BARRIER();
    ST_MARK(8);
    // ---------------- images [x columns 64 pt .. | dh1]: dW1 (the layer-1 bias gradient through the ones column of x)
    img_put8(st, pb, split8(dh[0], dh[1]), 64);
    img_put8(st, pb, split8(dh[2], dh[3]), 96);
    ST_MARK(12);
#pragma unroll
    for (int pt = 0; pt < NP; ++pt) {
#ifndef X6_EXP_NOPUTX
      img_put8(st, pb, split8(xv[4 * pt], xv[4 * pt + 1]), 0);
      img_put8(st, pb, split8(xv[4 * pt + 2], xv[4 * pt + 3]), 32);
#endif
      ST_MARK(13);
      if ((kone >> 6) == pt && q == 0) {              // ones column (same wave, later instruction: lands after the packed write)
        *reinterpret_cast<short*>(st + one_off) = (short)0x3F80;
        *reinterpret_cast<short*>(st + one_off + PLS) = 0;
        *reinterpret_cast<short*>(st + one_off + 2 * PLS) = 0;
      }
      ST_MARK(14);
      // x is consumed: the next iteration's tile and dY elements (unconditional; the last one re-reads its own)
#ifndef X6_EXP_NOX
      if (pt == NP - 1) issue_x(it + 1 < i_end ? it + 1 : it);
#endif
      ST_MARK(9);
      WG_BARRIER();
      ST_MARK(10);
#pragma unroll
      for (int kc = 0; kc < XR / 32; ++kc) {
        const F3 af = img_tr3(st, tA0, tA1, kc);
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
          const F3 b0 = img_tr3(st, tB0 ^ (32 * (2 * cp)), tB1 ^ (32 * (2 * cp)), kc);
          const F3 b1 = img_tr3(st, tB0 ^ (32 * (2 * cp + 1)), tB1 ^ (32 * (2 * cp + 1)), kc);
          X6_MM2(af, b0, b1, dW1[pt][2 * cp], dW1[pt][2 * cp + 1])
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      WG_BARRIER();
      ST_MARK(11);
    }
  }
#undef X6_MM2
  if (KC == 8) { ST_DUMP(15); }

  // ---------------- slab: [dW1 64 x (K1+1) | dW2 64 x 65 | dW3 16 x 65], bias gradient in the last column
  const int K1x = a.K1 + 1;
  float* s1 = a.ws + ((long)stripe * a.groups + g) * mlp3_slab_floats(a.K1, a.N3);
  float* s2 = s1 + (long)HD * K1x;
  float* s3 = s2 + (long)HD * (HD + 1);
#pragma unroll
  for (int pt = 0; pt < NP; ++pt)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int vc = 64 * pt + 16 * u + m;                          // virtual column of this lane
      const int k = vc == kone ? a.K1 : vcol(vc, a.x.k0, a.kpad);   // (the ones column carries the bias gradient)
      const bool ok = vc == kone || (k >= 0 && k < a.K1 && vc < a.KV);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (ok) s1[(long)(16 * tf + 4 * q + r) * K1x + k] = dW1[pt][u][r];
    }
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) s2[(16 * tf + 4 * q + r) * (HD + 1) + 16 * u + m] = dW2[u][r];
#pragma unroll
  for (int r = 0; r < 4; ++r) s3[(4 * q + r) * (HD + 1) + 16 * wave + m] = dW3[r];
  if (m == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s2[(16 * tf + 4 * q + r) * (HD + 1) + HD] = bs2[r];
      if (wave == 0) s3[(4 * q + r) * (HD + 1) + HD] = bs3[r];
    }
  }
}

inline size_t x6_fwd_lds(int KC, int CF, bool three) {
  const int KC2 = KC / 2;
  return (size_t)(4 * KC2 + (three ? 8 : 0) + 2) * 768 * 4 + (size_t)(KC - CF) * 32 * 4;
}
inline size_t x6_bwd_lds(int KC, int CF) { return (size_t)8 * 768 * 4 + 4 * 3 * 128 * 4 + (size_t)3 * XR * 256 + (size_t)(KC - CF) * 32 * 4; }

}  // namespace

ST_DEFINE_SETTER(marl_debug_stamps_mlp3x6)

// three-layer heads with up to 16 outputs and up to 192 input columns whose padded width leaves a free column for the ones column
extern "C" int marl_mlp3_x6_supported(const marl_src_t* x, int K1, int H1, int H2, int N3, int groups) {
  if (!marl_mlp3_supported(x, K1, H1, H2, N3, groups)) return 0;
  if (H2 != HD || N3 > 16) return 0;
  const int KV = K1 + kpad_of(x);
  const int KC = kc_bucket_x6(KV);
  if (KC > 12 || KV >= 16 * KC) return 0;
  const int CF = lead_chunks_nat(x);
  return x6_fwd_lds(KC, CF, true) <= 160 * 1024 && x6_bwd_lds(KC, CF) <= 160 * 1024;
}

extern "C" int marl_mlp3_x6_fwd_save(const marl_mlp3_weights_t* w, const marl_src_t* x, float* Y, long ldy, long gs_y,
                                     float* hsave, size_t hsave_floats, long M, int K1, int N3, int groups, void* stream) {
  if (M <= 0) return 0;
  if (!w->w2 || !marl_mlp3_x6_supported(x, K1, HD, HD, N3, groups)) return (int)hipErrorInvalidValue;
  if (hsave && (hsave_floats < marl_mlp3_save_floats(M, 1, groups) || !aligned16(hsave))) return (int)hipErrorInvalidValue;
  if (!aligned16(w->b1) || w->gs_b1 % 4 || !aligned16(w->b2) || w->gs_b2 % 4) return (int)hipErrorInvalidValue;
  Mlp3Args a;
  if (!fill_args(a, w, x, M, K1, N3, groups)) return (int)hipErrorInvalidValue;
  a.Y = Y; a.ldy = ldy; a.gs_y = gs_y; a.ws = nullptr; a.hs = hsave;
  const long tiles = (M + 15) / 16;
  a.nst = stripes((tiles + FNW - 1) / FNW, groups);
  const int KC = kc_bucket_x6(a.KV);
  a.CF = lead_chunks_nat(x);
  const size_t lds = x6_fwd_lds(KC, a.CF, true);
// (six leading full chunks = a 120-wide dense segment 0: the QPLEX heads on 2s3z-sized maps get the compile-time variants)
#define X6_PICK(K) (KC == 4 ? (const void*)K<4, true, -1> : KC == 8 ? (a.CF == 6 ? (const void*)K<8, true, 6> : (const void*)K<8, true, -1>) \
                    : (a.CF == 6 ? (const void*)K<12, true, 6> : (const void*)K<12, true, -1>))
  const void* fn = X6_PICK(mlp3x6_fwd_kernel);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  dim3 grid((unsigned)((a.nst + 7) / 8 * 8 * groups)), block(64 * FNW);
  void* kargs[] = {(void*)&a};
  e = hipLaunchKernel(fn, grid, block, kargs, lds, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  MARL_CHECK_LAUNCH();
  return 0;
}

extern "C" int marl_mlp3_x6_bwd_saved(const marl_mlp3_weights_t* w, const marl_src_t* x, const float* dY, long lddy, long gs_dy,
                                      const marl_mlp3_weights_t* grads, float* ws, size_t ws_bytes, const float* hsave,
                                      size_t hsave_floats, long M, int K1, int N3, int groups, void* stream) {
  if (M <= 0) return 0;
  if (!w->w2 || !grads->w2 || !hsave || !marl_mlp3_x6_supported(x, K1, HD, HD, N3, groups)) return (int)hipErrorInvalidValue;
  if (hsave_floats < marl_mlp3_save_floats(M, 1, groups) || !aligned16(hsave)) return (int)hipErrorInvalidValue;
  if (ws_bytes < marl_mlp3_bwd_workspace(M, K1, N3, groups)) return (int)hipErrorInvalidValue;
  Mlp3Args a;
  if (!fill_args(a, w, x, M, K1, N3, groups)) return (int)hipErrorInvalidValue;
  a.Y = const_cast<float*>(dY); a.ldy = lddy; a.gs_y = gs_dy; a.ws = ws; a.hs = const_cast<float*>(hsave);
  const int KC = kc_bucket_x6(a.KV);
  a.CF = lead_chunks_nat(x);
  a.nst = stripes((M + XR - 1) / XR, groups, 512);      // two workgroups per CU
  const size_t lds = x6_bwd_lds(KC, a.CF);
  const void* fn = X6_PICK(mlp3x6_bwd_kernel);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  dim3 grid((unsigned)((a.nst + 7) / 8 * 8 * groups)), block(64 * XNW);
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
