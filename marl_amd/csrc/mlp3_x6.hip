// Fused three-layer heads  y = W3 relu(W2 relu(W1 x + b1) + b2) + b3  (hidden width 64) with the multiplies done as an fp32-accurate
// SPLIT on the bf16 matrix cores ("bf16x6", opt-in: args.gemm_mode = "bf16x6"; the default path is mlp3_fused.hip on
// v_mfma_f32_16x16x4_f32).  Reference: the key / agents / action extractors of QPLEX's lambda-net, network/mixer.py:117-145,
// evaluated at :155-169.
//
// Every fp32 operand element is split EXACTLY into three bf16 terms, a = hi + mid + lo (round-to-nearest each: 8 + 8 + 8 significand
// bits), and a product a.b is the six bf16 products  mid.mid, hi.lo, lo.hi, hi.mid, mid.hi, hi.hi  accumulated in fp32 by
// v_mfma_f32_16x16x32_bf16 (smallest first); the dropped terms (mid.lo, lo.mid, lo.lo) are <= 2^-24 of |a.b|.  Measured against fp64:
// error / output scale 4.7e-7 at K = 64 where the fp32 MFMA has 6.8e-7 (profiles/r03_bf16x3_probe.txt).  Six such MFMAs cover a
// 16 x 16 x 32 block in ~100 pipe cycles; the eight v_mfma_f32_16x16x4_f32 of the same block take 256.
//
// Layouts (checked with integer data by tools/probe/x6_layout_probe.hip, profiles/r04_x6_layout_probe.txt):
//   * v_mfma_f32_16x16x32_bf16: lane (g = l >> 4, i = l & 15) holds A[i][slot j] and B[slot j][i], j = 0..7; D register r =
//     C[4g + r][i].  The slot -> k assignment is free as long as both operands use the same one.
//   * transposed formulation (as mlp3_fused.hip): out^T[feature][row] = W[feature][k] in^T[k][row]; the WEIGHTS are the A operand
//     (fragments pre-split once per workgroup, in LDS), the activations the B operand.  Two accumulator tiles of one layer
//     (features 32c + 4g + r and 32c + 16 + 4g + r of row i) ARE the 8 k-slots of the next layer's chunk c: slot j < 4 <->
//     k = 32c + 4g + j, slot j >= 4 <-> k = 32c + 16 + 4g + (j - 4) - the weight fragments are staged in that order.
//   * weight gradients reduce over ROWS: operands go through a [row][column] bf16 LDS image (each lane packs four consecutive
//     columns of its row: one 8-byte write per plane) and come back with ds_read_b64_tr_b16, which hands lane (g, i) the column i of
//     rows 8g .. 8g + 3 - the A / B fragment of a product that sums over the image's rows.  256-byte image rows, 16-byte chunks
//     XOR-swizzled by the row (cdna_hip_programming.md T10, image (b)).
#include "mlp3_common.h"

namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

struct F3 { i32x4 h, m, l; };       // 8 k-slots per lane, three bf16 planes
struct F3h { i32x2 h, m, l; };      // 4 k-slots per lane (v_mfma_f32_16x16x16_bf16: k = 4g + j)

// two fp32 values -> packed bf16 pairs of their hi / mid / lo terms (low half = first value); plain casts: hipcc emits
// v_cvt_pk_bf16_f32 (round to nearest even, NaN stays NaN)
__device__ __forceinline__ unsigned pk2(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf2));
}
__device__ __forceinline__ void split2(float x0, float x1, int& h, int& m, int& l) {
  const unsigned hp = pk2(x0, x1);
  const float r0 = x0 - __uint_as_float(hp << 16), r1 = x1 - __uint_as_float(hp & 0xffff0000u);      // exact
  const unsigned mp = pk2(r0, r1);
  const float s0 = r0 - __uint_as_float(mp << 16), s1 = r1 - __uint_as_float(mp & 0xffff0000u);      // exact, <= 8 bits left
  h = (int)hp; m = (int)mp; l = (int)pk2(s0, s1);
}
__device__ __forceinline__ F3h split4(const f32x4& a) {
  int h0, m0, l0, h1, m1, l1;
  split2(a[0], a[1], h0, m0, l0);
  split2(a[2], a[3], h1, m1, l1);
  F3h f;
  f.h = (i32x2){h0, h1}; f.m = (i32x2){m0, m1}; f.l = (i32x2){l0, l1};
  return f;
}
// slots 0..3 = a, slots 4..7 = b
__device__ __forceinline__ F3 split8(const f32x4& a, const f32x4& b) {
  const F3h x = split4(a), y = split4(b);
  F3 f;
  f.h = (i32x4){x.h[0], x.h[1], y.h[0], y.h[1]};
  f.m = (i32x4){x.m[0], x.m[1], y.m[0], y.m[1]};
  f.l = (i32x4){x.l[0], x.l[1], y.l[0], y.l[1]};
  return f;
}

__device__ __forceinline__ f32x4 mm(const i32x4& a, const i32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mmh(const i32x2& a, const i32x2& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}
// the six products, smallest first
__device__ __forceinline__ void mm6(const F3& a, const F3& b, f32x4& c) {
  c = mm(a.m, b.m, c); c = mm(a.h, b.l, c); c = mm(a.l, b.h, c);
  c = mm(a.h, b.m, c); c = mm(a.m, b.h, c); c = mm(a.h, b.h, c);
}
// four A fragments against one B fragment, the four accumulators round robin
__device__ __forceinline__ void mm6x4(const F3 (&a)[4], const F3& b, f32x4 (&c)[4]) {
#pragma unroll
  for (int t = 0; t < 4; ++t) c[t] = mm(a[t].m, b.m, c[t]);
#pragma unroll
  for (int t = 0; t < 4; ++t) c[t] = mm(a[t].h, b.l, c[t]);
#pragma unroll
  for (int t = 0; t < 4; ++t) c[t] = mm(a[t].l, b.h, c[t]);
#pragma unroll
  for (int t = 0; t < 4; ++t) c[t] = mm(a[t].h, b.m, c[t]);
#pragma unroll
  for (int t = 0; t < 4; ++t) c[t] = mm(a[t].m, b.h, c[t]);
#pragma unroll
  for (int t = 0; t < 4; ++t) c[t] = mm(a[t].h, b.h, c[t]);
}
__device__ __forceinline__ void mm6h(const F3h& a, const F3h& b, f32x4& c) {
  c = mmh(a.m, b.m, c); c = mmh(a.h, b.l, c); c = mmh(a.l, b.h, c);
  c = mmh(a.h, b.m, c); c = mmh(a.m, b.h, c); c = mmh(a.h, b.h, c);
}

// weight fragments in LDS: item (tile t, chunk c2) = [3 planes][64 lanes] 16 bytes
__device__ __forceinline__ F3 lds_f3(const int* base, int item, int lane) {
  const i32x4* p = reinterpret_cast<const i32x4*>(base) + item * 192 + lane;
  F3 f;
  f.h = p[0]; f.m = p[64]; f.l = p[128];
  return f;
}
// A fragments of W (TR = false: A[i = row 16t + m of W][slot <-> virtual column of W]) or of W^T (TR = true: A[i = column 16t + m of
// W][slot <-> row of W]), split into planes.  Slots: j < 4 <-> 32 c2 + 4q + j, j >= 4 <-> 32 c2 + 16 + 4q + (j - 4).
template <bool TR, int NTHR>
__device__ __forceinline__ void stage6(int* dst, const float* W, int ldw, int rows_valid, int K, int T, int KC2, int k0, int kpad) {
  for (int e = threadIdx.x; e < T * KC2 * 64; e += NTHR) {
    const int l = e & 63, tc = e >> 6, t = tc / KC2, c2 = tc - t * KC2;
    const int n = 16 * t + (l & 15), qq = l >> 4;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int col = 32 * c2 + (j < 4 ? 0 : 16) + 4 * qq + (j & 3);
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

template <int KC>
__device__ __forceinline__ f32x4 xv_or_zero(const f32x4 (&xv)[KC], int c) {      // (c is a constant once the caller's loop is unrolled)
  return c < KC ? xv[c < KC ? c : 0] : (f32x4){0.f, 0.f, 0.f, 0.f};
}

__host__ __device__ inline long x6_save_floats_per_tile(bool three) { return (three ? 2 : 1) * 2 * 3 * 256; }

// ------------------------------------------------------------------------------------------------- forward
// 8 waves, each walks its own 16-row tiles (x of the next tile in flight); hidden activations never leave the wave.  a.hs != NULL:
// the split planes of relu(h1), relu(h2) - the B fragments this kernel forms anyway - are kept for the backward,
// [head][tile][activation][chunk][plane][lane] 16 bytes (768 bytes per row and head).
template <int KC, bool THREE, int CFT>
__global__ __launch_bounds__(64 * FNW, 2) void mlp3x6_fwd_kernel(Mlp3Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int KC2 = (KC + 1) / 2;
  int stripe, g;
  if (!wg_map(a.groups, a.nst, stripe, g)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, m = lane & 15;
  int* W1s = reinterpret_cast<int*>(smem);          // [4][KC2] items of 768 dwords
  int* W2s = W1s + 4 * KC2 * 768;                   // [4][2]
  int* W3s = W2s + (THREE ? 8 * 768 : 0);           // [1][2]  (rows >= N3 zero)
  int* tab = W3s + 2 * 768;
  stage6<false, 64 * FNW>(W1s, a.W1 + g * a.gs_w1, a.K1, HD, a.K1, 4, KC2, a.x.k0, a.kpad);
  if (THREE) stage6<false, 64 * FNW>(W2s, a.W2 + g * a.gs_w2, HD, HD, HD, 4, 2, 1 << 30, 0);
  stage6<false, 64 * FNW>(W3s, a.W3 + g * a.gs_w3, HD, a.N3, HD, 1, 2, 1 << 30, 0);
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
  constexpr int NA = THREE ? 2 : 1;

  f32x4 xv[KC];
  long tile = t_begin + wave;
  if (tile >= t_end) return;
  XRow xr = x_row(a.x, tile * 16 + m, a.M);
  x_issue<KC, CFT>(xv, a.x, xr, tab, a.CF, lane);
  for (; tile < t_end; tile += FNW) {
    x_finish<KC, CFT>(xv, xr, tab, a.CF, lane);
    const bool live = (xr.flags & 1) != 0;
    float* y = Y + xr.rowc * a.ldy + 4 * q;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = b1v[t];
#pragma unroll
    for (int c2 = 0; c2 < KC2; ++c2) {
      const F3 xb = split8(xv[2 * c2], xv_or_zero<KC>(xv, 2 * c2 + 1));
      F3 wa[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) wa[t] = lds_f3(W1s, t * KC2 + c2, lane);
      mm6x4(wa, xb, acc);
      __builtin_amdgcn_sched_barrier(0);
    }
    {   // x is consumed: the next tile's loads (unconditional - the last iteration re-reads its own tile)
      const long nt = tile + FNW < t_end ? tile + FNW : tile;
      xr = x_row(a.x, nt * 16 + m, a.M);
      x_issue<KC, CFT>(xv, a.x, xr, tab, a.CF, lane);
    }
    F3 hb[2];
    hb[0] = split8(relu4(acc[0]), relu4(acc[1]));
    hb[1] = split8(relu4(acc[2]), relu4(acc[3]));
    i32x4* hp = a.hs ? reinterpret_cast<i32x4*>(a.hs) + (((long)g * tiles + tile) * NA) * 384 + lane : nullptr;
    if (hp) {
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        KEEP_ST(hb[c2].h, hp + c2 * 192); KEEP_ST(hb[c2].m, hp + c2 * 192 + 64); KEEP_ST(hb[c2].l, hp + c2 * 192 + 128);
      }
    }
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
      hb[0] = split8(relu4(acc[0]), relu4(acc[1]));
      hb[1] = split8(relu4(acc[2]), relu4(acc[3]));
      if (hp) {
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
          KEEP_ST(hb[c2].h, hp + 384 + c2 * 192); KEEP_ST(hb[c2].m, hp + 384 + c2 * 192 + 64); KEEP_ST(hb[c2].l, hp + 384 + c2 * 192 + 128);
        }
      }
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
// One workgroup of 8 waves per CU, 128 rows per iteration (wave w: the 16-row tile 8 it + w).
//   phase A (registers, per wave): dh2^T = relu'(h2) (W3^T dy^T), dh1^T = relu'(h1) (W2^T dh2^T) - the transposed chain with
//            pre-split W3^T / W2^T fragments from LDS; relu' from the sign of the kept hi plane; bias sums of layers 2, 3.
//   phase B (weight gradients, reduce over the 128 rows): operands through the [plane][128 rows][256 B] image, two 64-column
//            blocks per row:  [x columns 64 pt .. 64 pt + 63 | dh1]  (one pass per 64 columns of x: dW1; the bias gradient of layer 1
//            comes out of the same products through a ones column placed at the first free virtual column of x),
//            [h1 | dh2] (dW2), [h2 | dy] (dW3).  Wave w accumulates feature tile w & 3 x column tiles 2 (w >> 2), 2 (w >> 2) + 1.
constexpr int XR = 128;                               // rows per iteration
constexpr int XNW = 8;
__device__ __forceinline__ int st_off(int pl, int row, int ch, int sub) {
  return pl * (XR * 256) + row * 256 + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) + sub;
}
// a fragment's 8 slots of row `row` -> columns cb + 4q + (0..3) and cb + 16 + 4q + (0..3) of the image (cb a multiple of 32)
__device__ __forceinline__ void img_put8(char* st, const F3& f, int row, int cb, int q) {
  const int c0 = cb + 4 * q, c1 = c0 + 16, sub = 8 * (q & 1);
  const i32x4 pl[3] = {f.h, f.m, f.l};
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    *reinterpret_cast<i32x2*>(st + st_off(p, row, c0 >> 3, sub)) = (i32x2){pl[p][0], pl[p][1]};
    *reinterpret_cast<i32x2*>(st + st_off(p, row, c1 >> 3, sub)) = (i32x2){pl[p][2], pl[p][3]};
  }
}
__device__ __forceinline__ void img_put4(char* st, const F3h& f, int row, int cb, int q) {
  const int c0 = cb + 4 * q, sub = 8 * (q & 1);
  *reinterpret_cast<i32x2*>(st + st_off(0, row, c0 >> 3, sub)) = f.h;
  *reinterpret_cast<i32x2*>(st + st_off(1, row, c0 >> 3, sub)) = f.m;
  *reinterpret_cast<i32x2*>(st + st_off(2, row, c0 >> 3, sub)) = f.l;
}
// operand fragment of a product that sums over the image's rows rb + 8g + (0..7): lane (g, i) gets column c0 + i
__device__ __forceinline__ i32x4 img_tr(const char* st, int pl, int rb, int c0, int lane) {
  const int g = lane >> 4, i = lane & 15, qq = i >> 2, p = i & 3;
  const int r0 = rb + 8 * g + qq;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(st + st_off(pl, r0, (c0 >> 3) + (p >> 1), 8 * (p & 1))));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(st + st_off(pl, r0 + 4, (c0 >> 3) + (p >> 1), 8 * (p & 1))));
  return __builtin_bit_cast(i32x4, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ F3 img_tr3(const char* st, int rb, int c0, int lane) {
  F3 f;
  f.h = img_tr(st, 0, rb, c0, lane); f.m = img_tr(st, 1, rb, c0, lane); f.l = img_tr(st, 2, rb, c0, lane);
  return f;
}
// relu'(h) from the kept hi plane: slot s of the fragment (bf16 in half s & 1 of dword s >> 1) is positive
__device__ __forceinline__ bool slot_pos(const i32x4& hi, int s) {
  const unsigned d = (unsigned)hi[s >> 1];
  return (int)((s & 1) ? (d & 0xffff0000u) : (d << 16)) > 0;
}

template <int KC, bool THREE, int CFT>
__global__ __launch_bounds__(64 * XNW, 2) void mlp3x6_bwd_kernel(Mlp3Args a) {
  static_assert(THREE, "three-layer heads only");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NP = (KC + 3) / 4;                  // passes of 64 x columns
  int stripe, g;
  if (!wg_map(a.groups, a.nst, stripe, g)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, m = lane & 15;
  int* W2Ts = reinterpret_cast<int*>(smem);         // [4][2] items of 768 dwords: A[i = f1][slot <-> f2] = W2[f2][f1]
  int* W3Ts = W2Ts + 8 * 768;                       // [4 t][3 planes][64 lanes] 8 bytes: A[i = f2 = 16t + m][k = n3 = 4q + j] = W3[n3][f2]
  char* st = reinterpret_cast<char*>(W3Ts + 4 * 3 * 128);      // the image: 3 * 128 * 256 bytes
  int* tab = reinterpret_cast<int*>(st + 3 * XR * 256);
  const float* W2 = a.W2 + g * a.gs_w2;
  const float* W3 = a.W3 + g * a.gs_w3;
  stage6<true, 64 * XNW>(W2Ts, W2, HD, HD, HD, 4, 2, 1 << 30, 0);
  for (int e = tid; e < 4 * 64; e += 64 * XNW) {
    const int l = e & 63, t = e >> 6;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n3 = 4 * (l >> 4) + j;
      v[j] = n3 < a.N3 ? W3[(long)n3 * HD + 16 * t + (l & 15)] : 0.f;
    }
    const F3h f = split4(v);
    i32x2* p = reinterpret_cast<i32x2*>(W3Ts) + t * 192 + l;
    p[0] = f.h; p[64] = f.m; p[128] = f.l;
  }
  build_tab(tab, a.x, a.KV, a.kpad, a.CF, KC, 64 * XNW);
  __syncthreads();

  const long its = (a.M + XR - 1) / XR;
  const long per = (its + a.nst - 1) / a.nst;
  const long i_begin = (long)stripe * per;
  long i_end = i_begin + per; if (i_end > its) i_end = its;
  const float* dY = a.Y + g * a.gs_y;
  const long tiles = (a.M + 15) / 16;
  const i32x4* hs = reinterpret_cast<const i32x4*>(a.hs);

  const int tf = wave & 3, chh = wave >> 2;        // feature tile, column-tile pair of this wave's weight-gradient accumulators
  f32x4 dW1[NP][2], dW2[2], dW3 = {0.f, 0.f, 0.f, 0.f};
  f32x4 bs2[4];
  f32x4 bs3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < NP; ++p) { dW1[p][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; dW1[p][1] = dW1[p][0]; }
  dW2[0] = dW2[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 4; ++t) bs2[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // the ones column: first virtual column past the input (K1 + kpad), there is one because 16 KC > KV (marl_mlp3_x6_supported)
  const int kone = a.KV;

  f32x4 xv[KC];
  XRow xr;
  i32x4 hi1[2], hi2[2];            // kept hi planes of this wave's tile (relu masks), loaded an iteration ahead
  f32x4 dyn;                       // dY[row m][4q .. 4q + 3]
  auto issue_next = [&](long it_) __attribute__((always_inline)) {
    xr = x_row(a.x, (it_ * XNW + wave) * 16 + m, a.M);
    x_issue<KC, CFT>(xv, a.x, xr, tab, a.CF, lane);
    const bool lv = (xr.flags & 1) != 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) dyn[j] = (lv && 4 * q + j < a.N3) ? dY[xr.rowc * a.ldy + 4 * q + j] : 0.f;
    long tl = it_ * XNW + wave; if (tl > tiles - 1) tl = tiles - 1;
    const i32x4* hp = hs + (((long)g * tiles + tl) * 2) * 384 + lane;
    hi1[0] = KEEP_LD(hp); hi1[1] = KEEP_LD(hp + 192);
    hi2[0] = KEEP_LD(hp + 384); hi2[1] = KEEP_LD(hp + 384 + 192);
  };
  if (i_begin < i_end) issue_next(i_begin);
  const int row = 16 * wave + m;                   // this lane's row of the image
  for (long it = i_begin; it < i_end; ++it) {
    // ---------------- phase A
    x_finish<KC, CFT>(xv, xr, tab, a.CF, lane);
    long tl = it * XNW + wave; if (tl > tiles - 1) tl = tiles - 1;
    const i32x4* hp = hs + (((long)g * tiles + tl) * 2) * 384 + lane;
    const F3h dy3 = split4(dyn);
    bs3 += dyn;
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
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) dh[t][r] = slot_pos(hi2[t >> 1], (t & 1) * 4 + r) ? dh[t][r] : 0.f;
      bs2[t] += dh[t];
    }
    F3 dh2f[2];
    dh2f[0] = split8(dh[0], dh[1]);
    dh2f[1] = split8(dh[2], dh[3]);
    {
      f32x4 acc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        F3 wa[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) wa[t] = lds_f3(W2Ts, t * 2 + c2, lane);
        mm6x4(wa, dh2f[c2], acc);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dh[t][r] = slot_pos(hi1[t >> 1], (t & 1) * 4 + r) ? acc[t][r] : 0.f;
    }
    // image: [x columns of pass 0 | dh1]
    img_put8(st, split8(dh[0], dh[1]), row, 64, q);
    img_put8(st, split8(dh[2], dh[3]), row, 96, q);
#pragma unroll
    for (int pt = 0; pt < NP; ++pt) {
      img_put8(st, split8(xv_or_zero<KC>(xv, 4 * pt), xv_or_zero<KC>(xv, 4 * pt + 1)), row, 0, q);
      img_put8(st, split8(xv_or_zero<KC>(xv, 4 * pt + 2), xv_or_zero<KC>(xv, 4 * pt + 3)), row, 32, q);
      if ((kone >> 6) == pt && q == 0) {              // ones column (same wave, later instruction: lands after the packed write)
        const int c = kone & 63;
        *reinterpret_cast<short*>(st + st_off(0, row, c >> 3, 2 * (c & 7))) = (short)0x3F80;
        *reinterpret_cast<short*>(st + st_off(1, row, c >> 3, 2 * (c & 7))) = 0;
        *reinterpret_cast<short*>(st + st_off(2, row, c >> 3, 2 * (c & 7))) = 0;
      }
      if (pt == NP - 1) {
        // x is consumed: the next iteration's tile, dY elements and mask planes (unconditional; the last one re-reads its own)
        issue_next(it + 1 < i_end ? it + 1 : it);
      }
      WG_BARRIER();
      // dW1[feature tile tf][columns 64 pt + 16 (2 chh + u)] += dh1^T x over the 128 rows
#pragma unroll
      for (int kc = 0; kc < XR / 32; ++kc) {
        const F3 af = img_tr3(st, 32 * kc, 64 + 16 * tf, lane);
        const F3 b0 = img_tr3(st, 32 * kc, 16 * (2 * chh), lane);
        const F3 b1 = img_tr3(st, 32 * kc, 16 * (2 * chh + 1), lane);
        mm6(af, b0, dW1[pt][0]);
        mm6(af, b1, dW1[pt][1]);
      }
      WG_BARRIER();
    }
    // image: [h1 | dh2]   (all three planes of the kept h1)
    {
      F3 f0, f1;
      // (hi1 / hi2 already hold the NEXT tile's planes: all three are read again - L2 hits)
      f0.h = KEEP_LD(hp); f0.m = KEEP_LD(hp + 64); f0.l = KEEP_LD(hp + 128);
      f1.h = KEEP_LD(hp + 192); f1.m = KEEP_LD(hp + 192 + 64); f1.l = KEEP_LD(hp + 192 + 128);
      img_put8(st, f0, row, 0, q);
      img_put8(st, f1, row, 32, q);
      img_put8(st, dh2f[0], row, 64, q);
      img_put8(st, dh2f[1], row, 96, q);
    }
    WG_BARRIER();
#pragma unroll
    for (int kc = 0; kc < XR / 32; ++kc) {
      const F3 af = img_tr3(st, 32 * kc, 64 + 16 * tf, lane);
      const F3 b0 = img_tr3(st, 32 * kc, 16 * (2 * chh), lane);
      const F3 b1 = img_tr3(st, 32 * kc, 16 * (2 * chh + 1), lane);
      mm6(af, b0, dW2[0]);
      mm6(af, b1, dW2[1]);
    }
    WG_BARRIER();
    // image: [h2 | dy]
    {
      F3 f0, f1;
      f0.h = KEEP_LD(hp + 384); f0.m = KEEP_LD(hp + 384 + 64); f0.l = KEEP_LD(hp + 384 + 128);
      f1.h = KEEP_LD(hp + 384 + 192); f1.m = KEEP_LD(hp + 384 + 192 + 64); f1.l = KEEP_LD(hp + 384 + 192 + 128);
      img_put8(st, f0, row, 0, q);
      img_put8(st, f1, row, 32, q);
      img_put4(st, dy3, row, 64, q);
    }
    WG_BARRIER();
    if (wave < 4) {
#pragma unroll
      for (int kc = 0; kc < XR / 32; ++kc) {
        const F3 af = img_tr3(st, 32 * kc, 64, lane);
        const F3 bf = img_tr3(st, 32 * kc, 16 * wave, lane);
        mm6(af, bf, dW3);
      }
    }
    WG_BARRIER();
  }

  // ---------------- slab: [dW1 64 x (K1+1) | dW2 64 x 65 | dW3 16 x 65], bias gradient in the last column
  const int K1x = a.K1 + 1;
  float* s1 = a.ws + ((long)stripe * a.groups + g) * mlp3_slab_floats(a.K1, a.N3);
  float* s2 = s1 + (long)HD * K1x;
  float* s3 = s2 + (long)HD * (HD + 1);
#pragma unroll
  for (int pt = 0; pt < NP; ++pt)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int vc = 64 * pt + 16 * (2 * chh + u) + m;             // virtual column of this lane
      const int k = vc == kone ? a.K1 : vcol(vc, a.x.k0, a.kpad);   // (the ones column carries the bias gradient)
      const bool ok = vc == kone || (k >= 0 && k < a.K1 && vc < a.KV);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (ok) s1[(long)(16 * tf + 4 * q + r) * K1x + k] = dW1[pt][u][r];
    }
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) s2[(16 * tf + 4 * q + r) * (HD + 1) + 16 * (2 * chh + u) + m] = dW2[u][r];
  if (wave < 4) {
#pragma unroll
    for (int r = 0; r < 4; ++r) s3[(4 * q + r) * (HD + 1) + 16 * wave + m] = dW3[r];
  }
  // bias sums of layers 2 and 3: over the 16 rows of a lane group and the 8 waves, through the (now free) image
  float* red = reinterpret_cast<float*>(st);      // [8 waves][64 lanes][20]
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wave * 64 + lane) * 20 + 4 * t + r] = bs2[t][r];
#pragma unroll
  for (int r = 0; r < 4; ++r) red[(wave * 64 + lane) * 20 + 16 + r] = bs3[r];
  __syncthreads();
  if (tid < 80) {
    // feature f = 16t + 4q' + r of layer 2 (tid < 64) or output n3 = 4q' + r of layer 3: element e of the lanes (q', 0..15)
    const int f = tid < 64 ? tid : tid - 64;
    const int e = tid < 64 ? 4 * (f >> 4) + (f & 3) : 16 + (f & 3), qq = (f >> 2) & 3;
    float sum = 0.f;
    for (int w = 0; w < XNW; ++w)
      for (int mm_ = 0; mm_ < 16; ++mm_) sum += red[(w * 64 + qq * 16 + mm_) * 20 + e];
    if (tid < 64) s2[f * (HD + 1) + HD] = sum;
    else s3[f * (HD + 1) + HD] = sum;
  }
}

inline size_t x6_fwd_lds(int KC, int CF, bool three) {
  const int KC2 = (KC + 1) / 2;
  return (size_t)(4 * KC2 + (three ? 8 : 0) + 2) * 768 * 4 + (size_t)(KC - CF) * 32 * 4;
}
inline size_t x6_bwd_lds(int KC, int CF) { return (size_t)8 * 768 * 4 + 4 * 3 * 128 * 4 + (size_t)3 * XR * 256 + (size_t)(KC - CF) * 32 * 4; }

}  // namespace

// three-layer heads with up to 16 outputs and up to 192 input columns whose padded width leaves a free column for the ones column
extern "C" int marl_mlp3_x6_supported(const marl_src_t* x, int K1, int H1, int H2, int N3, int groups) {
  if (!marl_mlp3_supported(x, K1, H1, H2, N3, groups)) return 0;
  if (H2 != HD || N3 > 16) return 0;
  const int KV = K1 + kpad_of(x);
  const int KC = kc_bucket(KV);
  if (KC > 12 || KV >= 16 * KC) return 0;
  const int CF = lead_chunks(x);
  return x6_fwd_lds(KC, CF, true) <= 160 * 1024 && x6_bwd_lds(KC, CF) <= 160 * 1024;
}

extern "C" size_t marl_mlp3_x6_save_floats(long M, int groups) {
  return M <= 0 ? 0 : (size_t)groups * (size_t)((M + 15) / 16) * (size_t)x6_save_floats_per_tile(true);
}

extern "C" int marl_mlp3_x6_fwd_save(const marl_mlp3_weights_t* w, const marl_src_t* x, float* Y, long ldy, long gs_y,
                                     float* hsave, size_t hsave_floats, long M, int K1, int N3, int groups, void* stream) {
  if (M <= 0) return 0;
  if (!w->w2 || !marl_mlp3_x6_supported(x, K1, HD, HD, N3, groups)) return (int)hipErrorInvalidValue;
  if (hsave && (hsave_floats < marl_mlp3_x6_save_floats(M, groups) || !aligned16(hsave))) return (int)hipErrorInvalidValue;
  if (!aligned16(w->b1) || w->gs_b1 % 4 || !aligned16(w->b2) || w->gs_b2 % 4) return (int)hipErrorInvalidValue;
  Mlp3Args a;
  if (!fill_args(a, w, x, M, K1, N3, groups)) return (int)hipErrorInvalidValue;
  a.Y = Y; a.ldy = ldy; a.gs_y = gs_y; a.ws = nullptr; a.hs = hsave;
  const long tiles = (M + 15) / 16;
  a.nst = stripes((tiles + FNW - 1) / FNW, groups);
  const int KC = kc_bucket(a.KV);
  const size_t lds = x6_fwd_lds(KC, a.CF, true);
#define MLP3_PICK(K, ...) (KC == 4 ? (const void*)K<4, __VA_ARGS__> : KC == 8 ? (a.CF == 7 ? (const void*)K<8, MLP3_CF7(__VA_ARGS__)> : (const void*)K<8, __VA_ARGS__>) \
                           : KC == 11 ? (a.CF == 7 ? (const void*)K<11, MLP3_CF7(__VA_ARGS__)> : (const void*)K<11, __VA_ARGS__>) : (const void*)K<12, __VA_ARGS__>)
  const void* fn = MLP3_PICK(mlp3x6_fwd_kernel, true, -1);
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
  if (hsave_floats < marl_mlp3_x6_save_floats(M, groups) || !aligned16(hsave)) return (int)hipErrorInvalidValue;
  if (ws_bytes < marl_mlp3_bwd_workspace(M, K1, N3, groups)) return (int)hipErrorInvalidValue;
  Mlp3Args a;
  if (!fill_args(a, w, x, M, K1, N3, groups)) return (int)hipErrorInvalidValue;
  a.Y = const_cast<float*>(dY); a.ldy = lddy; a.gs_y = gs_dy; a.ws = ws; a.hs = const_cast<float*>(hsave);
  const int KC = kc_bucket(a.KV);
  a.nst = stripes((M + XR - 1) / XR, groups, 256);
  const size_t lds = x6_bwd_lds(KC, a.CF);
  const void* fn = MLP3_PICK(mlp3x6_bwd_kernel, true, -1);
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
