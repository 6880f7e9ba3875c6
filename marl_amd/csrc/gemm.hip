// Dense-layer kernels on the fp32 matrix cores (v_mfma_f32_16x16x4_f32), gfx950 only.
//
//   marl_linear      Y = act(X W^T + b)      X is a virtual concat (ConcatSrc); also used for
//                                            dX = dY W   (W read k-major, no transpose copy)
//   marl_linear_wgrad  dW += G^T X, db += colsum(G), G = dY * act'(Yact); slab partials then a
//                                            fixed-order reduce => bitwise reproducible
//
// Every mixer (QMIX hypernets, QPLEX lambda-net / transformation net, QTRAN joint-Q and V heads;
// reference network/mixer.py) is a composition of these plus the per-row kernels in mixers.hip.
// Operand fragments are loaded straight to VGPRs with the K-permutation of common.h, so no LDS
// round trip is needed: the weights are small and L2-resident, X is streamed once per column block.
#include <cstdlib>
#include "common.h"
#include "../../include/marl_hip.h"

namespace {

struct LinArgs {
  ConcatSrc x;
  const float* W; long ldw;
  const float* bias;
  float* Y; long ldy;
  int M, N, K;
  int act;        // 0 none, 1 relu
  float beta;     // Y = beta*Y + result
  int groups;     // blockIdx.z; per-group element strides below (0 = shared)
  long gs_x0, gs_x1, gs_w, gs_b, gs_y, gs_m0;
};

// block = 256 threads = 4 waves; wave tile = 32 rows x 16*NC cols (2 x NC MFMA tiles; NC = 5 for 64 < N <= 80: QTRAN's
// 78-wide encoders in ONE column block - X is read once instead of twice).
// K is walked in chunks of 16.  Chunks that lie entirely inside dense segment 0 (no relu gate) take the FAST path:
// operands go straight to VGPRs (AMODE 1: one 16-byte load per lane and row tile; AMODE 2: four dword loads, for
// row strides / bases that are not 16-byte aligned, e.g. S = 322) and are software-pipelined one chunk ahead in a
// second register set (static ping-pong, no copies).  The remaining chunks (other segments of the virtual concat,
// segment boundaries, gated inputs) take the guarded element path.
template <int AMODE, bool W_KMAJOR, bool BF, bool GATE, int NC>
__global__ __launch_bounds__(256) void linear_kernel(LinArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, m = lane & 15;
  const int g = blockIdx.z;
  ConcatSrc x = a.x;
  if (x.p0) x.p0 += g * a.gs_x0;
  if (x.p1) x.p1 += g * a.gs_x1;
  if (x.m0) x.m0 += g * a.gs_m0;
  const float* W = a.W + g * a.gs_w;
  const float* bias = a.bias ? a.bias + g * a.gs_b : nullptr;
  float* Y = a.Y + g * a.gs_y;

  const long row0 = (long)blockIdx.x * 128 + wave * 32;
  const int col0 = blockIdx.y * (16 * NC);
  if (row0 >= a.M) return;
  int ct_used = (a.N - col0 + 15) / 16;
  if (ct_used > NC) ct_used = NC;

  long arow[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    long rr = row0 + r * 16 + m;
    arow[r] = rr < a.M ? rr : a.M - 1;
  }
  int bcol[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    int cc = col0 + c * 16 + m;
    bcol[c] = cc < a.N ? cc : a.N - 1;      // clamped: tiles past N compute garbage that is never stored
  }

  f32x4 acc[2][NC];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int K = a.K;
  const int kfast = AMODE ? (x.k0 & ~15) : 0;     // chunks [0, kfast) are whole chunks of dense segment 0
  ConcatRow crow[2];
  const float* ap[2];                             // row base pointers of segment 0 (fast path)
  const float* mp[2];                             // ... and of its relu gate (GATE: value * (gate > 0), the dX = (dY*relu') W calls)
  bool aok[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    crow[r] = concat_row(x, arow[r]);
    aok[r] = crow[r].ok0;
    ap[r] = x.p0 + (aok[r] ? crow[r].r0 : 0) * x.ld0 + 4 * q;
    mp[r] = GATE ? x.m0 + (aok[r] ? crow[r].r0 : 0) * x.ldm0 + 4 * q : nullptr;
  }
  const bool wvec = !W_KMAJOR && (a.ldw % 4 == 0) && ((reinterpret_cast<uintptr_t>(W) & 15) == 0);
  // load() issues a FIXED number of unpredicated loads (invalid rows read row 0 and are zeroed in mma(), column
  // tiles past N read a clamped column): predicated or conditional loads keep the compiler from counting them
  auto load = [&](f32x4 (&av)[2], f32x4 (&mv)[2], f32x4 (&bv)[NC], int k0) __attribute__((always_inline)) {
    const int kk = k0 + 4 * q;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (AMODE == 1) av[r] = *reinterpret_cast<const f32x4*>(ap[r] + k0);
      else { av[r][0] = ap[r][k0]; av[r][1] = ap[r][k0 + 1]; av[r][2] = ap[r][k0 + 2]; av[r][3] = ap[r][k0 + 3]; }
      if (GATE) {
        if (AMODE == 1) mv[r] = *reinterpret_cast<const f32x4*>(mp[r] + k0);
        else { mv[r][0] = mp[r][k0]; mv[r][1] = mp[r][k0 + 1]; mv[r][2] = mp[r][k0 + 2]; mv[r][3] = mp[r][k0 + 3]; }
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (!W_KMAJOR) {
        const float* wp = W + (long)bcol[c] * a.ldw + kk;
        if (wvec) bv[c] = *reinterpret_cast<const f32x4*>(wp);
        else { bv[c][0] = wp[0]; bv[c][1] = wp[1]; bv[c][2] = wp[2]; bv[c][3] = wp[3]; }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[c][i] = W[(long)(kk + i) * a.ldw + bcol[c]];
      }
    }
  };
  auto mma = [&](const f32x4 (&av)[2], const f32x4 (&mv)[2], const f32x4 (&bv)[NC]) __attribute__((always_inline)) {
    f32x4 am[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      am[r] = aok[r] ? av[r] : (f32x4){0.f, 0.f, 0.f, 0.f};
      if (GATE) {
#pragma unroll
        for (int i = 0; i < 4; ++i) am[r][i] = mv[r][i] > 0.f ? am[r][i] : 0.f;
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (c >= ct_used) continue;
#pragma unroll
      for (int r = 0; r < 2; ++r) acc[r][c] = mm16x4<BF>(am[r], bv[c], acc[r][c]);
    }
  };
  if (AMODE && kfast > 0) {
    // the prefetch is issued unconditionally (the last chunk is simply loaded again): a conditional issue makes the
    // compiler wait vmcnt(0) - for the prefetch it has just issued - in front of every MFMA block
    f32x4 aA[2], bA[NC], aB[2], bB[NC], mA[2], mB[2];
    const int klast = kfast - 16;
    load(aA, mA, bA, 0);
    int k0 = 0;
    while (true) {
      load(aB, mB, bB, k0 + 16 < klast ? k0 + 16 : klast);
      mma(aA, mA, bA);
      k0 += 16;
      if (k0 >= kfast) break;
      load(aA, mA, bA, k0 + 16 < klast ? k0 + 16 : klast);
      mma(aB, mB, bB);
      k0 += 16;
      if (k0 >= kfast) break;
    }
  }
  for (int k0 = kfast; k0 < K; k0 += 16) {   // generic path: guarded element loads
    const int kk = k0 + 4 * q;
    f32x4 av[2], bv[NC];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) av[r][i] = (kk + i < K) ? concat_at(x, crow[r], kk + i) : 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (c < ct_used) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float w = 0.f;
          if (kk + i < K) w = W_KMAJOR ? W[(long)(kk + i) * a.ldw + bcol[c]] : W[(long)bcol[c] * a.ldw + kk + i];
          bv[c][i] = w;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (c < ct_used) {
#pragma unroll
        for (int r = 0; r < 2; ++r) acc[r][c] = mm16x4<BF>(av[r], bv[c], acc[r][c]);
      }
    }
  }

  // epilogue: D-layout -> Y (16 lanes = 64 contiguous bytes per row)
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    if (c >= ct_used) continue;
    const int col = col0 + c * 16 + m;
    if (col >= a.N) continue;
    const float b = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long row = row0 + r * 16 + 4 * q + i;
        if (row < a.M) {
          float v = acc[r][c][i] + b;
          if (a.act == 1) v = v > 0.f ? v : 0.f;
          float* y = Y + row * a.ldy + col;
          if (a.beta != 0.f) v += a.beta * *y;
          *y = v;
        }
      }
    }
  }
}

struct WgradArgs {
  const float* G; long ldg;          // dY [M,N]
  const float* Yact; long ldya;      // optional relu gate: G *= (Yact > 0)
  ConcatSrc x;                       // X [M,K] virtual
  float* ws;                         // [slabs][groups][N][K+1] partials (last col = bias grad)
  int M, N, K;
  int slabs;
  int nyb;                           // column blocks of N per group
  int groups;
  long gs_g, gs_ya, gs_x0, gs_x1;
  int gvec, xvec;                    // float4 operand loads allowed (alignment checked on the host)
};

// dW block (64 n x 64 k) per workgroup, rows of the slab streamed in 64-row chunks:
//   * staging: all 256 threads copy the chunk's G [64 x 64] and X [64 x 64] sub-matrices HBM -> registers
//     -> LDS with 16-byte accesses (coalesced; the virtual-concat / row-remap / relu-gate logic runs once
//     per element here), one chunk ahead of the MFMAs (double-buffered LDS, one barrier per chunk);
//   * compute: wave w owns n-tile w (16 rows of dW) x 4 k-tiles; A^T and B fragments are 4-row column
//     reads of the LDS tiles (row stride 68 floats: the two 16-lane groups of a half-wave hit disjoint banks);
//   * no cross-wave reduction: each wave writes its own part of the slab partial.
constexpr int WS_ = 68;          // LDS row stride (floats)
constexpr int WCH = 64;          // rows per chunk

template <bool BF>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][WCH * WS_];     // [buffer][G|X][row][col]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, m = lane & 15;
  const int g = blockIdx.y / a.nyb;
  const int n0 = (blockIdx.y % a.nyb) * 64;
  const int k0 = blockIdx.z * 64;
  const int Kext = a.K + 1;
  const float* G = a.G + g * a.gs_g;
  const float* Ya = a.Yact ? a.Yact + g * a.gs_ya : nullptr;
  ConcatSrc x = a.x;
  if (x.p0) x.p0 += g * a.gs_x0;
  if (x.p1) x.p1 += g * a.gs_x1;

  int kt_used = (Kext - k0 + 15) / 16; if (kt_used > 4) kt_used = 4;
  const bool my_n = n0 + 16 * wave < a.N;            // this wave's n-tile exists

  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const long chunks = ((long)a.M + WCH - 1) / WCH;
  const long per = (chunks + a.slabs - 1) / a.slabs;
  const long c_begin = (long)blockIdx.x * per;
  long c_end = c_begin + per; if (c_end > chunks) c_end = chunks;

  // staging assignment: element group e = tid + 256*i -> row e/16 of the chunk, columns 4*(e%16) .. +3.
  // fetch() only ISSUES loads (raw values, clamped addresses, no masking / gating / selecting on a loaded value -
  // the first use of a value makes the compiler wait for its load before issuing the next one, which serialised
  // every load of a chunk); stash() masks, gates and writes the LDS tile one chunk later.
  const bool nvb = a.gvec && n0 + 64 <= a.N;                       // uniform: the whole 64-column block is inside N
  f32x4 gq[4], yq[4], xq[4];
  int fq[4];                                                       // 1 row inside M, 2 dense row valid
  auto fetch = [&](long c) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i;
      const int rl = e >> 4, c4 = (e & 15) * 4;
      long row = c * WCH + rl;
      const bool live = row < a.M;
      if (!live) row = a.M - 1;
      const int nb = n0 + c4;
      if (nvb) {
        gq[i] = *reinterpret_cast<const f32x4*>(G + row * a.ldg + nb);
        if (Ya) yq[i] = *reinterpret_cast<const f32x4*>(Ya + row * a.ldya + nb);
      } else {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const int n = nb + cc < a.N ? nb + cc : a.N - 1;         // clamped; masked in stash()
          gq[i][cc] = G[row * a.ldg + n];
          if (Ya) yq[i][cc] = Ya[row * a.ldya + n];
        }
      }
      const ConcatRow cr = concat_row(x, row);
      const int kb = k0 + c4;
      // per ITEM (4 columns): inside dense segment 0 and ungated -> raw 16-byte / dword loads, masked in stash();
      // only items that straddle a segment boundary or lie in the one-hot / id / ones columns take the element path
      // (a per-BLOCK test sent every partial last block - e.g. columns 64..120 of a 120-wide state - down that path)
      const bool ins = x.p0 && !x.m0 && kb + 3 < x.k0;
      fq[i] = (live ? 1 : 0) | (cr.ok0 ? 2 : 0) | (ins ? 4 : 0);
      if (ins && a.xvec) {
        xq[i] = *reinterpret_cast<const f32x4*>(x.p0 + (cr.ok0 ? cr.r0 : 0) * x.ld0 + kb);
      } else if (ins) {
        const float* xp_ = x.p0 + (cr.ok0 ? cr.r0 : 0) * x.ld0 + kb;
        xq[i][0] = xp_[0]; xq[i][1] = xp_[1]; xq[i][2] = xp_[2]; xq[i][3] = xp_[3];
      } else {
        f32x4 xv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const int k = kb + cc;
          if (k < a.K) xv[cc] = concat_at(x, cr, k);
          else if (k == a.K) xv[cc] = 1.f;            // virtual ones column => bias gradient
        }
        xq[i] = xv;
      }
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i;
      const int rl = e >> 4, c4 = (e & 15) * 4;
      f32x4 gv = gq[i], xv = xq[i];
      if (Ya) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) gv[cc] = yq[i][cc] > 0.f ? gv[cc] : 0.f;
      }
      if (!nvb) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) gv[cc] = n0 + c4 + cc < a.N ? gv[cc] : 0.f;
      }
      if (!(fq[i] & 1)) gv = (f32x4){0.f, 0.f, 0.f, 0.f};          // row past M contributes nothing
      if ((fq[i] & 4) && !(fq[i] & 2)) xv = (f32x4){0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(&lds[buf][0][rl * WS_ + c4]) = gv;
      *reinterpret_cast<f32x4*>(&lds[buf][1][rl * WS_ + c4]) = xv;
    }
  };

  if (c_begin < c_end) {
    fetch(c_begin);
    stash(0);
  }
  __syncthreads();
  int buf = 0;
  for (long c = c_begin; c < c_end; ++c, buf ^= 1) {
    const bool more = c + 1 < c_end;
    if (more) fetch(c + 1);                    // global loads of the next chunk fly during the MFMAs
    if (my_n) {
      const float* Gs = &lds[buf][0][0];
      const float* Xs = &lds[buf][1][0];
#pragma unroll
      for (int sub = 0; sub < WCH / 16; ++sub) {
        f32x4 af, bf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rr = (sub * 16 + 4 * q + i) * WS_;
          af[i] = Gs[rr + 16 * wave + m];
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) bf[kt][i] = Xs[rr + 16 * kt + m];
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
          if (kt < kt_used) acc[kt] = mm16x4<BF>(af, bf[kt], acc[kt]);
      }
    }
    if (more) stash(buf ^ 1);
    __syncthreads();
  }

  // slab partial: D-layout tile (n = 16w + 4q + i, k = 16kt + m)
  float* ws = a.ws + ((long)blockIdx.x * a.groups + g) * (long)a.N * Kext;
  if (my_n) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = n0 + 16 * wave + 4 * q + i, k = k0 + 16 * kt + m;
        if (n < a.N && k < Kext) ws[(long)n * Kext + k] = acc[kt][i];
      }
  }
}


// ---------------------------------------------------------------------------------------------------
// Full-width weight gradient for narrow layers (N <= 80 outputs, K + 1 <= 80 inputs: QTRAN's 78 x 78 encoders and the
// 64 x 64 head layers): ONE workgroup column, so G and X are read from HBM exactly once (the 64 x 64-block kernel
// above re-reads G per 64 input columns and X per 64 outputs: 1.26 GB for 0.38 GB of operands on the encoders).
// Staging as above (16-byte coalesced, one chunk of 64 rows ahead, double-buffered LDS); the ROWS of a chunk are split
// over the four waves - wave w multiplies rows [16w, 16w+16) into all NTN x NTK output tiles (<= 25 accumulators) - and
// the waves' partial sums meet once at the end, in wave order, through LDS.
constexpr int WF = 80;           // tile width (columns) of both operands
constexpr int WFS = 84;          // LDS row stride: 4 rows apart = 16 banks apart (half-wave conflict-free)

template <bool GATE>
__global__ __launch_bounds__(256, 2) void wgrad_full_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];       // [G|X][64][WFS] (43 KB: two workgroups per CU); reused for the reduction
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, m = lane & 15;
  const int Kext = a.K + 1;
  const int ntn = (a.N + 15) / 16, ntk = (Kext + 15) / 16;
  const float* G = a.G;
  const float* Ya = GATE ? a.Yact : nullptr;
  const ConcatSrc& x = a.x;
  f32x4 acc[5][5];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const long chunks = ((long)a.M + WCH - 1) / WCH;
  const long per = (chunks + a.slabs - 1) / a.slabs;
  const long c_begin = (long)blockIdx.x * per;
  long c_end = c_begin + per; if (c_end > chunks) c_end = chunks;
  auto tile = [&](int, int which) { return lds + (which * WCH) * WFS; };

  // item e = tid + 256 * i (i < 5): row e / 20 of the chunk, columns 4 * (e % 20) .. + 3 of G and of X.
  // The columns of an item are the same in every chunk, so what they ARE is resolved once: kind 1 = four dense0
  // columns as one 16-byte load; otherwise one row base (2 action indices, 3 dense1, 4 dense0 element-wise, 0 none) +
  // four byte offsets + four codes (0xffff zero, 0xfffe one = the bias column, 0xfffd the loaded float, else the
  // action that makes a one-hot column 1).  The host only picks this kernel when no item needs two bases.
  // fetch() only ISSUES loads (raw bits): nothing selects on a loaded value before stash() - the first use of a value
  // makes the compiler wait for its load, which serialised every element load of the 64 x 64-block kernel's tail path.
  // (kept in LDS, 16 bytes per item: with 25 accumulator tiles the register file has no room for them - spilled
  // registers are reloaded through the same vmcnt queue as the prefetch and drain it)
  int* itab = reinterpret_cast<int*>(lds + 2 * WCH * WFS) + 2 * 2 * WCH * 4;      // behind the row table: [5][256] int4
  int it_kind[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int e = tid + 256 * i;
    const int c4 = (e % 20) * 4;
    int kind = 0;
    unsigned off[4], cmp[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      int k = c4 + cc;
      off[cc] = 0; cmp[cc] = 0xffffu;
      if (k > a.K) continue;
      if (k == a.K) { cmp[cc] = 0xfffeu; continue; }
      if (k < x.k0) { kind = 4; off[cc] = 4u * k; cmp[cc] = 0xfffdu; }
      else if (k - x.k0 < x.k1) { kind = 3; off[cc] = 4u * (k - x.k0); cmp[cc] = 0xfffdu; }
      else {
        k -= x.k0 + x.k1;
        const int j = k / x.hot_w;
        kind = 2; off[cc] = 4u * j; cmp[cc] = (unsigned)(k - j * x.hot_w);
      }
    }
    if (kind == 4 && a.xvec && c4 + 3 < x.k0) kind = 1;
    it_kind[i] = kind;
    int* t = itab + (i * 256 + tid) * 4;
    t[0] = (int)(off[0] | (off[1] << 16)); t[1] = (int)(off[2] | (off[3] << 16));
    t[2] = (int)(cmp[0] | (cmp[1] << 16)); t[3] = (int)(cmp[2] | (cmp[3] << 16));
  }
  f32x4 gq[5], yq[GATE ? 5 : 1], xq[5];        // xq: the 16-byte load of a kind-1 item or the four raw words of the others
  int fq[5];
  // Row arithmetic (clamp, row remap, episode map, index remap) ONCE PER ROW of a chunk, not once per 16-byte item:
  // 64 threads fill a small LDS table {G row offset, x.p0 row offset, idx row offset, p1 row offset (bytes), flags}
  // two chunks ahead; every item then starts from two 16-byte LDS reads.  (Done per item it was ~110 vector
  // instructions against the item's 20 MFMAs - the kernel ran at 1.3 TB/s, bound by that arithmetic.)
  long* rtab = reinterpret_cast<long*>(lds + 2 * WCH * WFS);      // [2 slots][64 rows][4 longs]; flags in the top byte of [3]
  auto rowinfo = [&](long c, int slot) {
    if (tid < WCH) {
      long row = c * WCH + tid;
      const bool live = row < a.M;
      if (!live) row = a.M - 1;
      const ConcatRow cr = concat_row(x, row);
      const long fl = (live ? 1 : 0) | (cr.ok0 ? 2 : 0) | (cr.oki ? 4 : 0);
      long* t = rtab + (slot * WCH + tid) * 4;
      t[0] = row;
      t[1] = (cr.ok0 ? cr.r0 : 0) * x.ld0 * 4;
      t[2] = (cr.oki ? cr.ri : 0) * (long)x.nhot * 4;
      t[3] = fl;
    }
  };
  auto fetch = [&](long c) {
    const long* tb = rtab + (int)(c & 1) * WCH * 4;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int e = tid + 256 * i;
      const int rl = e / 20, c4 = (e - rl * 20) * 4;
      const long row = tb[rl * 4 + 0], o0 = tb[rl * 4 + 1], oi = tb[rl * 4 + 2];
      fq[i] = (int)tb[rl * 4 + 3];
      const int cg = c4 < a.N ? c4 : 0;                 // column groups past N re-read group 0 and are zeroed in stash()
      gq[i] = *reinterpret_cast<const f32x4*>(G + row * a.ldg + cg);
      if (GATE) yq[GATE ? i : 0] = *reinterpret_cast<const f32x4*>(Ya + row * a.ldya + cg);
      const char* d0 = reinterpret_cast<const char*>(x.p0) + o0;
      const int kind = it_kind[i];
      if (kind == 1) {
        xq[i] = *reinterpret_cast<const f32x4*>(d0 + 4 * c4);
      } else if (kind >= 2) {
        const char* d1 = x.p1 ? reinterpret_cast<const char*>(x.p1 + row * x.ld1) : d0;
        const char* di = x.idx ? reinterpret_cast<const char*>(x.idx) + oi : d0;
        const char* base = kind == 2 ? di : (kind == 3 ? d1 : d0);
        const unsigned o01 = (unsigned)itab[(i * 256 + tid) * 4 + 0], o23 = (unsigned)itab[(i * 256 + tid) * 4 + 1];
        xq[i][0] = __int_as_float(*reinterpret_cast<const int*>(base + (o01 & 0xffffu)));
        xq[i][1] = __int_as_float(*reinterpret_cast<const int*>(base + (o01 >> 16)));
        xq[i][2] = __int_as_float(*reinterpret_cast<const int*>(base + (o23 & 0xffffu)));
        xq[i][3] = __int_as_float(*reinterpret_cast<const int*>(base + (o23 >> 16)));
      }
    }
  };
  auto stash = [&](int buf) {
    float* Gs = tile(buf, 0);
    float* Xs = tile(buf, 1);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int e = tid + 256 * i;
      const int rl = e / 20, c4 = (e - rl * 20) * 4;
      f32x4 gv = gq[i];
      if (GATE) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) gv[cc] = yq[GATE ? i : 0][cc] > 0.f ? gv[cc] : 0.f;
      }
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) gv[cc] = c4 + cc < a.N ? gv[cc] : 0.f;
      if (!(fq[i] & 1)) gv = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int kind = it_kind[i];
      const bool ok0 = (fq[i] & 2) != 0, oki = (fq[i] & 4) != 0;
      f32x4 xv;
      if (kind == 1) {
        xv = ok0 ? xq[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
      } else {
        const bool okd = kind == 3 || (kind == 4 && ok0);
        const unsigned c01 = (unsigned)itab[(i * 256 + tid) * 4 + 2], c23 = (unsigned)itab[(i * 256 + tid) * 4 + 3];
        const unsigned cm[4] = {c01 & 0xffffu, c01 >> 16, c23 & 0xffffu, c23 >> 16};
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const int raw = kind >= 2 ? __float_as_int(xq[i][cc]) : 0;
          float v = 0.f;
          if (cm[cc] == 0xfffeu) v = 1.f;
          else if (cm[cc] == 0xfffdu) v = okd ? __int_as_float(raw) : 0.f;
          else if (cm[cc] != 0xffffu) v = (oki && (unsigned)raw == cm[cc]) ? 1.f : 0.f;
          xv[cc] = v;
        }
      }
      *reinterpret_cast<f32x4*>(Gs + rl * WFS + c4) = gv;
      *reinterpret_cast<f32x4*>(Xs + rl * WFS + c4) = xv;
    }
  };

  // single LDS tile: registers -> LDS, barrier, next chunk's loads issued, MFMAs, barrier.  The loads of chunk c+1
  // fly during the MFMAs of chunk c and the second workgroup of the CU fills the barrier / latency gaps (one
  // double-buffered workgroup per CU left the memory system at 1.2 TB/s).
  if (c_begin < c_end) {
    rowinfo(c_begin, (int)(c_begin & 1));
    rowinfo(c_begin + 1, (int)((c_begin + 1) & 1));
    __syncthreads();
    fetch(c_begin);
  }
  for (long c = c_begin; c < c_end; ++c) {
    stash(0);
    __syncthreads();                               // tile of chunk c and the row table of chunk c+1 are in LDS
    if (c + 1 < c_end) fetch(c + 1);
    rowinfo(c + 2, (int)(c & 1));                  // slot of chunk c: its fetch happened an iteration ago
    const float* Gs = tile(0, 0) + (16 * wave + 4 * q) * WFS + m;
    const float* Xs = tile(0, 1) + (16 * wave + 4 * q) * WFS + m;
#pragma unroll
    for (int kt = 0; kt < 5; ++kt) {
      if (kt < ntk) {
        f32x4 bf;
#pragma unroll
        for (int i = 0; i < 4; ++i) bf[i] = Xs[i * WFS + 16 * kt];
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) {
          if (nt < ntn) {
            f32x4 af;
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = Gs[i * WFS + 16 * nt];
            acc[nt][kt] = mfma16x4(af, bf, acc[nt][kt]);
          }
        }
      }
    }
    __syncthreads();
  }

  // the four waves' partial sums, added in wave order through LDS (the staging buffers are free now)
  float* red = lds;                        // [WF][WFS]
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int nt = 0; nt < 5; ++nt)
#pragma unroll
        for (int kt = 0; kt < 5; ++kt)
          if (nt < ntn && kt < ntk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float* d = red + (16 * nt + 4 * q + i) * WFS + 16 * kt + m;
              *d = (w == 0 ? 0.f : *d) + acc[nt][kt][i];
            }
          }
    }
    __syncthreads();
  }
  float* ws = a.ws + (long)blockIdx.x * a.N * Kext;
  for (int e = tid; e < a.N * Kext; e += 256) {
    const int n = e / Kext, k = e - n * Kext;
    ws[e] = red[n * WFS + k];
  }
}

// ---------------------------------------------------------------------------------------------------
// Direct weight gradient for N == 64 outputs (fc1 of the agent: dW1 = dxp^T [obs | one-hot(u) | id] over
// B*T*N rows, the largest reduction of an update).  No LDS staging: the reduction index of dW = G^T X is
// the ROW, and fp32 MFMA 16x16x4 takes one k (= row) per lane quarter, so lane (q, m) feeds the matrix
// cores straight from two coalesced 16-byte loads of row r0+q:
//     G[row][4m..4m+3]   -> A operands of the 4 n-tiles   (tile j holds dW rows n = 4m'+j)
//     X[row][4m..4m+3]   -> B operands of 4 k-tiles       (tile j holds columns  k = 4m'+j)
// i.e. the tiles are taken over a permuted column set, undone when the slab is written.  Columns past the
// first 64 (or all of them when segment 0 is narrower than 64 / gated / unaligned) go through NTP "plain"
// 16-column tiles read element-wise from the virtual concat.  Per 4 rows a wave issues 2-3 loads and
// 4*(4*KP+NTP) MFMAs; rows are dealt to waves in 16-row blocks, one block prefetched in registers.
// The bias gradient is a VALU column sum of the G operands.  8 waves (2 per SIMD) reduce through LDS into
// one slab per workgroup; slabs are summed by wgrad_reduce_kernel (same layout as wgrad_kernel).
#ifndef WD_DU
#define WD_DU 4
#define WD_WAVES 8
#endif
constexpr int DU = WD_DU;         // k-steps (of 4 rows) per block
constexpr int WDW = WD_WAVES;     // waves per workgroup (8 = two per SIMD; 4 waves with a 32-row block in flight measured 1.6x slower)

template <int KP, int NTP>
__global__ __launch_bounds__(64 * WDW, 1) void wgrad_direct_kernel(WgradArgs a) {
  extern __shared__ float red[];  // [64][K+1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, m = lane & 15;
  const int K = a.K, Kx = K + 1;
  constexpr int NT = 4 * KP + NTP;
  f32x4 acc[4][NT];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[j][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  const long nblk = ((long)a.M + 4 * DU - 1) / (4 * DU);
  const long G = (long)gridDim.x * WDW;
  constexpr int NP = NTP > 0 ? NTP : 1;
  // per-lane description of its column in each plain tile: segment kind (0 dense, 2 one-hot, 3 agent id, -1 pad),
  // clamped dense column, one-hot block and the value that makes the element 1
  int pkind[NP], pcol[NP], pj[NP], pc[NP];
#pragma unroll
  for (int t = 0; t < NP; ++t) {
    int k = 64 * KP + 16 * t + m;
    pkind[t] = -1; pcol[t] = 0; pj[t] = 0; pc[t] = -1;
    if (k < a.x.k0) { pkind[t] = 0; pcol[t] = k; }
    else {
      k -= a.x.k0;
      const int hw = a.x.nhot * a.x.hot_w;
      if (k < hw) { pkind[t] = 2; pj[t] = k / a.x.hot_w; pc[t] = k - pj[t] * a.x.hot_w; }
      else if (k - hw < a.x.nid) { pkind[t] = 3; pc[t] = k - hw; }
    }
  }
  // every load below is issued UNCONDITIONALLY (blocks past the end read clamped rows and contribute zeros): with a
  // conditional prefetch the compiler cannot count the outstanding loads and waits vmcnt(0) - i.e. for the
  // prefetch it has just issued - in front of every MFMA block, which serialises streaming and math
  const int* idxp = a.x.idx ? a.x.idx : reinterpret_cast<const int*>(a.x.p0);
  const int nhot_e = a.x.nhot ? a.x.nhot : 1;
  // NOTHING in load() may consume a loaded value (no masking, no select): the first use of a value makes the
  // compiler wait for that load before it issues the next one, and a block's loads then pay their latencies one
  // after the other.  load() stores raw bits plus a small predicate word; mac() masks right before the MFMAs.
  f32x4 gA[DU], gB[DU], xA[DU], xB[DU];      // two register sets: the block in flight and the one being consumed
  int pA[DU][NP], pB[DU][NP];                // plain tiles: raw 32-bit words (dense float / action index / ready value)
  int fA, fB;                                // per k-step predicate bits: 1 row inside M, 2 dense row valid, 4 index row valid
  auto load = [&](long blk, f32x4 (&ga)[DU], f32x4 (&xb)[DU], int (&xp)[DU][NP], int& flags) __attribute__((always_inline)) {
    int fl = 0;
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      long row = blk * (4 * DU) + 4 * u + q;
      const bool live = row < a.M;
      if (!live) row = a.M - 1;
      ga[u] = *reinterpret_cast<const f32x4*>(a.G + row * a.ldg + 4 * m);
      const ConcatRow cr = concat_row(a.x, row);
      const long r0c = cr.ok0 ? cr.r0 : 0, ric = cr.oki ? cr.ri : 0;      // clamped: loads are never predicated
      fl |= ((live ? 1 : 0) | (cr.ok0 ? 2 : 0) | (cr.oki ? 4 : 0)) << (4 * u);
      if (KP) xb[u] = *reinterpret_cast<const f32x4*>(a.x.p0 + r0c * a.x.ld0 + 4 * m);
      // plain tiles: the lane's column (hence its segment) is fixed -> ONE raw 32-bit load per element from a
      // per-lane selected address (dense float, or the action index of its one-hot block); agent-id lanes need no
      // memory at all and store their finished value
#pragma unroll
      for (int t = 0; t < NTP; ++t) {
        const uintptr_t pd = reinterpret_cast<uintptr_t>(a.x.p0 + r0c * a.x.ld0 + pcol[t]);
        const uintptr_t po = reinterpret_cast<uintptr_t>(idxp + ric * nhot_e + pj[t]);
        const int raw = *reinterpret_cast<const int*>(pkind[t] == 2 ? po : pd);      // one load, address selected per lane
        xp[u][t] = pkind[t] == 3 ? __float_as_int(cr.nidx == pc[t] ? 1.f : 0.f) : raw;
      }
    }
    flags = fl;
  };
  auto mac = [&](const f32x4 (&ga)[DU], const f32x4 (&xb)[DU], const int (&xp)[DU][NP], int flags) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      const int f = flags >> (4 * u);
      f32x4 g = ga[u];
      if (!(f & 1)) g = (f32x4){0.f, 0.f, 0.f, 0.f};                       // row past M: contributes nothing
      bsum += g;
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
      if (KP) x = (f & 2) ? xb[u] : x;
      float xs[NP];
#pragma unroll
      for (int t = 0; t < NTP; ++t) {
        const int raw = xp[u][t];
        float v = 0.f;
        if (pkind[t] == 0) v = (f & 2) ? __int_as_float(raw) : 0.f;
        else if (pkind[t] == 2) v = ((f & 4) && raw == pc[t]) ? 1.f : 0.f;
        else if (pkind[t] == 3) v = __int_as_float(raw);
        xs[t] = v;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (KP) {
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[j][t] = mfma16(g[j], x[t], acc[j][t]);
        }
#pragma unroll
        for (int t = 0; t < NTP; ++t) acc[j][4 * KP + t] = mfma16(g[j], xs[t], acc[j][4 * KP + t]);
      }
    }
  };
  long blk = (long)blockIdx.x * WDW + wave;
  load(blk, gA, xA, pA, fA);
  while (blk < nblk) {
    load(blk + G, gB, xB, pB, fB);
    mac(gA, xA, pA, fA);
    blk += G;
    if (blk >= nblk) break;
    load(blk + G, gA, xA, pA, fA);
    mac(gB, xB, pB, fB);
    blk += G;
  }
  // ---- workgroup reduction through LDS (wave order fixed -> deterministic), then one slab
#pragma unroll
  for (int j = 0; j < 4; ++j) bsum[j] += __shfl_xor(bsum[j], 16, 64);
#pragma unroll
  for (int j = 0; j < 4; ++j) bsum[j] += __shfl_xor(bsum[j], 32, 64);
  for (int w = 0; w < WDW; ++w) {
    if (wave == w) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int n = 4 * (4 * q + i) + j;                               // D row -> permuted output row
            const int k = (KP && t < 4) ? 4 * m + t : 64 * KP + 16 * (t - 4 * KP) + m;
            if (k < K) {
              float* d = &red[n * Kx + k];
              *d = (w == 0 ? 0.f : *d) + acc[j][t][i];
            }
          }
        if (q == 0) {
          float* d = &red[(4 * m + j) * Kx + K];
          *d = (w == 0 ? 0.f : *d) + bsum[j];
        }
      }
    }
    __syncthreads();
  }
  float* slab = a.ws + (long)blockIdx.x * 64 * Kx;
  for (int e = tid; e < 64 * Kx; e += 64 * WDW) slab[e] = red[e];
}

template <int KP, int NTP>
inline int launch_wgrad_direct(const WgradArgs& a, hipStream_t s) {
  const size_t lds = (size_t)64 * (a.K + 1) * sizeof(float);
  hipLaunchKernelGGL((wgrad_direct_kernel<KP, NTP>), dim3(a.slabs), dim3(64 * WDW), lds, s, a);
  return 0;
}

// returns -1 when the shape is not covered (caller uses the LDS-staged kernel)
inline int try_wgrad_direct(const WgradArgs& a, hipStream_t s) {
  if (a.N != 64 || a.groups != 1 || a.Yact || !a.gvec || a.M < 4096 || a.x.k1 || a.x.m0 || a.x.k0 < 0 || !a.x.p0) return -1;
  const bool perm = a.xvec && a.x.k0 >= 64;
  const int KP = perm ? 1 : 0;
  const int rest = a.K - 64 * KP;
  const int NTP = (rest + 15) / 16;
  if (NTP > (KP ? 4 : 5) || KP + NTP == 0) return -1;
#define WD_CASE(kp, nt) if (KP == kp && NTP == nt) return launch_wgrad_direct<kp, nt>(a, s);
  WD_CASE(1, 0) WD_CASE(1, 1) WD_CASE(1, 2) WD_CASE(1, 3) WD_CASE(1, 4)
  WD_CASE(0, 1) WD_CASE(0, 2) WD_CASE(0, 3) WD_CASE(0, 4) WD_CASE(0, 5)
#undef WD_CASE
  return -1;
}

// ---------------------------------------------------------------------------------------------------
// Tall-skinny weight gradient for N == 64 outputs through LDS (fc1 of the agent: dW1 = dxp^T [obs | one-hot(u) | id] over
// B*T*N rows - 2.5 M rows at the headline workload).  The direct kernel above feeds the MFMAs straight from global loads and
// ends up paying load time PLUS matrix time.  Here the workgroup is split by role:
//   waves 4-7 (producers) stream 32-row chunks of G and X into LDS, two chunks of loads in flight (two named register sets),
//             registers -> LDS into the buffer the consumers are not reading; they also resolve the row remap / episode map /
//             action index of the rows two chunk pairs ahead into LDS tables (one producer wave per pair, in rotation), so no
//             dependent global load sits inside the staging loads;
//   waves 0-3 (consumers, one per SIMD) own output tile row tn = wave and ALL k tiles: per 4-row step one G read + KTT X reads
//             (plain 32-bit LDS reads, prefetched a step ahead) and KTT MFMAs - the matrix pipe never waits for a load phase.
// One LDS-only barrier per chunk joins the two roles.  (A first version had every wave load, multiply and store in turn: all
// waves sit in the same phase, stamps showed multiply 30 %, load issue 20 %, LDS write 15 %, barrier 30 % of a chunk - 0.61 ms
// against 0.57 ms of the direct kernel.)
// The reduction index of dW = G^T X is the row, and v_mfma_f32_16x16x4 takes one k (= row) per lane quarter, so lane (q, m)
// reads G[row 4s+q][16 tn + m] and X[row 4s+q][16 tk + m] from ROW-MAJOR tiles (row pitch = 16 mod 32 floats: the two quarters
// of a half-wave hit disjoint banks) - no transposition anywhere.  The one-hot / agent-id columns of the virtual concat live in
// a tail of the X tile that is zeroed once; a chunk sets its (at most two) ones per row and the next chunk landing in the
// buffer clears them.
constexpr int TGP = 64 + 16;      // LDS pitch of the G chunk
constexpr int TCH = 32;           // rows per chunk
constexpr int TPT = 256;          // producer threads

template <int KTT, int NX>        // k tiles (6: K <= 96, 10: K <= 160, 14: K <= 224); float4 of X a producer thread stages per chunk
__global__ __launch_bounds__(512, NX >= 7 ? 2 : 4) void wgrad_tall_kernel(WgradArgs a, int XP) {      // (NX = 7: 84 KB of LDS, one workgroup per CU)
  extern __shared__ __attribute__((aligned(16))) float tsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool cons = wave < 4;
  const int ptid = tid - 256;                                   // producer thread index (negative in consumers)
  const int q = lane >> 4, m = lane & 15;
  // k0: the dense width rounded up to whole float4 groups - a width that is not a multiple of 4 (QTRAN's 78-wide encoder sums) is
  // allowed for plain dense sources whose rows are padded to 16 bytes: the pad columns are staged like data (any value - they only
  // meet the accumulators of columns >= K, which are never stored)
  const int K = a.K, k0 = (a.x.k0 + 3) & ~3;
  const int BUF = TCH * (TGP + XP);
  long* rtab = reinterpret_cast<long*>(tsm + 2 * BUF);          // [8][TCH] source offset (floats) of the dense row, by chunk & 7
  int* htab = reinterpret_cast<int*>(rtab + 8 * TCH);           // [8][TCH] column of the one-hot 1 (or -1)
  int* itab = htab + 8 * TCH;                                   // [8][TCH] column of the agent-id 1 (or -1)
  const long per = ((long)a.M + gridDim.x - 1) / gridDim.x;
  const long r_begin = (long)blockIdx.x * per;
  long r_end = r_begin + per; if (r_end > a.M) r_end = a.M;
  const long nch = r_end > r_begin ? (r_end - r_begin + TCH - 1) / TCH : 0;
  const int D4 = k0 >> 2;                                       // float4 groups of the dense segment
  const int T4 = (XP - k0) >> 2;                                // float4 groups of the tail (one-hot, agent id, pad)
  const int di = TCH * D4, ti = TCH * T4;
  constexpr int NG = TCH * 16 / TPT;                            // float4 of G a producer thread stages per chunk

  // ---- row tables of a chunk PAIR (64 rows = the lanes of one producer wave): loads issued in the first half of a trip,
  // consumed at the end of its second half
  int rs_em = 0, rs_u = -1, rs_w = 0, rs_n = -1;
  bool rs_ok0 = false, rs_oki = false, rs_live = false;
  auto resolve_issue = [&](long ch) {                           // ch even: rows of chunks ch, ch + 1
    long row = r_begin + ch * TCH + lane;
    rs_live = row < r_end;
    if (row > (long)a.M - 1) row = (long)a.M - 1;
    const unsigned ur = (unsigned)row;
    unsigned e = ur; long w = 0;
    rs_ok0 = true;
    if (a.x.rpe0) { e = fastdiv(ur, a.x.fd0); w = (long)(ur - e * (unsigned)a.x.rpe0) + a.x.off0; rs_ok0 = w >= 0; }
    rs_w = (int)w;
    rs_em = (a.x.emap0 && a.x.rpe0) ? a.x.emap0[e] : (int)e;
    long ri = row; rs_oki = true;
    if (a.x.rpei) { const unsigned ei = fastdiv(ur, a.x.fdi); const long wi = (long)(ur - ei * (unsigned)a.x.rpei) + a.x.offi;
                    rs_oki = wi >= 0; ri = (long)ei * a.x.bsi + wi; }
    rs_u = (a.x.nhot && rs_oki) ? a.x.idx[ri * a.x.nhot] : -1;
    rs_n = a.x.nid ? (int)(ur - fastdiv(ur, a.x.fdn) * (unsigned)a.x.nid) : -1;
  };
  auto resolve_commit = [&](long ch) {
    const int b = (int)(ch & 7) * TCH + lane;                   // ch even: slots ch & 7 and (ch & 7) + 1 are adjacent
    const long r0 = a.x.rpe0 ? (long)rs_em * a.x.bs0 + rs_w : (long)rs_em;
    rtab[b] = rs_ok0 ? r0 * a.x.ld0 : -1;
    htab[b] = (rs_live && rs_u >= 0 && rs_u < a.x.hot_w) ? k0 + rs_u : -1;
    itab[b] = (rs_live && rs_n >= 0) ? k0 + a.x.nhot * a.x.hot_w + rs_n : -1;
  };

  // ---- producers: the (row, column) a thread stages is the same in every chunk
  int xr[NX], xc[NX];
  const float* gp[NG];
  int gr[NG], glim[NG];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int e = ptid + TPT * i;
    xr[i] = -1; xc[i] = 0;
    if (ptid >= 0 && e < di) { const int r = e / D4; xr[i] = r; xc[i] = (e - r * D4) * 4; }
  }
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const int e = (ptid < 0 ? 0 : ptid) + TPT * i;              // rows x 16 float4
    gr[i] = e >> 4;
    long row = r_begin + gr[i];
    if (row > (long)a.M - 1) row = (long)a.M - 1;
    gp[i] = a.G + row * a.ldg + (e & 15) * 4;
    const long left = r_end - r_begin - gr[i];                  // row gr[i] of chunk ch exists while ch * TCH < left
    glim[i] = left > 0 ? (int)((left + TCH - 1) / TCH) : 0;
  }
  const long gstep = (long)TCH * a.ldg;
  auto fetch = [&](f32x4 (&pg)[NG], f32x4 (&px)[NX], unsigned& zmask, long ch) __attribute__((always_inline)) {
    const int tb = (int)(ch & 7) * TCH;
    zmask = 0;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const bool live = (int)ch < glim[i];                      // the last chunk of a slab can be ragged; chunks past it are dead
      pg[i] = *reinterpret_cast<const f32x4*>(live ? gp[i] + ch * gstep : gp[i]);
      if (!live) zmask |= 0x100u << i;                          // rows past the slab contribute nothing
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {                              // unconditional (threads past the tile re-read its first float4):
      const long off = rtab[tb + (xr[i] < 0 ? 0 : xr[i])];      // a lane-conditional load makes hipcc wait for the previous set
      if (off < 0) zmask |= 1u << i;                            // a remapped row before the first slot reads as zero
      px[i] = *reinterpret_cast<const f32x4*>(a.x.p0 + (off < 0 ? 0 : off) + xc[i]);
    }
  };
  int old_h[2] = {-1, -1}, old_i[2] = {-1, -1};
  if (!cons)
    for (int e = ptid; e < 2 * ti; e += TPT) {                  // zero the tails of both buffers once
      const int bb = e >= ti ? 1 : 0, ee = e - bb * ti;
      const int r = ee / T4, c = k0 + (ee - r * T4) * 4;
      *reinterpret_cast<f32x4*>(tsm + bb * BUF + TCH * TGP + r * XP + c) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  auto stash = [&](const f32x4 (&pg)[NG], const f32x4 (&px)[NX], unsigned zmask, long ch, int b) __attribute__((always_inline)) {
    const int tb = (int)(ch & 7) * TCH;
    float* Gs = tsm + b * BUF;
    float* Xs = Gs + TCH * TGP;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int e = ptid + TPT * i;
      f32x4 v = pg[i];
      if (zmask & (0x100u << i)) v = (f32x4){0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(Gs + (e >> 4) * TGP + (e & 15) * 4) = v;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      if (xr[i] >= 0) {
        f32x4 v = px[i];
        if (zmask & (1u << i)) v = (f32x4){0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(Xs + xr[i] * XP + xc[i]) = v;
      }
    }
    if (ptid < TCH && T4 > 0) {
      float* row = Xs + ptid * XP;
      if (old_h[b] >= 0) row[old_h[b]] = 0.f;
      if (old_i[b] >= 0) row[old_i[b]] = 0.f;
      const int hc = htab[tb + ptid], ic = itab[tb + ptid];
      if (hc >= 0) row[hc] = 1.f;
      if (ic >= 0) row[ic] = 1.f;
      old_h[b] = hc; old_i[b] = ic;
    }
  };

  if (wave == 4) {
    resolve_issue(0); resolve_commit(0);
    resolve_issue(2); resolve_commit(2);
  }
  __syncthreads();
  // The two roles are separate loops (each with ONE barrier per chunk, so the counts match): in a shared loop the register
  // allocator keeps the accumulators and both staging sets live together and spills.
  if (!cons) {
    f32x4 gS0[NG], xS0[NX], gS1[NG], xS1[NX];
    unsigned z0 = 0, z1 = 0;
    fetch(gS0, xS0, z0, 0); stash(gS0, xS0, z0, 0, 0);
    fetch(gS0, xS0, z0, 1);
    ST_DECL(2);
    // Nothing in the loop is conditional but the resolver's turn: a skipped fetch or stash leaves hipcc's scoreboard with
    // "maybe pending" registers at the loop head, and it then drains the other set before issuing the next loads.  Chunks past
    // the slab are therefore staged like any other (dead rows: clamped addresses, zeros) into a buffer nobody reads.
    for (long ch = 0; ch < nch; ch += 2) {
      const bool mine = wave == 4 + (int)((ch >> 1) & 3);       // this trip's table resolver
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // chunk ch is in buffer 0; buffer 1 is free
      ST_MARK(0);
      if (mine) resolve_issue(ch + 4);                          // BEFORE the fetch: vmcnt retires in order
      fetch(gS1, xS1, z1, ch + 2);
      stash(gS0, xS0, z0, ch + 1, 1);
      ST_MARK(1);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // chunk ch + 1 is in buffer 1; buffer 0 is free
      ST_MARK(0);
      fetch(gS0, xS0, z0, ch + 3);
      stash(gS1, xS1, z1, ch + 2, 0);
      if (mine) resolve_commit(ch + 4);
      ST_MARK(1);
    }
    ST_DUMP(2);
    return;
  }

  // ---- consumers
  const int tn = wave & 3;
  f32x4 acc[KTT];
#pragma unroll
  for (int j = 0; j < KTT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bs = 0.f;
  ST_DECL(2);
  for (long ch = 0; ch < ((nch + 1) & ~1L); ++ch) {             // the producers' loop runs in pairs of chunks
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    ST_MARK(0);
    if (ch >= nch) break;
    // running LDS offsets (no multiply in the loop: hipcc turns `row * pitch` into a 64-bit mad whose unused high half can
    // land on a register with a global load pending, and then waits vmcnt(0) inside the MFMA loop)
    const float* Gs = tsm + (int)(ch & 1) * BUF + tn * 16 + m + q * TGP;
    const float* Xs = tsm + (int)(ch & 1) * BUF + TCH * TGP + m + q * XP;
    auto ld = [&](float& gv, float (&xv)[KTT]) __attribute__((always_inline)) {
      gv = Gs[0];
#pragma unroll
      for (int j = 0; j < KTT; ++j) xv[j] = Xs[16 * j];
      Gs += 4 * TGP; Xs += 4 * XP;
    };
    auto mac = [&](float gv, const float (&xv)[KTT]) __attribute__((always_inline)) {
      bs += gv;
#pragma unroll
      for (int j = 0; j < KTT; ++j) acc[j] = mfma16(gv, xv[j], acc[j]);
      __builtin_amdgcn_sched_barrier(0);
    };
    float gA, gB, xA[KTT], xB[KTT];
    ld(gA, xA);
#pragma unroll 1
    for (int st = 0; st < TCH / 4 - 2; st += 2) {
      ld(gB, xB);
      mac(gA, xA);
      ld(gA, xA);
      mac(gB, xB);
    }
    ld(gB, xB);
    mac(gA, xA);
    mac(gB, xB);
    ST_MARK(1);
  }
  ST_DUMP(2);
  const int Kx = K + 1;
  float* slab = a.ws + (long)blockIdx.x * 64 * Kx;
#pragma unroll
  for (int j = 0; j < KTT; ++j) {
    const int k = 16 * j + m;
    if (k < K) {
#pragma unroll
      for (int i = 0; i < 4; ++i) slab[(long)(16 * tn + 4 * q + i) * Kx + k] = acc[j][i];
    }
  }
  bs += __shfl_xor(bs, 16, 64);
  bs += __shfl_xor(bs, 32, 64);
  if (q == 0) slab[(long)(16 * tn + m) * Kx + K] = bs;
}

inline int tall_xp(int K) { const int w = (K + 15) / 16 * 16; return (w % 32 == 16) ? w : w + 16; }
// returns -1 when the shape is not covered
inline int try_wgrad_tall(WgradArgs& a, size_t ws_bytes, hipStream_t s) {
  if (!marl_switches()->wgrad_tall) return -1;      // A/B switch for measurements (common.h: MarlSwitches)
  if (a.N != 64 || a.groups != 1 || a.Yact || !a.gvec || !a.xvec || a.M < 4096 || a.x.k1 || a.x.m0 || !a.x.p0) return -1;
  if (a.x.nhot > 1 || a.x.k0 < 16 || a.x.k0 > 224) return -1;
  if ((a.x.k0 & 3) && (a.x.nhot || a.x.nid || (a.x.ld0 & 3))) return -1;      // ragged dense width: plain dense rows padded to 16 bytes only
  const int KT = (a.K + 15) / 16;
  if (KT > 14) return -1;
  const int XP = tall_xp(a.K);
  const size_t lds = (size_t)2 * TCH * (TGP + XP) * sizeof(float) + 8 * TCH * (sizeof(long) + 2 * sizeof(int));
  // (dense widths of 196 .. 224 columns - the 216 state columns of 3s5z under the first layer of QTRAN's heads - need seven
  // float4 per producer thread and 84 KB of LDS: one workgroup per CU instead of two, still 2.5x the generic kernel's rate)
  const bool wide = a.x.k0 > 192;
  if (lds > (wide ? 160 : 80) * 1024) return -1;
  // two slabs per CU when the caller's workspace holds them (marl_linear_wgrad_workspace sizes it so for N == 64)
  int slabs = a.slabs;
  if (slabs == 256 && ws_bytes >= (size_t)512 * 64 * (a.K + 1) * sizeof(float)) slabs = 512;
  a.slabs = slabs;
  const void* fn = KT <= 6 ? (const void*)wgrad_tall_kernel<6, 3> : KT <= 10 ? (const void*)wgrad_tall_kernel<10, 5>
                   : wide ? (const void*)wgrad_tall_kernel<14, 7> : (const void*)wgrad_tall_kernel<14, 6>;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  int xp = XP;
  void* kargs[] = {(void*)&a, (void*)&xp};
  e = hipLaunchKernel(fn, dim3(slabs), dim3(512), kargs, lds, s);
  if (e != hipSuccess) return (int)e;
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// One-output layers (the last layer of QTRAN's joint-Q / V heads, network/mixer.py:384-388 / :410-414, and of QMIX's hyper_b2,
// mixer.py:44-46): dW[k] = sum_m g[m] x[m][k] is a weighted column sum - HBM bound on reading x once.  The MFMA kernels above spend
// a 64-column output block on it (M = 76 800, K = 64: 49 us for 20 MB).  Here thread (rl, cg) walks rows rl, rl + RL, ... of its
// slab with one 16-byte load per row, four rows in flight; the RL partial sums of a column are added in a fixed order.
__global__ __launch_bounds__(256) void wgrad_thin_kernel(WgradArgs a) {
  __shared__ float part[256 * 4 + 256];
  const int tid = threadIdx.x;
  const int KG = (a.K + 3) >> 2, RL = 256 / KG;
  const int cg = tid % KG, rl = tid / KG;
  const long per = ((long)a.M + gridDim.x - 1) / gridDim.x;
  const long r_begin = (long)blockIdx.x * per;
  long r_end = r_begin + per; if (r_end > a.M) r_end = a.M;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float gs = 0.f;
  if (rl < RL) {
    const float* xp = a.x.p0 + 4 * cg;
    long r = r_begin + rl;
    for (; r + 3L * RL < r_end; r += 4L * RL) {
      float g[4]; f32x4 xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { g[u] = a.G[(r + (long)u * RL) * a.ldg]; xv[u] = *reinterpret_cast<const f32x4*>(xp + (r + (long)u * RL) * a.x.ld0); }
#pragma unroll
      for (int u = 0; u < 4; ++u) { acc += g[u] * xv[u]; gs += g[u]; }
    }
    for (; r < r_end; r += RL) {
      const float g = a.G[r * a.ldg];
      acc += g * *reinterpret_cast<const f32x4*>(xp + r * a.x.ld0);
      gs += g;
    }
  }
  *reinterpret_cast<f32x4*>(part + tid * 4) = acc;          // [rl][cg][4] = [rl][column]
  if (cg == 0 && rl < RL) part[1024 + rl] = gs;
  __syncthreads();
  float* slab = a.ws + (long)blockIdx.x * (a.K + 1);
  if (tid < a.K) {
    float t = 0.f;
    for (int j = 0; j < RL; ++j) t += part[(j * KG) * 4 + tid];
    slab[tid] = t;
  } else if (tid == a.K) {
    float t = 0.f;
    for (int j = 0; j < RL; ++j) t += part[1024 + j];
    slab[a.K] = t;
  }
}

// returns -1 when the shape is not covered
inline int try_wgrad_thin(WgradArgs& a, hipStream_t s) {
  if (a.N != 1 || a.groups != 1 || a.Yact || !a.xvec || a.M < 4096 || a.K > 252 || (a.x.ld0 & 3)) return -1;
  if (!a.x.p0 || a.x.k0 != a.K || a.x.k1 || a.x.nhot || a.x.nid || a.x.m0 || a.x.rpe0 || a.x.emap0) return -1;
  if (a.x.ld0 < (a.K + 3) / 4 * 4) return -1;                // the last 16-byte load of a row stays inside its pitch
  hipLaunchKernelGGL(wgrad_thin_kernel, dim3(a.slabs), dim3(256), 0, s, a);
  return 0;
}

struct WredArgs {
  const float* ws; float* dW; long lddw; float* db;
  int N, K, slabs, groups; long gs_dw, gs_db;
};

// 64 output elements per block, 16 slab groups per element: thread (e, sg) sums slabs sg, sg+16, ... in order (loads eight
// at a time in flight), then the 16 partial sums are added in a fixed order -> deterministic, short dependent chains
// (512 slabs: 38 us with 4 groups - the chain of 128 dependent adds was the whole kernel)
constexpr int RSG = 16;
__global__ __launch_bounds__(64 * RSG) void wgrad_reduce_kernel(WredArgs a) {
  __shared__ float part[RSG][64];
  const int Kext = a.K + 1;
  const long per = (long)a.N * Kext;
  const long total = per * a.groups;
  const int el = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const long e = (long)blockIdx.x * 64 + el;
  float s = 0.f;
  if (e < total) {
    const int g = (int)(e / per);
    const long r = e - (long)g * per;
    const long stride = (long)a.groups * per;
    const float* p = a.ws + (long)g * per + r;
    int sl = sg;
    for (; sl + 7 * RSG < a.slabs; sl += 8 * RSG) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(long)(sl + u * RSG) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; sl < a.slabs; sl += RSG) s += p[(long)sl * stride];
  }
  part[sg][el] = s;
  __syncthreads();
  if (sg == 0 && e < total) {
    const int g = (int)(e / per);
    const long r = e - (long)g * per;
    const int n = (int)(r / Kext), k = (int)(r - (long)n * Kext);
    float v = part[0][el];
#pragma unroll
    for (int u = 1; u < RSG; ++u) v += part[u][el];
    if (k < a.K) a.dW[g * a.gs_dw + (long)n * a.lddw + k] += v;
    else if (a.db) a.db[g * a.gs_db + n] += v;
  }
}

inline ConcatSrc to_src(const marl_src_t* s) {
  ConcatSrc c;
  c.p0 = s->p0; c.ld0 = s->ld0; c.k0 = s->k0;
  c.p1 = s->p1; c.ld1 = s->ld1; c.k1 = s->k1;
  c.idx = s->idx; c.nhot = s->nhot; c.hot_w = s->hot_w; c.nid = s->nid;
  c.m0 = s->m0; c.ldm0 = s->ldm0;
  c.rpe0 = s->rpe0; c.bs0 = s->bs0; c.off0 = s->off0;
  c.rpei = s->rpei; c.bsi = s->bsi; c.offi = s->offi;
  c.emap0 = s->emap0;
  c.fd0 = make_fastdiv((unsigned)(s->rpe0 > 0 ? s->rpe0 : 1));
  c.fdi = make_fastdiv((unsigned)(s->rpei > 0 ? s->rpei : 1));
  c.fdn = make_fastdiv((unsigned)(s->nid > 0 ? s->nid : 1));
  return c;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

ST_DEFINE_SETTER(marl_debug_stamps_wgrad)

extern "C" int marl_linear(const marl_src_t* x, const float* W, long ldw, int w_kmajor, const float* bias,
                           float* Y, long ldy, int M, int N, int K, int act, float beta,
                           const marl_group_t* grp, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  LinArgs a;
  a.x = to_src(x);
  if (concat_width(a.x) != K) return (int)hipErrorInvalidValue;
  a.W = W; a.ldw = ldw; a.bias = bias; a.Y = Y; a.ldy = ldy; a.M = M; a.N = N; a.K = K;
  a.act = act; a.beta = beta;
  a.groups = grp ? grp->groups : 1;
  a.gs_x0 = grp ? grp->gs_x0 : 0; a.gs_x1 = grp ? grp->gs_x1 : 0; a.gs_w = grp ? grp->gs_w : 0;
  a.gs_b = grp ? grp->gs_b : 0; a.gs_y = grp ? grp->gs_y : 0; a.gs_m0 = grp ? grp->gs_m0 : 0;
  // fast-path mode of the chunks inside dense segment 0: 1 = 16-byte loads, 2 = dword loads, 0 = none
  int amode = 0;
  const bool gate = a.x.m0 != nullptr;
  if (a.x.p0 && a.x.k0 >= 16) {
    bool al = (a.x.ld0 % 4 == 0) && aligned16(a.x.p0) && (a.gs_x0 % 4 == 0);
    if (gate) al = al && (a.x.ldm0 % 4 == 0) && aligned16(a.x.m0) && (a.gs_m0 % 4 == 0);
    amode = al ? 1 : 2;
  }
  const bool bf = (act & 0x100) != 0;        // bf16 operands, fp32 accumulate (mixer GEMMs, opt-in)
  const int nc = (!bf && N > 64 && N <= 80) ? 5 : 4;
  dim3 grid((M + 127) / 128, (N + 16 * nc - 1) / (16 * nc), a.groups), block(256);
  hipStream_t s = (hipStream_t)stream;
  a.act = act & 0xff;
#define LIN_GT(AM, KM, BFV, NCV) do { if (gate) hipLaunchKernelGGL((linear_kernel<AM, KM, BFV, true, NCV>), grid, block, 0, s, a); \
                                     else hipLaunchKernelGGL((linear_kernel<AM, KM, BFV, false, NCV>), grid, block, 0, s, a); } while (0)
#define LIN_KM(AM, BFV, NCV) do { if (w_kmajor) LIN_GT(AM, true, BFV, NCV); else LIN_GT(AM, false, BFV, NCV); } while (0)
#define LIN_AM(BFV, NCV) do { if (amode == 1) LIN_KM(1, BFV, NCV); else if (amode == 2) LIN_KM(2, BFV, NCV); else LIN_KM(0, BFV, NCV); } while (0)
  if (bf) LIN_AM(true, 4);
  else if (nc == 5) LIN_AM(false, 5);
  else LIN_AM(false, 4);
#undef LIN_AM
#undef LIN_KM
#undef LIN_GT
  MARL_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t marl_linear_wgrad_workspace(int M, int N, int K, int groups) {
  int slabs = marl_wgrad_slabs(M);
  if (groups == 1 && ((N <= 80 && K + 1 <= 80) || N == 64)) slabs *= 2;   // the full-width and the tall kernel run two workgroups per CU
  return (size_t)slabs * groups * N * (K + 4) * sizeof(float);      // + 3: column passes of the direct kernel each carry a bias column
}

extern "C" int marl_wgrad_slabs(int M) {
  long chunks = ((long)M + 63) / 64;
  long s = chunks / 8;            // >= 8 chunks of 64 rows per block
  if (s < 1) s = 1;
  if (s > 256) s = 256;
  return (int)s;
}

extern "C" int marl_linear_wgrad(const float* G, long ldg, const float* Yact, long ldya, const marl_src_t* x,
                                 float* dW, long lddw, float* db, int M, int N, int K, int flags,
                                 const marl_group_t* grp, float* ws, size_t ws_bytes, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int groups = grp ? grp->groups : 1;
  if (ws_bytes < marl_linear_wgrad_workspace(M, N, K, groups)) return (int)hipErrorInvalidValue;
  WgradArgs a;
  a.G = G; a.ldg = ldg; a.Yact = Yact; a.ldya = ldya; a.x = to_src(x);
  if (concat_width(a.x) != K) return (int)hipErrorInvalidValue;
  a.ws = ws; a.M = M; a.N = N; a.K = K; a.slabs = marl_wgrad_slabs(M);
  a.nyb = (N + 63) / 64; a.groups = groups;
  a.gs_g = grp ? grp->gs_y : 0; a.gs_ya = grp ? grp->gs_m0 : 0;
  a.gs_x0 = grp ? grp->gs_x0 : 0; a.gs_x1 = grp ? grp->gs_x1 : 0;
  a.gvec = (ldg % 4 == 0) && aligned16(G) && (a.gs_g % 4 == 0) &&
           (!Yact || ((ldya % 4 == 0) && aligned16(Yact) && (a.gs_ya % 4 == 0)));
  a.xvec = a.x.p0 && !a.x.m0 && (a.x.ld0 % 4 == 0) && aligned16(a.x.p0) && (a.gs_x0 % 4 == 0);
  hipStream_t s = (hipStream_t)stream;
  const bool bf = (flags & 1) != 0;           // bf16 operands, fp32 accumulate (mixer GEMMs, opt-in)
  if (!bf) {                                  // 64-output layers over many rows (fc1 of the agent): LDS-staged tall kernel
    const int slabs0 = a.slabs;
    const int rc = try_wgrad_tall(a, ws_bytes, s);
    if (rc > 0) return rc;
    if (rc == 0) {
      MARL_CHECK_LAUNCH();
      WredArgs r;
      r.ws = ws; r.dW = dW; r.lddw = lddw; r.db = db; r.N = N; r.K = K; r.slabs = a.slabs; r.groups = 1; r.gs_dw = 0; r.gs_db = 0;
      const long total = (long)N * (K + 1);
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64 * RSG), 0, s, r);
      MARL_CHECK_LAUNCH();
      return 0;
    }
    a.slabs = slabs0;
  }
  // Wide inputs of a 64-output layer (fc1 of the agent on 3s5z / MMM2: 150 / 204 columns - more accumulator tiles than
  // one pass of the direct kernel holds): column passes [0,64), [64,128) as permuted blocks and the rest as plain
  // tiles, each a direct launch over its own sub-source with its own slabs; G is re-read per pass (3 x 315 MB at
  // MMM2 / 1024 envs) but nothing is staged through LDS (the LDS-staged kernel took 1.35 ms there).
  if (!bf && a.N == 64 && groups == 1 && !Yact && a.gvec && a.xvec && a.M >= 4096 && !a.x.k1 && !a.x.m0 && a.x.k0 >= 64 &&
      K - 64 > 64) {
    const int c_rest = a.x.k0 >= 128 ? 128 : 64;
    if ((K - c_rest + 15) / 16 <= 5) {
      size_t ws_off = 0;
      int c0 = 0;
      for (int pass = 0; c0 < K; ++pass) {
        WgradArgs b = a;
        const bool perm = c0 < c_rest;
        const int kw = perm ? 64 : K - c0;
        b.x.p0 = a.x.p0 + c0;
        b.x.k0 = perm ? 64 : a.x.k0 - c0;
        if (perm) { b.x.idx = nullptr; b.x.nhot = 0; b.x.hot_w = 0; b.x.nid = 0; }
        b.K = kw;
        b.ws = ws + ws_off;
        if (try_wgrad_direct(b, s) != 0) return (int)hipErrorInvalidValue;
        MARL_CHECK_LAUNCH();
        WredArgs r;
        r.ws = b.ws; r.dW = dW + c0; r.lddw = lddw; r.db = pass == 0 ? db : nullptr; r.N = N; r.K = kw; r.slabs = a.slabs;
        r.groups = 1; r.gs_dw = 0; r.gs_db = 0;
        const long total = (long)N * (kw + 1);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64 * RSG), 0, s, r);
        MARL_CHECK_LAUNCH();
        ws_off += (size_t)a.slabs * N * (kw + 1);
        c0 += kw;
      }
      return 0;
    }
  }
  int done = try_wgrad_thin(a, s);            // fp32 either way: one output column is not matrix work
  if (done != 0) done = bf ? -1 : try_wgrad_direct(a, s);
  bool one_base = a.x.p0 && !a.x.m0 && !a.x.nid && a.x.k0 < 16384 && a.x.k1 < 16384 && a.x.nhot < 16384 && a.x.hot_w < 0xfff0;
  for (int c4 = 0; one_base && c4 < K; c4 += 4) {       // segment of the first and the last real column of each item
    auto seg = [&](int k) { return k < a.x.k0 ? 0 : (k - a.x.k0 < a.x.k1 ? 1 : 2); };
    const int kl = c4 + 3 < K ? c4 + 3 : K - 1;
    if (seg(c4) != seg(kl)) one_base = false;
  }
  if (done != 0 && !bf && groups == 1 && N <= WF && K + 1 <= WF && a.gvec && ldg >= (N + 3) / 4 * 4 &&
      (!Yact || ldya >= (N + 3) / 4 * 4) && M >= 2048 && one_base) {
    // narrow layer: one pass over G and X
    const size_t lds = (size_t)2 * WCH * WFS * sizeof(float) + (size_t)2 * WCH * 4 * sizeof(long) + (size_t)5 * 256 * 16;
    const long chunks = ((long)M + WCH - 1) / WCH;
    a.slabs = (int)(2L * a.slabs < chunks ? 2L * a.slabs : (chunks < 1 ? 1 : chunks));
    const void* fn = Yact ? (const void*)wgrad_full_kernel<true> : (const void*)wgrad_full_kernel<false>;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      // 67.5 KB > the 64 KB default
    if (e != hipSuccess) return (int)e;
    if (Yact) hipLaunchKernelGGL(wgrad_full_kernel<true>, dim3(a.slabs), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(wgrad_full_kernel<false>, dim3(a.slabs), dim3(256), lds, s, a);
    done = 0;
  }
  if (done != 0) {
    dim3 grid(a.slabs, a.nyb * groups, (K + 1 + 63) / 64), block(256);
    if (bf) hipLaunchKernelGGL(wgrad_kernel<true>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(wgrad_kernel<false>, grid, block, 0, s, a);
  }
  MARL_CHECK_LAUNCH();
  WredArgs r;
  r.ws = ws; r.dW = dW; r.lddw = lddw; r.db = db; r.N = N; r.K = K; r.slabs = a.slabs; r.groups = groups;
  r.gs_dw = grp ? grp->gs_w : 0; r.gs_db = grp ? grp->gs_b : 0;
  long total = (long)N * (K + 1) * groups;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64 * RSG), 0, s, r);
  MARL_CHECK_LAUNCH();
  return 0;
}
