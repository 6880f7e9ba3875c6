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

// block = 256 threads = 4 waves; wave tile = 32 rows x 64 cols (2 x 4 MFMA tiles)
template <bool VEC, bool W_KMAJOR>
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
  const int col0 = blockIdx.y * 64;
  if (row0 >= a.M) return;
  int ct_used = (a.N - col0 + 15) / 16;
  if (ct_used > 4) ct_used = 4;

  long arow[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    long rr = row0 + r * 16 + m;
    arow[r] = rr < a.M ? rr : a.M - 1;
  }
  int bcol[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    int cc = col0 + c * 16 + m;
    bcol[c] = cc < a.N ? cc : a.N - 1;
  }

  f32x4 acc[2][4];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int K = a.K;
  const int kfull = VEC ? (K & ~15) : 0;
  long asrc[2];   // source rows of dense segment 0 after the (T+1)-slot remap (vector path only)
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    bool ok;
    asrc[r] = remap_row(arow[r], x.rpe0, x.bs0, x.off0, ok);
  }
  for (int k0 = 0; k0 < kfull; k0 += 16) {   // vector path: 16 B per lane per operand
    const int kk = k0 + 4 * q;
    f32x4 av[2], bv[4];
#pragma unroll
    for (int r = 0; r < 2; ++r) av[r] = *reinterpret_cast<const f32x4*>(x.p0 + asrc[r] * x.ld0 + kk);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c < ct_used) {
        if (!W_KMAJOR) {
          bv[c] = *reinterpret_cast<const f32x4*>(W + (long)bcol[c] * a.ldw + kk);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) bv[c][i] = W[(long)(kk + i) * a.ldw + bcol[c]];
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c < ct_used) {
#pragma unroll
        for (int r = 0; r < 2; ++r) acc[r][c] = mfma16x4(av[r], bv[c], acc[r][c]);
      }
    }
  }
  for (int k0 = kfull; k0 < K; k0 += 16) {   // generic path: guarded element loads
    const int kk = k0 + 4 * q;
    f32x4 av[2], bv[4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) av[r][i] = (kk + i < K) ? concat_elem(x, arow[r], kk + i) : 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
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
    for (int c = 0; c < 4; ++c) {
      if (c < ct_used) {
#pragma unroll
        for (int r = 0; r < 2; ++r) acc[r][c] = mfma16x4(av[r], bv[c], acc[r][c]);
      }
    }
  }

  // epilogue: D-layout -> Y (16 lanes = 64 contiguous bytes per row)
#pragma unroll
  for (int c = 0; c < 4; ++c) {
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

// block = 4 waves; all waves own the same 64(n) x 64(k) tile of dW and split the slab's rows;
// operands are D-layout loads (4 rows per lane) which ARE the A^T / B fragments - no LDS staging.
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradArgs a) {
  __shared__ float red[4][64 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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

  int nt_used = (a.N - n0 + 15) / 16; if (nt_used > 4) nt_used = 4;
  int kt_used = (Kext - k0 + 15) / 16; if (kt_used > 4) kt_used = 4;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const long tiles = ((long)a.M + 15) / 16;
  const long per = (tiles + a.slabs - 1) / a.slabs;
  const long t_begin = (long)blockIdx.x * per;
  long t_end = t_begin + per; if (t_end > tiles) t_end = tiles;

  // Column mapping of the 4 MFMA tiles inside the 64-wide block.  "perm" (block wider than 16 columns):
  // tile c of lane m is column 4m + c, so ONE float4 per lane and row feeds all four tiles and 16 lanes
  // read 256 contiguous bytes; otherwise (narrow blocks, e.g. N = 1) tile c is columns 16c .. 16c+15.
  const bool gperm = (a.N - n0) > 16, xperm = (Kext - k0) > 16;
  const int gt_used = gperm ? 4 : nt_used, xt_used = xperm ? 4 : kt_used;
  const bool gvec = gperm && a.gvec, xvec = xperm && a.xvec;

  // operand tiles of row tile tt in accumulator layout; rows past M and columns past N / K+1 read zero
#define WG_LOAD_TILE(GV, XV, tt, en)                                                                    \
  {                                                                                                      \
    const long r0_ = (tt) * 16 + 4 * q;                                                                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                      \
      const long row = r0_ + i;                                                                          \
      const bool okr = (en) && row < a.M;                                                                \
      const ConcatRow cr = concat_row(x, okr ? row : 0);                                                 \
      f32x4 g4 = {0.f, 0.f, 0.f, 0.f}, x4 = {0.f, 0.f, 0.f, 0.f};                                        \
      if (okr) {                                                                                         \
        const int nb = n0 + 4 * m;                                                                       \
        if (gvec && nb + 3 < a.N) {                                                                      \
          g4 = *reinterpret_cast<const f32x4*>(G + row * a.ldg + nb);                                    \
          if (Ya) {                                                                                      \
            const f32x4 y4 = *reinterpret_cast<const f32x4*>(Ya + row * a.ldya + nb);                    \
            _Pragma("unroll") for (int c = 0; c < 4; ++c) g4[c] = y4[c] > 0.f ? g4[c] : 0.f;             \
          }                                                                                              \
        } else {                                                                                         \
          _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                \
            const int n = gperm ? nb + c : n0 + 16 * c + m;                                              \
            if (c < gt_used && n < a.N) {                                                                \
              float v = G[row * a.ldg + n];                                                              \
              if (Ya) v = Ya[row * a.ldya + n] > 0.f ? v : 0.f;                                          \
              g4[c] = v;                                                                                 \
            }                                                                                            \
          }                                                                                              \
        }                                                                                                \
        const int kb = k0 + 4 * m;                                                                       \
        if (xvec && cr.ok0 && kb + 3 < x.k0) {                                                           \
          x4 = *reinterpret_cast<const f32x4*>(x.p0 + cr.r0 * x.ld0 + kb);                               \
        } else {                                                                                         \
          _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                \
            const int k = xperm ? kb + c : k0 + 16 * c + m;                                              \
            if (c < xt_used) {                                                                           \
              if (k < a.K) x4[c] = concat_at(x, cr, k);                                                  \
              else if (k == a.K) x4[c] = 1.f;    /* virtual ones column => bias gradient */              \
            }                                                                                            \
          }                                                                                              \
        }                                                                                                \
      }                                                                                                  \
      _Pragma("unroll") for (int c = 0; c < 4; ++c) { GV[c][i] = g4[c]; XV[c][i] = x4[c]; }              \
    }                                                                                                    \
  }

  // (a register double-buffer of the next tile was tried: it halves occupancy (266 regs) and was 1.5x slower;
  //  two resident workgroups per CU hide the load latency better)
  for (long t = t_begin + wave; t < t_end; t += 4) {
    f32x4 gA[4], xA[4];
    WG_LOAD_TILE(gA, xA, t, true)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
        if (nt < gt_used && kt < xt_used) acc[nt][kt] = mfma16x4(gA[nt], xA[kt], acc[nt][kt]);
  }

  // cross-wave reduction of the 64x64 tile through LDS, then one slab partial per block
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int nl = gperm ? 4 * (4 * q + i) + nt : nt * 16 + 4 * q + i;      // local n (row of dW)
        const int kl = xperm ? 4 * m + kt : kt * 16 + m;                         // local k (column of dW)
        red[wave][nl * 64 + kl] = acc[nt][kt][i];
      }
  __syncthreads();
  float* ws = a.ws + ((long)blockIdx.x * a.groups + g) * (long)a.N * Kext;
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int n = n0 + e / 64, k = k0 + e % 64;
    if (n < a.N && k < Kext) ws[(long)n * Kext + k] = red[0][e] + red[1][e] + red[2][e] + red[3][e];
  }
}

struct WredArgs {
  const float* ws; float* dW; long lddw; float* db;
  int N, K, slabs, groups; long gs_dw, gs_db;
};

__global__ void wgrad_reduce_kernel(WredArgs a) {
  const int Kext = a.K + 1;
  const long per = (long)a.N * Kext;
  const long total = per * a.groups;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int g = e / per;
    const long r = e % per;
    const int n = r / Kext, k = r % Kext;
    float s = 0.f;
    for (int sl = 0; sl < a.slabs; ++sl) s += a.ws[((long)sl * a.groups + g) * per + r];
    if (k < a.K) a.dW[g * a.gs_dw + (long)n * a.lddw + k] += s;
    else if (a.db) a.db[g * a.gs_db + n] += s;
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
  return c;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

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
  bool vec = a.x.k0 >= 16 && !a.x.m0 && (a.x.rpe0 == 0 || a.x.off0 >= 0) && (a.x.ld0 % 4 == 0) && aligned16(a.x.p0) && (a.gs_x0 % 4 == 0);
  if (!w_kmajor) vec = vec && (ldw % 4 == 0) && aligned16(W) && (a.gs_w % 4 == 0);
  // the vector loop only covers whole 16-chunks that lie inside dense segment 0
  LinArgs av = a;
  dim3 grid((M + 127) / 128, (N + 63) / 64, a.groups), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (vec && a.x.k0 == K) {
    if (w_kmajor) hipLaunchKernelGGL((linear_kernel<true, true>), grid, block, 0, s, av);
    else hipLaunchKernelGGL((linear_kernel<true, false>), grid, block, 0, s, av);
  } else {
    if (w_kmajor) hipLaunchKernelGGL((linear_kernel<false, true>), grid, block, 0, s, av);
    else hipLaunchKernelGGL((linear_kernel<false, false>), grid, block, 0, s, av);
  }
  MARL_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t marl_linear_wgrad_workspace(int M, int N, int K, int groups) {
  int slabs = marl_wgrad_slabs(M);
  return (size_t)slabs * groups * N * (K + 1) * sizeof(float);
}

extern "C" int marl_wgrad_slabs(int M) {
  long tiles = ((long)M + 15) / 16;
  long s = tiles / 32;            // >= 32 row tiles (8 per wave) per block
  if (s < 1) s = 1;
  if (s > 256) s = 256;
  return (int)s;
}

extern "C" int marl_linear_wgrad(const float* G, long ldg, const float* Yact, long ldya, const marl_src_t* x,
                                 float* dW, long lddw, float* db, int M, int N, int K,
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
  dim3 grid(a.slabs, a.nyb * groups, (K + 1 + 63) / 64), block(256);
  hipLaunchKernelGGL(wgrad_kernel, grid, block, 0, s, a);
  MARL_CHECK_LAUNCH();
  WredArgs r;
  r.ws = ws; r.dW = dW; r.lddw = lddw; r.db = db; r.N = N; r.K = K; r.slabs = a.slabs; r.groups = groups;
  r.gs_dw = grp ? grp->gs_w : 0; r.gs_db = grp ? grp->gs_b : 0;
  long total = (long)N * (K + 1) * groups;
  int rb = (int)((total + 255) / 256); if (rb > 1024) rb = 1024;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rb), dim3(256), 0, s, r);
  MARL_CHECK_LAUNCH();
  return 0;
}
