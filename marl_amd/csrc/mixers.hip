// Per-row kernels of the learner: action-value selection, agent sums, the QMIX and QPLEX mixing
// epilogues (after their hypernet GEMMs) and the TD / QTRAN losses.  HBM-bound elementwise work:
// coalesced over the row axis, reductions by wave shuffles, deterministic two-stage sums.
#include "common.h"
#include "../../include/marl_hip.h"

namespace {

constexpr int TPB = 256;
inline int nblk(long n, int cap = 65535 * 16) {
  long b = (n + TPB - 1) / TPB;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

__global__ void q_gather_kernel(const float* q, const int* idx, const float* avail, float mask_val, float* out,
                                long rows, int A) {
  for (long r = (long)blockIdx.x * TPB + threadIdx.x; r < rows; r += (long)gridDim.x * TPB) {
    const int a = idx[r];
    float v = 0.f;
    if (a >= 0) v = (avail && avail[r * A + a] == 0.f) ? mask_val : q[r * A + a];
    out[r] = v;
  }
}

__global__ void q_masked_max_kernel(const float* q, const float* avail, float mask_val, float* out_max,
                                    int* out_arg, long rows, int A) {
  for (long r = (long)blockIdx.x * TPB + threadIdx.x; r < rows; r += (long)gridDim.x * TPB) {
    float best = 0.f;
    int arg = 0;
    for (int a = 0; a < A; ++a) {
      float v = q[r * A + a];
      if (avail && avail[r * A + a] == 0.f) v = mask_val;
      if (a == 0 || v > best) { best = v; arg = a; }   // strict >: first index wins ties (torch)
    }
    if (out_max) out_max[r] = best;
    if (out_arg) out_arg[r] = arg;
  }
}

// Double-Q selection in one pass (reference q_learner.py:104-117): arg = first-index argmax over the available
// actions of q_sel (the eval net on the next observations), out = q_val (target net) at arg, masked the same way.
// A wave stages 64 rows of each operand through LDS with fully coalesced loads (rows are A floats = 44 B for
// 2s3z, a thread-per-row global read pattern wastes most of every 64-byte request); stride A is odd or the
// tile is padded to an odd stride, so the per-row LDS reads are conflict-free.
constexpr int DS_ROWS = 64;      // rows per wave-tile
__global__ __launch_bounds__(256) void q_double_select_kernel(const float* q_sel, const float* q_val, const float* avail,
                                                              float mask_val, float* out_val, int* out_arg, long rows, int A, int vec) {
  extern __shared__ __attribute__((aligned(16))) float ds_smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int TS = (DS_ROWS * A + 3) & ~3;                  // floats per operand tile (16-byte multiple)
  float* Sq = ds_smem + (size_t)wave * 3 * TS;
  float* Sv = Sq + TS;
  float* Sa = Sv + TS;
  const long tiles = (rows + DS_ROWS - 1) / DS_ROWS;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < tiles; tile += (long)gridDim.x * 4) {
    const long r0 = tile * DS_ROWS;
    const int n = (int)((rows - r0 < DS_ROWS ? rows - r0 : DS_ROWS) * A);
    // the tile is one contiguous run of n floats (64 rows x A): copied as it lies, 16 bytes per lane, no index math
    // (r0 * A * 4 is a multiple of 16, the operands themselves are 16-byte aligned - checked on the host)
    const float* gq = q_sel + r0 * A;
    const float* gv = q_val + r0 * A;
    const float* ga = avail ? avail + r0 * A : nullptr;
    const int n4 = vec ? n >> 2 : 0;                   // (operands not 16-byte aligned: plain element copy)
    for (int e = lane; e < n4; e += 64) {
      reinterpret_cast<f32x4*>(Sq)[e] = reinterpret_cast<const f32x4*>(gq)[e];
      reinterpret_cast<f32x4*>(Sv)[e] = reinterpret_cast<const f32x4*>(gv)[e];
      reinterpret_cast<f32x4*>(Sa)[e] = ga ? reinterpret_cast<const f32x4*>(ga)[e] : (f32x4){1.f, 1.f, 1.f, 1.f};
    }
    for (int e = 4 * n4 + lane; e < n; e += 64) {
      Sq[e] = gq[e]; Sv[e] = gv[e]; Sa[e] = ga ? ga[e] : 1.f;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);                   // this wave's LDS writes (lgkmcnt(0)); no cross-wave sharing
    const long r = r0 + lane;
    if (r < rows) {
      float best = 0.f; int arg = 0;
      for (int a = 0; a < A; ++a) {
        float v = Sq[lane * A + a];
        if (Sa[lane * A + a] == 0.f) v = mask_val;
        if (a == 0 || v > best) { best = v; arg = a; }   // strict >: first index wins ties (torch)
      }
      out_val[r] = Sa[lane * A + arg] == 0.f ? mask_val : Sv[lane * A + arg];
      if (out_arg) out_arg[r] = arg;
    }
  }
}

// q_masked_max with the staging of q_double_select_kernel: a wave copies 64 rows of q (and avail) into LDS with 16-byte coalesced
// loads, then one lane per row scans its A values (the thread-per-row kernel above reads 44 - 72 byte rows at a lane stride of a
// row: 50 us for 614 400 rows x 14 actions where the bytes are worth 12 us).  Same results (strict >: first index wins ties).
__global__ __launch_bounds__(256) void q_masked_max_tiled_kernel(const float* q, const float* avail, float mask_val, float* out_max,
                                                                 int* out_arg, long rows, int A) {
  extern __shared__ __attribute__((aligned(16))) float ds_smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int TS = (DS_ROWS * A + 3) & ~3;
  float* Sq = ds_smem + (size_t)wave * 2 * TS;
  float* Sa = Sq + TS;
  const long tiles = (rows + DS_ROWS - 1) / DS_ROWS;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < tiles; tile += (long)gridDim.x * 4) {
    const long r0 = tile * DS_ROWS;
    const int n = (int)((rows - r0 < DS_ROWS ? rows - r0 : DS_ROWS) * A);
    const float* gq = q + r0 * A;
    const float* ga = avail ? avail + r0 * A : nullptr;
    const int n4 = n >> 2;
    for (int e = lane; e < n4; e += 64) {
      reinterpret_cast<f32x4*>(Sq)[e] = reinterpret_cast<const f32x4*>(gq)[e];
      reinterpret_cast<f32x4*>(Sa)[e] = ga ? reinterpret_cast<const f32x4*>(ga)[e] : (f32x4){1.f, 1.f, 1.f, 1.f};
    }
    for (int e = 4 * n4 + lane; e < n; e += 64) { Sq[e] = gq[e]; Sa[e] = ga ? ga[e] : 1.f; }
    __builtin_amdgcn_s_waitcnt(0xC07F);                   // this wave's LDS writes (lgkmcnt(0)); no cross-wave sharing
    const long r = r0 + lane;
    if (r < rows) {
      float best = 0.f; int arg = 0;
      for (int a = 0; a < A; ++a) {
        float v = Sq[lane * A + a];
        if (Sa[lane * A + a] == 0.f) v = mask_val;
        if (a == 0 || v > best) { best = v; arg = a; }
      }
      if (out_max) out_max[r] = best;
      if (out_arg) out_arg[r] = arg;
    }
  }
}

__global__ void q_scatter_kernel(float* dq, const int* idx1, const float* g1, const int* idx2, const float* g2,
                                 long rows, int A, int gdiv) {
  for (long r = (long)blockIdx.x * TPB + threadIdx.x; r < rows; r += (long)gridDim.x * TPB) {
    const int a1 = idx1 ? idx1[r] : -1;
    const int a2 = idx2 ? idx2[r] : -1;
    const float v1 = g1 ? g1[r / gdiv] : 0.f;
    const float v2 = g2 ? g2[r / gdiv] : 0.f;
    for (int a = 0; a < A; ++a) {
      float v = 0.f;
      if (a == a1) v += v1;
      if (a == a2) v += v2;
      dq[r * A + a] = v;
    }
  }
}

__global__ void vec_add_kernel(const float* a, const float* b, float* out, long n) {
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) out[i] = a[i] + b[i];
}

// in rows (r*N + n) of stride ld_in, out rows r of stride ld_out (strides >= D: intermediates are padded to 16-byte rows)
__global__ void agent_sum_kernel(const float* in, long ld_in, float* out, long ld_out, long rows, int N, int D) {
  const long total = rows * D;
  for (long e = (long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long)gridDim.x * TPB) {
    const long r = e / D;
    const int d = (int)(e - r * D);
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += in[(r * N + n) * ld_in + d];
    out[r * ld_out + d] = s;
  }
}

__global__ void agent_bcast_kernel(const float* in, long ld_in, float* out, long ld_out, long rows, int N, int D, int acc) {
  const long total = rows * N * D;
  for (long e = (long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long)gridDim.x * TPB) {
    const long rn = e / D;
    const int d = (int)(e - rn * D);
    const long r = rn / N;
    const float v = in[r * ld_in + d];
    float* o = out + rn * ld_out + d;
    *o = acc ? *o + v : v;
  }
}

// ---- QMIX: 32 lanes per row (lane = embed unit e), two rows per wave ---------------------------
// b2 == nullptr: the state-conditioned scalar bias is formed here from the relu'd hidden layer in hy's last E columns,
// b2 = w22 . hb + b22 (hyper_b2.2, network/mixer.py:46-47)
__global__ void qmix_mix_fwd_kernel(const float* hy, long ldh, const float* b2, const float* w22, const float* b22, const float* q,
                                    float* q_tot, long rows, int N, int E) {
  const int half = (threadIdx.x & 63) >> 5, l = threadIdx.x & 31;
  const long wave_id = ((long)blockIdx.x * TPB + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * TPB) >> 6;
  for (long r2 = wave_id; r2 * 2 < rows; r2 += nwaves) {
    const long r = r2 * 2 + half;
    float part = 0.f;
    if (r < rows) {
      const float* h = hy + r * ldh;
      for (int e = l; e < E; e += 32) {
        float a = h[N * E + e];                                  // b1
        for (int n = 0; n < N; ++n) a += q[r * N + n] * fabsf(h[n * E + e]);
        const float hid = a > 0.f ? a : (__expf(a) - 1.f);       // elu, alpha = 1
        part += hid * fabsf(h[N * E + E + e]);
        if (!b2) part += w22[e] * h[N * E + 2 * E + e];
      }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if (r < rows && l == 0) q_tot[r] = part + (b2 ? b2[r] : b22[0]);
  }
}

// w22 != nullptr: also d hb = dq_tot w22 (hb > 0) into dhy's last E columns (the backward of hyper_b2.2 and its relu)
__global__ void qmix_mix_bwd_kernel(const float* hy, long ldh, const float* q, const float* dq_tot, const float* w22, float* dhy,
                                    float* db2, float* dq, long rows, int N, int E) {
  const int half = (threadIdx.x & 63) >> 5, l = threadIdx.x & 31;
  const long wave_id = ((long)blockIdx.x * TPB + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * TPB) >> 6;
  for (long r2 = wave_id; r2 * 2 < rows; r2 += nwaves) {
    const long r = r2 * 2 + half;
    const bool ok = r < rows;
    const float g = ok ? dq_tot[r] : 0.f;
    float dqn[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) dqn[n] = 0.f;
    if (ok) {
      const float* h = hy + r * ldh;
      float* dh = dhy + r * ldh;
      for (int e = l; e < E; e += 32) {
        float a = h[N * E + e];
        for (int n = 0; n < N; ++n) a += q[r * N + n] * fabsf(h[n * E + e]);
        const float ex = __expf(a);
        const float hid = a > 0.f ? a : ex - 1.f;
        const float w2r = h[N * E + E + e];
        const float sgn2 = w2r > 0.f ? 1.f : (w2r < 0.f ? -1.f : 0.f);
        dh[N * E + E + e] = g * hid * sgn2;                       // d w2raw
        const float dpre = g * fabsf(w2r) * (a > 0.f ? 1.f : ex);
        dh[N * E + e] = dpre;                                     // d b1
        if (w22) dh[N * E + 2 * E + e] = h[N * E + 2 * E + e] > 0.f ? g * w22[e] : 0.f;
        for (int n = 0; n < N; ++n) {
          const float w1r = h[n * E + e];
          const float sgn1 = w1r > 0.f ? 1.f : (w1r < 0.f ? -1.f : 0.f);
          dh[n * E + e] = q[r * N + n] * dpre * sgn1;             // d w1raw
          if (n < 16) dqn[n] += fabsf(w1r) * dpre;
        }
      }
    }
    for (int n = 0; n < N && n < 16; ++n) {
      float v = dqn[n];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (ok && l == 0) dq[r * N + n] = v;
    }
    if (ok && l == 0) db2[r] = g;
  }
}

// ---- QPLEX --------------------------------------------------------------------------------------
__global__ void qplex_mix_fwd_kernel(const float* w_raw, const float* v, const float* q, const float* max_q,
                                     const float* key, const float* ag, const float* ac, float* v_tot,
                                     float* a_tot, float* lam_out, long rows, int N, int K, int weighted,
                                     int minus_one) {
  for (long r = (long)blockIdx.x * TPB + threadIdx.x; r < rows; r += (long)gridDim.x * TPB) {
    float vt = 0.f, at = 0.f;
    for (int i = 0; i < N; ++i) {
      const float w = fabsf(w_raw[r * N + i]) + 1e-10f;
      const float qi = q[r * N + i];
      const float qt = weighted ? w * qi + v[r * N + i] : qi;
      vt += qt;
      if (max_q) {
        const float mi = max_q[r * N + i];
        const float mt = weighted ? w * mi + v[r * N + i] : mi;
        float lam = 0.f;
        for (int k = 0; k < K; ++k) {
          const float kk = fabsf(key[r * K + k]) + 1e-10f;
          lam += kk * sigmoidf_(ag[(r * K + k) * N + i]) * sigmoidf_(ac[(r * K + k) * N + i]);
        }
        if (lam_out) lam_out[r * N + i] = lam;
        at += (qt - mt) * (minus_one ? lam - 1.f : lam);
      }
    }
    if (v_tot) v_tot[r] = vt;
    if (a_tot && max_q) a_tot[r] = at;
  }
}

__global__ void qplex_mix_bwd_kernel(const float* w_raw, const float* q, const float* max_q, const float* key,
                                     const float* ag, const float* ac, const float* g, float* dq, float* dw_raw,
                                     float* dv, float* dkey, float* dag, float* dac, long rows, int N, int K,
                                     int weighted, int minus_one) {
  for (long r = (long)blockIdx.x * TPB + threadIdx.x; r < rows; r += (long)gridDim.x * TPB) {
    const float gr = g[r];
    for (int k = 0; k < K; ++k) dkey[r * K + k] = 0.f;
    for (int i = 0; i < N; ++i) {
      const float wr = w_raw[r * N + i];
      const float w = fabsf(wr) + 1e-10f;
      const float qi = q[r * N + i];
      // v_tot path (adv is detached in the reference: mixer.py:237)
      dq[r * N + i] = weighted ? gr * w : gr;
      dw_raw[r * N + i] = weighted ? gr * qi * (wr > 0.f ? 1.f : (wr < 0.f ? -1.f : 0.f)) : 0.f;
      dv[r * N + i] = weighted ? gr : 0.f;
      // a_tot path: only lambda gets gradient
      const float mi = max_q[r * N + i];
      const float adv = weighted ? (w * qi - w * mi) : (qi - mi);
      const float dlam = gr * adv;
      for (int k = 0; k < K; ++k) {
        const float kr = key[r * K + k];
        const float kk = fabsf(kr) + 1e-10f;
        const float sa = sigmoidf_(ag[(r * K + k) * N + i]);
        const float sc = sigmoidf_(ac[(r * K + k) * N + i]);
        dkey[r * K + k] += dlam * sa * sc * (kr > 0.f ? 1.f : (kr < 0.f ? -1.f : 0.f));
        dag[(r * K + k) * N + i] = dlam * kk * sc * sa * (1.f - sa);
        dac[(r * K + k) * N + i] = dlam * kk * sa * sc * (1.f - sc);
      }
    }
  }
}

// Tiled forms of the two kernels above for the full head set (max_q != NULL): a workgroup owns QR consecutive rows,
// whose key / agents / action head outputs are CONTIGUOUS blocks of QR*K and QR*K*N floats - they are copied
// HBM <-> LDS with 16-byte coalesced accesses and all per-row arithmetic runs out of LDS.  (One thread per row read
// them with a 200-byte lane stride: 0.9 ms forward / 1.75 ms backward at 491 520 rows, 14-28x the HBM time.)
// Same operation order per row as the kernels above => bitwise identical results.
__device__ __forceinline__ void tile_in(float* dst, const float* src, long n_valid, long n_tile) {
  // n_tile floats of LDS; the first n_valid come from src (16-byte aligned, n_valid % 4 == 0 except in the last tile)
  const long n4 = n_valid >> 2;
  for (long e = threadIdx.x; e < n4; e += TPB)
    reinterpret_cast<f32x4*>(dst)[e] = reinterpret_cast<const f32x4*>(src)[e];
  for (long e = (n4 << 2) + threadIdx.x; e < n_tile; e += TPB) dst[e] = e < n_valid ? src[e] : 0.f;
}
__device__ __forceinline__ void tile_out(float* dst, const float* src, long n_valid) {
  const long n4 = n_valid >> 2;
  for (long e = threadIdx.x; e < n4; e += TPB)
    reinterpret_cast<f32x4*>(dst)[e] = reinterpret_cast<const f32x4*>(src)[e];
  for (long e = (n4 << 2) + threadIdx.x; e < n_valid; e += TPB) dst[e] = src[e];
}

__global__ __launch_bounds__(TPB) void qplex_mix_fwd_tiled_kernel(const float* w_raw, const float* v, const float* q,
                                                                  const float* max_q, const float* key, const float* ag,
                                                                  const float* ac, float* v_tot, float* a_tot,
                                                                  float* lam_out, long rows, int N, int K, int weighted,
                                                                  int minus_one, int QR) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int KN = K * N;
  float* ags = sm;                      // [QR][K][N]
  float* acs = ags + QR * KN;           // [QR][K][N]
  float* keys = acs + QR * KN;          // [QR][K]
  float* qts = keys + QR * K;           // [QR][N]  weighted q
  float* ads = qts + QR * N;            // [QR][N]  (qt - mt) * lambda term
  for (long r0 = (long)blockIdx.x * QR; r0 < rows; r0 += (long)gridDim.x * QR) {
    const long nr = rows - r0 < QR ? rows - r0 : QR;
    tile_in(ags, ag + r0 * KN, nr * KN, (long)QR * KN);
    tile_in(acs, ac + r0 * KN, nr * KN, (long)QR * KN);
    tile_in(keys, key + r0 * K, nr * K, (long)QR * K);
    __syncthreads();
    for (int e = threadIdx.x; e < nr * N; e += TPB) {
      const int rl = e / N, i = e - rl * N;
      const long gi = r0 * N + e;
      const float w = fabsf(w_raw[gi]) + 1e-10f;
      const float qi = q[gi];
      const float qt = weighted ? w * qi + v[gi] : qi;
      const float mi = max_q[gi];
      const float mt = weighted ? w * mi + v[gi] : mi;
      float lam = 0.f;
      for (int k = 0; k < K; ++k) {
        const float kk = fabsf(keys[rl * K + k]) + 1e-10f;
        lam += kk * sigmoidf_(ags[(rl * K + k) * N + i]) * sigmoidf_(acs[(rl * K + k) * N + i]);
      }
      if (lam_out) lam_out[gi] = lam;
      qts[e] = qt;
      ads[e] = (qt - mt) * (minus_one ? lam - 1.f : lam);
    }
    __syncthreads();
    for (int rl = threadIdx.x; rl < nr; rl += TPB) {
      float vt = 0.f, at = 0.f;
      for (int i = 0; i < N; ++i) { vt += qts[rl * N + i]; at += ads[rl * N + i]; }
      if (v_tot) v_tot[r0 + rl] = vt;
      if (a_tot) a_tot[r0 + rl] = at;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(TPB) void qplex_mix_bwd_tiled_kernel(const float* w_raw, const float* q, const float* max_q,
                                                                  const float* key, const float* ag, const float* ac,
                                                                  const float* g, float* dq, float* dw_raw, float* dv,
                                                                  float* dkey, float* dag, float* dac, long rows, int N,
                                                                  int K, int weighted, int QR) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int KN = K * N;
  float* ags = sm;                      // [QR][K][N]  in: agents heads, out: their gradient
  float* acs = ags + QR * KN;
  float* keys = acs + QR * KN;          // [QR][K]     in: key heads, out: their gradient
  float* dls = keys + QR * K;           // [QR][N]     dL/dlambda
  for (long r0 = (long)blockIdx.x * QR; r0 < rows; r0 += (long)gridDim.x * QR) {
    const long nr = rows - r0 < QR ? rows - r0 : QR;
    tile_in(ags, ag + r0 * KN, nr * KN, (long)QR * KN);
    tile_in(acs, ac + r0 * KN, nr * KN, (long)QR * KN);
    tile_in(keys, key + r0 * K, nr * K, (long)QR * K);
    for (int e = threadIdx.x; e < nr * N; e += TPB) {
      const int rl = e / N;
      const long gi = r0 * N + e;
      const float gr = g[r0 + rl];
      const float wr = w_raw[gi];
      const float w = fabsf(wr) + 1e-10f;
      const float qi = q[gi];
      dq[gi] = weighted ? gr * w : gr;
      dw_raw[gi] = weighted ? gr * qi * (wr > 0.f ? 1.f : (wr < 0.f ? -1.f : 0.f)) : 0.f;
      dv[gi] = weighted ? gr : 0.f;
      const float mi = max_q[gi];
      const float adv = weighted ? (w * qi - w * mi) : (qi - mi);
      dls[e] = gr * adv;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nr * K; e += TPB) {      // (row, head): agents in order, as in the row-per-thread kernel
      const int rl = e / K;
      const float kr = keys[e];
      const float kk = fabsf(kr) + 1e-10f;
      const float sg = kr > 0.f ? 1.f : (kr < 0.f ? -1.f : 0.f);
      float dk = 0.f;
      for (int i = 0; i < N; ++i) {
        const float dlam = dls[rl * N + i];
        const float sa = sigmoidf_(ags[e * N + i]);
        const float sc = sigmoidf_(acs[e * N + i]);
        dk += dlam * sa * sc * sg;
        ags[e * N + i] = dlam * kk * sc * sa * (1.f - sa);
        acs[e * N + i] = dlam * kk * sa * sc * (1.f - sc);
      }
      keys[e] = dk;
    }
    __syncthreads();
    tile_out(dag + r0 * KN, ags, nr * KN);
    tile_out(dac + r0 * KN, acs, nr * KN);
    tile_out(dkey + r0 * K, keys, nr * K);
    __syncthreads();
  }
}

// rows per workgroup of the tiled kernels (0: shape not covered, use the row-per-thread kernels)
inline int qplex_tile_rows(int N, int K, const void* a, const void* b, const void* c) {
  if (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) != 0) return 0;
  for (int qr = 128; qr >= 32; qr >>= 1)
    if ((size_t)qr * (2 * K * N + K + 2 * N) * 4 <= 64 * 1024) return qr;
  return 0;
}

// ---- deterministic two-stage sums ------------------------------------------------------------------
template <int NV>
__device__ __forceinline__ void block_partials(float (&v)[NV], float* ws) {
  __shared__ float sh[NV][TPB / 64];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float s = wave_sum(v[i]);
    if ((threadIdx.x & 63) == 0) sh[i][threadIdx.x >> 6] = s;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    float s = 0.f;
    for (int w = 0; w < TPB / 64; ++w) s += sh[threadIdx.x][w];
    ws[(long)blockIdx.x * NV + threadIdx.x] = s;
  }
}

__global__ void finish_sums_kernel(const float* ws, int nblocks, int nv, float* out) {
  __shared__ float sh[TPB];
  for (int i = 0; i < nv; ++i) {
    float s = 0.f;
    for (int b = threadIdx.x; b < nblocks; b += TPB) s += ws[(long)b * nv + i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = TPB / 2; o > 0; o >>= 1) {
      if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[i] = sh[0];
    __syncthreads();
  }
}

__global__ void td_loss_kernel(const float* q_tot, const float* q_tgt, const float* r, const float* term,
                               const float* padded, float gamma, float* dq_tot, float* ws, long rows) {
  float acc[2] = {0.f, 0.f};
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < rows; i += (long)gridDim.x * TPB) {
    const float mask = 1.f - padded[i];
    const float target = r[i] + gamma * q_tgt[i] * (1.f - term[i]);
    const float td = target - q_tot[i];
    const float mtd = mask * td;
    acc[0] += mtd * mtd;
    acc[1] += mask;
    dq_tot[i] = -2.f * mask * mtd;
  }
  block_partials<2>(acc, ws);
}

__global__ void qtran_loss_kernel(const float* jq, const float* jq_tgt, const float* v, const float* jq_hat,
                                  const float* qs_opt, const float* qs_nopt, const float* r, const float* term,
                                  const float* padded, float gamma, float lam_opt, float lam_nopt, float* d_jq,
                                  float* d_v, float* d_qs_opt, float* d_qs_nopt, float* ws, long rows) {
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < rows; i += (long)gridDim.x * TPB) {
    const float mask = 1.f - padded[i];
    const float y = r[i] + gamma * jq_tgt[i] * (1.f - term[i]);
    const float td = (jq[i] - y) * mask;
    const float opt = (qs_opt[i] - jq_hat[i] + v[i]) * mask;
    float nraw = qs_nopt[i] - jq[i] + v[i];
    nraw = nraw < 0.f ? nraw : 0.f;
    const float nopt = nraw * mask;
    acc[0] += td * td; acc[1] += opt * opt; acc[2] += nopt * nopt; acc[3] += mask;
    d_jq[i] = 2.f * mask * td;
    const float go = lam_opt * 2.f * mask * opt, gn = lam_nopt * 2.f * mask * nopt;
    d_v[i] = go + gn;
    d_qs_opt[i] = go;
    d_qs_nopt[i] = gn;
  }
  block_partials<4>(acc, ws);
}

inline int loss_blocks(long rows) {
  long b = (rows + TPB - 1) / TPB;
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int marl_q_gather(const float* q, const int* idx, const float* avail, float mask_val, float* out,
                             long rows, int A, void* stream) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(q_gather_kernel, dim3(nblk(rows)), dim3(TPB), 0, (hipStream_t)stream, q, idx, avail, mask_val,
                     out, rows, A);
  MARL_CHECK_LAUNCH();
  return 0;
}

extern "C" int marl_q_masked_max(const float* q, const float* avail, float mask_val, float* out_max, int* out_arg,
                                 long rows, int A, void* stream) {
  if (rows <= 0) return 0;
  // 16-byte aligned operands (64 rows x A floats is a multiple of 16 bytes): the LDS-staged kernel
  const size_t lds = (size_t)4 * 2 * ((DS_ROWS * A + 3) & ~3) * sizeof(float);
  if (((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(avail)) & 15) == 0 && lds <= 64 * 1024 && rows >= 4096) {
    const long tiles = (rows + DS_ROWS - 1) / DS_ROWS;
    long nb = (tiles + 3) / 4;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(q_masked_max_tiled_kernel, dim3((unsigned)nb), dim3(256), lds, (hipStream_t)stream, q, avail, mask_val,
                       out_max, out_arg, rows, A);
    MARL_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(q_masked_max_kernel, dim3(nblk(rows)), dim3(TPB), 0, (hipStream_t)stream, q, avail, mask_val,
                     out_max, out_arg, rows, A);
  MARL_CHECK_LAUNCH();
  return 0;
}

extern "C" int marl_q_double_select(const float* q_sel, const float* q_val, const float* avail, float mask_val,
                                    float* out_val, int* out_arg, long rows, int A, void* stream) {
  if (rows <= 0) return 0;
  const long tiles = (rows + DS_ROWS - 1) / DS_ROWS;
  long nb = (tiles + 3) / 4;
  if (nb > 2048) nb = 2048;
  const size_t lds = (size_t)4 * 3 * ((DS_ROWS * A + 3) & ~3) * sizeof(float);
  if (lds > 64 * 1024) return (int)hipErrorInvalidValue;
  const int vec = ((reinterpret_cast<uintptr_t>(q_sel) | reinterpret_cast<uintptr_t>(q_val) | reinterpret_cast<uintptr_t>(avail)) & 15) == 0;
  hipLaunchKernelGGL(q_double_select_kernel, dim3((unsigned)nb), dim3(256), lds, (hipStream_t)stream, q_sel, q_val, avail,
                     mask_val, out_val, out_arg, rows, A, vec);
  MARL_CHECK_LAUNCH();
  return 0;
}

extern "C" int marl_q_scatter(float* dq, const int* idx1, const float* g1, const int* idx2, const float* g2,
                              long rows, int A, int gdiv, void* stream) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(q_scatter_kernel, dim3(nblk(rows)), dim3(TPB), 0, (hipStream_t)stream, dq, idx1, g1, idx2, g2,
                     rows, A, gdiv < 1 ? 1 : gdiv);
  MARL_CHECK_LAUNCH();
  return 0;
}

extern "C" int marl_vec_add(const float* a, const float* b, float* out, long n, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(vec_add_kernel, dim3(nblk(n, 4096)), dim3(TPB), 0, (hipStream_t)stream, a, b, out, n);
  MARL_CHECK_LAUNCH();
  return 0;
}

extern "C" int marl_agent_sum(const float* in, long ld_in, float* out, long ld_out, long rows, int N, int D, void* stream) {
  if (rows <= 0) return 0;
  if (ld_in < D || ld_out < D) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(agent_sum_kernel, dim3(nblk(rows * D)), dim3(TPB), 0, (hipStream_t)stream, in, ld_in, out, ld_out, rows, N, D);
  MARL_CHECK_LAUNCH();
  return 0;
}

extern "C" int marl_agent_bcast(const float* in, long ld_in, float* out, long ld_out, long rows, int N, int D, int accumulate,
                                void* stream) {
  if (rows <= 0) return 0;
  if (ld_in < D || ld_out < D) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(agent_bcast_kernel, dim3(nblk(rows * N * D)), dim3(TPB), 0, (hipStream_t)stream, in, ld_in, out, ld_out,
                     rows, N, D, accumulate);
  MARL_CHECK_LAUNCH();
  return 0;
}

extern "C" int marl_qmix_mix_fwd(const float* hy, long ldh, const float* b2, const float* w22, const float* b22, const float* q,
                                 float* q_tot, long rows, int N, int E, void* stream) {
  if (rows <= 0) return 0;
  if (!b2 && (!w22 || !b22)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(qmix_mix_fwd_kernel, dim3(nblk((rows + 1) / 2 * 64, 8192)), dim3(TPB), 0, (hipStream_t)stream,
                     hy, ldh, b2, w22, b22, q, q_tot, rows, N, E);
  MARL_CHECK_LAUNCH();
  return 0;
}

extern "C" int marl_qmix_mix_bwd(const float* hy, long ldh, const float* q, const float* dq_tot, const float* w22, float* dhy,
                                 float* db2, float* dq, long rows, int N, int E, void* stream) {
  if (rows <= 0) return 0;
  if (N > 16) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(qmix_mix_bwd_kernel, dim3(nblk((rows + 1) / 2 * 64, 8192)), dim3(TPB), 0, (hipStream_t)stream,
                     hy, ldh, q, dq_tot, w22, dhy, db2, dq, rows, N, E);
  MARL_CHECK_LAUNCH();
  return 0;
}

extern "C" int marl_qplex_mix_fwd(const float* w_raw, const float* v, const float* q, const float* max_q,
                                  const float* key, const float* ag, const float* ac, float* v_tot, float* a_tot,
                                  float* lam_out, long rows, int N, int K, int weighted_head, int minus_one,
                                  void* stream) {
  if (rows <= 0) return 0;
  const int qr = (max_q && key && ag && ac) ? qplex_tile_rows(N, K, key, ag, ac) : 0;
  if (qr) {
    long nb = (rows + qr - 1) / qr; if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(qplex_mix_fwd_tiled_kernel, dim3((unsigned)nb), dim3(TPB), (size_t)qr * (2 * K * N + K + 2 * N) * 4,
                       (hipStream_t)stream, w_raw, v, q, max_q, key, ag, ac, v_tot, a_tot, lam_out, rows, N, K,
                       weighted_head, minus_one, qr);
    MARL_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(qplex_mix_fwd_kernel, dim3(nblk(rows)), dim3(TPB), 0, (hipStream_t)stream, w_raw, v, q, max_q,
                     key, ag, ac, v_tot, a_tot, lam_out, rows, N, K, weighted_head, minus_one);
  MARL_CHECK_LAUNCH();
  return 0;
}

extern "C" int marl_qplex_mix_bwd(const float* w_raw, const float* q, const float* max_q, const float* key,
                                  const float* ag, const float* ac, const float* g, float* dq, float* dw_raw,
                                  float* dv, float* dkey, float* dag, float* dac, long rows, int N, int K,
                                  int weighted_head, int minus_one, void* stream) {
  if (rows <= 0) return 0;
  int qr = qplex_tile_rows(N, K, key, ag, ac);
  if (qr && ((reinterpret_cast<uintptr_t>(dkey) | reinterpret_cast<uintptr_t>(dag) | reinterpret_cast<uintptr_t>(dac)) & 15) != 0) qr = 0;
  if (qr) {
    long nb = (rows + qr - 1) / qr; if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(qplex_mix_bwd_tiled_kernel, dim3((unsigned)nb), dim3(TPB), (size_t)qr * (2 * K * N + K + 2 * N) * 4,
                       (hipStream_t)stream, w_raw, q, max_q, key, ag, ac, g, dq, dw_raw, dv, dkey, dag, dac, rows, N, K,
                       weighted_head, qr);
    MARL_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(qplex_mix_bwd_kernel, dim3(nblk(rows)), dim3(TPB), 0, (hipStream_t)stream, w_raw, q, max_q, key,
                     ag, ac, g, dq, dw_raw, dv, dkey, dag, dac, rows, N, K, weighted_head, minus_one);
  MARL_CHECK_LAUNCH();
  return 0;
}

// get_max_episode_len (algorithm/q_learner.py:49-66) in one launch: per episode the first step with terminated == 1,
// max over episodes of (that step + 1); episodes that never terminate contribute nothing (quirk Q2).  One wave per
// episode at a time scans its row 64 steps at a time (ballot + first set bit) and keeps a running max; ONE atomicMax per
// workgroup (4096 episodes each with its own atomic on the one output word took 48 us).
__global__ __launch_bounds__(TPB) void first_term_kernel(const float* term, long ld, int E, int T, int* out) {
  __shared__ int wmax[TPB / 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long wave_id = ((long)blockIdx.x * TPB + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * TPB) >> 6;
  int best = 0;
  for (long e = wave_id; e < E; e += nwaves) {
    int first = 0;
    for (int t0 = 0; t0 < T && first == 0; t0 += 64) {
      const int t = t0 + lane;
      const bool hit = t < T && term[e * ld + t] == 1.f;
      const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
      if (m) first = t0 + __builtin_ctzll(m) + 1;
    }
    best = first > best ? first : best;
  }
  if (lane == 0) wmax[wv] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    int b = wmax[0];
    for (int i = 1; i < TPB / 64; ++i) b = wmax[i] > b ? wmax[i] : b;
    if (b > 0) atomicMax(out, b);
  }
}

extern "C" int marl_first_terminated_len(const float* term, long ld, int E, int T, int* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(out, 0, sizeof(int), s);
  if (e != hipSuccess) return (int)e;
  if (E <= 0 || T <= 0) return 0;
  long nb = ((long)E * 64 + TPB - 1) / TPB; if (nb > 256) nb = 256;
  hipLaunchKernelGGL(first_term_kernel, dim3((unsigned)nb), dim3(TPB), 0, s, term, ld, E, T, out);
  MARL_CHECK_LAUNCH();
  return 0;
}

// ReplayBuffer.sample (common/replaybuffer.py:54-60) for a device-resident ring: the per-step arrays of the sampled
// episodes in ONE launch (was seven index_select launches + an int32 cast + a clamp + the re-packing of next-step avail).
// Block (b, chunk): a 1024-element chunk of episode idx[b]'s avail slots 1..T; chunk 0 also copies the small arrays.
namespace {
struct GatherArgs {
  const long long* idx;
  const int *u_src, *length_src, *won_src;
  const float *r_src, *term_src, *padded_src, *avail_src;
  int *o_map, *u, *u_act, *length, *won;
  float *r, *term, *padded, *avail_next, *avail_cur;
  int B, T, N, A;
};
__global__ __launch_bounds__(TPB) void replay_gather_kernel(GatherArgs a) {
  const int b = blockIdx.x;
  const long e = (long)a.idx[b];
  const int TNA = a.T * a.N * a.A;
  const float* av = a.avail_src + (e * (a.T + 1) + 1) * (long)a.N * a.A;
  float* ao = a.avail_next + (long)b * TNA;
  for (int i = blockIdx.y * 1024 + threadIdx.x; i < TNA && i < (blockIdx.y + 1) * 1024; i += TPB) ao[i] = av[i];
  if (a.avail_cur) {                 // slots 0..T-1, zeros from the episode's end on (the learners' `avail`: hostutil.DeviceBatch)
    const int NA = a.N * a.A;
    const long live = (long)a.length_src[e] * NA;
    const float* ac = av - NA;
    float* co = a.avail_cur + (long)b * TNA;
    for (int i = blockIdx.y * 1024 + threadIdx.x; i < TNA && i < (blockIdx.y + 1) * 1024; i += TPB) co[i] = i < live ? ac[i] : 0.f;
  }
  if (blockIdx.y != 0) return;
  const int TN = a.T * a.N;
  for (int i = threadIdx.x; i < TN; i += TPB) {
    const int v = a.u_src[e * TN + i];
    a.u[(long)b * TN + i] = v;
    a.u_act[(long)b * TN + i] = v < 0 ? 0 : v;
  }
  for (int i = threadIdx.x; i < a.T; i += TPB) {
    a.r[(long)b * a.T + i] = a.r_src[e * a.T + i];
    a.term[(long)b * a.T + i] = a.term_src[e * a.T + i];
    a.padded[(long)b * a.T + i] = a.padded_src[e * a.T + i];
  }
  if (threadIdx.x == 0) {
    a.o_map[b] = (int)e;
    a.length[b] = a.length_src[e];
    a.won[b] = a.won_src[e];
  }
}
}  // namespace

extern "C" int marl_replay_gather(const long long* idx, int B, int T, int N, int A, const int* u_src, const float* r_src,
                                  const float* term_src, const float* padded_src, const int* length_src,
                                  const int* won_src, const float* avail_src, int* o_map, int* u, int* u_act, float* r,
                                  float* term, float* padded, int* length, int* won, float* avail_next, float* avail_cur,
                                  void* stream) {
  if (B <= 0 || T <= 0) return 0;
  GatherArgs a;
  a.idx = idx; a.u_src = u_src; a.length_src = length_src; a.won_src = won_src;
  a.r_src = r_src; a.term_src = term_src; a.padded_src = padded_src; a.avail_src = avail_src;
  a.o_map = o_map; a.u = u; a.u_act = u_act; a.length = length; a.won = won;
  a.r = r; a.term = term; a.padded = padded; a.avail_next = avail_next; a.avail_cur = avail_cur;
  a.B = B; a.T = T; a.N = N; a.A = A;
  const int chunks = (T * N * A + 1023) / 1024;
  hipLaunchKernelGGL(replay_gather_kernel, dim3((unsigned)B, (unsigned)(chunks > 0 ? chunks : 1)), dim3(TPB), 0, (hipStream_t)stream, a);
  MARL_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t marl_loss_workspace(long rows) { return (size_t)1024 * 4 * sizeof(float); }

extern "C" int marl_td_loss(const float* q_tot, const float* q_tot_tgt, const float* r, const float* term,
                            const float* padded, float gamma, float* dq_tot, float* out2, float* ws, long rows,
                            void* stream) {
  if (rows <= 0) return 0;
  const int nb = loss_blocks(rows);
  hipLaunchKernelGGL(td_loss_kernel, dim3(nb), dim3(TPB), 0, (hipStream_t)stream, q_tot, q_tot_tgt, r, term, padded,
                     gamma, dq_tot, ws, rows);
  MARL_CHECK_LAUNCH();
  hipLaunchKernelGGL(finish_sums_kernel, dim3(1), dim3(TPB), 0, (hipStream_t)stream, (const float*)ws, nb, 2, out2);
  MARL_CHECK_LAUNCH();
  return 0;
}

extern "C" int marl_qtran_loss(const float* jq, const float* jq_tgt, const float* v, const float* jq_hat,
                               const float* qsum_opt, const float* qsum_nopt, const float* r, const float* term,
                               const float* padded, float gamma, float lam_opt, float lam_nopt, float* d_jq,
                               float* d_v, float* d_qsum_opt, float* d_qsum_nopt, float* out4, float* ws, long rows,
                               void* stream) {
  if (rows <= 0) return 0;
  const int nb = loss_blocks(rows);
  hipLaunchKernelGGL(qtran_loss_kernel, dim3(nb), dim3(TPB), 0, (hipStream_t)stream, jq, jq_tgt, v, jq_hat, qsum_opt,
                     qsum_nopt, r, term, padded, gamma, lam_opt, lam_nopt, d_jq, d_v, d_qsum_opt, d_qsum_nopt, ws,
                     rows);
  MARL_CHECK_LAUNCH();
  hipLaunchKernelGGL(finish_sums_kernel, dim3(1), dim3(TPB), 0, (hipStream_t)stream, (const float*)ws, nb, 4, out4);
  MARL_CHECK_LAUNCH();
  return 0;
}
