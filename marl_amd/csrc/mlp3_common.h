// Shared pieces of the fused 64-wide head kernels (mlp3_fused.hip: fp32 MFMA; mlp3_x6.hip: bf16 x 6-product split):
// argument block, the virtual-concat x tile (row lookup, loads, selects), weight-fragment staging, workgroup -> (stripe, head)
// map, the slab reduction and the host-side shape helpers.  Included inside each file (anonymous namespace: one copy per object).
#pragma once
#include "common.h"
#include "../../include/marl_hip.h"

namespace {


constexpr int HD = 64;            // hidden width
constexpr int RS = 68;            // row stride (floats) of the [feature][64 rows] stage tiles
constexpr int FNW = 8;            // forward: waves per workgroup
constexpr int BNW = 4;            // backward: waves per workgroup (= 16-row tiles per iteration)

// kept activations stream through HBM once each way (A/B: -DMARL_KEEP_TEMPORAL uses ordinary accesses)
#ifdef MARL_KEEP_TEMPORAL
#define KEEP_ST(v, p) (*(p) = (v))
#define KEEP_LD(p) (*(p))
#else
#define KEEP_ST(v, p) __builtin_nontemporal_store((v), (p))
#define KEEP_LD(p) __builtin_nontemporal_load(p)
#endif
#define WG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

struct Mlp3Args {
  ConcatSrc x;
  const float *W1, *b1, *W2, *b2, *W3, *b3;
  long gs_w1, gs_b1, gs_w2, gs_b2, gs_w3, gs_b3;     // element strides between heads
  float* Y; long ldy, gs_y;                          // forward: outputs; backward: dY (read only)
  float* ws;                                         // backward: [slab][group][slab_floats]
  float* hs;                                         // kept hidden activations [group][16-row tile][plane][64 lanes] f32x4, or NULL
  long M;
  int K1, N3, groups, nst, CF;                       // nst stripes per group; CF leading 16-byte-loadable chunks
  int kpad, KV;                                      // virtual K axis: kpad zero columns after dense0 (so that k0 + kpad is a multiple of 4), KV = K1 + kpad
  int n3t, yvec;                                     // WIDE kernels: output tiles of 16 (N3 up to 160); Y rows take 16-byte accesses
};
// The kept-activation backward runs TWO workgroups per CU up to this chunk count (256 registers per wave, <= 80 KB of LDS each:
// more than 8 chunks of x^T go through the stage 6 at a time): four barriers per 64 rows with one wave per SIMD left the
// matrix pipe idle 43 % of the time; two independent workgroups fill each other's exchange phases (key / agents / action head
// backward 2.06 / 2.15 / 2.63 -> 1.71 / 1.74 / 2.21 ms; -DMLP3_BWD2_KC=0 is the one-workgroup form).
#ifndef MLP3_BWD2_KC
#define MLP3_BWD2_KC 11
#endif
constexpr int NTW = 10;           // output tiles of the WIDE variants (hypernet heads of QMIX with two_hyper_layers: N*E = 160 columns)

// virtual column (K axis of the kernels) -> column of W1 / dW1, or -1 for a pad column.  A lane's four consecutive columns must
// come from ONE segment of the concat; a dense0 width that is not a multiple of 4 (MMM2: 322 state columns) is padded in the
// kernels' own K axis instead of asking the caller for another weight layout.
__host__ __device__ inline int vcol(int k, int k0, int kpad) { return k < k0 ? k : (k < k0 + kpad ? -1 : k - kpad); }

__host__ __device__ inline long mlp3_slab_floats(int K1, int N3) {
  const long n3p = N3 <= 16 ? 16 : (N3 + 15) / 16 * 16;
  return (long)HD * (K1 + 1) + (long)HD * (HD + 1) + n3p * (HD + 1);
}

// workgroup -> (stripe, head): the `groups` heads of one stripe of rows run on the SAME XCD (blockIdx % 8) next to
// each other in time, so the stripe's x rows are fetched from HBM once and hit that XCD's L2 for the other heads
__device__ __forceinline__ bool wg_map(int groups, int nst, int& stripe, int& g) {
  const int L = blockIdx.x, xcd = L & 7, r = L >> 3;
  g = r % groups;
  stripe = (r / groups) * 8 + xcd;
  return stripe < nst;
}

// Table of the generic chunks (those not wholly inside the 16-byte aligned part of dense0), two int4 per (chunk, q = lane / 16) -
// a lane's columns 16c + 4q + i do not depend on its row, and the 16 lanes of a q read one address (LDS broadcast):
//   [0] byte offsets of the lane's four elements from the base of its group's source row
//   [1] {cmp0 | cmp1 << 16, cmp2 | cmp3 << 16, kind, -}   kind: 0 zero, 1 dense0, 2 one-hot index, 3 dense1
// Segment starts are multiples of 4 on the kernels' K axis (dense0 padded by `kpad` zero columns, dense1 checked on the host), so
// the four elements k = 16c+4q+0..3 of a lane come from ONE source: a lane selects one row base per chunk and the element loads
// are base + offset - no per-element branching.
__device__ __forceinline__ void build_tab(int* tab, const ConcatSrc& x, int KV, int kpad, int CF, int KC, int nthreads) {
  for (int e = threadIdx.x; e < (KC - CF) * 4; e += nthreads) {
    const int gc = e >> 2, qq = e & 3;
    const int kb = 16 * (CF + gc) + 4 * qq;
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

// weight fragments (A operands), fragment-major: [(t * KCn + c) * 64 + lane] f32x4 = W[16t + m][16c + 4q + 0..3]
// (all of a thread's loads are issued - unconditionally, indices clamped - before its first LDS store: a load - mask - store
// loop serialises one memory round trip per item, which at the small shards is a visible part of the launch)
template <int NITEMS, int NTHR>
__device__ __forceinline__ void stage_w(float* dst, const float* W, int ldw, int rows_valid, int K, int KCn, int k0 = 1 << 30, int kpad = 0) {
  constexpr int NIT = (NITEMS + NTHR - 1) / NTHR;
  f32x4 v[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    int e = threadIdx.x + NTHR * it; if (e > NITEMS - 1) e = NITEMS - 1;
    const int l = e & 63, tc = e >> 6, t = tc / KCn, c = tc - t * KCn;
    int n = 16 * t + (l & 15); if (n > rows_valid - 1) n = rows_valid - 1;
    const int kq = 16 * c + 4 * (l >> 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int kr = vcol(kq + i, k0, kpad);
      kr = kr < 0 ? 0 : (kr < K ? kr : K - 1);
      v[it][i] = W[(long)n * ldw + kr];
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e = threadIdx.x + NTHR * it;
    const int l = e & 63, tc = e >> 6, t = tc / KCn, c = tc - t * KCn;
    const int n = 16 * t + (l & 15), kq = 16 * c + 4 * (l >> 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kr = vcol(kq + i, k0, kpad);
      v[it][i] = (n < rows_valid && kr >= 0 && kr < K) ? v[it][i] : 0.f;
    }
    if (e < NITEMS) *reinterpret_cast<f32x4*>(dst + (long)e * 4) = v[it];
  }
}
// transposed fragments: [(t * 4 + c) * 64 + lane] f32x4 = W[16c + 4q + i][16t + m]   (A operand of dX^T = W^T dY^T)
template <int NTHR>
__device__ __forceinline__ void stage_wT(float* dst, const float* W, int ldw) {
  constexpr int NIT = (16 * 64 + NTHR - 1) / NTHR;
  f32x4 v[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    int e = threadIdx.x + NTHR * it; if (e > 16 * 64 - 1) e = 16 * 64 - 1;
    const int l = e & 63, tc = e >> 6, t = tc >> 2, c = tc & 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[it][i] = W[(long)(16 * c + 4 * (l >> 4) + i) * ldw + 16 * t + (l & 15)];
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e = threadIdx.x + NTHR * it;
    if (e < 16 * 64) *reinterpret_cast<f32x4*>(dst + (long)e * 4) = v[it];
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
// CFT >= 0: the number of leading 16-byte-loadable chunks is a compile-time constant (QPLEX on 2s3z: 7 = a 120-wide state):
// with a runtime CF every chunk of the unrolled loops carried a branch, the table reads of the generic path and their waits -
// ~250 vector instructions per 16-row tile around 208 MFMAs (PMC: 1.04 non-MFMA vector instructions per MFMA in the forward)
template <int KC, int CFT>
__device__ __forceinline__ void x_issue(f32x4 (&xv)[KC], const ConcatSrc& x, const XRow& r, const int* tab, int CFr, int lane) {
  const int CF = CFT >= 0 ? CFT : CFr;
  const int q = lane >> 4;
  const char* d0 = reinterpret_cast<const char*>(x.p0 + r.r0c * x.ld0);
  const char* d1 = x.p1 ? reinterpret_cast<const char*>(x.p1 + r.rowc * x.ld1) : d0;
  const char* di = x.idx ? reinterpret_cast<const char*>(x.idx + r.ric * x.nhot) : d0;
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    if (c < CF) {
      xv[c] = *reinterpret_cast<const f32x4*>(d0 + 64 * c + 16 * q);
    } else {
      const int* t = tab + ((c - CF) * 4 + q) * 8;
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
template <int KC, int CFT>
__device__ __forceinline__ void x_finish(f32x4 (&xv)[KC], const XRow& r, const int* tab, int CFr, int lane) {
  const int CF = CFT >= 0 ? CFT : CFr;
  const bool ok0 = (r.flags & 2) != 0, oki = (r.flags & 4) != 0;
  // rows that read as zero (remap before the first slot) are rare: one wave-uniform test instead of 4 selects per chunk
  const bool any_bad0 = __builtin_amdgcn_ballot_w64(!ok0) != 0;
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    if (c < CF) {
      if (any_bad0 && !ok0) xv[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    } else {
      const int4 t1 = *reinterpret_cast<const int4*>(tab + ((c - CF) * 4 + (lane >> 4)) * 8 + 4);
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

struct Mlp3RedArgs {
  const float* ws;
  float *dW1, *db1, *dW2, *db2, *dW3, *db3;
  long gs_w1, gs_b1, gs_w2, gs_b2, gs_w3, gs_b3;
  int K1, N3, groups, nst;
};

// grads += sum over the stripes' slabs, in stripe order (deterministic).  Heads that SHARE their first layer(s) (element stride 0
// between groups: one wide head evaluated as column blocks of 160 outputs) have those gradients summed over the groups by
// the thread of group 0, in group order.
__global__ __launch_bounds__(256) void mlp3_reduce_kernel(Mlp3RedArgs a) {
  const long SZ = mlp3_slab_floats(a.K1, a.N3);
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= SZ * a.groups) return;
  const int g = (int)(e / SZ);
  long r = e - (long)g * SZ;
  const int K1x = a.K1 + 1;
  const bool first = r < (long)HD * K1x, second = !first && r < (long)HD * K1x + HD * (HD + 1);
  const bool shared = a.groups > 1 && ((first && a.gs_w1 == 0 && a.gs_b1 == 0) || (second && a.gs_w2 == 0 && a.gs_b2 == 0));
  if (shared && g != 0) return;
  float s = 0.f;
  for (int gg = g; gg < (shared ? a.groups : g + 1); ++gg)      // (same order of additions; eight reads in flight: common.h)
    s = slab_acc(s, a.ws + (long)gg * SZ + r, (long)a.groups * SZ, 0, 1, a.nst);
  if (first) {
    const int n = (int)(r / K1x), k = (int)(r - (long)n * K1x);
    if (k < a.K1) a.dW1[g * a.gs_w1 + (long)n * a.K1 + k] += s;
    else a.db1[g * a.gs_b1 + n] += s;
    return;
  }
  r -= (long)HD * K1x;
  if (second) {
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

#define MLP3_CF7_(T3, CF, ...) T3, 7, ##__VA_ARGS__
#define MLP3_CF7(...) MLP3_CF7_(__VA_ARGS__)
// instantiated chunk counts: 4, 8, 11 (QPLEX [state 120 | one-hot 55] exactly), 12; 16, 24, 32 (K1 up to 512: the backward of
// these exists only for kept activations)
inline int kc_bucket(int KV) {
  const int kc = (KV + 15) / 16;
  return kc == 11 ? 11 : kc <= 12 ? (kc + 3) / 4 * 4 : kc <= 16 ? 16 : (kc + 7) / 8 * 8;
}
inline int kpad_of(const marl_src_t* x) { return (4 - x->k0 % 4) % 4; }
inline size_t fwd_lds(int KC, int CF, bool wide) {
  return (size_t)(4 * KC * 256 + 16 * 256 + (wide ? NTW : 1) * 4 * 256 + (wide ? 16 * NTW : 0)) * 4 + (size_t)(KC - CF) * 32 * 4;
}
inline size_t bwd_lds(int KC, int CF, bool kept, bool wide, bool three) {
  const bool two = kept && !wide && KC <= MLP3_BWD2_KC;
  const int KH = KC > 24 ? 16 : (two && KC > 8) ? 6 : KC;
  const int SF2 = (three ? 192 : 64) + 16 * (wide ? NTW : 1);
  const int SF = (16 * KH + HD) > SF2 ? (16 * KH + HD) : SF2;
  return (size_t)((kept ? 0 : 4 * KC * 256 + 16 * 256) + (three || !wide ? 16 * 256 : 0) + (wide ? NTW * 4 * 256 : 16 * 64) + SF * RS) * 4 +
         (size_t)(KC - CF) * 32 * 4;
}
inline int lead_chunks(const marl_src_t* x) {
  const bool al = x->p0 && (x->ld0 % 4 == 0) && aligned16(x->p0);
  return al ? x->k0 / 16 : 0;
}
// stripes per group: a multiple of 8 (one per XCD and round, see wg_map) with stripes * groups <= 256 workgroups, so
// that every XCD gets the same number of workgroups and all of them are resident at once (one per CU): 26 stripes
// x 10 heads = 260 workgroups ran as two rounds and took twice as long as 24 x 10
inline int stripes(long units, int groups, int wgs = 256) {
  long n = wgs / groups / 8 * 8;
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
  a.kpad = kpad_of(x); a.KV = K1 + a.kpad;
  a.CF = lead_chunks(x);
  a.n3t = (N3 + 15) / 16; a.yvec = 0;
  return true;
}

}  // namespace
