// Shared device helpers for the gfx950 kernels (wave64, fp32 MFMA 16x16x4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MARL_WAVE 64

// Experiment switches (A/B measurements and variant tests): ONE table per process, set through marl_experiment_set() (optim.hip;
// marl_amd/experiments.py reads the MARL_* environment once at import and forwards it).  No launch path reads the environment.
//   fwd_xs (1): the double-Q unroll reads the eval unroll's input-side gate sums;  fwd_dma (0): LDS-DMA observation tile of the
//   saving unroll;  fwd_w2l (1): six prefetch registers / fc2 fragments in LDS for wide observations;  bwd_pipe_max_rt (4): row
//   tiles up to which the pipelined BPTT runs;  wgrad_tall (1): LDS-staged tall weight-gradient kernel;  wide_res (1) / wide_res32
//   (0): resident-weights forward of the wide-state QMIX mixer (16- / 32-row tiles);  rollout_v1 (0 = by batch size): 1 forces the split
//   whole-rollout kernel of round 5 (rollout_x6_v1.hip: four barriers per lock-step, at most three row tiles per workgroup), 2 the one
//   of round 6 (rollout_x6.hip: three barriers, up to five tiles);  unroll_r6 (1): non-saving split unrolls of large batches on
//   agent_x6p.hip (the round-6 decomposition: five row tiles per workgroup, two barriers per step)
struct MarlSwitches {
  int fwd_xs, fwd_dma, fwd_w2l, bwd_pipe_max_rt, wgrad_tall, wide_res, wide_res32, rollout_v1, unroll_r6;
};
extern "C" const MarlSwitches* marl_switches(void);      // optim.hip

// LDS row pitches of the activation tiles, in floats beyond the tile width - the DEFAULT (+8) for files that do not choose their
// own: rollout_fused.hip keeps it (0-1 % ahead there), agent.hip overrides it with +4 (agent.hip:16-23: +8 costs the wide MMM2
// tiles 6 % through the LDS it takes from the row-tile count).  Why +8: a wave reads an MFMA operand fragment with one
// ds_read_b128 per lane (row m, columns 4q..4q+3 of a 16-chunk); gfx950 serves that instruction in four groups of 16 lanes over
// 64 banks, and a pitch of 8 (mod 16) floats spreads every group over all banks (the usual "+4" is a 2-way conflict on every
// such read; tools/lds_pitch.py), while the accumulator-layout accesses (row 4q+i, column m: ds_read/write_b32) become 2-way -
// free for the writes, and the reads are few.  It halves the conflict share the counters show (profiles/archive/r03_pmc_pitch8.json)
// and is time-neutral on 2s3z-sized tiles (profiles/archive/r03_ab_variants.txt).
#ifndef MARL_PAD_H
#define MARL_PAD_H 8
#endif
#ifndef MARL_PAD_K
#define MARL_PAD_K 8
#endif
#ifndef MARL_PAD_G
#define MARL_PAD_G 8
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
// 16-byte integer LDS reads use THIS type, not HIP's int4: int4 is a union-based struct, its loads carry "may alias anything"
// type information, and hipcc then waits vmcnt(0) in front of every such LDS read while an LDS-DMA (global_load_lds) may be
// pending - draining the DMA (and the wave's stores) right there.  Scalar int / ext-vector loads do not get that wait.
typedef int i32x4 __attribute__((ext_vector_type(4)));

// D = A(16x4) * B(4x16) + C, exact f32 (v_mfma_f32_16x16x4_f32).
// lane l: A[row l&15][k l>>4], B[k l>>4][col l&15]; D reg r: D[row 4*(l>>4)+r][col l&15].
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// K-permutation trick: a k-chunk of 16 is consumed as 4 MFMA steps; at step i lane (q=l>>4)
// supplies element k0+4q+i for BOTH operands, so each lane's 4 operands are 16 contiguous bytes.
__device__ __forceinline__ f32x4 mfma16x4(const f32x4& a, const f32x4& b, f32x4 c) {
  c = mfma16(a[0], b[0], c);
  c = mfma16(a[1], b[1], c);
  c = mfma16(a[2], b[2], c);
  c = mfma16(a[3], b[3], c);
  return c;
}

// Interleaved forms for INDEPENDENT accumulators.  v_mfma_f32_16x16x4_f32 issues every 32 cycles but its result can feed the
// next MFMA only after 40: four back-to-back MFMAs on ONE accumulator (mfma16x4 above, which is how hipcc emits it - it keeps
// the source order) run at 80 % of the pipe rate.  Round-robin over two to four accumulators keeps the pipe full; the order
// of the four products inside each accumulator is unchanged, so results are bit-identical to the plain form.
#ifdef MARL_NO_INTERLEAVE      // A/B builds (tools/build_variant.sh): the plain order
__device__ __forceinline__ void mfma16x4_il2(const f32x4& a0, const f32x4& b0, f32x4& c0, const f32x4& a1, const f32x4& b1, f32x4& c1) {
  c0 = mfma16x4(a0, b0, c0); c1 = mfma16x4(a1, b1, c1);
}
__device__ __forceinline__ void mfma16x4_il3(const f32x4& a0, const f32x4& b0, f32x4& c0, const f32x4& a1, const f32x4& b1, f32x4& c1,
                                             const f32x4& a2, const f32x4& b2, f32x4& c2) {
  c0 = mfma16x4(a0, b0, c0); c1 = mfma16x4(a1, b1, c1); c2 = mfma16x4(a2, b2, c2);
}
__device__ __forceinline__ void mfma16x4_il4(const f32x4& a0, const f32x4& b0, f32x4& c0, const f32x4& a1, const f32x4& b1, f32x4& c1,
                                             const f32x4& a2, const f32x4& b2, f32x4& c2, const f32x4& a3, const f32x4& b3, f32x4& c3) {
  c0 = mfma16x4(a0, b0, c0); c1 = mfma16x4(a1, b1, c1); c2 = mfma16x4(a2, b2, c2); c3 = mfma16x4(a3, b3, c3);
}
#else
__device__ __forceinline__ void mfma16x4_il2(const f32x4& a0, const f32x4& b0, f32x4& c0, const f32x4& a1, const f32x4& b1, f32x4& c1) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { c0 = mfma16(a0[i], b0[i], c0); c1 = mfma16(a1[i], b1[i], c1); }
}
__device__ __forceinline__ void mfma16x4_il3(const f32x4& a0, const f32x4& b0, f32x4& c0, const f32x4& a1, const f32x4& b1, f32x4& c1,
                                             const f32x4& a2, const f32x4& b2, f32x4& c2) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { c0 = mfma16(a0[i], b0[i], c0); c1 = mfma16(a1[i], b1[i], c1); c2 = mfma16(a2[i], b2[i], c2); }
}
__device__ __forceinline__ void mfma16x4_il4(const f32x4& a0, const f32x4& b0, f32x4& c0, const f32x4& a1, const f32x4& b1, f32x4& c1,
                                             const f32x4& a2, const f32x4& b2, f32x4& c2, const f32x4& a3, const f32x4& b3, f32x4& c3) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    c0 = mfma16(a0[i], b0[i], c0); c1 = mfma16(a1[i], b1[i], c1); c2 = mfma16(a2[i], b2[i], c2); c3 = mfma16(a3[i], b3[i], c3);
  }
}

#endif

// bf16 operands, fp32 accumulate (v_mfma_f32_16x16x16_bf16): with the K-permutation above one instruction replaces
// the four fp32 MFMAs of a 16-chunk.  Opt-in for the MIXER GEMMs only (BASELINE config 5, "bf16 mixer with MFMA").
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16x4_bf16(const f32x4& a, const f32x4& b, f32x4 c) {
  const s16x4 ab = __builtin_bit_cast(s16x4, __builtin_convertvector(a, bf16x4_t));      // round to nearest even
  const s16x4 bb = __builtin_bit_cast(s16x4, __builtin_convertvector(b, bf16x4_t));
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab, bb, c, 0, 0, 0);
}
template <bool BF>
__device__ __forceinline__ f32x4 mm16x4(const f32x4& a, const f32x4& b, f32x4 c) {
  if (BF) return mfma16x4_bf16(a, b, c);
  return mfma16x4(a, b, c);
}

// v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division sequence: abs error ~1e-7 on (0,1)
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
  // 1 - 2/(e^{2x}+1): saturates cleanly, abs error ~1e-7
  float e = __expf(2.0f * x);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// ---- GRU gate math on PRE-SCALED accumulators (every GRU forward kernel: unroll, pipelined unroll, rollout).
// fp32 MFMAs and vector instructions share the SIMD's issue, so a vector instruction saved is matrix time
// gained.  sigmoid(x) = 1 / (1 + 2^(-x log2 e)) and tanh(y) = 1 - 2 / (2^(2 y log2 e) + 1): the kernels multiply the r / z rows of
// W_ih, W_hh and their biases by -log2(e) and the candidate rows by 2 log2(e) ONCE, when the fragments are loaded into
// registers, so the accumulators arrive as the exponents and the per-element multiplies (5 of 17 vector instructions per
// element) are gone.  Scaling a weight rounds it once more (relative 6e-8, the size of the products' own rounding).
#define MARL_NLOG2E (-1.4426950408889634f)
#define MARL_2LOG2E (2.8853900817779268f)
#define MARL_INV_2LOG2E (0.34657359027997264f)
// Measured (same box, alternating, profiles/archive/r03_prescale_ab.txt): the rollout kernel gains 1.5 % from this; of the learner's unroll
// kernels the two-action-tile instantiations (AC = 2: MMM2, 18 actions) gain 10 % and the one-tile ones (2s3z, 3s5z) LOSE 2 % (the
// saving unroll also has to un-scale the plane it stores for BPTT) - those use gru_point_plain().  GRU_PRE: agent.hip decides by AC.
// -DMARL_NO_PRESCALE: plain form everywhere (A/B).
__device__ __forceinline__ void gru_point_plain(float ar, float az, float ain, float ahn, float hp, float& r, float& z, float& n, float& h) {
  r = sigmoidf_(ar); z = sigmoidf_(az);
  n = tanhf_(ain + r * ahn);
  h = (1.f - z) * n + z * hp;
}
#ifdef MARL_NO_PRESCALE
#undef MARL_INV_2LOG2E
#define MARL_INV_2LOG2E (1.0f)
__device__ __forceinline__ void gru_prescale(f32x4 (&)[3][4], f32x4 (&)[3][4], float&, float&, float&, float&) {}
__device__ __forceinline__ void gru_point(float ar, float az, float ain, float ahn, float hp, float& r, float& z, float& n, float& h) {
  gru_point_plain(ar, az, ain, ahn, hp, r, z, n, h);
}
#else
__device__ __forceinline__ void gru_prescale(f32x4 (&wih)[3][4], f32x4 (&whh)[3][4], float& b_r, float& b_z, float& b_in, float& b_hn) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    wih[0][c] *= MARL_NLOG2E; whh[0][c] *= MARL_NLOG2E;
    wih[1][c] *= MARL_NLOG2E; whh[1][c] *= MARL_NLOG2E;
    wih[2][c] *= MARL_2LOG2E; whh[2][c] *= MARL_2LOG2E;
  }
  b_r *= MARL_NLOG2E; b_z *= MARL_NLOG2E; b_in *= MARL_2LOG2E; b_hn *= MARL_2LOG2E;
}
// ar, az: -log2(e) x (gate pre-activation);  ain, ahn: 2 log2(e) x (input-side / hidden-side part of the candidate's)
__device__ __forceinline__ void gru_point(float ar, float az, float ain, float ahn, float hp, float& r, float& z, float& n, float& h) {
  r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ar));
  z = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(az));
  const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(r, ahn, ain));
  n = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
  h = __builtin_fmaf(z, hp - n, n);                       // (1 - z) n + z h_prev
}
#endif

// s = p[w0 * stride] + p[(w0 + step) * stride] + .. (w < n), added in that order, eight reads in flight.  The slab reduce kernels
// sum hundreds of slabs per element: written as `for (..) s += p[..]` every read waits for the previous one (the add keeps the
// order), and the kernel runs at one HBM latency per slab - 26 us for a 7 MB reduce (round 5, qtran_reduce_kernel: 27 -> 7 us).
__device__ __forceinline__ float slab_acc(float s, const float* p, long stride, int w0, int step, int n) {
  int w = w0;
  for (; w + 7 * step < n; w += 8 * step) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(long)(w + u * step) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; w < n; w += step) s += p[(long)w * stride];
  return s;
}
__device__ __forceinline__ float slab_sum(const float* p, long stride, int w0, int step, int n) { return slab_acc(0.f, p, stride, w0, step, n); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Virtual row-major matrix [M, K] = [dense0 | dense1 | nhot one-hot blocks | agent-id block].
// Lets GEMMs consume concatenations ([h|onehot(u)], [s|enc], [s|joint one-hot], [obs|u|id])
// without materialising them.
// exact unsigned 32-bit division by an invariant divisor (Granlund-Montgomery): the row remaps and the
// agent-id column need row / d and row % d per row; a hardware-less 64-bit '/' costs hundreds of cycles
struct FastDiv { unsigned d, m, s; };
__host__ inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f; f.d = d ? d : 1; f.m = 0; f.s = 0;
  if (f.d == 1) return f;
  unsigned s = 0; while ((1ull << s) < f.d) ++s;
  f.s = s;
  f.m = (unsigned)(((1ull << 32) * ((1ull << s) - f.d)) / f.d + 1ull);
  return f;
}
__device__ __forceinline__ unsigned fastdiv(unsigned n, const FastDiv& f) {
  if (f.d == 1) return n;
  const unsigned t = __umulhi(f.m, n);
  return (t + ((n - t) >> 1)) >> (f.s - 1);
}

struct ConcatSrc {
  const float* p0; long ld0; int k0;        // dense segment 0 (k0 columns)
  const float* p1; long ld1; int k1;        // dense segment 1
  const int* idx; int nhot; int hot_w;      // one-hot blocks: column j*hot_w + idx[row*nhot+j] is 1
  int nid;                                  // identity block: column (row % nid) is 1
  const float* m0; long ldm0;               // optional gate on segment 0: value * (m0 > 0)
  long rpe0, bs0, off0;                     // row remap of p0/m0: (row/rpe)*bs + row%rpe + off
  long rpei, bsi, offi;                     // row remap of idx; (row%rpe)+off < 0 reads "none"
  FastDiv fd0, fdi, fdn;                    // dividers for rpe0, rpei, nid (rows < 2^31)
  const int* emap0;                         // optional episode map of the p0 remap
};

__device__ __forceinline__ long remap_row(long row, long rpe, long bs, long off, bool& valid) {
  valid = true;
  if (rpe == 0) return row;
  const long e = row / rpe, w = row - e * rpe + off;
  valid = w >= 0;
  return e * bs + w;
}

__device__ __forceinline__ float concat_elem(const ConcatSrc& s, long row, int k) {
  if (k < s.k0) {
    bool ok;
    long r0;
    if (s.emap0 && s.rpe0) {
      const long e = row / s.rpe0, w = row - e * s.rpe0 + s.off0;
      ok = w >= 0;
      r0 = (long)s.emap0[e] * s.bs0 + w;
    } else {
      r0 = remap_row(row, s.rpe0, s.bs0, s.off0, ok);
    }
    if (!ok) return 0.f;
    float v = s.p0[r0 * s.ld0 + k];
    if (s.m0) v = s.m0[r0 * s.ldm0 + k] > 0.f ? v : 0.f;
    return v;
  }
  k -= s.k0;
  if (k < s.k1) return s.p1[row * s.ld1 + k];
  k -= s.k1;
  int hw = s.nhot * s.hot_w;
  if (k < hw) {
    int j = k / s.hot_w;
    bool ok;
    const long ri = remap_row(row, s.rpei, s.bsi, s.offi, ok);
    if (!ok) return 0.f;
    int a = s.idx[ri * s.nhot + j];
    return (a == k - j * s.hot_w) ? 1.f : 0.f;
  }
  k -= hw;
  if (k < s.nid) return ((int)(row % s.nid) == k) ? 1.f : 0.f;
  return 0.f;
}

// Two-stage form for inner loops: the row-dependent part (remaps) once per row, then cheap per-column reads.
struct ConcatRow { long r0, ri, row; int nidx; bool ok0, oki; };

__device__ __forceinline__ long remap_row_fast(unsigned row, long rpe, long bs, long off, const FastDiv& f, bool& valid) {
  valid = true;
  if (rpe == 0) return row;
  const unsigned e = fastdiv(row, f);
  const long w = (long)(row - e * (unsigned)rpe) + off;
  valid = w >= 0;
  return (long)e * bs + w;
}

__device__ __forceinline__ ConcatRow concat_row(const ConcatSrc& s, long row) {
  ConcatRow c;
  c.row = row;
  if (s.emap0 && s.rpe0) {
    const unsigned e = fastdiv((unsigned)row, s.fd0);
    const long w = (long)((unsigned)row - e * (unsigned)s.rpe0) + s.off0;
    c.ok0 = w >= 0;
    c.r0 = (long)s.emap0[e] * s.bs0 + w;
  } else {
    c.r0 = remap_row_fast((unsigned)row, s.rpe0, s.bs0, s.off0, s.fd0, c.ok0);
  }
  c.ri = remap_row_fast((unsigned)row, s.rpei, s.bsi, s.offi, s.fdi, c.oki);
  c.nidx = s.nid ? (int)((unsigned)row - fastdiv((unsigned)row, s.fdn) * (unsigned)s.nid) : 0;
  return c;
}

__device__ __forceinline__ float concat_at(const ConcatSrc& s, const ConcatRow& c, int k) {
  if (k < s.k0) {
    if (!c.ok0) return 0.f;
    float v = s.p0[c.r0 * s.ld0 + k];
    if (s.m0) v = s.m0[c.r0 * s.ldm0 + k] > 0.f ? v : 0.f;
    return v;
  }
  k -= s.k0;
  if (k < s.k1) return s.p1[c.row * s.ld1 + k];
  k -= s.k1;
  const int hw = s.nhot * s.hot_w;
  if (k < hw) {
    if (!c.oki) return 0.f;
    const int j = k / s.hot_w;
    return (s.idx[c.ri * s.nhot + j] == k - j * s.hot_w) ? 1.f : 0.f;
  }
  k -= hw;
  if (k < s.nid) return (c.nidx == k) ? 1.f : 0.f;
  return 0.f;
}

__host__ __device__ inline int concat_width(const ConcatSrc& s) { return s.k0 + s.k1 + s.nhot * s.hot_w + s.nid; }

// ---- diagnostic build only (make stamps -> libmarl_hip_stamps.so): per-segment s_memtime sums of every wave of
// workgroup 0, written to a buffer of their own (tools/stamps.py).  No stamp executes in the product library.
#ifdef MARL_STAMPS
static __device__ unsigned long long* marl_stamp_buf;   // one per translation unit
#define ST_DEFINE_SETTER(name) extern "C" int name(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(marl_stamp_buf), &p, sizeof(p)); }
#define ST_NOW(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define ST_DECL(n) unsigned long long st_sum_[n]; for (int i_ = 0; i_ < (n); ++i_) st_sum_[i_] = 0ull; unsigned long long st_prev_; ST_NOW(st_prev_)
#define ST_MARK(i) do { unsigned long long n_; ST_NOW(n_); st_sum_[i] += n_ - st_prev_; st_prev_ = n_; } while (0)
#define ST_DUMP_AT(n, col) do { if (marl_stamp_buf && blockIdx.x == 0 && (threadIdx.x & 63) == 0) for (int i_ = 0; i_ < (n); ++i_) marl_stamp_buf[(threadIdx.x >> 6) * 16 + (col) + i_] = st_sum_[i_]; } while (0)
#define ST_DUMP(n) ST_DUMP_AT(n, 0)
#else
#define ST_DEFINE_SETTER(name)
#define ST_DECL(n)
#define ST_MARK(i)
#define ST_DUMP(n)
#define ST_DUMP_AT(n, col)
#endif

#define MARL_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
