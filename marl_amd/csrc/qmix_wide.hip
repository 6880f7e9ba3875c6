// Fused QMIX mixer for WIDE states (reference network/mixer.py:57-80 at S = 322, N = 10: MMM2, BASELINE config 5).
//
// qmix_fused.hip keeps the hypernet weights in registers, which stops at S <= 128 / 256 output columns; here the
// concatenated hypernet [ w1 (N*E) | b1 (E) | w2 (E) | h = hyper_b2.0 (E) ] x S is 416 x 322 = 536 KB.  It is packed
// once per call into MFMA-fragment order (1 KB per (column tile, k-chunk), L2 resident) and STREAMED through LDS: a
// workgroup owns blocks of 128 (episode, step) rows, every wave one 16-row tile x ALL 26 column tiles (104 accumulator
// registers), so every weight byte read from L2 feeds 128 rows and the mixing arithmetic of a row is wave local.  The
// 416-wide hypernet output never reaches HBM in the forward pass.
//   forward : q_tot = sum_e elu(sum_n q_n |w1[n,e]| + b1_e) |w2_e| + (relu(h) . w_b2 + b_b2)
//   backward: recomputes the tile, forms d(hypernet output) in accumulator layout and writes it (rows x C) for the
//             weight-gradient GEMM below; dq and the hyper_b2.2 gradients come out of the same pass.
//   wgrad   : dW[C][S] += dhy^T s over all rows - a tall-skinny GEMM: 2 column groups x 128 row slabs, operands
//             staged row-major in LDS (double buffered), MFMA operands are plain 32-bit LDS reads (k = row), each wave
//             keeps (all column tiles of the group) x (its k tiles) accumulators in registers; slabs + fixed-order
//             reduce scatter into the four weight / bias gradients.
// BF = true: the hypernet GEMM takes bf16 operands (state tile rounded once when it is written to LDS, weights packed
// as bf16) on v_mfma_f32_16x16x32_bf16 with fp32 accumulation - "bf16 mixer with MFMA"; then the kernel is bound by
// reading the states from HBM.  Mixing and the gradients stay fp32; the weight-gradient GEMM stays fp32 too unless the caller
// also sets flags & 2 (qmix_wide_wgrad_bf16_kernel: dhy and the states rounded to bf16 where a chunk is staged).
#include "common.h"
#include <cstdlib>
#include "../../include/marl_hip.h"

namespace {

constexpr int E = 32;
constexpr int NW = 8;             // waves per workgroup

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

#define WG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

struct WideArgs {
  const float* Wp;                // packed weights: fp32 [NCT][KC][64] f32x4, bf16 [NCT][KC32][64] 8 x bf16
  const float* Bc;                // [C] concatenated biases
  const float *wb2, *bb2;         // hyper_b2.2: (1,E), (1)
  ConcatSrc s;                    // state rows
  const float* q;                 // (rows, N)
  const float* g;                 // (rows) dL/dq_tot            (backward)
  float* q_tot;                   // (rows)                      (forward)
  float* dq;                      // (rows, N)                   (backward)
  float* dhy;                     // (rows, C) d(hypernet out)   (backward)
  float* slab;                    // [grid][E + 3] hyper_b2.2 gradient partials | loss numerator | sum(mask)
  // LOSS variant (backward with the TD loss folded in, as qmix_fused.hip): g is not read
  const float *lr, *lterm, *lpadded, *lq_tgt;
  float gamma;
  long rows;
  int N, S, C, NCT, KC;           // KC: k-chunks of 16 (fp32) or 32 (bf16)
};

struct PackArgs {
  const float* W[4]; const float* Bv[4];
  float* Wp; float* Bc;
  int N, S, C, NCT, KC, bf;
};

// column of the concatenated hypernet -> (segment, row inside the segment)
__device__ __forceinline__ void seg_of(int col, int NE, int& seg, int& r) {
  if (col < NE) { seg = 0; r = col; }
  else if (col < NE + E) { seg = 1; r = col - NE; }
  else if (col < NE + 2 * E) { seg = 2; r = col - NE - E; }
  else { seg = 3; r = col - NE - 2 * E; }
}

__global__ __launch_bounds__(256) void qmix_pack_kernel(PackArgs a) {
  const int NE = a.N * E;
  const long total = (long)a.NCT * a.KC * 64;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int l = (int)(e & 63);
    const long tc = e >> 6;
    const int ct = (int)(tc / a.KC), kc = (int)(tc - (long)ct * a.KC);
    const int col = 16 * ct + (l & 15), qq = l >> 4;
    int seg = 0, r = 0;
    const bool okc = col < a.C;
    if (okc) seg_of(col, NE, seg, r);
    const float* Wr = okc ? a.W[seg] + (long)r * a.S : nullptr;
    if (a.bf) {
      bf16x8_t v;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 32 * kc + 8 * qq + j;
        v[j] = (__bf16)((okc && k < a.S) ? Wr[k] : 0.f);
      }
      *reinterpret_cast<bf16x8_t*>(a.Wp + e * 4) = v;
    } else {
      f32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = 16 * kc + 4 * qq + i;
        v[i] = (okc && k < a.S) ? Wr[k] : 0.f;
      }
      *reinterpret_cast<f32x4*>(a.Wp + e * 4) = v;
    }
  }
  for (int c = blockIdx.x * 256 + threadIdx.x; c < a.C; c += gridDim.x * 256) {
    int seg, r;
    seg_of(c, NE, seg, r);
    a.Bc[c] = a.Bv[seg][r];
  }
}

__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
__device__ __forceinline__ float sum16(float v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
  return v;
}
// sum over each aligned group of 16 lanes with DPP moves (no LDS crossbar round trips); every lane gets the total
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_sum16(float v) {
  v += dpp_f<0xB1>(v);      // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);      // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);     // row_half_mirror: the other quad of each 8 lanes
  v += dpp_f<0x140>(v);     // row_mirror: the other half of the row
  return v;
}
__device__ __forceinline__ float sum32(float v) { v = sum16(v); v += __shfl_xor(v, 16, 64); return v; }

constexpr int NCTM = 26;          // column tiles of the concatenated hypernet (N <= 10: 16 * 26 = 416 columns)
constexpr int RB = 16 * NW;       // rows per block: one 16-row tile per wave

__host__ __device__ inline size_t wide_lds(int NCT) {
  // weight chunk x 2 [NCT][64] f32x4 | Qs [RB][16] | Gs [RB] | Ls [4][RB] | red [NW][E + 3]
  return (size_t)(2 * NCT * 256 + RB * 16 + RB + 4 * RB + NW * (E + 3)) * 4;
}

// One GEMM-shaped kernel: a workgroup walks blocks of 128 (episode, step) rows; wave w owns rows [16w, 16w + 16) of the
// block and ALL column tiles (<= 26 accumulator tiles = 104 registers), so the whole mixing arithmetic of a row - the sums
// over agents and over the 32 embedding units - stays inside one wave: no LDS scratch and no barrier in the epilogue.
// Per k-chunk the packed weight fragments of all tiles (26 KB) are staged once per workgroup into LDS (double buffered,
// one barrier per chunk; next chunk's fragments in flight in registers) and read by every wave; the state operand goes
// straight from HBM to registers (each wave reads only its own 16 rows; one chunk ahead).
// NCTT: compile-time number of column tiles (26 = MMM2's 10 agents: no per-tile branches, tile roles known statically) or
// 0 = read it from the arguments.
// LOSS (with BWD): the TD loss is folded in as in qmix_fused.hip - q_tot of a row is complete inside its wave, so dL/dq_tot is formed
// there (second pass over the cheap per-element math instead of more live registers) and the eval mixer's forward launch,
// the loss launch and its reduction are not needed.
template <bool BWD, bool BF, int NCTT, bool LOSS = false>
__global__ __launch_bounds__(64 * NW, 2) void qmix_wide_kernel(WideArgs a) {
  static_assert(!LOSS || BWD, "the loss is folded into the backward kernel");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q4 = lane >> 4, m = lane & 15;
  const int S = a.S, C = a.C, KC = a.KC;
  const int NCT = NCTT > 0 ? NCTT : a.NCT;
  const int N = NCTT > 0 ? (NCTT - 6) / 2 : a.N;
  float* Wl[2] = {smem, smem + NCT * 256};                              // [NCT][64] f32x4 (fp32) / 8 x bf16
  float* Qs = smem + 2 * NCT * 256;                                     // [RB][16]
  float* Gs = Qs + RB * 16;                                             // [RB]
  float* Ls = Gs + RB;                                                  // [4][RB] reward | terminated | padded | target q_tot (LOSS)
  float* red = Ls + 4 * RB;                                             // [NW][E + 3]
  const f32x4* Wp4 = reinterpret_cast<const f32x4*>(a.Wp);              // 16 bytes per (tile, chunk, lane) in both modes
  const int witems = NCT * 64;
  const int S4x4 = ((S + 3) >> 2) * 4;                                  // readable floats of a state row
  const float wb2lo = a.wb2[m], wb2hi = a.wb2[16 + m];
  const float bb2 = a.bb2[0];
  float acc_wb2[2] = {0.f, 0.f}, acc_bb2 = 0.f;
  float acc_ln = 0.f, acc_lm = 0.f;                                     // LOSS: sum (mask td)^2, sum mask (lanes m == 0)
  const long nblk = (a.rows + RB - 1) / RB;

  f32x4 wpf[4];                                                         // this thread's share of the next weight chunk
  auto wfetch = [&](int kc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int e = tid + 64 * NW * i;
      if (e > witems - 1) e = witems - 1;
      wpf[i] = Wp4[((long)(e >> 6) * KC + kc) * 64 + (e & 63)];
    }
  };
  auto wstash = [&](int b) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 64 * NW * i;
      if (e < witems) *reinterpret_cast<f32x4*>(Wl[b] + e * 4) = wpf[i];
    }
  };

  for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long row0 = blk * RB;
    const long rowm = row0 + 16 * wave + m;                             // the row this lane feeds as A operand
    const long rowc = rowm < a.rows ? rowm : a.rows - 1;
    const ConcatRow cr = concat_row(a.s, rowc);
    const float* srow = a.s.p0 + cr.r0 * a.s.ld0;
    // state fragments of chunk kc: fp32 4 floats at 16 kc + 4 q, bf16 8 floats at 32 kc + 8 q (zero past the row)
    auto aload = [&](f32x4 (&v)[2], int kc) __attribute__((always_inline)) {
      if (BF) {
        const int k0 = 32 * kc + 8 * q4;
        v[0] = k0 < S4x4 ? *reinterpret_cast<const f32x4*>(srow + k0) : (f32x4){0.f, 0.f, 0.f, 0.f};
        v[1] = k0 + 4 < S4x4 ? *reinterpret_cast<const f32x4*>(srow + k0 + 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
      } else {
        const int k0 = 16 * kc + 4 * q4;
        v[0] = k0 < S4x4 ? *reinterpret_cast<const f32x4*>(srow + k0) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    };
    // the block's q / g tiles (read in the epilogue, after at least one barrier of the chunk loop)
    for (int e = tid; e < RB * N; e += 64 * NW) {
      const long row = row0 + e / N;
      Qs[(e / N) * 16 + e % N] = row < a.rows ? a.q[row * N + e % N] : 0.f;
    }
    if (BWD && tid < RB) {
      const long row = row0 + tid;
      if (LOSS) {
        const bool ok = row < a.rows;                                   // rows past the batch: padded
        Ls[tid] = ok ? a.lr[row] : 0.f; Ls[RB + tid] = ok ? a.lterm[row] : 0.f;
        Ls[2 * RB + tid] = ok ? a.lpadded[row] : 1.f; Ls[3 * RB + tid] = ok ? a.lq_tgt[row] : 0.f;
      } else Gs[tid] = row < a.rows ? a.g[row] : 0.f;
    }

    f32x4 acc[NCTM];
#pragma unroll
    for (int ct = 0; ct < NCTM; ++ct) {
      const float bv = ct < NCT ? a.Bc[16 * ct + m] : 0.f;
      acc[ct] = (f32x4){bv, bv, bv, bv};
    }
    // state fragments travel FOUR chunks ahead of their use in four named register sets (a chunk of a 128-row block is
    // ~1 us of work for the workgroup, an HBM miss under load takes two or more: one chunk ahead - the first version -
    // left every chunk waiting for its own loads, the kernel ran at 0.2 of the HBM rate with both pipes idle)
    f32x4 a0[2], a1[2], a2[2], a3[2];
    const int KCm = KC - 1;
    wfetch(0);
    aload(a0, 0);
    aload(a1, 1 < KCm ? 1 : KCm);
    aload(a2, 2 < KCm ? 2 : KCm);
    wstash(0);
    if (KC > 1) wfetch(1);
    auto mac = [&](const f32x4 (&av)[2], int b) __attribute__((always_inline)) {
      const float* wl = Wl[b] + lane * 4;
      if (BF) {
        bf16x8_t a8;
        const bf16x4_t lo = __builtin_convertvector(av[0], bf16x4_t), hi = __builtin_convertvector(av[1], bf16x4_t);
#pragma unroll
        for (int i = 0; i < 4; ++i) { a8[i] = lo[i]; a8[4 + i] = hi[i]; }
#pragma unroll
        for (int ct = 0; ct < NCTM; ++ct)
          if (ct < NCT) {
            const bf16x8_t w8 = *reinterpret_cast<const bf16x8_t*>(wl + ct * 256);
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, w8, acc[ct], 0, 0, 0);
          }
      } else {
#pragma unroll
        for (int ct = 0; ct < NCTM; ++ct)
          if (ct < NCT) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(wl + ct * 256);
            acc[ct] = mfma16x4(av[0], w4, acc[ct]);
          }
      }
    };
    // chunk loop: buffer c & 1 holds the weights of chunk c; chunk c + 1 is in wpf and goes to the other buffer after the
    // barrier.  One step: barrier, stash + fetch weights, issue the state loads of chunk c + 3, multiply chunk c.
#define WIDE_STEP(CUR, NXT3, c_)                                                    \
    if ((c_) < KC) {                                                                \
      WG_BARRIER();          /* chunk c is in Wl[c & 1]; everybody is done with the other buffer */ \
      if ((c_) + 1 < KC) wstash(((c_) + 1) & 1);                                    \
      if ((c_) + 2 < KC) wfetch((c_) + 2);                                          \
      aload(NXT3, (c_) + 3 < KCm ? (c_) + 3 : KCm);                                 \
      mac(CUR, (c_) & 1);                                                           \
    }
    for (int kc = 0; kc < KC; kc += 4) {
      WIDE_STEP(a0, a3, kc)
      WIDE_STEP(a1, a0, kc + 1)
      WIDE_STEP(a2, a1, kc + 2)
      WIDE_STEP(a3, a2, kc + 3)
    }
#undef WIDE_STEP
    // ---- mixing, wave local.  acc[ct][i]: row 16 wave + 4 q + i, column 16 ct + m; column tile ct -> agent ct / 2, e-half
    // ct & 1 for the w1 tiles; then b1 (2N, 2N+1), w2 (2N+2, 2N+3), h (2N+4, 2N+5)
    const float* qrow = Qs + (16 * wave + 4 * q4) * 16;
    float pa[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i) pa[h][i] = 0.f;
#pragma unroll
    for (int ct = 0; ct < NCTM - 6; ++ct)
      if (ct < 2 * N) {
#pragma unroll
        for (int i = 0; i < 4; ++i) pa[ct & 1][i] += qrow[i * 16 + (ct >> 1)] * fabsf(acc[ct][i]);
      }
    // the six tail tiles sit at a runtime position 2N: select them with compile-time indices
    f32x4 tb1[2], tw2[2], th[2];
#pragma unroll
    for (int ct = 0; ct < NCTM; ++ct) {
      const int rel = ct - 2 * N;
      if (rel == 0) tb1[0] = acc[ct]; else if (rel == 1) tb1[1] = acc[ct];
      else if (rel == 2) tw2[0] = acc[ct]; else if (rel == 3) tw2[1] = acc[ct];
      else if (rel == 4) th[0] = acc[ct]; else if (rel == 5) th[1] = acc[ct];
    }
    float dpre[2][4], hidv[2][4], tot[4], gr[4] = {0.f, 0.f, 0.f, 0.f};
    // one pass over the per-element mixing math; with_grad also forms dpre / hid / the hyper_b2.2 gradient from gr
    auto mix_pass = [&](const bool with_grad) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) tot[i] = 0.f;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float ae = pa[h][i] + tb1[h][i];
          const float ex = __expf(ae);
          const float hid = ae > 0.f ? ae : ex - 1.f;                       // elu, alpha = 1
          const float w2 = fabsf(tw2[h][i]), hb = fmaxf(th[h][i], 0.f);
          tot[i] += hid * w2 + hb * (h ? wb2hi : wb2lo);
          if (with_grad) {
            dpre[h][i] = gr[i] * w2 * (ae > 0.f ? 1.f : ex);
            hidv[h][i] = hid;
            acc_wb2[h] += gr[i] * hb;
          }
        }
    };
    const long rbase = row0 + 16 * wave + 4 * q4;
    if (BWD && !LOSS) {
#pragma unroll
      for (int i = 0; i < 4; ++i) gr[i] = Gs[16 * wave + 4 * q4 + i];
    }
    mix_pass(BWD && !LOSS);
    if (!BWD) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float t = sum16(tot[i]);
        if (m == 0 && rbase + i < a.rows) a.q_tot[rbase + i] = t + bb2;
      }
    } else {
      if (LOSS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rl = 16 * wave + 4 * q4 + i;
          const float qt = sum16(tot[i]) + bb2;
          const float mask = 1.f - Ls[2 * RB + rl];
          const float target = Ls[rl] + a.gamma * Ls[3 * RB + rl] * (1.f - Ls[RB + rl]);
          const float mtd = mask * (target - qt);
          gr[i] = -2.f * mask * mtd;
          if (m == 0) {
            acc_ln += mtd * mtd; acc_lm += mask;
            if (a.q_tot && rbase + i < a.rows) a.q_tot[rbase + i] = qt;
          }
        }
        mix_pass(true);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (m == 0) acc_bb2 += gr[i];
      // d(hypernet output) in place, dq_n = sum_e |w1[n,e]| dpre_e
#pragma unroll
      for (int ct = 0; ct < NCTM; ++ct) {
        if (ct >= NCT) continue;
        const int rel = ct - 2 * N, h = ct & 1;
        f32x4 v;
        if (rel < 0) {
          float dqp[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float o = acc[ct][i];
            v[i] = qrow[i * 16 + (ct >> 1)] * dpre[h][i] * sgn(o);
            dqp[i] = fabsf(o) * dpre[h][i];
          }
          if (h == 1) {                                   // both halves of agent ct / 2 are in this wave: ct - 1 and ct
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float t = sum16(dqp[i] + fabsf(acc[ct - 1 < 0 ? 0 : ct - 1][i]) * dpre[0][i]);
              if (m == 0 && rbase + i < a.rows) a.dq[(rbase + i) * N + (ct >> 1)] = t;
            }
          }
        } else if (rel < 2) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = dpre[h][i];
        } else if (rel < 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = gr[i] * hidv[h][i] * sgn(acc[ct][i]);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = acc[ct][i] > 0.f ? gr[i] * (h ? wb2hi : wb2lo) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (rbase + i < a.rows) a.dhy[(rbase + i) * C + 16 * ct + m] = v[i];
      }
    }
    WG_BARRIER();                                        // Qs / Gs / both weight buffers are free for the next block
  }
  if (BWD) {
    // hyper_b2.2 gradient partials: rows 4q + i summed over the lane quarters, then the waves in fixed order
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      acc_wb2[h] += __shfl_xor(acc_wb2[h], 16, 64);
      acc_wb2[h] += __shfl_xor(acc_wb2[h], 32, 64);
    }
    acc_bb2 += __shfl_xor(acc_bb2, 16, 64);
    acc_bb2 += __shfl_xor(acc_bb2, 32, 64);
    acc_ln += __shfl_xor(acc_ln, 16, 64); acc_ln += __shfl_xor(acc_ln, 32, 64);       // lanes m == 0 of the four row groups
    acc_lm += __shfl_xor(acc_lm, 16, 64); acc_lm += __shfl_xor(acc_lm, 32, 64);
    __syncthreads();
    if (q4 == 0) { red[wave * (E + 3) + m] = acc_wb2[0]; red[wave * (E + 3) + 16 + m] = acc_wb2[1]; }
    if (lane == 0) { red[wave * (E + 3) + E] = acc_bb2; red[wave * (E + 3) + E + 1] = acc_ln; red[wave * (E + 3) + E + 2] = acc_lm; }
    __syncthreads();
    if (tid < E + 3) {
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) tot += red[w * (E + 3) + tid];
      a.slab[(long)blockIdx.x * (E + 3) + tid] = tot;
    }
  }
}

// ------------------------------------------------------------------------------------------------- weight gradient
// dW[c][k] = sum_rows dhy[row][c] s[row][k], db[c] = sum_rows dhy[row][c].  grid = (row slabs, column groups of 7 tiles):
// the groups of one row slab are gridDim.x blocks apart (a multiple of 8: same XCD), so the slab's state rows come from
// HBM once and hit that XCD's L2 for the other groups.
constexpr int WCH = 32;           // rows per staged chunk
constexpr int WNT = 7;            // column tiles per group: 7 x 3 accumulator tiles per wave (84 registers; 13 x 3 spilled)
constexpr int WKT = 3;            // k tiles per wave (8 waves x 3 cover S <= 384)

struct WideWgArgs {
  const float* dhy; ConcatSrc s; float* ws;      // slabs [nslab][C][Kx], Kx = 16 KT + 1 (bias in the last column)
  long rows; int S, C, KT, nslab;
};
__host__ __device__ inline int wg_gp() { return 16 * WNT + 32; }               // LDS pitch of the dhy chunk: 144 = 16 (mod 32)
__host__ __device__ inline int wg_xp(int KT) { const int w = 16 * KT; return (w % 32 == 16) ? w : w + 16; }

__global__ __launch_bounds__(64 * NW, 2) void qmix_wide_wgrad_kernel(WideWgArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, m = lane & 15;
  const int GP = wg_gp(), XP = wg_xp(a.KT);
  // buffer b: dhy chunk at smem + b * BUF, state chunk behind it (plain offsets from the LDS base - a runtime-indexed array
  // of pointers made the compiler fall back to FLAT loads, whose waits also drain the global prefetch)
  const int BUF = WCH * (GP + XP);
  // source offset (floats) of every row of a chunk, resolved ONE CHUNK AHEAD by 32 threads: the row remap / episode map
  // costs a dependent global load per row, and resolving it inside the staging loads serialised six L2 round trips per chunk
  long* rtab = reinterpret_cast<long*>(smem + 2 * BUF);                     // [2][WCH]
  const int grp = blockIdx.y, col0 = grp * 16 * WNT;
  const int S4 = (a.S + 3) >> 2;
  const long per = (a.rows + a.nslab - 1) / a.nslab;
  const long r_begin = (long)blockIdx.x * per;
  long r_end = r_begin + per; if (r_end > a.rows) r_end = a.rows;
  const long nch = r_end > r_begin ? (r_end - r_begin + WCH - 1) / WCH : 0;
  // staging items: dhy chunk 32 rows x 52 float4 (columns of this group), state chunk 32 rows x S4 float4
  constexpr int G4 = 4 * WNT;
  const int gi = WCH * G4, xi = WCH * S4;
  constexpr int NG = (WCH * G4 + 64 * NW - 1) / (64 * NW);       // 2
  constexpr int NX = 6;                                          // covers S <= 384
  f32x4 pg[NG], px[NX];
  const float invS4 = 1.0f / (float)S4;
  auto fetch = [&](long ch) {
    const long rb = r_begin + ch * WCH;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      int e = tid + 64 * NW * i;
      if (e > gi - 1) e = gi - 1;
      const int r = e / G4, c4 = (e - r * G4) * 4;
      long row = rb + r;
      const bool live = row < r_end && col0 + c4 < a.C;
      if (row > a.rows - 1) row = a.rows - 1;
      const int cc = col0 + c4 < a.C ? col0 + c4 : 0;
      f32x4 v = *reinterpret_cast<const f32x4*>(a.dhy + row * a.C + cc);
      if (!live) v = (f32x4){0.f, 0.f, 0.f, 0.f};
      pg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      int e = tid + 64 * NW * i;
      if (e > xi - 1) e = xi - 1;
      const int r = (int)(((float)e + 0.5f) * invS4);
      const int c4 = (e - r * S4) * 4;
      px[i] = *reinterpret_cast<const f32x4*>(a.s.p0 + rtab[(ch & 1) * WCH + r] + c4);
    }
  };
  auto resolve = [&](long ch) {                    // threads 0..31: table of chunk ch
    if (tid < WCH) {
      long row = r_begin + ch * WCH + tid;
      if (row > a.rows - 1) row = a.rows - 1;
      const ConcatRow cr = concat_row(a.s, row);
      rtab[(ch & 1) * WCH + tid] = cr.r0 * a.s.ld0;
    }
  };
  auto stash = [&](int b) {
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int e = tid + 64 * NW * i;
      if (e < gi) { const int r = e / G4, c4 = (e - r * G4) * 4; *reinterpret_cast<f32x4*>(smem + b * BUF + r * GP + c4) = pg[i]; }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = tid + 64 * NW * i;
      if (e < xi) {
        const int r = (int)(((float)e + 0.5f) * invS4);
        const int c4 = (e - r * S4) * 4;
        f32x4 v = px[i];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) if (c4 + cc >= a.S) v[cc] = 0.f;
        *reinterpret_cast<f32x4*>(smem + b * BUF + WCH * GP + r * XP + c4) = v;
      }
    }
  };
  // zero the never-written pad columns of both buffers (k padding of the state chunk)
  for (int b = 0; b < 2; ++b)
    for (int e = tid; e < WCH * (XP - 4 * S4); e += 64 * NW)
      smem[b * BUF + WCH * GP + (e / (XP - 4 * S4)) * XP + 4 * S4 + e % (XP - 4 * S4)] = 0.f;
  f32x4 acc[WNT][WKT];
  float bs[WNT];
#pragma unroll
  for (int t = 0; t < WNT; ++t) {
    bs[t] = 0.f;
#pragma unroll
    for (int k = 0; k < WKT; ++k) acc[t][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  resolve(0);
  resolve(1);
  __syncthreads();
  if (nch > 0) fetch(0);
  __syncthreads();
  if (nch > 0) stash(0);
  for (long ch = 0; ch < nch; ++ch) {
    const int b = (int)(ch & 1);
    WG_BARRIER();                                  // chunk ch is in buffer b; buffer b^1 is free (read two chunks ago)
    if (ch + 1 < nch) fetch(ch + 1);               // (row table of chunk ch + 1: written before this barrier)
    if (ch + 2 < nch) resolve(ch + 2);             // overwrites the table of chunk ch, whose loads were issued long ago
    const float* G = smem + b * BUF;
    const float* X = G + WCH * GP;
    // operands of step st+1 are read while step st multiplies (two named register sets, no copies)
    auto ld = [&](float (&gv)[WNT], float (&xv)[WKT], int st) __attribute__((always_inline)) {
      const int row = 4 * st + q;                  // MFMA k index = lane quarter = one row of the chunk
#pragma unroll
      for (int k = 0; k < WKT; ++k) {
        const int kt = wave + NW * k;
        xv[k] = X[row * XP + 16 * (kt < a.KT ? kt : 0) + m];
      }
#pragma unroll
      for (int t = 0; t < WNT; ++t) gv[t] = G[row * GP + 16 * t + m];
    };
    auto mac = [&](const float (&gv)[WNT], const float (&xv)[WKT]) __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < WNT; ++t) {
        if (wave == 0) bs[t] += gv[t];
#pragma unroll
        for (int k = 0; k < WKT; ++k) acc[t][k] = mfma16(gv[t], xv[k], acc[t][k]);   // k tiles past KT: a clamped operand,
      }                                                                                // a never-stored accumulator (no branch:
      __builtin_amdgcn_sched_barrier(0);                                               // those waves wait at the barrier anyway)
    };
    float gA[WNT], xA[WKT], gB[WNT], xB[WKT];
    ld(gA, xA, 0);
#pragma unroll 1
    for (int st = 0; st < WCH / 4; st += 2) {
      ld(gB, xB, st + 1);
      mac(gA, xA);
      ld(gA, xA, st + 2 < WCH / 4 ? st + 2 : st + 1);
      mac(gB, xB);
    }
    if (ch + 1 < nch) stash(b ^ 1);
  }
  // ---- slab: rows = columns of the hypernet output, bias gradient in column 16 KT
  const int Kx = 16 * a.KT + 1;
  float* slab = a.ws + (long)blockIdx.x * a.C * Kx;
#pragma unroll
  for (int t = 0; t < WNT; ++t) {
#pragma unroll
    for (int k = 0; k < WKT; ++k) {
      const int kt = wave + NW * k;
      if (kt >= a.KT) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = col0 + 16 * t + 4 * q + i;
        if (c < a.C) slab[(long)c * Kx + 16 * kt + m] = acc[t][k][i];
      }
    }
    if (wave == 0) {
      float v = bs[t];
      v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
      const int c = col0 + 16 * t + m;
      if (q == 0 && c < a.C) slab[(long)c * Kx + 16 * a.KT] = v;
    }
  }
}

// The same reduction with bf16 operands (flags & 2 beside flags & 1; BASELINE config 5 "bf16 mixer with MFMA"): dhy and the states are rounded to
// bf16 where they are staged, a 32-row chunk is ONE k-step of v_mfma_f32_16x16x32_bf16 (21 MFMAs per wave and chunk instead of
// 168 fp32 ones), and - the reduction index being the row - both operands come out of row-major [row][column] bf16 images
// through ds_read_b64_tr_b16 (lane (g, i) receives column i of rows 8g .. 8g + 3: cdna_hip_programming.md T10).  The kernel is
// then bound by streaming dhy and the states (55 KB per chunk and workgroup), not by the multiplies.  fp32 accumulation; the
// bias gradient stays an exact fp32 column sum of dhy (taken where the chunk is staged).
typedef short wg_s16x4 __attribute__((ext_vector_type(4)));
typedef short wg_s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf4 __attribute__((ext_vector_type(4)));
typedef int wg_i32x2 __attribute__((ext_vector_type(2)));
__host__ __device__ inline int wgb_gb() { return 2 * 16 * WNT; }                  // bytes per image row of the dhy chunk (224)
__host__ __device__ inline int wgb_xb(int KT) { return 2 * 16 * KT; }             // ... of the state chunk (672 at MMM2)

__global__ __launch_bounds__(64 * NW, 2) void qmix_wide_wgrad_bf16_kernel(WideWgArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, m = lane & 15;
  const int GB = wgb_gb(), XB = wgb_xb(a.KT);
  const int BUF = WCH * (GB + XB);                   // bytes per buffer
  char* img = reinterpret_cast<char*>(smem);
  long* rtab = reinterpret_cast<long*>(img + 2 * BUF);                      // [2][WCH] source offsets of the chunk's rows
  const int grp = blockIdx.y, col0 = grp * 16 * WNT;
  const int S4 = (a.S + 3) >> 2;
  const long per = (a.rows + a.nslab - 1) / a.nslab;
  const long r_begin = (long)blockIdx.x * per;
  long r_end = r_begin + per; if (r_end > a.rows) r_end = a.rows;
  const long nch = r_end > r_begin ? (r_end - r_begin + WCH - 1) / WCH : 0;
  constexpr int G4 = 4 * WNT;
  const int gi = WCH * G4, xi = WCH * S4;
  constexpr int NG = (WCH * G4 + 64 * NW - 1) / (64 * NW);       // 2
  constexpr int NX = 6;                                          // covers S <= 384
  f32x4 pg[NG], px[NX], bsum[NG];
#pragma unroll
  for (int i = 0; i < NG; ++i) bsum[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float invS4 = 1.0f / (float)S4;
  auto fetch = [&](long ch) {
    const long rb = r_begin + ch * WCH;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      int e = tid + 64 * NW * i;
      if (e > gi - 1) e = gi - 1;
      const int r = e / G4, c4 = (e - r * G4) * 4;
      long row = rb + r;
      const bool live = row < r_end && col0 + c4 < a.C && tid + 64 * NW * i < gi;
      if (row > a.rows - 1) row = a.rows - 1;
      const int cc = col0 + c4 < a.C ? col0 + c4 : 0;
      f32x4 v = *reinterpret_cast<const f32x4*>(a.dhy + row * a.C + cc);
      if (!live) v = (f32x4){0.f, 0.f, 0.f, 0.f};
      pg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      int e = tid + 64 * NW * i;
      if (e > xi - 1) e = xi - 1;
      const int r = (int)(((float)e + 0.5f) * invS4);
      const int c4 = (e - r * S4) * 4;
      px[i] = *reinterpret_cast<const f32x4*>(a.s.p0 + rtab[(ch & 1) * WCH + r] + c4);
    }
  };
  auto resolve = [&](long ch) {
    if (tid < WCH) {
      long row = r_begin + ch * WCH + tid;
      if (row > a.rows - 1) row = a.rows - 1;
      const ConcatRow cr = concat_row(a.s, row);
      rtab[(ch & 1) * WCH + tid] = cr.r0 * a.s.ld0;
    }
  };
  auto pack4 = [](const f32x4& v) { return __builtin_bit_cast(wg_i32x2, __builtin_convertvector(v, wg_bf4)); };
  auto stash = [&](int b) {
    char* G = img + b * BUF;
    char* X = G + WCH * GB;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int e = tid + 64 * NW * i;
      bsum[i] += pg[i];                              // exact fp32 column sums (rows past the slab were zeroed in fetch)
      if (e < gi) { const int r = e / G4, c4 = (e - r * G4) * 4; *reinterpret_cast<wg_i32x2*>(G + r * GB + 2 * c4) = pack4(pg[i]); }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = tid + 64 * NW * i;
      if (e < xi) {
        const int r = (int)(((float)e + 0.5f) * invS4);
        const int c4 = (e - r * S4) * 4;
        f32x4 v = px[i];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) if (c4 + cc >= a.S) v[cc] = 0.f;
        *reinterpret_cast<wg_i32x2*>(X + r * XB + 2 * c4) = pack4(v);
      }
    }
  };
  // zero the never-written pad columns of both state images (k padding up to 16 KT)
  for (int b = 0; b < 2; ++b)
    for (int e = tid; e < WCH * (16 * a.KT - 4 * S4); e += 64 * NW) {
      const int w = 16 * a.KT - 4 * S4;
      *reinterpret_cast<short*>(img + b * BUF + WCH * GB + (e / w) * XB + 2 * (4 * S4 + e % w)) = 0;
    }
  f32x4 acc[WNT][WKT];
#pragma unroll
  for (int t = 0; t < WNT; ++t)
#pragma unroll
    for (int k = 0; k < WKT; ++k) acc[t][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // transposed-read addresses of this lane: rows 8g + qq (+ 4), 8 bytes at column c0 + 4p
  const int qq = m >> 2, pp = m & 3;
  const int gA = (8 * q + qq) * GB + 8 * pp, xA = (8 * q + qq) * XB + 8 * pp;
  typedef __attribute__((address_space(3))) wg_s16x4 lds_s16x4;
  auto tr8 = [&](const char* base, int off, int pitch) __attribute__((always_inline)) {
    const wg_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off));
    const wg_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off + 4 * pitch));
    return __builtin_bit_cast(wg_bf8, (wg_s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  };
  resolve(0);
  resolve(1);
  __syncthreads();
  if (nch > 0) fetch(0);
  __syncthreads();
  if (nch > 0) stash(0);
  for (long ch = 0; ch < nch; ++ch) {
    const int b = (int)(ch & 1);
    WG_BARRIER();                                  // chunk ch is in buffer b; buffer b^1 is free (read two chunks ago)
    if (ch + 1 < nch) fetch(ch + 1);
    if (ch + 2 < nch) resolve(ch + 2);
    const char* G = img + b * BUF;
    const char* X = G + WCH * GB;
    wg_bf8 xb[WKT];
#pragma unroll
    for (int k = 0; k < WKT; ++k) {
      const int kt = wave + NW * k;
      xb[k] = tr8(X, xA + 32 * (kt < a.KT ? kt : 0), XB);
    }
#pragma unroll
    for (int t = 0; t < WNT; ++t) {
      const wg_bf8 ga = tr8(G, gA + 32 * t, GB);
#pragma unroll
      for (int k = 0; k < WKT; ++k) acc[t][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga, xb[k], acc[t][k], 0, 0, 0);
    }
    if (ch + 1 < nch) stash(b ^ 1);
  }
  // ---- slab: rows = columns of the hypernet output, bias gradient in column 16 KT
  const int Kx = 16 * a.KT + 1;
  float* slab = a.ws + (long)blockIdx.x * a.C * Kx;
#pragma unroll
  for (int t = 0; t < WNT; ++t)
#pragma unroll
    for (int k = 0; k < WKT; ++k) {
      const int kt = wave + NW * k;
      if (kt >= a.KT) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = col0 + 16 * t + 4 * q + i;
        if (c < a.C) slab[(long)c * Kx + 16 * kt + m] = acc[t][k][i];
      }
    }
  // bias: the per-thread column sums (thread e -> chunk row e / G4, columns 4 (e % G4) ..) added over the 32 chunk rows
  __syncthreads();
  float* red = smem;                                 // [WCH][16 WNT]
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const int e = tid + 64 * NW * i;
    if (e < gi) { const int r = e / G4, c4 = (e - r * G4) * 4; *reinterpret_cast<f32x4*>(red + r * (16 * WNT) + c4) = bsum[i]; }
  }
  __syncthreads();
  if (tid < 16 * WNT) {
    float v = 0.f;
    for (int r = 0; r < WCH; ++r) v += red[r * (16 * WNT) + tid];
    const int c = col0 + tid;
    if (c < a.C) slab[(long)c * Kx + 16 * a.KT] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Forward with bf16 operands and the weights RESIDENT in LDS (BASELINE config 5: "bf16 mixer with MFMA" - the kernel SURVEY 8d
// prices against the HBM read of the states).  The streaming kernel above re-stages all 26 column tiles per 128-row block
// (one barrier and one L2 round trip per k-chunk) and, at 216 registers, runs ONE workgroup per CU: every chunk waited for
// its own loads - 88 us = 0.22 of the HBM rate with the matrix pipe 17 % busy.  Here the column tiles are split by
// embedding half (e < 16 / e >= 16): a workgroup keeps the N + 3 tiles of ITS half - w1[n, half], b1, w2, h - for the whole
// k range in LDS (13 tiles x 11 chunks x 1 KB = 143 KB at MMM2), loaded once, and then only streams states: no barrier in the
// row loop, one load stream per wave, so the state fragments of the NEXT row tile (the whole k range: 22 x 16 B per lane)
// are in flight while the current one is multiplied - each register pair is re-issued as soon as its chunk is consumed.
// The mixing math of a half is wave local as before; a row's q_tot is the sum of its two halves, formed by one float atomic
// each on a zeroed output (two addends: the result does not depend on their order).  The two workgroups of a row group sit
// on the same XCD (blockIdx % 8), so the second read of a state row is an L2 hit.
#ifndef RES_SB
#define RES_SB 1
#endif
template <int NH, int KCT>
__global__ __launch_bounds__(64 * NW, 2) void qmix_wide_res_fwd_kernel(WideArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int N = NH - 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q4 = lane >> 4, m = lane & 15;
  const int xcd = blockIdx.x & 7, rr = blockIdx.x >> 3, h = rr & 1;
  const int grp = (rr >> 1) * 8 + xcd, ngrp = (int)(gridDim.x >> 1);
  ST_DECL(5);
  float* Wl = smem;                                   // [NH][KCT][64] 8 x bf16
  float* Qw = smem + NH * KCT * 256 + wave * 256;     // this wave's [16][16] q tile
  float* Bl = smem + NH * KCT * 256 + NW * 256;       // [NH][16] biases of this half's columns
  const f32x4* Wp4 = reinterpret_cast<const f32x4*>(a.Wp);
  auto ct_of = [&](int j) { return j < N ? 2 * j + h : 2 * N + 2 * (j - N) + h; };      // w1[n, half] | b1 | w2 | h tiles of this half
  {
    // all of a thread's loads in flight before the first LDS store (a load - wait - store loop serialised 18 memory round trips)
    constexpr int NIT = (NH * KCT * 64 + 64 * NW - 1) / (64 * NW);
    f32x4 wv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int e = tid + 64 * NW * it; if (e > NH * KCT * 64 - 1) e = NH * KCT * 64 - 1;
      const int j = e / (KCT * 64), rem = e - j * (KCT * 64);
      wv[it] = Wp4[(long)ct_of(j) * (KCT * 64) + rem];
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = tid + 64 * NW * it;
      if (e < NH * KCT * 64) *reinterpret_cast<f32x4*>(Wl + (long)e * 4) = wv[it];
    }
  }
  if (tid < NH * 16) Bl[tid] = a.Bc[16 * ct_of(tid >> 4) + (tid & 15)];
  const float wb2 = a.wb2[16 * h + m];
  const float bb2 = h == 0 ? a.bb2[0] : 0.f;          // added once per row (by the first half)
  __syncthreads();

  const long tiles = (a.rows + 15) >> 4;
  const long per = (tiles + ngrp - 1) / ngrp;
  const long t_begin = (long)grp * per;
  const long t_end = t_begin + per < tiles ? t_begin + per : tiles;
  long tile = t_begin + wave;
  if (tile >= t_end) return;
  const int S4x4 = ((a.S + 3) >> 2) * 4;              // readable floats of a state row
  // fragment loads: chunk kc, half hf = 4 floats at column 32 kc + 8 q + 4 hf - an immediate offset from (row + 8 q); only
  // the LAST chunk can run past the row: its two offsets are clamped into it (the weights of columns >= S are zero)
  const int kl0 = (32 * (KCT - 1) + 8 * q4 < S4x4 ? 32 * (KCT - 1) + 8 * q4 : S4x4 - 4) - 8 * q4;
  const int kl1 = (32 * (KCT - 1) + 8 * q4 + 4 < S4x4 ? 32 * (KCT - 1) + 8 * q4 + 4 : S4x4 - 4) - 8 * q4;
  auto kof = [&](int kc, int hf) { return kc < KCT - 1 ? 32 * kc + 4 * hf : (hf ? kl1 : kl0); };
  auto srow_of = [&](long tl) -> const float* {
    long row = tl * 16 + m;
    if (row > a.rows - 1) row = a.rows - 1;
    const ConcatRow cr = concat_row(a.s, row);
    return a.s.p0 + cr.r0 * a.s.ld0 + 8 * q4;
  };
  f32x4 st[KCT][2];
  const float* srow = srow_of(tile);
#pragma unroll
  for (int kc = 0; kc < KCT; ++kc) {
    st[kc][0] = *reinterpret_cast<const f32x4*>(srow + kof(kc, 0));
    st[kc][1] = *reinterpret_cast<const f32x4*>(srow + kof(kc, 1));
  }
  // the row address of the tile after next is resolved a tile ahead of the loads that use it (episode map: a dependent load)
  const float* srow_n = srow_of(tile + NW < t_end ? tile + NW : tile);
  ST_MARK(4);       // prologue (weights -> LDS, first loads)
  for (; tile < t_end; tile += NW) {
    const long t2 = tile + 2 * NW < t_end ? tile + 2 * NW : (tile + NW < t_end ? tile + NW : tile);
    const float* srow_n2 = srow_of(t2);
    // q of this wave's 16 rows (issued first: consumed in the epilogue, when it is the oldest load in flight)
    float qv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      int e = lane + 64 * i; if (e > 16 * N - 1) e = 16 * N - 1;
      long row = tile * 16 + e / N; if (row > a.rows - 1) row = a.rows - 1;
      qv[i] = a.q[row * N + e % N];
    }
    f32x4 acc[NH];
#pragma unroll
    for (int j = 0; j < NH; ++j) { const float b = Bl[16 * j + m]; acc[j] = (f32x4){b, b, b, b}; }
    ST_MARK(0);
#pragma unroll
    for (int kc = 0; kc < KCT; ++kc) {
      bf16x8_t a8;
      const bf16x4_t lo = __builtin_convertvector(st[kc][0], bf16x4_t), hi = __builtin_convertvector(st[kc][1], bf16x4_t);
#pragma unroll
      for (int i = 0; i < 4; ++i) { a8[i] = lo[i]; a8[4 + i] = hi[i]; }
      // this chunk's registers are free: the same chunk of the next row tile goes into them (unconditional, clamped)
      st[kc][0] = *reinterpret_cast<const f32x4*>(srow_n + kof(kc, 0));
      st[kc][1] = *reinterpret_cast<const f32x4*>(srow_n + kof(kc, 1));
      const float* wl = Wl + (kc * 64 + lane) * 4;
#pragma unroll
      for (int j = 0; j < NH; ++j) {
        const bf16x8_t w8 = *reinterpret_cast<const bf16x8_t*>(wl + j * (KCT * 256));
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, w8, acc[j], 0, 0, 0);
      }
      if (RES_SB) __builtin_amdgcn_sched_barrier(0);      // (keeps the chunk order: loads of the next tile issued chunk by chunk)
    }
    ST_MARK(1);
    // ---- mixing of this half, wave local: acc[j][i] = row 4 q + i, embedding unit 16 h + m
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int e = lane + 64 * i;
      if (e < 16 * N) Qw[(e / N) * 16 + e % N] = qv[i];
    }
    ST_MARK(2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's own LDS writes (no other wave touches Qw)
    const float* qrow = Qw + 4 * q4 * 16;
    const long rbase = tile * 16 + 4 * q4;
    float tsum[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // the row's q values: three 16-byte LDS reads (same address in all 16 lanes of the row group: a broadcast)
      f32x4 qa[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) qa[u] = *reinterpret_cast<const f32x4*>(qrow + i * 16 + 4 * u);
      float pa = 0.f;
#pragma unroll
      for (int n = 0; n < N; ++n) pa += qa[n >> 2][n & 3] * fabsf(acc[n][i]);
      const float ae = pa + acc[N][i];
      const float ex = __expf(ae);
      const float hid = ae > 0.f ? ae : ex - 1.f;                         // elu, alpha = 1
      tsum[i] = row_sum16(hid * fabsf(acc[N + 1][i]) + fmaxf(acc[N + 2][i], 0.f) * wb2) + bb2;
    }
    if (m == 0) {                                    // one divergent region for the four rows' atomics
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (rbase + i < a.rows) atomicAdd(a.q_tot + rbase + i, tsum[i]);
    }
    srow_n = srow_n2;
    ST_MARK(3);
  }
  ST_DUMP(5);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same forward on 32-row tiles (v_mfma_f32_32x32x16_bf16) in TRANSPOSED form: out^T[column][row] = W[column][k] s^T[k][row].
// Why: with 16-row tiles every 1 KB weight-fragment read from LDS feeds ONE MFMA, so the resident kernel above needs the LDS's full
// 256 B/clk just to keep the bf16 pipe busy (stamps: 47 cycles per MFMA); a 32 x 32 x 16 MFMA does twice the multiplies per
// fragment byte.  Transposed, because the 32 x 32 accumulator has its COLUMN index on the lane and its row index in the 16
// registers: with the batch row on the lane, everything a row's mixing needs - the 32 hypernet columns of a tile - sits in
// that lane's registers (two lanes per row: l and l + 32 hold eight embedding units each), so the sums over agents and
// embedding units are plain register arithmetic and one cross-lane add; q is loaded straight into registers, no LDS tile.
// Column tiles of an embedding half (16 units): five tiles [agent 2t | agent 2t+1], one [b1 | w2], one [h | 0] - 7 x 21
// k-chunks x 1 KB = 147 KB of bf16 fragments, written by the workgroup itself from the fp32 parameters (no pack launch).
struct Wide32Args {
  const float* W[4]; const float* Bv[4];      // w1 (N*E, S) | b1 (E, S) | w2 (E, S) | h (E, S) and their biases
  float* Wp;                                  // packed bf16 fragments [2 halves][7][KCT][64] x 16 B (pack kernel -> main kernel)
  const float *wb2, *bb2;
  ConcatSrc s;
  const float* q;
  float* q_tot;
  long rows;
  int S;
};
typedef float f32x16 __attribute__((ext_vector_type(16)));

// column i of tile t of embedding half eh -> (segment, row of the segment), or none
__device__ __forceinline__ void wide32_col(int eh, int t, int i, int& seg, int& row) {
  const int g = i >> 4, e = 16 * eh + (i & 15);
  if (t < 5) { seg = 0; row = (2 * t + g) * E + e; }
  else if (t == 5) { seg = g ? 2 : 1; row = e; }
  else { seg = g ? -1 : 3; row = e; }
}
// fp32 parameters -> bf16 A fragments of the 32 x 32 x 16 MFMA: item (half, tile, chunk, lane l) = W[column l & 31][16 kc + 8 (l >> 5) + 0..7]
__global__ __launch_bounds__(256) void qmix_pack32_kernel(Wide32Args a, int KCT) {
  const long total = 2L * 7 * KCT * 64;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int l = (int)(e & 63);
    const long tc = e >> 6;
    const int kc = (int)(tc % KCT), t = (int)((tc / KCT) % 7), eh = (int)(tc / (7L * KCT));
    int seg, row;
    wide32_col(eh, t, l & 31, seg, row);
    const int k0 = 16 * kc + 8 * (l >> 5);
    bf16x8_t v;
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (__bf16)((seg >= 0 && k0 + u < a.S) ? a.W[seg][(long)row * a.S + k0 + u] : 0.f);
    *reinterpret_cast<bf16x8_t*>(a.Wp + e * 4) = v;
  }
}

template <int KCT /* 16-wide k chunks */, int PF /* chunks of state prefetch in flight; KCT % PF == 0 */>
__global__ __launch_bounds__(64 * NW, 2) void qmix_wide_res32_fwd_kernel(Wide32Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int N = 10, NT = 7;
  static_assert(KCT % PF == 0, "the prefetch slots rotate with the chunk index");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hk = lane >> 5;
  const int xcd = blockIdx.x & 7, rr = blockIdx.x >> 3, eh = rr & 1;          // eh: embedding half of this workgroup
  const int grp = (rr >> 1) * 8 + xcd, ngrp = (int)(gridDim.x >> 1);
  ST_DECL(5);
  float* Wl = smem;                                   // [NT][KCT][64] 8 x bf16 (A fragments: lane l = column l & 31, k = 8 (l >> 5) ..)
  float* Bl = smem + NT * KCT * 256;                  // [NT][2][16]: bias of the column a lane half holds in register r
  {
    // this half's fragment image (written by qmix_pack32_kernel): all of a thread's loads in flight before the first LDS store
    constexpr int NIT = (NT * KCT * 64 + 64 * NW - 1) / (64 * NW);
    const f32x4* Wp4 = reinterpret_cast<const f32x4*>(a.Wp) + (long)eh * (NT * KCT * 64);
    f32x4 wv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int e = tid + 64 * NW * it; if (e > NT * KCT * 64 - 1) e = NT * KCT * 64 - 1;
      wv[it] = Wp4[e];
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = tid + 64 * NW * it;
      if (e < NT * KCT * 64) *reinterpret_cast<f32x4*>(Wl + (long)e * 4) = wv[it];
    }
  }
  if (tid < NT * 32) {
    const int t = tid >> 5, h2 = (tid >> 4) & 1, r = tid & 15;
    int seg, row;
    wide32_col(eh, t, (r & 3) + 8 * (r >> 2) + 4 * h2, seg, row);
    Bl[tid] = seg >= 0 ? a.Bv[seg][row] : 0.f;
  }
  // hyper_b2.2 weights of the eight embedding units of this lane half: unit u -> e = 16 eh + (u & 3) + 8 (u >> 2) + 4 hk
  float wb2v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) wb2v[u] = a.wb2[16 * eh + (u & 3) + 8 * (u >> 2) + 4 * hk];
  const float bb2 = eh == 0 ? a.bb2[0] : 0.f;
  __syncthreads();

  const long tiles = (a.rows + 31) >> 5;
  const long per = (tiles + ngrp - 1) / ngrp;
  const long t_begin = (long)grp * per;
  const long t_end = t_begin + per < tiles ? t_begin + per : tiles;
  long tile = t_begin + wave;
  if (tile >= t_end) return;
  const int S4x4 = ((a.S + 3) >> 2) * 4;
  // B fragment of chunk kc: 8 floats at column 16 kc + 8 hk of the lane's row - two 16-byte loads at immediate offsets from
  // (row + 8 hk); only the last chunk can run past the row: clamped into it (the weights of columns >= S are zero)
  const int kb = 16 * (KCT - 1) + 8 * hk;
  const int kl0 = (kb < S4x4 ? kb : S4x4 - 4) - 8 * hk, kl1 = (kb + 4 < S4x4 ? kb + 4 : S4x4 - 4) - 8 * hk;
  auto kof = [&](int kc, int hf) { return kc < KCT - 1 ? 16 * kc + 4 * hf : (hf ? kl1 : kl0); };
  auto rowc_of = [&](long tl) { long row = tl * 32 + j; return row > a.rows - 1 ? a.rows - 1 : row; };
  auto srow_of = [&](long tl) -> const float* {
    const ConcatRow cr = concat_row(a.s, rowc_of(tl));
    return a.s.p0 + cr.r0 * a.s.ld0 + 8 * hk;
  };
  f32x4 st[PF][2];
  const float* srow = srow_of(tile);
#pragma unroll
  for (int kc = 0; kc < PF; ++kc) {
    st[kc][0] = *reinterpret_cast<const f32x4*>(srow + kof(kc, 0));
    st[kc][1] = *reinterpret_cast<const f32x4*>(srow + kof(kc, 1));
  }
  const float* srow_n = srow_of(tile + NW < t_end ? tile + NW : tile);
  ST_MARK(4);       // prologue (weights -> LDS, first loads)
  for (; tile < t_end; tile += NW) {
    const long t2 = tile + 2 * NW < t_end ? tile + 2 * NW : (tile + NW < t_end ? tile + NW : tile);
    const float* srow_n2 = srow_of(t2);            // (episode map: a dependent load - resolved a tile ahead of its use)
    const long rowc = rowc_of(tile);
    // q of this lane's row: floats 0-3 | 4-7 | 6-9 (three in-row 16-byte loads), consumed in the epilogue
    const float* qp = a.q + rowc * N;
    const f32x4 qA = *reinterpret_cast<const f32x4*>(qp), qB = *reinterpret_cast<const f32x4*>(qp + 4);
    const f32x4 qC = *reinterpret_cast<const f32x4*>(qp + 6);
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(Bl + (t * 2 + hk) * 16 + 4 * r4);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][4 * r4 + i] = b[i];
      }
    }
    ST_MARK(0);
#pragma unroll
    for (int kc = 0; kc < KCT; ++kc) {
      bf16x8_t b8;
      const bf16x4_t lo = __builtin_convertvector(st[kc % PF][0], bf16x4_t), hi = __builtin_convertvector(st[kc % PF][1], bf16x4_t);
#pragma unroll
      for (int i = 0; i < 4; ++i) { b8[i] = lo[i]; b8[4 + i] = hi[i]; }
      // the slot is free: chunk kc + PF goes into it - of this tile, or of the next one (unconditional, clamped)
      const float* nsrc = kc + PF < KCT ? srow : srow_n;
      const int nkc = kc + PF < KCT ? kc + PF : kc + PF - KCT;
      st[kc % PF][0] = *reinterpret_cast<const f32x4*>(nsrc + kof(nkc, 0));
      st[kc % PF][1] = *reinterpret_cast<const f32x4*>(nsrc + kof(nkc, 1));
      const float* wl = Wl + (kc * 64 + lane) * 4;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const bf16x8_t a8 = *reinterpret_cast<const bf16x8_t*>(wl + t * (KCT * 256));
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    ST_MARK(1);
    // ---- mixing, in registers: register u (first 16 columns of a tile) / 8 + u (second 16) = embedding unit u of this lane half
    float tot = 0.f;
    const float qv[N] = {qA[0], qA[1], qA[2], qA[3], qB[0], qB[1], qB[2], qB[3], qC[2], qC[3]};
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float pa = 0.f;
#pragma unroll
      for (int t = 0; t < 5; ++t) pa += qv[2 * t] * fabsf(acc[t][u]) + qv[2 * t + 1] * fabsf(acc[t][8 + u]);
      const float ae = pa + acc[5][u];
      const float ex = __expf(ae);
      const float hid = ae > 0.f ? ae : ex - 1.f;                         // elu, alpha = 1
      tot += hid * fabsf(acc[5][8 + u]) + fmaxf(acc[6][u], 0.f) * wb2v[u];
    }
    ST_MARK(2);
    tot += __shfl_xor(tot, 32, 64);                                       // the row's other eight units of this half
    if (hk == 0 && tile * 32 + j < a.rows) atomicAdd(a.q_tot + tile * 32 + j, tot + bb2);
    srow = srow_n;
    srow_n = srow_n2;
    ST_MARK(3);
  }
  ST_DUMP(5);
}

struct WideRedArgs {
  const float* ws; const float* slab2; int nslab, nwg, N, S, C, KT;
  float *dW[4], *dB[4], *dwb2, *dbb2;
  float* loss2;       // [sum (mask td)^2 | sum mask] accumulated into (LOSS variant) or null
};
constexpr int RSG = 8;            // slab groups per output element (fixed summation order -> deterministic)
__global__ __launch_bounds__(64 * RSG) void qmix_wide_reduce_kernel(WideRedArgs a) {
  __shared__ float part[RSG][64];
  const int Kx = 16 * a.KT + 1;
  const long n1 = (long)a.C * Kx;
  const int el = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const long e = (long)blockIdx.x * 64 + el;
  float s = 0.f;
  if (e < n1) s = slab_sum(a.ws + e, n1, sg, RSG, a.nslab);
  else if (e < n1 + E + 3) s = slab_sum(a.slab2 + (e - n1), E + 3, sg, RSG, a.nwg);
  part[sg][el] = s;
  __syncthreads();
  if (sg != 0) return;
  s = 0.f;
#pragma unroll
  for (int g = 0; g < RSG; ++g) s += part[g][el];
  if (e < n1) {
    const int col = (int)(e / Kx), k = (int)(e - (long)col * Kx);
    int seg, r;
    seg_of(col, a.N * E, seg, r);
    if (k < a.S) a.dW[seg][(long)r * a.S + k] += s;
    else if (k == 16 * a.KT) a.dB[seg][r] += s;
  } else if (e < n1 + E + 3) {
    const long t = e - n1;
    if (t < E) a.dwb2[t] += s;
    else if (t == E) a.dbb2[0] += s;
    else if (a.loss2) a.loss2[t - E - 1] += s;
  }
}

inline bool supported(int N, int S, int Eq) {
  const int C = N * E + 3 * E;
  return Eq == E && N >= 1 && N <= 16 && C <= 16 * NCTM && S >= 4 && S <= 384;       // S: k tiles of the weight-gradient GEMM (8 waves x 3)
}
inline ConcatSrc state_src(const marl_src_t* s) {
  ConcatSrc c;
  c.p0 = s->p0; c.ld0 = s->ld0; c.k0 = s->k0; c.p1 = nullptr; c.ld1 = 0; c.k1 = 0;
  c.idx = nullptr; c.nhot = 0; c.hot_w = 0; c.nid = 0; c.m0 = nullptr; c.ldm0 = 0;
  c.rpe0 = s->rpe0; c.bs0 = s->bs0; c.off0 = s->off0; c.rpei = 0; c.bsi = 0; c.offi = 0;
  c.fd0 = make_fastdiv((unsigned)(s->rpe0 > 0 ? s->rpe0 : 1));
  c.fdi = make_fastdiv(1); c.fdn = make_fastdiv(1);
  c.emap0 = s->emap0;
  return c;
}
inline bool src_ok(const marl_src_t* s, int S) {
  // dense segment 0 only; every row starts on a 16-byte boundary and holds (S rounded up to 4) readable floats
  return s->p0 && s->k0 == S && !s->k1 && !s->nhot && !s->nid && !s->m0 && (s->ld0 % 4 == 0) && s->ld0 >= (S + 3) / 4 * 4 &&
         ((reinterpret_cast<uintptr_t>(s->p0) & 15) == 0);
}
inline int kc_of(int S, bool bf) { return bf ? (S + 31) / 32 : (S + 15) / 16; }
inline size_t packed_floats(int N, int S) {                  // room for either packing (+ the bias vector)
  const int C = N * E + 3 * E, NCT = (C + 15) / 16;
  return (size_t)NCT * ((S + 15) / 16) * 256 + (size_t)((C + 3) / 4 * 4);
}
inline int wg_groups(int C) { return ((C + 15) / 16 + WNT - 1) / WNT; }
inline int wg_slabs(long rows) { long n = (rows + 8 * WCH - 1) / (8 * WCH); if (n > 64) n = 64; return (int)(n < 1 ? 1 : n); }
inline unsigned grid_for(long rows) {
  const long nblk = (rows + RB - 1) / RB;
  return (unsigned)(nblk < 256 ? nblk : 256);
}

int pack(const marl_qmix_weights_t* w, int N, int S, bool bf, float* ws, hipStream_t st, WideArgs& a) {
  PackArgs p;
  p.W[0] = w->w1; p.Bv[0] = w->w1_b; p.W[1] = w->b1; p.Bv[1] = w->b1_b; p.W[2] = w->w2; p.Bv[2] = w->w2_b;
  p.W[3] = w->h; p.Bv[3] = w->h_b;
  p.N = N; p.S = S; p.C = N * E + 3 * E; p.NCT = (p.C + 15) / 16; p.KC = kc_of(S, bf); p.bf = bf ? 1 : 0;
  p.Wp = ws; p.Bc = ws + (size_t)p.NCT * ((S + 15) / 16) * 256;
  const long total = (long)p.NCT * p.KC * 64;
  hipLaunchKernelGGL(qmix_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p);
  MARL_CHECK_LAUNCH();
  a.Wp = p.Wp; a.Bc = p.Bc; a.wb2 = w->b2_w; a.bb2 = w->b2_b;
  a.N = N; a.S = S; a.C = p.C; a.NCT = p.NCT; a.KC = p.KC;
  return 0;
}

template <typename K>
int launch_main(K fn, const WideArgs& a, unsigned grid, bool bf, hipStream_t st) {
  (void)bf;
  const size_t lds = wide_lds(a.NCT);
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  void* kargs[] = {(void*)&a};
  e = hipLaunchKernel((const void*)fn, dim3(grid), dim3(64 * NW), kargs, lds, st);
  if (e != hipSuccess) return (int)e;
  MARL_CHECK_LAUNCH();
  return 0;
}

}  // namespace

ST_DEFINE_SETTER(marl_debug_stamps_wide)

extern "C" int marl_qmix_wide_supported(int N, int S, int Eq) { return supported(N, S, Eq) ? 1 : 0; }

extern "C" size_t marl_qmix_wide_workspace(long rows, int N, int S, int backward) {
  const int C = N * E + 3 * E;
  size_t f = packed_floats(N, S);
  if (backward) {
    const int KT = (S + 15) / 16;
    f += (size_t)rows * C + (size_t)wg_slabs(rows) * C * (16 * KT + 1) + (size_t)256 * (E + 3) + 64;
  }
  return f * sizeof(float);
}

// which forward kernel marl_qmix_wide_fwd launches for this shape (the prefix of its name in a rocprofv3 kernel trace), so that a
// caller that times the call can name the kernel it timed (bench.py); the same dispatch rules as below
extern "C" const char* marl_qmix_wide_fwd_kernel(long rows, int N, int S, int flags) {
  const bool bf = (flags & 1) != 0;
  const bool res_off = !marl_switches()->wide_res, res32_on = marl_switches()->wide_res32 == 1;
  if (bf && N == 10 && (S + 15) / 16 == 21 && rows >= 128L * 16 * 16 && !res_off && res32_on) return "qmix_wide_res32_fwd_kernel";
  if (bf && N == 10 && (S + 31) / 32 == 11 && rows >= 128L * 16 * 16 && !res_off) return "qmix_wide_res_fwd_kernel";
  return "qmix_wide_kernel<false";
}

extern "C" int marl_qmix_wide_fwd(const marl_qmix_weights_t* w, const marl_src_t* s, const float* q, float* q_tot,
                                  float* ws, size_t ws_bytes, long rows, int N, int S, int Eq, int flags, void* stream) {
  if (rows <= 0) return 0;
  if (!supported(N, S, Eq) || !src_ok(s, S)) return (int)hipErrorInvalidValue;
  if (ws_bytes < marl_qmix_wide_workspace(rows, N, S, 0) || (reinterpret_cast<uintptr_t>(ws) & 15)) return (int)hipErrorInvalidValue;
  const bool bf = (flags & 1) != 0;
  hipStream_t st = (hipStream_t)stream;
  // bf16, 10 agents, 21 / 11 k-chunks (MMM2) and enough row tiles for 128 row groups: the resident-weights kernels
  const bool res_off = !marl_switches()->wide_res;      // A/B switches (common.h: MarlSwitches)
  // (the 32-row-tile variant is opt-in: measured 65 us against 60 us for the 16-row one - both run their chunk loops with the
  // bf16 pipe ~75 % busy, the 32 x 32 tiles pay 8 % padding and a longer dependent chain per tile)
  const bool res32_on = marl_switches()->wide_res32 == 1;
  if (bf && N == 10 && (S + 15) / 16 == 21 && rows >= 128L * 16 * 16 && !res_off && res32_on) {
    Wide32Args b = {};
    b.W[0] = w->w1; b.Bv[0] = w->w1_b; b.W[1] = w->b1; b.Bv[1] = w->b1_b; b.W[2] = w->w2; b.Bv[2] = w->w2_b; b.W[3] = w->h; b.Bv[3] = w->h_b;
    b.wb2 = w->b2_w; b.bb2 = w->b2_b; b.s = state_src(s); b.q = q; b.q_tot = q_tot; b.rows = rows; b.S = S;
    b.Wp = ws;                                   // 2 x 7 x 21 KB = 294 KB of the workspace (packed_floats covers 26 x 21 KB)
    const size_t lds = (size_t)(7 * 21 * 256 + 7 * 32) * 4;
    hipLaunchKernelGGL(qmix_pack32_kernel, dim3(74), dim3(256), 0, st, b, 21);
    MARL_CHECK_LAUNCH();
    hipError_t e = hipMemsetAsync(q_tot, 0, (size_t)rows * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)qmix_wide_res32_fwd_kernel<21, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    void* kargs[] = {(void*)&b};
    e = hipLaunchKernel((const void*)qmix_wide_res32_fwd_kernel<21, 7>, dim3(256), dim3(64 * NW), kargs, lds, st);
    if (e != hipSuccess) return (int)e;
    MARL_CHECK_LAUNCH();
    return 0;
  }
  WideArgs a = {};
  int rc = pack(w, N, S, bf, ws, st, a);
  if (rc) return rc;
  a.s = state_src(s); a.q = q; a.q_tot = q_tot; a.rows = rows;
  const unsigned grid = grid_for(rows);
  if (bf && N == 10 && a.KC == 11 && rows >= 128L * 16 * 16 && !res_off) {
    const size_t lds = (size_t)(13 * 11 * 256 + NW * 256 + 13 * 16) * 4;
    hipError_t e = hipMemsetAsync(q_tot, 0, (size_t)rows * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)qmix_wide_res_fwd_kernel<13, 11>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    void* kargs[] = {(void*)&a};
    e = hipLaunchKernel((const void*)qmix_wide_res_fwd_kernel<13, 11>, dim3(256), dim3(64 * NW), kargs, lds, st);
    if (e != hipSuccess) return (int)e;
    MARL_CHECK_LAUNCH();
    return 0;
  }
  if (a.NCT == NCTM) return bf ? launch_main(qmix_wide_kernel<false, true, NCTM>, a, grid, true, st)
                               : launch_main(qmix_wide_kernel<false, false, NCTM>, a, grid, false, st);
  return bf ? launch_main(qmix_wide_kernel<false, true, 0>, a, grid, true, st)
            : launch_main(qmix_wide_kernel<false, false, 0>, a, grid, false, st);
}

struct WideLoss { const float *q_tot_tgt, *r, *term, *padded; float gamma; float* q_tot; float* loss2; };

static int wide_bwd_impl(const marl_qmix_weights_t* w, const marl_src_t* s, const float* q, const float* dq_tot, const WideLoss* L,
                         float* dq, const marl_qmix_weights_t* grads, float* ws, size_t ws_bytes, long rows,
                         int N, int S, int Eq, int flags, void* stream) {
  if (rows <= 0) return 0;
  if (!supported(N, S, Eq) || !src_ok(s, S)) return (int)hipErrorInvalidValue;
  if (ws_bytes < marl_qmix_wide_workspace(rows, N, S, 1) || (reinterpret_cast<uintptr_t>(ws) & 15)) return (int)hipErrorInvalidValue;
  const bool bf = (flags & 1) != 0;
  hipStream_t st = (hipStream_t)stream;
  WideArgs a = {};
  int rc = pack(w, N, S, bf, ws, st, a);
  if (rc) return rc;
  const int C = a.C, KT = (S + 15) / 16;
  float* dhy = ws + (packed_floats(N, S) + 15) / 16 * 16;
  const int nslab = wg_slabs(rows);
  float* wslab = dhy + (size_t)rows * C;
  float* bslab = wslab + (size_t)nslab * C * (16 * KT + 1);
  a.s = state_src(s); a.q = q; a.g = dq_tot; a.dq = dq; a.dhy = dhy; a.slab = bslab; a.rows = rows;
  if (L) { a.lr = L->r; a.lterm = L->term; a.lpadded = L->padded; a.lq_tgt = L->q_tot_tgt; a.gamma = L->gamma; a.q_tot = L->q_tot; }
  const unsigned grid = grid_for(rows);
  if (L) {
    if (a.NCT == NCTM) rc = bf ? launch_main(qmix_wide_kernel<true, true, NCTM, true>, a, grid, true, st)
                               : launch_main(qmix_wide_kernel<true, false, NCTM, true>, a, grid, false, st);
    else rc = bf ? launch_main(qmix_wide_kernel<true, true, 0, true>, a, grid, true, st)
                 : launch_main(qmix_wide_kernel<true, false, 0, true>, a, grid, false, st);
  } else {
    if (a.NCT == NCTM) rc = bf ? launch_main(qmix_wide_kernel<true, true, NCTM>, a, grid, true, st)
                               : launch_main(qmix_wide_kernel<true, false, NCTM>, a, grid, false, st);
    else rc = bf ? launch_main(qmix_wide_kernel<true, true, 0>, a, grid, true, st)
                 : launch_main(qmix_wide_kernel<true, false, 0>, a, grid, false, st);
  }
  if (rc) return rc;
  WideWgArgs g;
  g.dhy = dhy; g.s = a.s; g.ws = wslab; g.rows = rows; g.S = S; g.C = C; g.KT = KT; g.nslab = nslab;
  if (bf && (flags & 2)) {      // bf16 weight-gradient operands (own flag bit): one k-step of the bf16 MFMA per 32-row chunk, transposed LDS reads
    size_t lds = (size_t)2 * WCH * (wgb_gb() + wgb_xb(KT)) + 2 * WCH * sizeof(long);
    const size_t red = (size_t)WCH * 16 * WNT * sizeof(float);
    if (lds < red) lds = red;
    hipError_t e = hipFuncSetAttribute((const void*)qmix_wide_wgrad_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(qmix_wide_wgrad_bf16_kernel, dim3(nslab, wg_groups(C)), dim3(64 * NW), lds, st, g);
    MARL_CHECK_LAUNCH();
  } else {
    const size_t lds = (size_t)2 * WCH * (wg_gp() + wg_xp(KT)) * sizeof(float) + 2 * WCH * sizeof(long);
    hipError_t e = hipFuncSetAttribute((const void*)qmix_wide_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(qmix_wide_wgrad_kernel, dim3(nslab, wg_groups(C)), dim3(64 * NW), lds, st, g);
    MARL_CHECK_LAUNCH();
  }
  WideRedArgs r;
  r.ws = wslab; r.slab2 = bslab; r.nslab = nslab; r.nwg = (int)grid; r.N = N; r.S = S; r.C = C; r.KT = KT;
  r.loss2 = L ? L->loss2 : nullptr;
  r.dW[0] = const_cast<float*>(grads->w1); r.dB[0] = const_cast<float*>(grads->w1_b);
  r.dW[1] = const_cast<float*>(grads->b1); r.dB[1] = const_cast<float*>(grads->b1_b);
  r.dW[2] = const_cast<float*>(grads->w2); r.dB[2] = const_cast<float*>(grads->w2_b);
  r.dW[3] = const_cast<float*>(grads->h); r.dB[3] = const_cast<float*>(grads->h_b);
  r.dwb2 = const_cast<float*>(grads->b2_w); r.dbb2 = const_cast<float*>(grads->b2_b);
  const long total = (long)C * (16 * KT + 1) + E + 3;
  hipLaunchKernelGGL(qmix_wide_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64 * RSG), 0, st, r);
  MARL_CHECK_LAUNCH();
  return 0;
}

extern "C" int marl_qmix_wide_bwd(const marl_qmix_weights_t* w, const marl_src_t* s, const float* q, const float* dq_tot,
                                  float* dq, const marl_qmix_weights_t* grads, float* ws, size_t ws_bytes, long rows,
                                  int N, int S, int Eq, int flags, void* stream) {
  return wide_bwd_impl(w, s, q, dq_tot, nullptr, dq, grads, ws, ws_bytes, rows, N, S, Eq, flags, stream);
}

extern "C" int marl_qmix_wide_loss_bwd(const marl_qmix_weights_t* w, const marl_src_t* s, const float* q, const float* q_tot_tgt,
                                       const float* r, const float* term, const float* padded, float gamma, float* q_tot,
                                       float* dq, const marl_qmix_weights_t* grads, float* loss2, float* ws, size_t ws_bytes,
                                       long rows, int N, int S, int Eq, int flags, void* stream) {
  if (!q_tot_tgt || !r || !term || !padded || !loss2) return (int)hipErrorInvalidValue;
  WideLoss L = {q_tot_tgt, r, term, padded, gamma, q_tot, loss2};
  return wide_bwd_impl(w, s, q, nullptr, &L, dq, grads, ws, ws_bytes, rows, N, S, Eq, flags, stream);
}
