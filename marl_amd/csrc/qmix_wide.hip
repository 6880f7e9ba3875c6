// Fused QMIX mixer for WIDE states (reference network/mixer.py:57-80 at S = 322, N = 10: MMM2, BASELINE config 5).
//
// qmix_fused.hip keeps the hypernet weights in registers, which stops at S <= 128 / 256 output columns; here the
// concatenated hypernet [ w1 (N*E) | b1 (E) | w2 (E) | h = hyper_b2.0 (E) ] x S is 416 x 322 = 536 KB.  It is packed
// once per call into MFMA-fragment order (1 KB per (column tile, k-chunk), L2 resident) and STREAMED: a workgroup owns
// a block of RB = 64 (episode, step) rows - the state tile sits in LDS - and wave w owns column tiles w, w+8, w+16
// (, w+24): per k-chunk it reads its own fragments straight from L2 (one chunk ahead, two named register sets) and
// multiplies them into RB/16 row tiles, so every weight byte read from L2 feeds 64 rows.  The 416-wide hypernet
// output never reaches HBM in the forward pass; the mixing arithmetic runs on the accumulators as in qmix_fused.hip.
//   forward : q_tot = sum_e elu(sum_n q_n |w1[n,e]| + b1_e) |w2_e| + (relu(h) . w_b2 + b_b2)
//   backward: recomputes the tile, forms d(hypernet output) in accumulator layout and writes it (rows x C) for the
//             weight-gradient GEMM below; dq and the hyper_b2.2 gradients come out of the same pass.
//   wgrad   : dW[C][S] += dhy^T s over all rows - a tall-skinny GEMM: 2 column groups x 128 row slabs, operands
//             staged row-major in LDS (double buffered), MFMA operands are plain 32-bit LDS reads (k = row), each wave
//             keeps (all column tiles of the group) x (its k tiles) accumulators in registers; slabs + fixed-order
//             reduce scatter into the four weight / bias gradients.
// BF = true: the hypernet GEMM takes bf16 operands (state tile rounded once when it is written to LDS, weights packed
// as bf16) on v_mfma_f32_16x16x32_bf16 with fp32 accumulation - "bf16 mixer with MFMA"; then the kernel is bound by
// reading the states from HBM.  Everything else (mixing, gradients, the weight-gradient GEMM) stays fp32.
#include "common.h"
#include "../../include/marl_hip.h"

namespace {

constexpr int E = 32;
constexpr int NW = 8;             // waves per workgroup
constexpr int NTW = 4;            // column tiles per wave (upper bound: C <= 512)
constexpr int RTMAX = 4;          // row tiles per block (upper bound; the kernel is instantiated for 2 and 4)

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
  float* slab;                    // [grid][E + 1] hyper_b2.2 gradient partials
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
__device__ __forceinline__ float sum32(float v) { v = sum16(v); v += __shfl_xor(v, 16, 64); return v; }

__host__ __device__ inline int ss_f32(int KC) { return 16 * KC + 4; }          // LDS row pitch of the fp32 state tile (floats)
__host__ __device__ inline int ss_bf16(int KC) { return 32 * KC + 8; }         // ... of the bf16 tile (bf16 elements)
__host__ __device__ inline size_t tile_bytes(int KC, bool bf, int RB) {
  const size_t st = bf ? (size_t)RB * ss_bf16(KC) * 2 : (size_t)RB * ss_f32(KC) * 4;
  const size_t pa = (size_t)NW * RB * E * 4;                                     // scratch overlaying the state tile
  return st > pa ? st : pa;
}
__host__ __device__ inline size_t wide_lds(int KC, bool bf, int RB) {
  // state tile / PA | W2A | HBA | DPRE | HID | Qs [RB][16] | DQH [RB][16][2] | Gs [RB] | red [NW][E + 1]
  return tile_bytes(KC, bf, RB) + (size_t)(4 * RB * E + RB * 16 + RB * 32 + RB + NW * (E + 1)) * 4;
}

template <bool BWD, bool BF, int RT>
__global__ __launch_bounds__(64 * NW, 2) void qmix_wide_kernel(WideArgs a) {
  constexpr int RB = 16 * RT;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q4 = lane >> 4, m = lane & 15;
  const int N = a.N, S = a.S, C = a.C, NE = N * E, KC = a.KC;
  float* St = reinterpret_cast<float*>(smem_raw);                       // fp32 state tile [RB][SSF]
  __bf16* Sb = reinterpret_cast<__bf16*>(smem_raw);                     // bf16 state tile [RB][SSB]
  float* PA = reinterpret_cast<float*>(smem_raw);                       // [NW][RB][E]  (after the GEMM)
  float* W2A = reinterpret_cast<float*>(smem_raw + tile_bytes(KC, BF, RB)); // [RB][E] |w2|
  float* HBA = W2A + RB * E;                                            // [RB][E] relu(h)
  float* DPRE = HBA + RB * E;                                           // [RB][E] dL/da_e
  float* HID = DPRE + RB * E;                                           // [RB][E] elu(a_e)
  float* Qs = HID + RB * E;                                             // [RB][16]
  float* DQH = Qs + RB * 16;                                            // [RB][16][2] halves of dq
  float* Gs = DQH + RB * 32;                                            // [RB]
  float* red = Gs + RB;                                                 // [NW][E + 1]
  const int SSF = ss_f32(KC), SSB = ss_bf16(KC);

  // ---- this wave's column tiles: ct = wave + 8 j
  int kind[NTW], nn[NTW], eh[NTW];
  float bias[NTW], wb2c[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int ct = wave + NW * j, col0 = 16 * ct;
    int k = -1;
    if (ct < a.NCT) {
      if (col0 < NE) k = 0;
      else if (col0 < NE + E) k = 1;
      else if (col0 < NE + 2 * E) k = 2;
      else if (col0 < C) k = 3;
    }
    kind[j] = k; nn[j] = k == 0 ? col0 / E : 0; eh[j] = ct & 1;
    bias[j] = k >= 0 ? a.Bc[col0 + m] : 0.f;
    wb2c[j] = a.wb2[16 * eh[j] + m];
  }
  const float wb2_l = a.wb2[lane & 31];
  const float bb2 = a.bb2[0];
  float acc_wb2 = 0.f, acc_bb2 = 0.f;

  // ---- state block staging: thread -> (row, float4 column) items over the WHOLE row pitch of the LDS tile: columns below
  // S come from HBM (one block ahead, in registers), the k-padding columns are rewritten as zeros with every block (the
  // epilogue scratch overlays the tile).  Rows past the end of the batch are clamped to the last row.
  const int S4 = (S + 3) >> 2;
  const int W4 = (BF ? SSB : SSF) >> 2;      // float4 groups per tile row
  const int items = RB * W4;
  constexpr int NPF = 3 * RT;                // prefetch registers (float4) per thread: covers S <= 384
  f32x4 pf[NPF];
  const float invW4 = 1.0f / (float)W4;
  const long nblk = (a.rows + RB - 1) / RB;
  auto fetch = [&](long blk) {
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      int e = tid + 64 * NW * i;
      if (e > items - 1) e = items - 1;
      const int r = (int)(((float)e + 0.5f) * invW4);
      int g4 = e - r * W4;
      if (g4 > S4 - 1) g4 = S4 - 1;                        // pad groups re-read the last real one (value unused)
      long row = blk * RB + r;
      if (row > a.rows - 1) row = a.rows - 1;
      const ConcatRow cr = concat_row(a.s, row);
      pf[i] = *reinterpret_cast<const f32x4*>(a.s.p0 + cr.r0 * a.s.ld0 + 4 * g4);   // rows are padded to 16 bytes (host check)
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      const int e = tid + 64 * NW * i;
      if (e < items) {
        const int r = (int)(((float)e + 0.5f) * invW4);
        const int c4 = (e - r * W4) * 4;
        f32x4 v = pf[i];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) if (c4 + cc >= S) v[cc] = 0.f;      // row padding and k padding
        if (BF) {
          const bf16x4_t b = __builtin_convertvector(v, bf16x4_t);
          *reinterpret_cast<bf16x4_t*>(Sb + r * SSB + c4) = b;
        } else {
          *reinterpret_cast<f32x4*>(St + r * SSF + c4) = v;
        }
      }
    }
  };
  // q / g of the block: thread -> up to two q elements, threads 0..RB-1 one g element
  float pq[2] = {0.f, 0.f}, pg = 0.f;
  auto fetch_qg = [&](long blk) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + 64 * NW * i;
      pq[i] = 0.f;
      if (e < RB * N) {
        const long row = blk * RB + e / N;
        if (row < a.rows) pq[i] = a.q[row * N + e % N];
      }
    }
    if (BWD) {
      pg = 0.f;
      const long row = blk * RB + tid;
      if (tid < RB && row < a.rows) pg = a.g[row];
    }
  };
  auto stash_qg = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + 64 * NW * i;
      if (e < RB * N) Qs[(e / N) * 16 + e % N] = pq[i];
    }
    if (BWD && tid < RB) Gs[tid] = pg;
  };

  long blk = blockIdx.x;
  if (blk < nblk) { fetch(blk); fetch_qg(blk); stash(); stash_qg(); }

  for (; blk < nblk; blk += gridDim.x) {
    const long row0 = blk * RB;
    WG_BARRIER();                                        // (A) state tile, Qs, Gs of this block are in LDS
    const long nb = blk + gridDim.x;
    if (nb < nblk) { fetch(nb); fetch_qg(nb); }           // next block's loads fly during the GEMM
    // ---- hypernet GEMM: acc[rt][j] = out[rows 16rt + 4q + i][cols 16ct_j + m]
    f32x4 acc[RT][NTW];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[rt][j] = (f32x4){bias[j], bias[j], bias[j], bias[j]};
    if (BF) {
      const bf16x8_t* Wp8 = reinterpret_cast<const bf16x8_t*>(a.Wp);
      auto wload = [&](bf16x8_t (&w)[NTW], int kc) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          const int ct = wave + NW * j;
          w[j] = Wp8[((long)(ct < a.NCT ? ct : 0) * KC + kc) * 64 + lane];
        }
      };
      auto mac = [&](const bf16x8_t (&w)[NTW], int kc) __attribute__((always_inline)) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const bf16x8_t av = *reinterpret_cast<const bf16x8_t*>(Sb + (16 * rt + m) * SSB + 32 * kc + 8 * q4);
#pragma unroll
          for (int j = 0; j < NTW; ++j)
            if (kind[j] >= 0) acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, w[j], acc[rt][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      bf16x8_t wA[NTW], wB[NTW];
      wload(wA, 0);
      for (int kc = 0; kc < KC; kc += 2) {
        wload(wB, kc + 1 < KC ? kc + 1 : KC - 1);
        mac(wA, kc);
        if (kc + 1 < KC) {
          wload(wA, kc + 2 < KC ? kc + 2 : KC - 1);
          mac(wB, kc + 1);
        }
      }
    } else {
      const f32x4* Wp4 = reinterpret_cast<const f32x4*>(a.Wp);
      auto wload = [&](f32x4 (&w)[NTW], int kc) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          const int ct = wave + NW * j;
          w[j] = Wp4[((long)(ct < a.NCT ? ct : 0) * KC + kc) * 64 + lane];
        }
      };
      auto mac = [&](const f32x4 (&w)[NTW], int kc) __attribute__((always_inline)) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const f32x4 av = *reinterpret_cast<const f32x4*>(St + (16 * rt + m) * SSF + 16 * kc + 4 * q4);
#pragma unroll
          for (int j = 0; j < NTW; ++j)
            if (kind[j] >= 0) acc[rt][j] = mfma16x4(av, w[j], acc[rt][j]);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      f32x4 wA[NTW], wB[NTW];
      wload(wA, 0);
      for (int kc = 0; kc < KC; kc += 2) {
        wload(wB, kc + 1 < KC ? kc + 1 : KC - 1);
        mac(wA, kc);
        if (kc + 1 < KC) {
          wload(wA, kc + 2 < KC ? kc + 2 : KC - 1);
          mac(wB, kc + 1);
        }
      }
    }
    WG_BARRIER();                                        // (B) every wave is done with the state tile: PA may overlay it
    // ---- partial pre-activations a_e = b1_e + sum_n q_n |w1[n,e]| of this wave's tiles; |w2|, relu(h) to LDS
    {
      float pa[2][RT][4];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int i = 0; i < 4; ++i) pa[h][rt][i] = 0.f;
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 16 * rt + 4 * q4 + i;
            const float v = acc[rt][j][i];
            if (kind[j] == 0) pa[eh[j]][rt][i] += Qs[r * 16 + nn[j]] * fabsf(v);
            else if (kind[j] == 1) pa[eh[j]][rt][i] += v;
            else if (kind[j] == 2) W2A[r * E + 16 * eh[j] + m] = fabsf(v);
            else if (kind[j] == 3) HBA[r * E + 16 * eh[j] + m] = fmaxf(v, 0.f);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int i = 0; i < 4; ++i) PA[(wave * RB + 16 * rt + 4 * q4 + i) * E + 16 * h + m] = pa[h][rt][i];
    }
    WG_BARRIER();                                        // (C)
    // ---- finish: wave w takes rows [8w, 8w + 8), two rows per pass (lane = e of its half)
#pragma unroll
    for (int p = 0; p < RB / NW / 2; ++p) {
      const int r = (RB / NW) * wave + 2 * p + (lane >> 5), e = lane & 31;
      float ae = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) ae += PA[(w * RB + r) * E + e];
      const float ex = __expf(ae);
      const float hid = ae > 0.f ? ae : ex - 1.f;                 // elu, alpha = 1
      const float w2 = W2A[r * E + e], hb = HBA[r * E + e];
      const float tot = sum32(hid * w2 + hb * wb2_l);
      if (!BWD) {
        if (e == 0 && row0 + r < a.rows) a.q_tot[row0 + r] = tot + bb2;
      } else {
        const float gr = Gs[r];
        DPRE[r * E + e] = gr * w2 * (ae > 0.f ? 1.f : ex);
        HID[r * E + e] = hid;
        acc_wb2 += gr * hb;
        if (e == 0) acc_bb2 += gr;
      }
    }
    if (BWD) {
      WG_BARRIER();                                      // (D)
      // ---- d(hypernet output) in accumulator layout -> HBM; halves of dq -> LDS
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        if (kind[j] < 0) continue;
        const int ecol = 16 * eh[j] + m, col = 16 * (wave + NW * j) + m;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 16 * rt + 4 * q4 + i;
            const float o = acc[rt][j][i];
            float v = 0.f;
            if (kind[j] == 0) {
              const float dp = DPRE[r * E + ecol];
              v = Qs[r * 16 + nn[j]] * dp * sgn(o);
              const float hsum = sum16(fabsf(o) * dp);                   // this tile's half of dq_n
              if (m == 0) DQH[(r * 16 + nn[j]) * 2 + eh[j]] = hsum;
            } else if (kind[j] == 1) v = DPRE[r * E + ecol];
            else if (kind[j] == 2) v = Gs[r] * HID[r * E + ecol] * sgn(o);
            else v = o > 0.f ? Gs[r] * wb2c[j] : 0.f;
            if (row0 + r < a.rows) a.dhy[(row0 + r) * C + col] = v;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      WG_BARRIER();                                      // (D2) both halves of every dq_n are in LDS
      for (int e = tid; e < RB * N; e += 64 * NW) {
        const int r = e / N, n = e - r * N;
        if (row0 + r < a.rows) a.dq[(row0 + r) * N + n] = DQH[(r * 16 + n) * 2] + DQH[(r * 16 + n) * 2 + 1];
      }
    }
    WG_BARRIER();                                        // (E) scratch is free: the next block's tile may land
    if (nb < nblk) { stash(); stash_qg(); }
  }
  if (BWD) {
    // hyper_b2.2 gradient partials: lanes e of both halves, then the waves in fixed order
    const float v = acc_wb2 + __shfl_xor(acc_wb2, 32, 64);
    const float b = acc_bb2 + __shfl_xor(acc_bb2, 32, 64);
    __syncthreads();
    if (lane < 32) red[wave * (E + 1) + lane] = v;
    if (lane == 0) red[wave * (E + 1) + E] = b;
    __syncthreads();
    if (tid < E + 1) {
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) tot += red[w * (E + 1) + tid];
      a.slab[(long)blockIdx.x * (E + 1) + tid] = tot;
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
  float* Gb[2] = {smem, smem + WCH * GP + WCH * XP};
  float* Xb[2] = {smem + WCH * GP, smem + 2 * WCH * GP + WCH * XP};
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
      long row = rb + r;
      if (row > a.rows - 1) row = a.rows - 1;
      const ConcatRow cr = concat_row(a.s, row);
      px[i] = *reinterpret_cast<const f32x4*>(a.s.p0 + cr.r0 * a.s.ld0 + c4);
    }
  };
  auto stash = [&](int b) {
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int e = tid + 64 * NW * i;
      if (e < gi) { const int r = e / G4, c4 = (e - r * G4) * 4; *reinterpret_cast<f32x4*>(Gb[b] + r * GP + c4) = pg[i]; }
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
        *reinterpret_cast<f32x4*>(Xb[b] + r * XP + c4) = v;
      }
    }
  };
  // zero the never-written pad columns of both buffers (k padding of the state chunk)
  for (int b = 0; b < 2; ++b)
    for (int e = tid; e < WCH * (XP - 4 * S4); e += 64 * NW) Xb[b][(e / (XP - 4 * S4)) * XP + 4 * S4 + e % (XP - 4 * S4)] = 0.f;
  f32x4 acc[WNT][WKT];
  float bs[WNT];
#pragma unroll
  for (int t = 0; t < WNT; ++t) {
    bs[t] = 0.f;
#pragma unroll
    for (int k = 0; k < WKT; ++k) acc[t][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  if (nch > 0) fetch(0);
  __syncthreads();
  if (nch > 0) stash(0);
  for (long ch = 0; ch < nch; ++ch) {
    const int b = (int)(ch & 1);
    WG_BARRIER();                                  // chunk ch is in buffer b; buffer b^1 is free (read two chunks ago)
    if (ch + 1 < nch) fetch(ch + 1);
    const float* G = Gb[b];
    const float* X = Xb[b];
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
        for (int k = 0; k < WKT; ++k)
          if (wave + NW * k < a.KT) acc[t][k] = mfma16(gv[t], xv[k], acc[t][k]);
      }
      __builtin_amdgcn_sched_barrier(0);
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

struct WideRedArgs {
  const float* ws; const float* slab2; int nslab, nwg, N, S, C, KT;
  float *dW[4], *dB[4], *dwb2, *dbb2;
};
constexpr int RSG = 8;            // slab groups per output element (fixed summation order -> deterministic)
__global__ __launch_bounds__(64 * RSG) void qmix_wide_reduce_kernel(WideRedArgs a) {
  __shared__ float part[RSG][64];
  const int Kx = 16 * a.KT + 1;
  const long n1 = (long)a.C * Kx;
  const int el = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const long e = (long)blockIdx.x * 64 + el;
  float s = 0.f;
  if (e < n1) {
    for (int w = sg; w < a.nslab; w += RSG) s += a.ws[(long)w * n1 + e];
  } else if (e < n1 + E + 1) {
    for (int w = sg; w < a.nwg; w += RSG) s += a.slab2[(long)w * (E + 1) + (e - n1)];
  }
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
  } else if (e < n1 + E + 1) {
    const long t = e - n1;
    if (t < E) a.dwb2[t] += s; else a.dbb2[0] += s;
  }
}

inline bool supported(int N, int S, int Eq) {
  const int C = N * E + 3 * E;
  return Eq == E && N >= 1 && N <= 16 && C <= 16 * NW * NTW && N <= 10 && S >= 4 && S <= 352;   // S: prefetch registers of the state tile
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
inline int rt_of(bool bf) { (void)bf; return 2; }      // row tiles per block: the bf16 GEMM is ~16x shorter per row, so it needs
                                                      // bigger blocks to keep the L2 weight stream below the HBM state stream
inline unsigned grid_for(long rows, int RB) {
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
  const size_t lds = wide_lds(a.KC, bf, 16 * rt_of(bf));
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

extern "C" int marl_qmix_wide_supported(int N, int S, int Eq) { return supported(N, S, Eq) ? 1 : 0; }

extern "C" size_t marl_qmix_wide_workspace(long rows, int N, int S, int backward) {
  const int C = N * E + 3 * E;
  size_t f = packed_floats(N, S);
  if (backward) {
    const int KT = (S + 15) / 16;
    f += (size_t)rows * C + (size_t)wg_slabs(rows) * C * (16 * KT + 1) + (size_t)256 * (E + 1) + 64;
  }
  return f * sizeof(float);
}

extern "C" int marl_qmix_wide_fwd(const marl_qmix_weights_t* w, const marl_src_t* s, const float* q, float* q_tot,
                                  float* ws, size_t ws_bytes, long rows, int N, int S, int Eq, int flags, void* stream) {
  if (rows <= 0) return 0;
  if (!supported(N, S, Eq) || !src_ok(s, S)) return (int)hipErrorInvalidValue;
  if (ws_bytes < marl_qmix_wide_workspace(rows, N, S, 0) || (reinterpret_cast<uintptr_t>(ws) & 15)) return (int)hipErrorInvalidValue;
  const bool bf = (flags & 1) != 0;
  hipStream_t st = (hipStream_t)stream;
  WideArgs a = {};
  int rc = pack(w, N, S, bf, ws, st, a);
  if (rc) return rc;
  a.s = state_src(s); a.q = q; a.q_tot = q_tot; a.rows = rows;
  const unsigned grid = grid_for(rows, 16 * rt_of(bf));
  return bf ? launch_main(qmix_wide_kernel<false, true, 2>, a, grid, true, st)
            : launch_main(qmix_wide_kernel<false, false, 2>, a, grid, false, st);
}

extern "C" int marl_qmix_wide_bwd(const marl_qmix_weights_t* w, const marl_src_t* s, const float* q, const float* dq_tot,
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
  const unsigned grid = grid_for(rows, 16 * rt_of(bf));
  rc = bf ? launch_main(qmix_wide_kernel<true, true, 2>, a, grid, true, st)
          : launch_main(qmix_wide_kernel<true, false, 2>, a, grid, false, st);
  if (rc) return rc;
  WideWgArgs g;
  g.dhy = dhy; g.s = a.s; g.ws = wslab; g.rows = rows; g.S = S; g.C = C; g.KT = KT; g.nslab = nslab;
  const size_t lds = (size_t)2 * WCH * (wg_gp() + wg_xp(KT)) * sizeof(float);
  hipError_t e = hipFuncSetAttribute((const void*)qmix_wide_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(qmix_wide_wgrad_kernel, dim3(nslab, wg_groups(C)), dim3(64 * NW), lds, st, g);
  MARL_CHECK_LAUNCH();
  WideRedArgs r;
  r.ws = wslab; r.slab2 = bslab; r.nslab = nslab; r.nwg = (int)grid; r.N = N; r.S = S; r.C = C; r.KT = KT;
  r.dW[0] = const_cast<float*>(grads->w1); r.dB[0] = const_cast<float*>(grads->w1_b);
  r.dW[1] = const_cast<float*>(grads->b1); r.dB[1] = const_cast<float*>(grads->b1_b);
  r.dW[2] = const_cast<float*>(grads->w2); r.dB[2] = const_cast<float*>(grads->w2_b);
  r.dW[3] = const_cast<float*>(grads->h); r.dB[3] = const_cast<float*>(grads->h_b);
  r.dwb2 = const_cast<float*>(grads->b2_w); r.dbb2 = const_cast<float*>(grads->b2_b);
  const long total = (long)C * (16 * KT + 1) + E + 1;
  hipLaunchKernelGGL(qmix_wide_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64 * RSG), 0, st, r);
  MARL_CHECK_LAUNCH();
  return 0;
}
