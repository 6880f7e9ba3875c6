// Fused QMIX mixer (reference network/mixer.py:57-80) for gfx950: the four state-conditioned
// hypernetworks and the mixing arithmetic in ONE kernel, forward and backward.
//
// Column order of the fused hypernet output (C = N*E + 3E columns, E = 32):
//     [ w1 (N*E, agent-major) | b1 (E) | w2 (E) | h = hyper_b2.0 (E) ]
// A workgroup (4 waves, one per SIMD) walks 16-row tiles of (episode, step) rows.  Wave w owns column
// tiles 4w..4w+3; its slice of the hypernet weights lives in registers for the whole launch as MFMA
// B-fragments (4 tiles x 8 k-chunks), the state tile is staged through LDS one tile ahead.  The 256-wide
// hypernet output is never written to HBM:
//   forward : q_tot = sum_e elu(sum_n q_n |w1[n,e]| + b1_e) |w2_e| + (relu(h) . w_b2 + b_b2)
//   backward: recomputes the hypernet tile, forms d(hypernet output) in accumulator layout - which IS the
//             A^T operand of  dW += d(out)^T [s | 1]  - and keeps dW (64 x 128 per wave) in registers;
//             one partial slab per workgroup, fixed-order reduce.
// Supported when E == 32, N*E + 3E <= 256 and S <= 128 (QMIX on 2s3z / matrix game); other shapes use
// the generic marl_linear composition.
#include "common.h"
#include "../../include/marl_hip.h"

namespace {

constexpr int E = 32;
constexpr int KCQ = 8;            // k-chunks of 16 (S padded to 128)
constexpr int SS = 128 + 4;       // LDS row stride of the state tile

#define WG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

struct QmixArgs {
  const float *W[4], *Bv[4];      // segment weights (rows x S) and biases: w1, b1, w2, h
  const float *wb2, *bb2;         // hyper_b2.2: (1,E), (1)
  ConcatSrc s;                    // state rows (dense segment 0, optional (T+1)-slot remap)
  const float* q;                 // (rows, N)
  const float* g;                 // (rows) dL/dq_tot (backward)
  float* q_tot;                   // (rows) (forward)
  float* dq;                      // (rows, N) (backward)
  float* ws;                      // [nwg][slab] (backward)
  // LOSS variant (backward with the TD loss folded in): g is not read; dL/dq_tot is formed per row from these
  const float *lr, *lterm, *lpadded, *lq_tgt;   // (rows) reward, terminated, padded, target-network q_tot of the next state
  float gamma;
  long rows;
  int N, S, C;
};

__host__ __device__ inline long qmix_slab_floats(int C, int S) { return (long)C * (S + 1) + (E + 1) + 2; }   // + [sum (mask td)^2 | sum mask] of the LOSS variant

__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
__device__ __forceinline__ float sum16(float v) {     // over the 16 lanes of a quarter-wave
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
  return v;
}
__device__ __forceinline__ float sum32(float v) { v = sum16(v); v += __shfl_xor(v, 16, 64); return v; }

// TPW column tiles per wave, NW = 16/TPW waves.  Two waves share each SIMD (8 waves x 2 tiles, or two 4-wave
// workgroups per CU): one wave's finishing math / LDS waits / barrier skew hide behind the other's MFMAs.
// LOSS (with BWD): the TD loss of q_learner.py:112-127 is folded in.  The backward pass recomputes q_tot anyway, so the
// separate forward launch of the eval mixer, the loss launch and its reduction disappear: per row
//     target = r + gamma q_tot_target (1 - terminated),  td = mask (target - q_tot),  dL/dq_tot = -2 mask td
// with mask = 1 - padded (un-normalised: the division by the global sum(mask) is folded into the optimizer step); the loss
// numerator and sum(mask) go through the slab like the weight gradients (fixed summation order).
template <bool BWD, int TPW, bool LOSS = false>
__global__ __launch_bounds__(64 * (16 / TPW), 2) void qmix_fused_kernel(QmixArgs a) {
  static_assert(!LOSS || BWD, "the loss is folded into the backward kernel");
  constexpr int NW = 16 / TPW, QNT = 64 * NW;
  __shared__ __attribute__((aligned(16))) float Ss[2][16 * SS];   // state tile, double buffered
  __shared__ float PA[NW][16][E];      // per-wave partial sums of the pre-activation a_e
  __shared__ float W2A[16][E];        // |w2|
  __shared__ float HBA[16][E];        // relu(h)
  __shared__ float DPRE[16][E];       // dL/da_e          (backward)
  __shared__ float HID[16][E];        // elu(a_e)         (backward)
  __shared__ float Qs2[2][16][16];    // q tile (double buffered with the state tile)
  __shared__ float Gs2[2][16];        // dL/dq_tot tile   (backward)
  __shared__ float Ls2[2][4][16];     // reward | terminated | padded | target q_tot of the tile (LOSS)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q4 = lane >> 4, m = lane & 15;
  const int N = a.N, S = a.S, C = a.C, NE = N * E;

  // ---- this wave's 4 column tiles: kind (0 w1, 1 b1, 2 w2, 3 h, -1 unused), agent n, e-half
  int kind[TPW], nn[TPW], eh[TPW];
  f32x4 wq[TPW][KCQ];
  float bias[TPW];
#pragma unroll
  for (int c = 0; c < TPW; ++c) {
    const int gt = TPW * wave + c, col0 = 16 * gt;
    int k = -1, seg_col = 0;
    if (col0 < NE) { k = 0; seg_col = col0; }
    else if (col0 < NE + E) { k = 1; seg_col = col0 - NE; }
    else if (col0 < NE + 2 * E) { k = 2; seg_col = col0 - NE - E; }
    else if (col0 < C) { k = 3; seg_col = col0 - NE - 2 * E; }
    kind[c] = k; nn[c] = k == 0 ? col0 / E : 0; eh[c] = (col0 % E) / 16;
    const float* Wp = k >= 0 ? a.W[k] + (long)(seg_col + m) * S : nullptr;
    bias[c] = k >= 0 ? a.Bv[k][seg_col + m] : 0.f;
#pragma unroll
    for (int kc = 0; kc < KCQ; ++kc)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kk = 16 * kc + 4 * q4 + i;
        wq[c][kc][i] = (k >= 0 && kk < S) ? Wp[kk] : 0.f;
      }
  }
  f32x4 accW[BWD ? TPW : 1][BWD ? KCQ : 1];
  float sbW[TPW];
#pragma unroll
  for (int c = 0; c < TPW; ++c) sbW[c] = 0.f;
  float acc_wb2 = 0.f, acc_bb2 = 0.f;      // hyper_b2.2 gradients (finishing lanes)
  float acc_ln = 0.f, acc_lm = 0.f;        // LOSS: sum (mask td)^2, sum mask (lanes e == 0 of the finishing rows)
  if (BWD) {
#pragma unroll
    for (int c = 0; c < TPW; ++c)
#pragma unroll
      for (int kc = 0; kc < KCQ; ++kc) accW[c][kc] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const float wb2_l = a.wb2[lane & 31];
  const float bb2 = a.bb2[0];

  // ---- state tile staging: thread -> (row tid/16 .. , float4 column); 16 rows x 32 float4 = 512 = 2 per thread
  const long tiles = (a.rows + 15) / 16;
  constexpr int NPF = 512 / QNT;
  f32x4 pf[NPF];
  auto fetch = [&](long tile) {
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      const int e = tid + QNT * i;
      const int r = e >> 5, c4 = (e & 31) * 4;
      long row = tile * 16 + r;
      if (row > a.rows - 1) row = a.rows - 1;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c4 < S) {
        const ConcatRow cr = concat_row(a.s, row);
        const float* p = a.s.p0 + cr.r0 * a.s.ld0 + c4;
        if (c4 + 3 < S) v = *reinterpret_cast<const f32x4*>(p);
        else {
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) if (c4 + cc < S) v[cc] = p[cc];
        }
      }
      pf[i] = v;
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      const int e = tid + QNT * i;
      *reinterpret_cast<f32x4*>(&Ss[buf][(e >> 5) * SS + (e & 31) * 4]) = pf[i];
    }
  };
  // q / g elements of this thread, also one tile ahead (threads 0..16N-1: q, threads 192..207: g)
  float pq = 0.f, pg = 0.f, pl[4] = {0.f, 0.f, 0.f, 0.f};
  const int qr = tid / N, qn = tid - qr * N;
  auto fetch_qg = [&](long tile) {
    pq = 0.f; pg = 0.f;
    if (tid < 16 * N) {
      const long row = tile * 16 + qr;
      if (row < a.rows) pq = a.q[row * N + qn];
    }
    if (BWD && tid >= 192 && tid < 208) {
      const long row = tile * 16 + (tid - 192);
      if (LOSS) {
        pl[0] = pl[1] = pl[3] = 0.f; pl[2] = 1.f;        // rows past the batch: padded
        if (row < a.rows) { pl[0] = a.lr[row]; pl[1] = a.lterm[row]; pl[2] = a.lpadded[row]; pl[3] = a.lq_tgt[row]; }
      } else if (row < a.rows) pg = a.g[row];
    }
  };
  float wb2c[TPW];                                   // hyper_b2.2 weight of this lane's column in each tile
#pragma unroll
  for (int c = 0; c < TPW; ++c) wb2c[c] = a.wb2[16 * eh[c] + m];
  long tile = blockIdx.x;
  if (tile < tiles) { fetch(tile); fetch_qg(tile); stash(0); }
  int buf = 0;
  ST_DECL(8);
  // barriers below only order LDS traffic (s_waitcnt lgkmcnt): the prefetch loads of the next tile stay in
  // flight across them - a __syncthreads() would drain vmcnt and expose the HBM latency on every tile
  for (; tile < tiles; tile += gridDim.x, buf ^= 1) {
    const long row0 = tile * 16;
    float (*Qs)[16] = Qs2[buf];
    float* Gs = Gs2[buf];
    if (tid < 16 * N) Qs[qr][qn] = pq;
    if (BWD && tid >= 192 && tid < 208) {
      if (LOSS) {
#pragma unroll
        for (int k = 0; k < 4; ++k) Ls2[buf][k][tid - 192] = pl[k];
      } else Gs[tid - 192] = pg;
    }
    const long nt = tile + gridDim.x;
    if (nt < tiles) { fetch(nt); fetch_qg(nt); }
    ST_MARK(0);
    WG_BARRIER();                                  // Ss[buf], Qs, Gs ready
    ST_MARK(1);
    // ---- hypernet tile: out[row 4q+i][col 16gt+m], 4 column tiles x 8 k-chunks
    f32x4 acc[TPW];
#pragma unroll
    for (int c = 0; c < TPW; ++c) acc[c] = (f32x4){bias[c], bias[c], bias[c], bias[c]};
    const float* sr = &Ss[buf][m * SS + 4 * q4];
#pragma unroll
    for (int kc = 0; kc < KCQ; ++kc) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(sr + 16 * kc);
#pragma unroll
      for (int c = 0; c < TPW; ++c) acc[c] = mfma16x4(a4, wq[c][kc], acc[c]);
    }
    ST_MARK(2);
    // ---- partial pre-activations: a_e = b1_e + sum_n q_n |w1[n,e]|  (this wave's agents / b1 tiles)
    float pa[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int c = 0; c < TPW; ++c) {
      if (kind[c] == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) pa[eh[c]][i] += Qs[4 * q4 + i][nn[c]] * fabsf(acc[c][i]);
      } else if (kind[c] == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) pa[eh[c]][i] += acc[c][i];
      } else if (kind[c] == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) W2A[4 * q4 + i][16 * eh[c] + m] = fabsf(acc[c][i]);
      } else if (kind[c] == 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) HBA[4 * q4 + i][16 * eh[c] + m] = fmaxf(acc[c][i], 0.f);
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i) PA[wave][4 * q4 + i][16 * h + m] = pa[h][i];
    ST_MARK(3);
    WG_BARRIER();
    ST_MARK(4);
    // ---- finish: wave w takes rows (16/NW)w .., two rows per pass (lane = e of row `half`)
#pragma unroll
    for (int p = 0; p < 8 / NW; ++p) {
      const int r = (16 / NW) * wave + 2 * p + (lane >> 5), e = lane & 31;
      float ae = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) ae += PA[w][r][e];
      const float ex = __expf(ae);
      const float hid = ae > 0.f ? ae : ex - 1.f;                 // elu, alpha = 1
      const float w2 = W2A[r][e], hb = HBA[r][e];
      const float tot = sum32(hid * w2 + hb * wb2_l);
      if (!BWD) {
        if (e == 0 && row0 + r < a.rows) a.q_tot[row0 + r] = tot + bb2;
      } else {
        float gr;
        if (LOSS) {
          const float qt = tot + bb2;
          const float mask = 1.f - Ls2[buf][2][r];
          const float target = Ls2[buf][0][r] + a.gamma * Ls2[buf][3][r] * (1.f - Ls2[buf][1][r]);
          const float mtd = mask * (target - qt);
          gr = -2.f * mask * mtd;
          if (e == 0) {
            acc_ln += mtd * mtd; acc_lm += mask;
            Gs[r] = gr;                                  // read by every wave after the barrier below
            if (a.q_tot && row0 + r < a.rows) a.q_tot[row0 + r] = qt;
          }
        } else gr = Gs[r];
        DPRE[r][e] = gr * w2 * (ae > 0.f ? 1.f : ex);
        HID[r][e] = hid;
        acc_wb2 += gr * hb;
        if (e == 0) acc_bb2 += gr;
      }
    }
    ST_MARK(5);
    if (BWD) {
      WG_BARRIER();
      // ---- d(hypernet output) in accumulator layout, dq, then dW += dhy^T [s | 1]
      f32x4 dhy[TPW];
#pragma unroll
      for (int c = 0; c < TPW; ++c) {
        const int ecol = 16 * eh[c] + m;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * q4 + i;
          float v = 0.f;
          if (kind[c] == 0) v = Qs[r][nn[c]] * DPRE[r][ecol] * sgn(acc[c][i]);
          else if (kind[c] == 1) v = DPRE[r][ecol];
          else if (kind[c] == 2) v = Gs[r] * HID[r][ecol] * sgn(acc[c][i]);
          else if (kind[c] == 3) v = acc[c][i] > 0.f ? Gs[r] * wb2c[c] : 0.f;
          dhy[c][i] = v;
        }
        sbW[c] += dhy[c][0] + dhy[c][1] + dhy[c][2] + dhy[c][3];
      }
      // dq_n = sum_e |w1[n,e]| dpre_e : tiles (2n, 2n+1) sit in the same wave as (c, c+1), c even
#pragma unroll
      for (int c = 0; c < TPW; c += 2) {
        if (kind[c] == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * q4 + i;
            float v = fabsf(acc[c][i]) * DPRE[r][m] + fabsf(acc[c + 1][i]) * DPRE[r][16 + m];
            v = sum16(v);
            if (m == 0 && row0 + r < a.rows) a.dq[(row0 + r) * N + nn[c]] = v;
          }
        }
      }
      const float* sd = &Ss[buf][(4 * q4) * SS + m];
#pragma unroll
      for (int kc = 0; kc < KCQ; ++kc) {
        f32x4 sD;
#pragma unroll
        for (int i = 0; i < 4; ++i) sD[i] = sd[i * SS + 16 * kc];
#pragma unroll
        for (int c = 0; c < TPW; ++c) accW[c][kc] = mfma16x4(dhy[c], sD, accW[c][kc]);
      }
    }
    ST_MARK(6);
    if (nt < tiles) stash(buf ^ 1);
    ST_MARK(7);
    // the next iteration's first barrier orders these LDS writes before their readers; the small tiles
    // (PA, W2A, ...) are rewritten only after that barrier too
  }
  ST_DUMP(8);
  if (BWD) {
    float* slab = a.ws + (long)blockIdx.x * qmix_slab_floats(C, S);
    const int Sx = S + 1;
#pragma unroll
    for (int c = 0; c < TPW; ++c) {
      if (kind[c] < 0) continue;
      const int col0 = 16 * (TPW * wave + c);
#pragma unroll
      for (int kc = 0; kc < KCQ; ++kc)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int k = 16 * kc + m;
          if (k < S) slab[(long)(col0 + 4 * q4 + i) * Sx + k] = accW[c][kc][i];     // dW[col][k]: D rows = columns of dhy
        }
      float sb = sbW[c];
      sb += __shfl_xor(sb, 16, 64); sb += __shfl_xor(sb, 32, 64);                    // over the 4 row groups
      if (q4 == 0) slab[(long)(col0 + m) * Sx + S] = sb;
    }
    // hyper_b2.2 partials: lanes e of both halves, then the 4 waves in fixed order through LDS
    float v = acc_wb2 + __shfl_xor(acc_wb2, 32, 64);
    float b = acc_bb2 + __shfl_xor(acc_bb2, 32, 64);
    __syncthreads();
    float* wred = &PA[0][0][0];                 // reuse: [4][E+1]
    if (lane < 32) wred[wave * (E + 1) + lane] = v;
    if (lane == 0) wred[wave * (E + 1) + E] = b;
    __syncthreads();
    if (tid < E + 1) {
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) tot += wred[w * (E + 1) + tid];
      slab[(long)C * Sx + tid] = tot;
    }
    // loss partials: the two finishing lanes (e == 0 of each half) of every wave, waves in fixed order
    __syncthreads();
    if (LOSS) {
      const float ln = acc_ln + __shfl_xor(acc_ln, 32, 64), lm = acc_lm + __shfl_xor(acc_lm, 32, 64);
      if (lane == 0) { wred[2 * wave] = ln; wred[2 * wave + 1] = lm; }
      __syncthreads();
      if (tid < 2) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += wred[2 * w + tid];
        slab[(long)C * Sx + E + 1 + tid] = tot;
      }
    } else if (tid < 2) slab[(long)C * Sx + E + 1 + tid] = 0.f;
  }
}

struct QmixRedArgs {
  const float* ws; int nwg; int N, S, C;
  float *dW[4], *dB[4], *dwb2, *dbb2;
  float* loss2;       // [sum (mask td)^2 | sum mask] accumulated into (LOSS variant) or null
};

constexpr int RSG = 16;            // slab groups per output element (fixed summation order -> deterministic)

__global__ __launch_bounds__(64 * RSG) void qmix_fused_reduce_kernel(QmixRedArgs a) {
  __shared__ float part[RSG][64];
  const long slab = qmix_slab_floats(a.C, a.S);
  const int el = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const long e = (long)blockIdx.x * 64 + el;
  float s = 0.f;
  if (e < slab)
    for (int w = sg; w < a.nwg; w += RSG) s += a.ws[(long)w * slab + e];
  part[sg][el] = s;
  __syncthreads();
  if (sg != 0 || e >= slab) return;
  s = 0.f;
#pragma unroll
  for (int g = 0; g < RSG; ++g) s += part[g][el];
  const int Sx = a.S + 1, NE = a.N * E;
  if (e < (long)a.C * Sx) {
    const int col = (int)(e / Sx), k = (int)(e - (long)col * Sx);
    int seg, sc;
    if (col < NE) { seg = 0; sc = col; }
    else if (col < NE + E) { seg = 1; sc = col - NE; }
    else if (col < NE + 2 * E) { seg = 2; sc = col - NE - E; }
    else { seg = 3; sc = col - NE - 2 * E; }
    if (k < a.S) a.dW[seg][(long)sc * a.S + k] += s;
    else a.dB[seg][sc] += s;
  } else {
    // tail: [dwb2 (E) | dbb2]
    const long tpos = e - (long)a.C * Sx;
    if (tpos < E) a.dwb2[tpos] += s;
    else if (tpos == E) a.dbb2[0] += s;
    else if (a.loss2) a.loss2[tpos - E - 1] += s;
  }
}

}  // namespace
ST_DEFINE_SETTER(marl_debug_stamps_qmix)
namespace {

inline bool supported(int N, int S, int Eq) { return Eq == E && N * E + 3 * E <= 256 && S <= 128 && N <= 16 && S >= 1; }

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

inline int fill(QmixArgs& a, const marl_qmix_weights_t* w, const marl_src_t* s, const float* q, long rows, int N, int S) {
  a.W[0] = w->w1; a.Bv[0] = w->w1_b; a.W[1] = w->b1; a.Bv[1] = w->b1_b; a.W[2] = w->w2; a.Bv[2] = w->w2_b;
  a.W[3] = w->h; a.Bv[3] = w->h_b; a.wb2 = w->b2_w; a.bb2 = w->b2_b;
  a.s = state_src(s);
  if (a.s.k0 != S || s->k1 || s->nhot || s->nid || s->m0 || (a.s.ld0 % 4) || ((uintptr_t)a.s.p0 & 15)) return 1;
  a.q = q; a.rows = rows; a.N = N; a.S = S; a.C = N * E + 3 * E;
  return 0;
}

#ifndef FWD_TPW
#define FWD_TPW 4
#endif

inline unsigned grid_for(long rows, int per_cu = 1) {
  long tiles = (rows + 15) / 16;
  return (unsigned)(tiles < 256 * per_cu ? tiles : 256 * per_cu);
}

}  // namespace

extern "C" int marl_qmix_fused_supported(int N, int S, int Eq) { return supported(N, S, Eq) ? 1 : 0; }

extern "C" size_t marl_qmix_fused_workspace(long rows, int N, int S) {
  return (size_t)grid_for(rows) * qmix_slab_floats(N * E + 3 * E, S) * sizeof(float);
}

extern "C" int marl_qmix_fused_fwd(const marl_qmix_weights_t* w, const marl_src_t* s, const float* q, float* q_tot,
                                   long rows, int N, int S, int Eq, void* stream) {
  if (rows <= 0) return 0;
  if (!supported(N, S, Eq)) return (int)hipErrorInvalidValue;
  QmixArgs a;
  if (fill(a, w, s, q, rows, N, S)) return (int)hipErrorInvalidValue;
  a.g = nullptr; a.q_tot = q_tot; a.dq = nullptr; a.ws = nullptr;
  a.lr = a.lterm = a.lpadded = a.lq_tgt = nullptr; a.gamma = 0.f;
  hipLaunchKernelGGL((qmix_fused_kernel<false, FWD_TPW>), dim3(grid_for(rows, 2)), dim3(64 * (16 / FWD_TPW)), 0,
                     (hipStream_t)stream, a);
  MARL_CHECK_LAUNCH();
  return 0;
}

static int qmix_bwd_launch(QmixArgs& a, const marl_qmix_weights_t* grads, float* loss2, float* ws, long rows, int N, int S,
                           bool loss, hipStream_t st) {
  const unsigned nwg = grid_for(rows);
  if (loss) hipLaunchKernelGGL((qmix_fused_kernel<true, 2, true>), dim3(nwg), dim3(512), 0, st, a);
  else hipLaunchKernelGGL((qmix_fused_kernel<true, 2, false>), dim3(nwg), dim3(512), 0, st, a);
  MARL_CHECK_LAUNCH();
  QmixRedArgs r;
  r.ws = ws; r.nwg = (int)nwg; r.N = N; r.S = S; r.C = a.C; r.loss2 = loss2;
  r.dW[0] = const_cast<float*>(grads->w1); r.dB[0] = const_cast<float*>(grads->w1_b);
  r.dW[1] = const_cast<float*>(grads->b1); r.dB[1] = const_cast<float*>(grads->b1_b);
  r.dW[2] = const_cast<float*>(grads->w2); r.dB[2] = const_cast<float*>(grads->w2_b);
  r.dW[3] = const_cast<float*>(grads->h); r.dB[3] = const_cast<float*>(grads->h_b);
  r.dwb2 = const_cast<float*>(grads->b2_w); r.dbb2 = const_cast<float*>(grads->b2_b);
  const long slab = qmix_slab_floats(a.C, S);
  hipLaunchKernelGGL(qmix_fused_reduce_kernel, dim3((unsigned)((slab + 63) / 64)), dim3(64 * RSG), 0, st, r);
  MARL_CHECK_LAUNCH();
  return 0;
}

extern "C" int marl_qmix_fused_bwd(const marl_qmix_weights_t* w, const marl_src_t* s, const float* q,
                                   const float* dq_tot, float* dq, const marl_qmix_weights_t* grads, float* ws,
                                   size_t ws_bytes, long rows, int N, int S, int Eq, void* stream) {
  if (rows <= 0) return 0;
  if (!supported(N, S, Eq)) return (int)hipErrorInvalidValue;
  if (ws_bytes < marl_qmix_fused_workspace(rows, N, S)) return (int)hipErrorInvalidValue;
  QmixArgs a;
  if (fill(a, w, s, q, rows, N, S)) return (int)hipErrorInvalidValue;
  a.g = dq_tot; a.q_tot = nullptr; a.dq = dq; a.ws = ws;
  a.lr = a.lterm = a.lpadded = a.lq_tgt = nullptr; a.gamma = 0.f;
  return qmix_bwd_launch(a, grads, nullptr, ws, rows, N, S, false, (hipStream_t)stream);
}

extern "C" int marl_qmix_fused_loss_bwd(const marl_qmix_weights_t* w, const marl_src_t* s, const float* q,
                                        const float* q_tot_tgt, const float* r, const float* term, const float* padded,
                                        float gamma, float* q_tot, float* dq, const marl_qmix_weights_t* grads,
                                        float* loss2, float* ws, size_t ws_bytes, long rows, int N, int S, int Eq,
                                        void* stream) {
  if (rows <= 0) return 0;
  if (!supported(N, S, Eq) || !q_tot_tgt || !r || !term || !padded || !loss2) return (int)hipErrorInvalidValue;
  if (ws_bytes < marl_qmix_fused_workspace(rows, N, S)) return (int)hipErrorInvalidValue;
  QmixArgs a;
  if (fill(a, w, s, q, rows, N, S)) return (int)hipErrorInvalidValue;
  a.g = nullptr; a.q_tot = q_tot; a.dq = dq; a.ws = ws;
  a.lr = r; a.lterm = term; a.lpadded = padded; a.lq_tgt = q_tot_tgt; a.gamma = gamma;
  return qmix_bwd_launch(a, grads, loss2, ws, rows, N, S, true, (hipStream_t)stream);
}
