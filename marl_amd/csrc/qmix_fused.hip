// Fused QMIX mixer (reference network/mixer.py:57-80) for gfx950: the four state-conditioned
// hypernetworks and the mixing arithmetic in ONE kernel, forward and backward.
//
// The fused hypernet output has C = N*E + 3E columns (E = 32): [ w1 (N*E, agent-major) | b1 | w2 | h = hyper_b2.0 ].
// A workgroup of NW waves walks 16-row tiles of (episode, step) rows.  The columns are dealt to the waves BY MIXING
// EMBEDDING INDEX e, not by segment: wave w owns e in [EW w, EW (w+1)), EW = 32 / NW, in all eight column groups
//     group 0..4 = w1 of agent 0..4 (absent agents: zero weights), 5 = b1, 6 = w2, 7 = h
// so its 8 EW columns are 16-column MFMA tiles of 16 / EW groups each (NW = 8: tiles {agents 0-3}, {agent 4, b1, w2, h};
// NW = 4: {0,1}, {2,3}, {4, b1}, {w2, h}); the wave's slice of the weights lives in registers for the whole launch as
// B-fragments.  With that deal everything a mixing element e needs sits in ONE wave, in accumulator registers:
//   a_e  = b1_e + sum_n q_n |w1[n,e]|        : per-lane fma, then a sum over the lane groups of a 16-lane row (DPP rotations)
//   term = elu(a_e) |w2_e| , relu(h_e) wb2_e : same lanes; the row sum over the wave's e is four DPP adds
// and the only cross-wave exchange per tile is the 16 partial q_tot values of each wave (one LDS write, one barrier).
// The backward pass continues in the same registers: dL/da_e, d(hypernet output) in accumulator layout - which IS the
// A^T operand of dW += d(out)^T [s | 1] - with dW (8 EW x 128 per wave) in registers, one slab per workgroup and a
// fixed-order reduce; the bias gradients come out of the same MFMAs through a ones column appended to the state tile.
// fp32 MFMA and VALU instructions do not overlap on a SIMD of this chip (tools/probe/coissue_probe.hip), so the cost of a
// tile is its MFMA cycles PLUS the issue cycles of everything else: the point of this deal is the instruction count
// (the by-segment deal it replaces needed ~3x the VALU instructions, LDS round trips for a_e and a second barrier).
// Supported when E == 32, N <= 5 and S <= 128 (QMIX on 2s3z / matrix game); wide states: qmix_wide.hip.
// X6 (the *_x6 entry points, args.gemm_mode = "bf16x6"): the two GEMMs of a tile - hypernet output and dW - with every fp32 product as
// six bf16 MFMA products (x6.h).  Same column deal, same epilogue, same accumulator layouts: the state tile lives in LDS as three
// bf16 planes (each element split once, where the tile is stashed), the weights are pre-split B fragments in registers
// (v_mfma_f32_16x16x32_bf16, four 32-wide k chunks), and the weight gradient contracts over the tile's 16 rows on
// v_mfma_f32_16x16x16_bf16: its A operand IS split4(d(out)) of the accumulator layout, its B operand comes from the row-major
// state planes through transposed LDS reads (ds_read_b64_tr_b16: rows 4g .. 4g+3 of a column per lane).
#include "x6.h"
#include "../../include/marl_hip.h"

namespace {

constexpr int E = 32;
constexpr int KCQ = 8;            // k-chunks of 16 (S padded to 128)
constexpr int SS = 128 + 4;       // LDS row stride of the state tile
constexpr int NAG = 5;            // agent groups (N <= 5); groups 5, 6, 7 = b1, w2, h
constexpr int PP = 136;           // X6: pitch (bf16 elements) of the state planes: 272-byte rows spread the 16-byte fragment reads of a 16-lane group over all banks

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

// cross-lane moves inside a 16-lane row (DPP: no LDS round trip; the compiler folds them into the consuming v_add)
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E;          // quad_perm [1,0,3,2], [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;
constexpr int DPP_ROR4 = 0x124, DPP_ROR8 = 0x128;
// sum over the lanes of a row that hold the same e (lane groups EW apart): every lane ends with the total
template <int EW>
__device__ __forceinline__ float group_sum(float v) {
  if (EW == 4) v += dpp<DPP_ROR4>(v);
  v += dpp<DPP_ROR8>(v);
  return v;
}
// sum over the EW lanes of a group / over all 16 lanes of a row
template <int EW>
__device__ __forceinline__ float e_sum(float v) {
  v += dpp<DPP_XOR1>(v);
  v += dpp<DPP_XOR2>(v);
  if (EW == 8) v += dpp<DPP_HALF_MIRROR>(v);
  return v;
}
__device__ __forceinline__ float row_sum(float v) {
  v += dpp<DPP_XOR1>(v);
  v += dpp<DPP_XOR2>(v);
  v += dpp<DPP_HALF_MIRROR>(v);
  v += dpp<DPP_MIRROR>(v);
  return v;
}
// sign(x) clamped to [lo, 1]: lo = -1 -> sign, 0 -> step, 1 -> 1 (exact for normal x; sign(0) = 0 like torch.abs')
__device__ __forceinline__ float sign_clamp(float x, float lo) { return __builtin_amdgcn_fmed3f(x * 0x1p126f, lo, 1.0f); }

// NW waves; EW = 32/NW mixing elements per wave; a 16-column tile holds GPT = 16/EW groups; TPW = 8/GPT tiles per wave.
// LOSS (with BWD): the TD loss of q_learner.py:112-127 is folded in.  The backward pass recomputes q_tot anyway, so the
// separate forward launch of the eval mixer, the loss launch and its reduction disappear: per row
//     target = r + gamma q_tot_target (1 - terminated),  td = mask (target - q_tot),  dL/dq_tot = -2 mask td
// with mask = 1 - padded (un-normalised: the division by the global sum(mask) is folded into the optimizer step); the loss
// numerator and sum(mask) go through the slab like the weight gradients (fixed summation order).
template <bool BWD, int NW, bool LOSS = false, bool X6 = false>
__global__ __launch_bounds__(64 * NW, 2) void qmix_fused_kernel(QmixArgs a) {
  static_assert(!LOSS || BWD, "the loss is folded into the backward kernel");
  static_assert(!X6 || NW == 8, "the split variant runs eight waves (two column tiles per wave: 96 registers of pre-split weights)");
  constexpr int QNT = 64 * NW, EW = 32 / NW, GPT = 16 / EW, TPW = 8 / GPT, LT = TPW - 1;
  // state tile, double buffered: fp32 rows, or (X6) three bf16 planes [buffer][plane][16 rows][PP]
  constexpr int SRAW = X6 ? 2 * 3 * 16 * PP * 2 : 2 * 16 * SS * 4;
  __shared__ __attribute__((aligned(16))) char Sraw[SRAW];
  float (*Ss)[16 * SS] = reinterpret_cast<float (*)[16 * SS]>(Sraw);
  short* const Sp = reinterpret_cast<short*>(Sraw);
  constexpr int PLN = 16 * PP, PBUF = 3 * PLN;                    // elements per plane / per buffer
  // X6 backward: the weight fragments of the last k chunk as ready fragments in LDS (the wave runs at the 256-register ceiling:
  // 72 instead of 96 registers of weights beside 64 of dW accumulators and the epilogue's working set - no spills)
  constexpr int KR = (X6 && BWD) ? (LOSS ? 2 : 3) : 4;            // k chunks whose fragments stay in registers
  constexpr int KL = 4 - KR;                                      // ... and in LDS: [wave][tile][chunk][plane][lane]
  __shared__ __attribute__((aligned(16))) i32x4 WL[KL ? NW * TPW * KL * 3 * 64 : 1];
  __shared__ __attribute__((aligned(16))) float Qt2[2][8][16];    // q tile, transposed: rows 0..4 agents, 5 = ones (b1), 6, 7 = zeros
  __shared__ __attribute__((aligned(16))) float QT[NW][16];       // per-wave partial q_tot
  __shared__ __attribute__((aligned(16))) float DQP[NW][NAG][16]; // per-wave partial dq (backward)
  __shared__ __attribute__((aligned(16))) float Gs2[2][16];       // dL/dq_tot tile   (backward)
  __shared__ __attribute__((aligned(16))) float Ls2[2][4][16];    // reward | terminated | padded | target q_tot of the tile (LOSS)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q4 = lane >> 4, m = lane & 15;
  const int el = m & (EW - 1), gl = m / EW, e = EW * wave + el;
  const int N = a.N, S = a.S, C = a.C, NE = N * E;
  const bool ones = S < 128;                    // room for the ones column -> bias gradients from the dW MFMAs

  // ---- this lane's column in each tile: group g = GPT c + gl, mixing element e
  f32x4 wq[X6 ? 1 : TPW][X6 ? 1 : KCQ];
  F3 wq6[X6 ? TPW : 1][X6 ? KR : 1];      // X6: B fragments, lane (g, column of lane m): W[column][32 kc + 8g .. + 7], split once
  float bias[TPW];
  int grp[TPW];
#pragma unroll
  for (int c = 0; c < TPW; ++c) {
    const int g = GPT * c + gl;
    grp[c] = g;
    const float *Wp = nullptr, *Bp = nullptr;
    if (g < NAG) { if (g < N) { Wp = a.W[0] + (long)(g * E + e) * S; Bp = a.Bv[0] + g * E + e; } }
    else { Wp = a.W[g - NAG + 1] + (long)e * S; Bp = a.Bv[g - NAG + 1] + e; }
    bias[c] = Bp ? *Bp : 0.f;
    if (X6) {
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        float v[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const int kk = 32 * kc + 8 * q4 + jj;
          v[jj] = (Wp && kk < S) ? Wp[kk] : 0.f;
        }
        const F3 f = split8((f32x4){v[0], v[1], v[2], v[3]}, (f32x4){v[4], v[5], v[6], v[7]});
        if (kc < KR) wq6[c][kc < KR ? kc : 0] = f;
        else { i32x4* wl = WL + (((wave * TPW + c) * (KL ? KL : 1) + (kc - KR)) * 3) * 64 + lane; wl[0] = f.h; wl[64] = f.m; wl[128] = f.l; }
      }
    } else {
#pragma unroll
      for (int kc = 0; kc < KCQ; ++kc)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int kk = 16 * kc + 4 * q4 + i;
          wq[c][kc][i] = (Wp && kk < S) ? Wp[kk] : 0.f;
        }
    }
  }
  const bool is6 = grp[LT] == 6, is7 = grp[LT] == 7;
  const float wb2e = a.wb2[e];
  const float cw = is7 ? wb2e : 0.f;
  const float bb2 = a.bb2[0];
  // the tile that mixes agent and b1 columns (group 5 sits in tile 5 / GPT): |x| for agents, x for b1
  constexpr int MT = NAG / GPT;
  const unsigned absmask = grp[MT] < NAG ? 0x7fffffffu : 0xffffffffu;
  float lo_c[TPW];                               // lower clamp of the sign factor of d(out): -1 |.| columns, 1 b1, 0 relu
#pragma unroll
  for (int c = 0; c < TPW; ++c) lo_c[c] = grp[c] == NAG ? 1.f : (grp[c] == 7 ? 0.f : -1.f);

  f32x4 accW[BWD ? TPW : 1][BWD ? KCQ : 1];
  float sbW[TPW];
#pragma unroll
  for (int c = 0; c < TPW; ++c) sbW[c] = 0.f;
  float acc_wb2 = 0.f, acc_bb2 = 0.f;      // hyper_b2.2 gradients (rows 4q..4q+3 of every tile; lanes of group 7 / any lane)
  float acc_ln = 0.f, acc_lm = 0.f;        // LOSS: sum (mask td)^2, sum mask
  if (BWD) {
#pragma unroll
    for (int c = 0; c < TPW; ++c)
#pragma unroll
      for (int kc = 0; kc < KCQ; ++kc) accW[c][kc] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // constant rows of the transposed q tile
  if (tid < 2 * 8 * 16) {
    const int r8 = (tid >> 4) & 7;
    (&Qt2[0][0][0])[tid] = r8 == NAG ? 1.f : 0.f;
  }

  // ---- state tile staging: thread -> (row, float4 column); 16 rows x 32 float4 = 512.  The row remap of the source
  // ((T+1)-slot storage, optional episode map of an in-place replay sample) is resolved ONE TILE EARLIER than the loads
  // that use it: emap0[e] -> address -> state row is a dependent chain, and resolving it in the same iteration put a
  // full memory latency at the top of every tile.  Columns >= S: the lane reads column 0 of its row instead (finite
  // values; their weights are zero and their dW columns are never written), no per-lane branch.
  const int tiles = (int)((a.rows + 15) / 16);
  constexpr int NPF = 512 / QNT;
  f32x4 pf[NPF];
  int pe0 = 0, pe1 = 0, pw0 = 0, pw1 = 0;                 // episode slot / row inside the episode block, one tile ahead
  const int fr = tid >> 5, fc4 = (tid & 31) * 4;
  const bool ones_lane = BWD && ones && fc4 == (S & ~3);
  const int ones_j = S & 3;
  const int nrows = (int)a.rows;                         // rows < 2^31 (32-bit row arithmetic in the loop)
  const unsigned last_row = (unsigned)(nrows - 1);
  const unsigned rpe = (unsigned)a.s.rpe0;
  const int roff = rpe ? (int)a.s.off0 : 0;
  const FastDiv fd = a.s.fd0;
  const int* emap = a.s.emap0;
  const float* sp0 = a.s.p0 + (fc4 < S ? fc4 : 0);
  const long ld = a.s.ld0, blk = (long)a.s.bs0 * a.s.ld0;  // floats per row / per episode block
  auto map_one = [&](int tile, int i, int& pe, int& pw) {
    unsigned row = (unsigned)tile * 16u + (unsigned)(fr + (QNT / 32) * i);
    row = row < last_row ? row : last_row;
    const unsigned ep = rpe ? fastdiv(row, fd) : 0u;
    pw = (int)(row - ep * rpe) + roff;
    pe = (int)ep;
    if (emap) pe = emap[ep];
  };
  auto map_issue = [&](int tile) {
    map_one(tile, 0, pe0, pw0);
    if (NPF > 1) map_one(tile, 1, pe1, pw1);
  };
  auto fetch_one = [&](int pe, int pw) {
    f32x4 v = *reinterpret_cast<const f32x4*>(sp0 + ((long)pe * blk + (long)pw * ld));
    if (BWD) {                                            // [s | 1]
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (ones_lane && j == ones_j) ? 1.f : v[j];
    }
    return v;
  };
  auto fetch = [&](int) {
    pf[0] = fetch_one(pe0, pw0);
    if (NPF > 1) pf[NPF - 1] = fetch_one(pe1, pw1);
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      if (X6) {                                           // split once, here: 8 bytes per plane
        const F3h f = split4(pf[i]);
        short* p = Sp + buf * PBUF + (fr + (QNT / 32) * i) * PP + fc4;
        *reinterpret_cast<i32x2*>(p) = f.h;
        *reinterpret_cast<i32x2*>(p + PLN) = f.m;
        *reinterpret_cast<i32x2*>(p + 2 * PLN) = f.l;
      } else *reinterpret_cast<f32x4*>(&Ss[buf][(fr + (QNT / 32) * i) * SS + fc4]) = pf[i];
    }
  };
  // q / g elements of this thread, also one tile ahead (threads 0..16N-1: q, threads 192..207: g)
  float pq = 0.f, pg = 0.f, pl[4] = {0.f, 0.f, 0.f, 0.f};
  const int qr = tid / N, qn = tid - qr * N;
  auto fetch_qg = [&](int tile) {
    pq = 0.f; pg = 0.f;
    if (tid < 16 * N) {
      const int row = tile * 16 + qr;
      if (row < nrows) pq = a.q[(unsigned)(row * N + qn)];
    }
    if (BWD && tid >= 192 && tid < 208) {
      const int row = tile * 16 + (tid - 192);
      if (LOSS) {
        pl[0] = pl[1] = pl[3] = 0.f; pl[2] = 1.f;        // rows past the batch: padded
        if (row < nrows) { pl[0] = a.lr[row]; pl[1] = a.lterm[row]; pl[2] = a.lpadded[row]; pl[3] = a.lq_tgt[row]; }
      } else if (row < nrows) pg = a.g[row];
    }
  };
  // dq of the previous tile: the waves' partials in fixed order (threads 0..16N-1, after the next barrier)
  auto flush_dq = [&](int prow0) {
    if (tid < 16 * N) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += DQP[w][qn][qr];
      if (prow0 + qr < nrows) a.dq[(unsigned)((prow0 + qr) * N + qn)] = v;
    }
  };
  int tile = blockIdx.x;
  if (tile < tiles) {
    map_issue(tile); fetch(tile); fetch_qg(tile);
    if (tile + (int)gridDim.x < tiles) map_issue(tile + gridDim.x);
    stash(0);
  }
  int buf = 0;
  int prow0 = -1;
  ST_DECL(8);
  // barriers below only order LDS traffic (s_waitcnt lgkmcnt): the prefetch loads of the next tile stay in
  // flight across them - a __syncthreads() would drain vmcnt and expose the HBM latency on every tile
  __syncthreads();                                  // constant rows of Qt2
  for (; tile < tiles; tile += gridDim.x, buf ^= 1) {
    const int row0 = tile * 16;
    if (tid < 16 * N) Qt2[buf][qn][qr] = pq;
    if (BWD && tid >= 192 && tid < 208) {
      if (LOSS) {
#pragma unroll
        for (int k = 0; k < 4; ++k) Ls2[buf][k][tid - 192] = pl[k];
      } else Gs2[buf][tid - 192] = pg;
    }
    const int nt = tile + gridDim.x;
    if (nt < tiles) {
      fetch(nt); fetch_qg(nt);
      if (nt + (int)gridDim.x < tiles) map_issue(nt + gridDim.x);
    }
    ST_MARK(0);
    WG_BARRIER();                                  // Ss[buf], Qt2[buf], Gs2 / Ls2 ready; DQP of the previous tile complete
    ST_MARK(1);
    if (BWD && prow0 >= 0) flush_dq(prow0);
    // ---- hypernet tile: out[row 4q+i][column (group, e) of lane m], TPW tiles x 8 k-chunks
    f32x4 acc[TPW];
#pragma unroll
    for (int c = 0; c < TPW; ++c) acc[c] = (f32x4){bias[c], bias[c], bias[c], bias[c]};
    if (X6) {
      // A fragments: row m, columns 32 kc + 8g .. + 7 of each plane (one 16-byte read per plane); even / odd k chunks on
      // separate accumulators: four independent chains for the wave's two column tiles
      f32x4 accb[TPW];
#pragma unroll
      for (int c = 0; c < TPW; ++c) accb[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const short* sp = Sp + buf * PBUF + m * PP + 8 * q4;
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        F3 xa;
        xa.h = *reinterpret_cast<const i32x4*>(sp + 32 * kc);
        xa.m = *reinterpret_cast<const i32x4*>(sp + PLN + 32 * kc);
        xa.l = *reinterpret_cast<const i32x4*>(sp + 2 * PLN + 32 * kc);
        F3 wk[TPW];
#pragma unroll
        for (int c = 0; c < TPW; ++c) {
          if (kc < KR) wk[c] = wq6[c][kc < KR ? kc : 0];
          else { const i32x4* wl = WL + (((wave * TPW + c) * (KL ? KL : 1) + (kc - KR)) * 3) * 64 + lane; wk[c].h = wl[0]; wk[c].m = wl[64]; wk[c].l = wl[128]; }
        }
#define OP(p_, q_) _Pragma("unroll") for (int c = 0; c < TPW; ++c) { if (kc & 1) accb[c] = mm(xa.p_, wk[c].q_, accb[c]); else acc[c] = mm(xa.p_, wk[c].q_, acc[c]); }
        OP(m, m) OP(h, l) OP(l, h) OP(h, m) OP(m, h) OP(h, h)
#undef OP
      }
#pragma unroll
      for (int c = 0; c < TPW; ++c) acc[c] += accb[c];
    } else {
      const float* sr = &Ss[buf][m * SS + 4 * q4];
#pragma unroll
      for (int kc = 0; kc < KCQ; ++kc) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(sr + 16 * kc);
        // round robin over the wave's column tiles: back-to-back MFMAs on ONE accumulator run at 80 % of the pipe (common.h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int c = 0; c < TPW; ++c) acc[c] = mfma16(a4[i], wq[c][kc][i], acc[c]);
      }
    }
    ST_MARK(2);
    __builtin_amdgcn_sched_barrier(0);             // the MFMA run first, the epilogue after it (no fine interleaving)
    // ---- a_e = b1_e + sum_n q_n |w1[n,e]| for rows 4q..4q+3: the lane's groups, then the groups of the row
    f32x4 qv[TPW];                                 // q of the lane's agent (1 for b1, 0 for w2 / h / absent agents)
    f32x4 p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c <= MT; ++c) {
      qv[c] = *reinterpret_cast<const f32x4*>(&Qt2[buf][grp[c]][4 * q4]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float av = acc[c][i];                 // (a bit_cast of the vector ELEMENT expression reads element 0)
        const float x = c < MT ? fabsf(av) : __builtin_bit_cast(float, __builtin_bit_cast(unsigned, av) & absmask);
        p[i] = c == 0 ? x * qv[c][i] : fmaf(x, qv[c][i], p[i]);
      }
    }
    f32x4 ex, hid, t4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      p[i] = group_sum<EW>(p[i]);
      ex[i] = __expf(p[i]);
      hid[i] = p[i] > 0.f ? p[i] : ex[i] - 1.f;                    // elu, alpha = 1
      // q_tot terms of the last tile's lanes: elu(a_e) |w2_e| (group 6), relu(h_e) wb2_e (group 7), 0 elsewhere
      const float z = is6 ? fabsf(acc[LT][i]) : fmaxf(acc[LT][i], 0.f);
      t4[i] = row_sum(z * (is6 ? hid[i] : cw));
    }
    if (m == 0) *reinterpret_cast<f32x4*>(&QT[wave][4 * q4]) = t4;
    ST_MARK(3);
    WG_BARRIER();
    ST_MARK(4);
    if (!BWD) {
      if (tid < 16) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += QT[w][tid];
        if (row0 + tid < nrows) a.q_tot[row0 + tid] = tot + bb2;
      }
      ST_MARK(5);
    } else {
      // ---- dL/dq_tot of rows 4q..4q+3 (every lane: the partials of all waves in fixed order)
      f32x4 qt = *reinterpret_cast<const f32x4*>(&QT[0][4 * q4]);
#pragma unroll
      for (int w = 1; w < NW; ++w) qt += *reinterpret_cast<const f32x4*>(&QT[w][4 * q4]);
      f32x4 gr;
      if (LOSS) {
        const f32x4 lr = *reinterpret_cast<const f32x4*>(&Ls2[buf][0][4 * q4]);
        const f32x4 lt = *reinterpret_cast<const f32x4*>(&Ls2[buf][1][4 * q4]);
        const f32x4 lp = *reinterpret_cast<const f32x4*>(&Ls2[buf][2][4 * q4]);
        const f32x4 lq = *reinterpret_cast<const f32x4*>(&Ls2[buf][3][4 * q4]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          qt[i] += bb2;
          const float mask = 1.f - lp[i];
          const float target = lr[i] + a.gamma * lq[i] * (1.f - lt[i]);
          const float mtd = mask * (target - qt[i]);
          gr[i] = -2.f * mask * mtd;
          acc_ln += mtd * mtd; acc_lm += mask;
        }
        if (a.q_tot && wave == 0 && m == 0) {
          if (row0 + 16 <= nrows) *reinterpret_cast<f32x4*>(&a.q_tot[row0 + 4 * q4]) = qt;
          else {
#pragma unroll
            for (int i = 0; i < 4; ++i) if (row0 + 4 * q4 + i < nrows) a.q_tot[row0 + 4 * q4 + i] = qt[i];
          }
        }
      } else gr = *reinterpret_cast<const f32x4*>(&Gs2[buf][4 * q4]);
      // ---- dL/da_e, d(hypernet output) in accumulator layout, partial dq
      f32x4 dpre, dhy[TPW];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float w2 = group_sum<EW>(is6 ? fabsf(acc[LT][i]) : 0.f);      // |w2_e| to every group of the row
        dpre[i] = gr[i] * w2 * (p[i] > 0.f ? 1.f : ex[i]);
        acc_wb2 += gr[i] * fmaxf(acc[LT][i], 0.f);                           // meaningful in the lanes of group 7
        acc_bb2 += gr[i];
      }
#pragma unroll
      for (int c = 0; c < TPW; ++c) {
        const bool lo_all = GPT * c + GPT - 1 <= NAG, hi_all = GPT * c > NAG;    // tile of agent / b1 columns, of w2 / h columns
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float x, y;
          if (lo_all) { x = qv[c][i]; y = dpre[i]; }
          else if (hi_all) { x = gr[i]; y = is6 ? hid[i] : wb2e; }
          else {
            const bool lo = grp[c] <= NAG;
            x = lo ? qv[c][i] : gr[i];
            y = lo ? dpre[i] : (is6 ? hid[i] : wb2e);
          }
          dhy[c][i] = (x * y) * sign_clamp(acc[c][i], lo_c[c]);
        }
        if (!ones) sbW[c] += dhy[c][0] + dhy[c][1] + dhy[c][2] + dhy[c][3];
        if (GPT * c < NAG) {                         // dq_n = sum_e |w1[n,e]| dL/da_e: this wave's EW elements
          f32x4 d4;
#pragma unroll
          for (int i = 0; i < 4; ++i) d4[i] = e_sum<EW>(fabsf(acc[c][i]) * dpre[i]);
          if (el == 0 && grp[c] < N) *reinterpret_cast<f32x4*>(&DQP[wave][grp[c]][4 * q4]) = d4;
        }
      }
      ST_MARK(5);
      __builtin_amdgcn_sched_barrier(0);
      // ---- dW += d(out)^T [s | 1]
      if (X6) {
        // contraction over the tile's 16 rows on v_mfma_f32_16x16x16_bf16 (k slot j of lane (g, .) = row 4g + j): the A operand is
        // split4 of d(out) as it sits in the accumulator layout; the B operand (rows 4g .. 4g+3 of state column 16 kc + lane) comes
        // from the row-major planes by transposed reads: lane (g, i = 4 qq + p) passes &plane[4g + qq][16 kc + 4p]
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        F3h dA[TPW];
#pragma unroll
        for (int c = 0; c < TPW; ++c) dA[c] = split4(dhy[c]);
        const short* tb = Sp + buf * PBUF + (4 * q4 + (m >> 2)) * PP + 4 * (m & 3);
#pragma unroll
        for (int kc = 0; kc < KCQ; ++kc) {
          F3h sb;
          sb.h = __builtin_bit_cast(i32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + 16 * kc)));
          sb.m = __builtin_bit_cast(i32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + PLN + 16 * kc)));
          sb.l = __builtin_bit_cast(i32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + 2 * PLN + 16 * kc)));
#define OP(p_, q_) _Pragma("unroll") for (int c = 0; c < TPW; ++c) accW[c][kc] = mmh(dA[c].p_, sb.q_, accW[c][kc]);
          OP(m, m) OP(h, l) OP(l, h) OP(h, m) OP(m, h) OP(h, h)
#undef OP
        }
      } else {
        const float* sd = &Ss[buf][(4 * q4) * SS + m];
#pragma unroll
        for (int kc = 0; kc < KCQ; ++kc) {
          f32x4 sD;
#pragma unroll
          for (int i = 0; i < 4; ++i) sD[i] = sd[i * SS + 16 * kc];
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < TPW; ++c) accW[c][kc] = mfma16(dhy[c][i], sD[i], accW[c][kc]);
        }
      }
      prow0 = row0;
    }
    ST_MARK(6);
    if (nt < tiles) stash(buf ^ 1);
    ST_MARK(7);
    // the next iteration's first barrier orders these LDS writes before their readers; QT / DQP are rewritten
    // only after that barrier too
  }
  ST_DUMP_AT(8, BWD ? 8 : 0);
  if (BWD) {
    __syncthreads();
    if (prow0 >= 0) flush_dq(prow0);
    float* slab = a.ws + (long)blockIdx.x * qmix_slab_floats(C, S);
    const int Sx = S + 1;
    // accumulator rows 4q+i of tile c are the columns (group GPT c + (4q+i) / EW, e = EW wave + (4q+i) % EW)
#pragma unroll
    for (int c = 0; c < TPW; ++c) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int cc = 4 * q4 + i, g = GPT * c + cc / EW, ee = EW * wave + (cc & (EW - 1));
        if (g < NAG && g >= N) continue;
        const int col = g < NAG ? g * E + ee : NE + (g - NAG) * E + ee;
#pragma unroll
        for (int kc = 0; kc < KCQ; ++kc) {
          const int k = 16 * kc + m;
          if (k < S || (ones && k == S)) slab[(long)col * Sx + k] = accW[c][kc][i];     // dW[col][k]; k == S: bias
        }
      }
      if (!ones) {
        float sb = sbW[c];
        sb += __shfl_xor(sb, 16, 64); sb += __shfl_xor(sb, 32, 64);                      // over the 4 row groups
        const int g = grp[c];
        if (q4 == 0 && !(g < NAG && g >= N)) slab[(long)(g < NAG ? g * E + e : NE + (g - NAG) * E + e) * Sx + S] = sb;
      }
    }
    // hyper_b2.2: weight gradient of e in the lanes of group 7 (rows 4q..4q+3 -> sum over q), bias / loss in wave 0
    float v = acc_wb2;
    v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    if (is7 && q4 == 0) slab[(long)C * Sx + e] = v;
    float b = acc_bb2, ln = acc_ln, lm = acc_lm;
    b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
    ln += __shfl_xor(ln, 16, 64); ln += __shfl_xor(ln, 32, 64);
    lm += __shfl_xor(lm, 16, 64); lm += __shfl_xor(lm, 32, 64);
    if (tid == 0) {
      slab[(long)C * Sx + E] = b;
      slab[(long)C * Sx + E + 1] = LOSS ? ln : 0.f;
      slab[(long)C * Sx + E + 2] = LOSS ? lm : 0.f;
    }
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
  if (e < slab) s = slab_sum(a.ws + e, slab, sg, RSG, a.nwg);
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

inline bool supported(int N, int S, int Eq) { return Eq == E && N >= 1 && N <= NAG && S <= 128 && S >= 1; }

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

inline unsigned grid_for(long rows, int per_cu = 1) {
  long tiles = (rows + 15) / 16;
  return (unsigned)(tiles < 256 * per_cu ? tiles : 256 * per_cu);
}

}  // namespace

extern "C" int marl_qmix_fused_supported(int N, int S, int Eq) { return supported(N, S, Eq) ? 1 : 0; }

extern "C" size_t marl_qmix_fused_workspace(long rows, int N, int S) {
  return (size_t)grid_for(rows) * qmix_slab_floats(N * E + 3 * E, S) * sizeof(float);
}

static int qmix_fwd_impl(const marl_qmix_weights_t* w, const marl_src_t* s, const float* q, float* q_tot,
                         long rows, int N, int S, int Eq, bool x6, void* stream) {
  if (rows <= 0) return 0;
  if (!supported(N, S, Eq)) return (int)hipErrorInvalidValue;
  QmixArgs a;
  if (fill(a, w, s, q, rows, N, S)) return (int)hipErrorInvalidValue;
  a.g = nullptr; a.q_tot = q_tot; a.dq = nullptr; a.ws = nullptr;
  a.lr = a.lterm = a.lpadded = a.lq_tgt = nullptr; a.gamma = 0.f;
  if (x6) hipLaunchKernelGGL((qmix_fused_kernel<false, 8, false, true>), dim3(grid_for(rows, 1)), dim3(512), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((qmix_fused_kernel<false, 4>), dim3(grid_for(rows, 2)), dim3(256), 0, (hipStream_t)stream, a);
  MARL_CHECK_LAUNCH();
  return 0;
}
extern "C" int marl_qmix_fused_fwd(const marl_qmix_weights_t* w, const marl_src_t* s, const float* q, float* q_tot,
                                   long rows, int N, int S, int Eq, void* stream) {
  return qmix_fwd_impl(w, s, q, q_tot, rows, N, S, Eq, false, stream);
}
extern "C" int marl_qmix_fused_fwd_x6(const marl_qmix_weights_t* w, const marl_src_t* s, const float* q, float* q_tot,
                                      long rows, int N, int S, int Eq, void* stream) {
  return qmix_fwd_impl(w, s, q, q_tot, rows, N, S, Eq, true, stream);
}

static int qmix_bwd_launch(QmixArgs& a, const marl_qmix_weights_t* grads, float* loss2, float* ws, long rows, int N, int S,
                           bool loss, bool x6, hipStream_t st) {
  const unsigned nwg = grid_for(rows);
  if (x6) {
    if (loss) hipLaunchKernelGGL((qmix_fused_kernel<true, 8, true, true>), dim3(nwg), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((qmix_fused_kernel<true, 8, false, true>), dim3(nwg), dim3(512), 0, st, a);
  } else if (loss) hipLaunchKernelGGL((qmix_fused_kernel<true, 8, true>), dim3(nwg), dim3(512), 0, st, a);
  else hipLaunchKernelGGL((qmix_fused_kernel<true, 8, false>), dim3(nwg), dim3(512), 0, st, a);
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

static int qmix_bwd_impl(const marl_qmix_weights_t* w, const marl_src_t* s, const float* q, const float* dq_tot, float* dq,
                         const marl_qmix_weights_t* grads, float* ws, size_t ws_bytes, long rows, int N, int S, int Eq, bool x6,
                         void* stream) {
  if (rows <= 0) return 0;
  if (!supported(N, S, Eq)) return (int)hipErrorInvalidValue;
  if (ws_bytes < marl_qmix_fused_workspace(rows, N, S)) return (int)hipErrorInvalidValue;
  QmixArgs a;
  if (fill(a, w, s, q, rows, N, S)) return (int)hipErrorInvalidValue;
  a.g = dq_tot; a.q_tot = nullptr; a.dq = dq; a.ws = ws;
  a.lr = a.lterm = a.lpadded = a.lq_tgt = nullptr; a.gamma = 0.f;
  return qmix_bwd_launch(a, grads, nullptr, ws, rows, N, S, false, x6, (hipStream_t)stream);
}
extern "C" int marl_qmix_fused_bwd(const marl_qmix_weights_t* w, const marl_src_t* s, const float* q,
                                   const float* dq_tot, float* dq, const marl_qmix_weights_t* grads, float* ws,
                                   size_t ws_bytes, long rows, int N, int S, int Eq, void* stream) {
  return qmix_bwd_impl(w, s, q, dq_tot, dq, grads, ws, ws_bytes, rows, N, S, Eq, false, stream);
}
extern "C" int marl_qmix_fused_bwd_x6(const marl_qmix_weights_t* w, const marl_src_t* s, const float* q,
                                      const float* dq_tot, float* dq, const marl_qmix_weights_t* grads, float* ws,
                                      size_t ws_bytes, long rows, int N, int S, int Eq, void* stream) {
  return qmix_bwd_impl(w, s, q, dq_tot, dq, grads, ws, ws_bytes, rows, N, S, Eq, true, stream);
}

static int qmix_loss_bwd_impl(const marl_qmix_weights_t* w, const marl_src_t* s, const float* q,
                              const float* q_tot_tgt, const float* r, const float* term, const float* padded,
                              float gamma, float* q_tot, float* dq, const marl_qmix_weights_t* grads,
                              float* loss2, float* ws, size_t ws_bytes, long rows, int N, int S, int Eq, bool x6,
                              void* stream) {
  if (rows <= 0) return 0;
  if (!supported(N, S, Eq) || !q_tot_tgt || !r || !term || !padded || !loss2) return (int)hipErrorInvalidValue;
  if (reinterpret_cast<uintptr_t>(q_tot) & 15) return (int)hipErrorInvalidValue;       // written 16 bytes per lane
  if (ws_bytes < marl_qmix_fused_workspace(rows, N, S)) return (int)hipErrorInvalidValue;
  QmixArgs a;
  if (fill(a, w, s, q, rows, N, S)) return (int)hipErrorInvalidValue;
  a.g = nullptr; a.q_tot = q_tot; a.dq = dq; a.ws = ws;
  a.lr = r; a.lterm = term; a.lpadded = padded; a.lq_tgt = q_tot_tgt; a.gamma = gamma;
  return qmix_bwd_launch(a, grads, loss2, ws, rows, N, S, true, x6, (hipStream_t)stream);
}
extern "C" int marl_qmix_fused_loss_bwd(const marl_qmix_weights_t* w, const marl_src_t* s, const float* q,
                                        const float* q_tot_tgt, const float* r, const float* term, const float* padded,
                                        float gamma, float* q_tot, float* dq, const marl_qmix_weights_t* grads,
                                        float* loss2, float* ws, size_t ws_bytes, long rows, int N, int S, int Eq,
                                        void* stream) {
  return qmix_loss_bwd_impl(w, s, q, q_tot_tgt, r, term, padded, gamma, q_tot, dq, grads, loss2, ws, ws_bytes, rows, N, S, Eq, false, stream);
}
extern "C" int marl_qmix_fused_loss_bwd_x6(const marl_qmix_weights_t* w, const marl_src_t* s, const float* q,
                                           const float* q_tot_tgt, const float* r, const float* term, const float* padded,
                                           float gamma, float* q_tot, float* dq, const marl_qmix_weights_t* grads,
                                           float* loss2, float* ws, size_t ws_bytes, long rows, int N, int S, int Eq,
                                           void* stream) {
  return qmix_loss_bwd_impl(w, s, q, q_tot_tgt, r, term, padded, gamma, q_tot, dq, grads, loss2, ws, ws_bytes, rows, N, S, Eq, true, stream);
}
