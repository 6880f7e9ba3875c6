// Persistent GRU-agent unroll kernels for gfx950 (reference network/q_network.py:16-21,
// controller/share_params.py:84-168).  One launch runs all T steps.
//
// Decomposition (MI355X-first, not one-thread-per-row):
//   * a workgroup owns a block of (episode, agent) rows = RT tiles of 16 rows and keeps them
//     for the whole unroll; 512 threads = 8 waves = 2 teams x 4 hidden-unit slices, two waves per SIMD;
//   * wave (team, w) owns hidden units [16w, 16w+16): its slice of W_ih / W_hh / W_2 lives in
//     VGPRs for the whole kernel as MFMA B-fragments (fc1's slice lives in LDS because its K
//     depends on the map), so the only per-step operand traffic is the activation tiles in LDS;
//     the teams split the row tiles (forward) or the products (backward);
//   * all products run on v_mfma_f32_16x16x4_f32 (exact fp32, the fp32 roofline of the chip);
//   * every thread carries 1/512 of the NEXT step's observation tile in float4 prefetch registers, issued a full
//     step ahead (there is no loader wave), and the one-hot(last action) / agent-id columns are kept in place in the
//     LDS input tile, so HBM latency never sits on the recurrent critical path;
//   * GRU pointwise math is done on the accumulator (D) layout in registers; activations that the
//     backward pass needs are written once, one 16-byte store per lane and plane (tile layout, below).
// LDS tile pitches of THIS file: +4 floats.  The +8 of common.h halves the bank-conflict share of the b128 fragment reads
// (tools/lds_pitch.py) but is time-neutral on 2s3z-sized tiles and COSTS the wide ones: QMIX on MMM2 / 1024 envs 159.6 -> 169.2
// updates/s with +4 here, QTRAN-base 3s5z 282 -> 284, QMIX 2s3z within +-0.3 % (same box, alternating: profiles/archive/r03_prescale_ab.txt, 8).
#ifndef MARL_PAD_H
#define MARL_PAD_H 4
#define MARL_PAD_K 4
#define MARL_PAD_G 4
#endif
#include "common.h"
#include <cstdlib>
#include "../../include/marl_hip.h"

namespace {

constexpr int H = 64;
constexpr int HS = H + MARL_PAD_H;      // LDS row stride of 64-wide tiles (floats); +4 spreads banks

struct FwdArgs {
  const float *W1, *b1, *Wih, *Whh, *bih, *bhh, *W2, *b2;
  const float* obs;       // row (b,t,n) at (b*obs_bs + (t+obs_t0)*N + n)*O
  long obs_bs; int obs_t0;
  const int* ufed;        // action fed back at step t: ufed[b*u_bs + (t+u_t0)*N + n]; <0 / t+u_t0<0 = none
  long u_bs; int u_t0;
  const int* ep_len;      // per-episode length or null: observations of steps t >= ep_len[b] read as zero
  const int* ep_map;      // per-episode storage index of obs or null (replay samples read in place)
  const float* h0;        // (B*N,64) or null (zeros)
  float* q;               // (B,T,N,A)
  float* hs;              // (B,T,N,64) or null
  float* h_last;          // (B*N,64) or null
  float* saved;           // [T][B*N][6][64]: hprev,x,r,z,n,hn per row-step (time-major), or null
  int B, T, N, O, A, I, KC, RT;
  int has_act, has_id;
  int vload;              // obs rows are 16-B aligned multiples of 4 floats: vector prefetch path
  float* gi_out;          // [T][R][3][64] or null: the input-side gate sums bias + x W_ih (r | z | n) of every row-step are stored
  const float* gi_in;     // XS kernels: gi_out of an earlier unroll of the SAME weights whose step t+1 input is this unroll's step t input
  long R;                 // B*N rows
};

// workgroup barrier that only drains LDS traffic: global stores (saved activations) and the
// prefetch loads stay in flight across it (a __syncthreads() would wait vmcnt(0) every time)
#define WG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

constexpr int FNT = 512;       // 8 waves: two teams of 4, two waves per SIMD
constexpr int NLDW = 4;        // float4 prefetch registers per thread (one step's obs tile per workgroup)

// ---------------------------------------------------------------------------------------------
// 8 compute waves = 2 teams x 4 hidden-unit slices.  Team k owns row tiles k, k+2, ...; both teams
// keep the same weight fragments in registers.  Two waves share each SIMD, so one wave's pointwise
// GRU math / LDS waits / global stores overlap the other's MFMAs (measured with one wave per SIMD:
// MFMA pipe 40 % busy, 28 % of wave time parked, 30 % VALU).  Every thread also carries 1/512 of the
// NEXT step's observation tile in 4 float4 registers, issued a full step ahead.
// Saved activations (and the input-side gate sums), TILE layout: [T][row tile][plane][column tile c][lane = 16 q + m][i].
// Lane (q, m) of a wave holds rows 4q+i (i = 0..3) at column 16c+m of a 16-row tile in its accumulator registers, so every
// plane of a tile is ONE 16-byte store / load per lane (1 KB per wave-instruction, fully coalesced) in the forward AND in the
// backward kernels - the row-major layout cost four 4-byte accesses per plane (36 stores per row tile in the saving unroll,
// 40 loads per row tile in BPTT; on a part where fp32 MFMAs and everything else issue one after the other that is time).
// Row tiles are global (row / 16), so kernels with different rows-per-workgroup agree; rows past the batch in the last tile
// have slots of their own.
__device__ __forceinline__ long sv_off(long tile_t /* t * n_tiles + tile */, int planes, int plane, int c, int lane) {
  return ((tile_t * planes + plane) * 4 + c) * 256 + lane * 4;
}

// store helpers: uniform base pointer (SGPR pair) + 32-bit byte offset -> saddr-form global stores,
// one v_lshl_add per element instead of 64-bit address arithmetic per store
__device__ __forceinline__ void st32(float* base, unsigned byte_off, float v) {
  *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// NL: float4 prefetch registers per thread for the next step's observation tile (4; 6 for wide observations - MMM2's
// O = 176 - where 4 would cap a workgroup at two row tiles and push a 640-tile shard into a second round of workgroups)
// XS: everything that depends only on the INPUT of a step - fc1 and the input-side gate sums bias + x W_ih, 288 of a row tile's
// 496 multiplies per step - is READ from what another unroll stored (a.gi_in = that unroll's gi_out, its step t+1) instead of
// being recomputed: the double-Q pass of a Q-learning update (q_learner.py:104-110) feeds the eval network the observations of
// steps 1..T right after the eval pass fed it steps 0..T-1 - same weights, same inputs, and every GRU kernel accumulates the
// input-side products before the hidden-side ones, so the stored sums are exactly the accumulators this pass would hold after
// its own input-side products (bit-identical).  Exceptions, where the whole workgroup computes the step as usual: the last
// step (a new observation), and any step t at which one of its rows has ep_len - 1 == t - there the earlier unroll (its step
// t+1 = ep_len) saw the zero padding and this one sees the final observation.  The observation prefetch therefore keeps
// running; what is saved is the multiplies.
// HALF: wide observations (MMM2: O = 176) - the NL prefetch registers hold one COLUMN HALF of the next step's observation
// tile at a time: the left half is committed (and the right half's loads issued) after the first barrier of a step, the right
// half committed (and the next step's left half issued) at the end of the gate phase - both inside the window in which the
// input tile may be rewritten, both with a gate phase or more between issue and use.  The same registers then cover twice
// the rows (a six-register variant spilled).  Non-saving unrolls only.
// DMA: the observation part of the input tile is filled by LDS-DMA (global_load_lds_dwordx4: global -> LDS with no register
// destination) instead of through prefetch registers.  The register path keeps one step's observation tile of the WORKGROUP
// in NL x 512 float4 registers, which caps the rows of a workgroup at 2048 * NL / O floats - for wide observations (MMM2:
// O = 176) two row tiles, i.e. a 640-tile shard runs as 320 workgroups = two rounds on 256 CUs.  With DMA a workgroup takes
// as many row tiles as its LDS holds (MMM2: three -> 214 workgroups, one round).  One wave-instruction writes 64 consecutive
// 16-byte slots of the tile image (M0 base + lane * 16) from per-lane source addresses: lane -> (row, column group) of the
// slot it covers, slots of the one-hot / agent-id / padding columns are masked off, rows past their episode end are zero-filled
// with a plain LDS store.  The waves of team 1 issue the DMAs of step t+1 right after the barrier that ends fc1(t) and wait for
// them (s_waitcnt vmcnt(0)) before the barrier that ends the gate phase: that team has the fewer row tiles (or as many), so
// the wait - which also covers its activation stores - falls into its slack.  hipcc inserts no waits of its own around an LDS-DMA
// (checked in the ISA); the two explicit ones below are what orders it.
// W2L: the fc2 fragments live in LDS instead of 16 * AC registers (read once per row tile in the short fc2 phase).  The
// registers pay for a wider observation prefetch (NL = 6): the activation-saving unroll of wide observations with two action
// tiles (MMM2: O = 176, A = 18) was held to two row tiles per workgroup by its four prefetch registers - 640 row tiles ran as
// 320 workgroups, two rounds on 256 CUs; with three tiles per workgroup it is 214 workgroups, one round.
template <int AC, bool SAVE, bool VL, int NL = NLDW, bool XS = false, bool HALF = false, bool DMA = false, bool W2L = false>
__global__ __launch_bounds__(FNT, 2) void agent_fwd_kernel(FwdArgs a) {
  static_assert(!XS || (VL && !SAVE && NL == NLDW && !HALF), "XS: vector path, no saving");
  static_assert(!HALF || VL, "HALF: vector path");
  static_assert(!DMA || (VL && !XS && !HALF), "DMA: vector path, plain schedule");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int team = wave >> 2, ws = wave & 3;
  const int q = lane >> 4, m = lane & 15;
  const long NTILES = (a.R + 15) >> 4;               // global 16-row tiles (saved-activation layout)
  const int RTW = (int)((NTILES - (long)blockIdx.x * a.RT) < a.RT ? (NTILES - (long)blockIdx.x * a.RT) : a.RT);   // REAL row tiles of this workgroup: the last one may hold fewer (whole tiles past the batch are not processed)
  const int rows = a.RT * 16;
  const int KP = a.KC * 16, KS = KP + MARL_PAD_K;
  // LDS carve (all offsets multiples of 4 floats)
  float* In = smem;                                   // [rows][KS]  (first: the LDS-DMA destinations stay below 64 KB)
  float* W1s = In + rows * KS;                        // [4][KC][64] f32x4
  float* Xt = W1s + 4 * a.KC * 64 * 4;                // [rows][HS]
  float* Ha = Xt + rows * HS;                         // [rows][HS] x2
  float* Hb = Ha + rows * HS;
  long* rowobs = reinterpret_cast<long*>(Hb + rows * HS);    // [rows]: (b*obs_bs + n) * O
  long* rowu = rowobs + rows;                                // [rows]: b*u_bs + n
  int* rowidx = reinterpret_cast<int*>(rowu + rows);         // [rows]: output row b*T*N + n
  int* rown = rowidx + rows;                                 // [rows]: n
  int* rowlen = rown + rows;                                 // [rows]: episode length (INT_MAX if none)
  int* rowrho = rowlen + rows;                               // [rows]: b*N + n
  int* xmask = rowrho + rows;                                // [T] (XS): step t is computed in full - the last step, or some row of this workgroup has ep_len - 1 == t
  int* ulds = xmask + ((a.T + 3) & ~3);                      // [2][rows] (DMA): actions fed at a step, by step parity
  float* W2s = reinterpret_cast<float*>(ulds + 2 * rows);    // [AC][4][64] f32x4 (W2L): fc2 fragments

  // Rows past the end of the batch (last workgroup only) are CLAMPED to the last valid row: they load
  // the same inputs, compute the same values and store them to the same addresses, so no per-lane
  // validity predicate is needed anywhere in the step loop.
  const long row0 = (long)blockIdx.x * rows;
  for (int r = tid; r < rows; r += FNT) {
    long rho = row0 + r;
    if (rho > a.R - 1) rho = a.R - 1;
    const long b = rho / a.N;
    const int n = (int)(rho % a.N);
    rowidx[r] = (int)(b * a.T * a.N + n);
    rowobs[r] = ((a.ep_map ? (long)a.ep_map[b] : b) * a.obs_bs + n) * a.O;
    rowu[r] = b * a.u_bs + n;
    rown[r] = n;
    rowlen[r] = a.ep_len ? a.ep_len[b] : 0x7fffffff;
    rowrho[r] = (int)rho;
  }
  __syncthreads();
  for (int e = tid; e < rows * H; e += FNT) {       // initial hidden tile
    int r = e / H, k = e % H;
    Ha[r * HS + k] = a.h0 ? a.h0[(long)rowrho[r] * H + k] : 0.f;
  }

  // ---- input tile [obs | onehot(ufed) | id | 0-pad]
  const int O = a.O;
  const int O4 = HALF ? O >> 3 : O >> 2, n4 = rows * O4;          // float4 groups per row (HALF: of one column half)
  const int hoff = HALF ? 4 * O4 : 0;                               // float offset of the right column half
  const float invO4 = 1.0f / (float)(O4 > 0 ? O4 : 1);
  f32x4 pf[NL];
  int pt = 0;                         // step the prefetch registers belong to
  int pu = -1, pu_lds = -1;           // action fed at the step being prefetched / one-hot column currently set in LDS
  // the element -> (row, column group) map of the prefetch is the same every step: resolve it once
  long goff[NL]; int loff[NL], plen[NL];
  // DMA: 16-byte slots per row of the tile image, wave-instructions (64 slots) that cover it
  const int SPR = KS >> 2, nchunks = (rows * SPR + 63) >> 6;
  const float invSPR = 1.0f / (float)SPR;
  auto dma_fill = [&](int t, int k0, int kstep) {      // observations of step t -> In (k0, kstep: this wave's share of the chunks)
    const long toff = (long)(t + a.obs_t0) * a.N * O;
    for (int k = k0; k < nchunks; k += kstep) {
      const int sl = k * 64 + lane;
      const int r = (int)(((float)sl + 0.5f) * invSPR);
      const int c4 = sl - r * SPR;
      const bool act = r < rows && c4 < (O >> 2);
      const int rc = act ? r : 0;
      const float* src = a.obs + rowobs[rc] + 4 * c4 + toff;
      if (act) {
        if (t < rowlen[rc]) {
          __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(In + k * 256), 16, 0, 0);
        } else {
          *reinterpret_cast<f32x4*>(In + sl * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};      // steps past the episode end feed zeros
        }
      }
    }
  };
  // ... and the actions fed back (one per row and step) travel the same way, as 4-byte DMAs into a table: the step loop of the
  // DMA kernels then holds NO ordinary load, and hipcc - which cannot count vmcnt past an LDS-DMA and would wait vmcnt(0),
  // i.e. for every store of the wave, where such a load's result is used - inserts no wait of its own
  auto dma_u = [&](int t) {                            // actions fed at step t -> ulds[t & 1]  (waves of team 1)
    int* dst = ulds + (t & 1) * rows;
    const bool has = a.ufed && t + a.u_t0 >= 0;
    for (int k = ws; k * 64 < rows; k += 4) {
      const int r = k * 64 + lane;
      if (r < rows) {
        if (has) __builtin_amdgcn_global_load_lds(a.ufed + rowu[r] + (long)(t + a.u_t0) * a.N, (__attribute__((address_space(3))) void*)(dst + k * 64), 4, 0, 0);
        else dst[r] = -1;
      }
    }
  };
  auto flip_u = [&](int t) {                           // one-hot(last action) column of the input tile -> step t's
    if (a.has_act && tid < rows) {
      const int u = ulds[(t & 1) * rows + tid];
      const int pn = (u >= 0 && u < a.A) ? u : -1;
      if (pn != pu_lds) {
        if (pu_lds >= 0) In[tid * KS + O + pu_lds] = 0.f;
        if (pn >= 0) In[tid * KS + O + pn] = 1.f;
        pu_lds = pn;
      }
    }
  };
  if (!DMA)
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    // elements past the tile are clamped to its last one (same value, same address): branch-free step loop
    int e = tid + FNT * i;
    if (e > n4 - 1) e = n4 - 1;
    if (e < 0) e = 0;
    const int r = (int)(((float)e + 0.5f) * invO4);
    const int k4 = e - r * O4;
    loff[i] = r * KS + 4 * k4;
    goff[i] = rowobs[r] + 4 * k4;
    plen[i] = rowlen[r];
  }
  const long urow = tid < rows ? rowu[tid] : 0;
  // XS: the stored input-side sums of the NEXT row tile this wave will process, in accumulator layout (rows 4q+i, column j);
  // issued a whole tile ahead and moved into the accumulators at the top of the tile (by then they have landed)
  f32x4 gB[3];
  auto gissue = [&](int ts, int rt) {
    const float* gp = a.gi_in + sv_off((long)ts * NTILES + (long)blockIdx.x * a.RT + rt, 3, 0, ws, lane);
    gB[0] = *reinterpret_cast<const f32x4*>(gp);
    gB[1] = *reinterpret_cast<const f32x4*>(gp + 1024);
    gB[2] = *reinterpret_cast<const f32x4*>(gp + 2 * 1024);
  };
  const int mylen = tid < rows ? rowlen[tid] : 0x7fffffff;
  if (XS) for (int e = tid; e < a.T; e += FNT) xmask[e] = e == a.T - 1 ? 1 : 0;      // (before the barrier of the constant columns)
  auto issue = [&](int t, int h = 0) {   // start the loads of step t's observations (vector path; h: column half)
    const long toff = (long)(t + a.obs_t0) * a.N * O + (h ? hoff : 0);
    if (!DMA) {
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        // always loaded (the record has every slot); steps past the episode end are zeroed at commit
        pf[i] = *reinterpret_cast<const f32x4*>(a.obs + goff[i] + toff);
      }
    }
    pt = t;
    if (h) return;
    int u = -1;
    if (tid < rows && a.ufed && t + a.u_t0 >= 0) u = a.ufed[urow + (long)(t + a.u_t0) * a.N];
    pu = u;
  };
  auto commit = [&](int h = 0) {      // registers -> LDS tile; the one-hot(last action) column is flipped in place
    if (!DMA) {
#pragma unroll
      for (int i = 0; i < NL; ++i)
        *reinterpret_cast<f32x4*>(In + loff[i] + (h ? hoff : 0)) = pt < plen[i] ? pf[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if (XS && n4 > NL * FNT) {
      // the reuse variant refills the input tile for the few steps it computes in full only, so it is not held to the rows
      // the prefetch registers cover: the rest of a larger tile (wide observations) is fetched here, synchronously
      const long toff = (long)(pt + a.obs_t0) * a.N * O;
      for (int e = NL * FNT + tid; e < n4; e += FNT) {
        const int r = (int)(((float)e + 0.5f) * invO4);
        const int k4 = e - r * O4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (pt < rowlen[r]) v = *reinterpret_cast<const f32x4*>(a.obs + rowobs[r] + 4 * k4 + toff);
        *reinterpret_cast<f32x4*>(In + r * KS + 4 * k4) = v;
      }
    }
    if (h) return;
    if (a.has_act && tid < rows) {
      const int pn = (pu >= 0 && pu < a.A) ? pu : -1;
      if (pn != pu_lds) {
        if (pu_lds >= 0) In[tid * KS + O + pu_lds] = 0.f;
        if (pn >= 0) In[tid * KS + O + pn] = 1.f;
        pu_lds = pn;
      }
    }
  };
  // constant columns of the input tile: empty one-hot, agent id, zero pad (vector path; written once)
  if (VL) {
    for (int e = tid; e < rows * (KP - O); e += FNT) {
      const int r = e / (KP - O), k = O + e % (KP - O);
      float v = 0.f;
      if (a.has_id && k >= a.I - a.N && k < a.I) v = (rown[r] == k - (a.I - a.N)) ? 1.f : 0.f;
      In[r * KS + k] = v;
    }
    __syncthreads();
  }
  auto load_generic = [&](int t) {    // element loads, no run-ahead (obs width not a multiple of 4 / unaligned)
    for (int e = tid; e < rows * KP; e += FNT) {
      const int r = e / KP, k = e - r * KP;
      float v = 0.f;
      if (k < O) {
        if (t < rowlen[r]) v = a.obs[rowobs[r] + (long)(t + a.obs_t0) * a.N * O + k];
      } else if (a.has_act && k < O + a.A) {
        int u = -1;
        if (a.ufed && t + a.u_t0 >= 0) u = a.ufed[rowu[r] + (long)(t + a.u_t0) * a.N];
        v = (u == k - O) ? 1.f : 0.f;
      } else if (a.has_id && k >= a.I - a.N && k < a.I) {
        v = (rown[r] == k - (a.I - a.N)) ? 1.f : 0.f;
      }
      In[r * KS + k] = v;
    }
  };
  // NOTE the prefetch is issued UNCONDITIONALLY every step (the step index is clamped): a conditional issue makes
  // the prefetch registers a phi of (loaded, old) and the compiler then drains vmcnt right after the loads to copy
  if (HALF) { issue(0, 0); commit(0); issue(0, 1); commit(1); issue(a.T > 1 ? 1 : 0, 0); }
  else if (VL) {
    if (DMA) {                                // (waited for below, in front of the barrier that publishes the staged weights)
      dma_fill(0, wave, 8);
      if (team == 1) { dma_u(0); dma_u(a.T > 1 ? 1 : 0); }
    } else { issue(0); commit(); issue(a.T > 1 ? 1 : 0); }
  }
  else load_generic(0);
  if (XS) {
    if (team < RTW) gissue(1, team);
    if (mylen >= 1 && mylen - 1 < a.T) xmask[mylen - 1] = 1;      // visible after the barrier below
  }

  // ---- stage weights: fc1 slice -> LDS fragments (team 0 writes, both teams read); GRU / fc2 -> registers
  f32x4 wih[3][4], whh[3][4], w2[AC][4];
  float bias_r, bias_z, bias_in, bias_hn, bias1, bias2[AC];
  const int j = 16 * ws + m;
  {
    if (team == 0) {
      for (int c = 0; c < a.KC; ++c) {
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int k = 16 * c + 4 * q + i;
          v[i] = k < a.I ? a.W1[(long)j * a.I + k] : 0.f;
        }
        *reinterpret_cast<f32x4*>(W1s + ((ws * a.KC + c) * 64 + lane) * 4) = v;
      }
    }
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        wih[g][c] = *reinterpret_cast<const f32x4*>(a.Wih + (long)(g * H + j) * H + 16 * c + 4 * q);
        whh[g][c] = *reinterpret_cast<const f32x4*>(a.Whh + (long)(g * H + j) * H + 16 * c + 4 * q);
      }
#pragma unroll
    for (int ac = 0; ac < AC; ++ac) {
      int arow = 16 * ac + m; if (arow >= a.A) arow = a.A - 1;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(a.W2 + (long)arow * H + 16 * c + 4 * q);
        if (W2L) { if (wave == 0) *reinterpret_cast<f32x4*>(W2s + ((ac * 4 + c) * 64 + lane) * 4) = wv; }
        else w2[ac][c] = wv;
      }
      bias2[ac] = a.b2[arow];
    }
    bias1 = a.b1[j];
    bias_r = a.bih[j] + a.bhh[j];
    bias_z = a.bih[H + j] + a.bhh[H + j];
    bias_in = a.bih[2 * H + j];
    bias_hn = a.bhh[2 * H + j];
    if (AC == 2) gru_prescale(wih, whh, bias_r, bias_z, bias_in, bias_hn);      // gate math on pre-scaled accumulators: see GRU_PRE in common.h
  }
  if (DMA) {
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's DMAs have landed
    WG_BARRIER();
    flip_u(0);
  }
  WG_BARRIER();   // input tile of step 0 and the fc1 fragments are in LDS

  // uniform base pointers of the per-step outputs; per-element offsets are 32-bit byte offsets
  // saved activations: [T][R][6][64] (time-major, the 6 planes of one row-step contiguous): one base
  // pointer per step + a per-row 32-bit offset; the plane is an immediate offset of the store
  const unsigned jb = (unsigned)j * 4u;
  float* Hp = Ha;
  float* Hn = Hb;
#ifdef MARL_PRIO_YOUNG
  if (wave >= 4) __builtin_amdgcn_s_setprio(MARL_PRIO_YOUNG);      // A/B: static priority for the younger half of the workgroup
#endif
  ST_DECL(6);
  for (int t = 0; t < a.T; ++t) {
    const unsigned trow = (unsigned)t * (unsigned)a.N;
    const long svt = SAVE ? (long)t * NTILES + (long)blockIdx.x * a.RT : 0;      // (step, first row tile of this workgroup)
    // ---------------- phase 1: x = relu(fc1(in))  (two of the team's row tiles in flight)
    const bool xread = XS && xmask[t] == 0;       // this step's input-side work is read, not computed
    for (int rt = xread ? RTW : team; rt < RTW; rt += 4) {
      const bool two = rt + 2 < RTW;
      f32x4 acc0 = {bias1, bias1, bias1, bias1}, acc1 = acc0;
      const float* in0 = In + (rt * 16 + m) * KS + 4 * q;
      const float* in1 = in0 + 32 * KS;
      const float* wf = W1s + (ws * a.KC * 64 + lane) * 4;
      for (int c = 0; c < a.KC; ++c) {
        f32x4 bv = *reinterpret_cast<const f32x4*>(wf + c * 256);
        f32x4 a0 = *reinterpret_cast<const f32x4*>(in0 + 16 * c);
        if (two) {
          f32x4 a1 = *reinterpret_cast<const f32x4*>(in1 + 16 * c);
          mfma16x4_il2(a0, bv, acc0, a1, bv, acc1);
        } else {
          acc0 = mfma16x4(a0, bv, acc0);
        }
      }
      const int r0 = rt * 16 + 4 * q;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc0[i] = fmaxf(acc0[i], 0.f);
        Xt[(r0 + i) * HS + j] = acc0[i];
      }
      if (SAVE) *reinterpret_cast<f32x4*>(a.saved + sv_off(svt + rt, 6, 1, ws, lane)) = acc0;
      if (two) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc1[i] = fmaxf(acc1[i], 0.f);
          Xt[(r0 + 32 + i) * HS + j] = acc1[i];
        }
        if (SAVE) *reinterpret_cast<f32x4*>(a.saved + sv_off(svt + rt + 2, 6, 1, ws, lane)) = acc1;
      }
    }
    ST_MARK(0);
    // XS, step read rather than computed: nothing touched the input tile or Xt, so there is nothing for this barrier to
    // order (one barrier per step: the hidden tiles are double-buffered, see the note at the end of the loop)
    if (!xread) WG_BARRIER();
    ST_MARK(1);
    // the input tile has been consumed: refill it for step t+1, start the loads of step t+2
    if (XS) {
      // only the steps computed in full read the input tile: its refill and the observation loads run for those alone
      // (conditional loads cost a register copy + wait where they are issued - here that is the rare path)
      if (t + 1 < a.T && xmask[t + 1]) commit();
      if (t + 2 < a.T && xmask[t + 2]) issue(t + 2);
    } else
    if (HALF) {
      commit(0);                                  // left half of step t+1's input; its right half travels during the gates
      issue(t + 1 < a.T ? t + 1 : a.T - 1, 1);
    } else if (VL) {
      if (DMA) {
        flip_u(t + 1 < a.T ? t + 1 : a.T - 1);
        if (team == 1) {                          // observations of step t+1 and actions of step t+2, in flight during the gates
          // waves 4-7 are the younger half of the workgroup and lose the issue arbitration against their SIMD partners'
          // MFMA streams (profiles/archive/r03_phase_probe.txt): without the priority the ~40 address instructions per DMA crawl
          // through the partners' gate phase (stamps: 45 % of a step for nine DMAs)
          __builtin_amdgcn_s_setprio(3);
          dma_fill(t + 1 < a.T ? t + 1 : a.T - 1, ws, 4);
          dma_u(t + 2 < a.T ? t + 2 : a.T - 1);
          __builtin_amdgcn_s_setprio(0);
        }
      } else {
        commit();                                 // (after the last step this writes a tile nobody reads)
        issue(t + 2 < a.T ? t + 2 : a.T - 1);
      }
    } else if (t + 1 < a.T) {
      load_generic(t + 1);
    }
    ST_MARK(2);
    // ---------------- phase 2: GRU gates + pointwise update
    const bool last = (t == a.T - 1) && a.h_last;
    for (int rt = team; rt < RTW; rt += 2) {
      f32x4 ar = {bias_r, bias_r, bias_r, bias_r};
      f32x4 az = {bias_z, bias_z, bias_z, bias_z};
      f32x4 ain = {bias_in, bias_in, bias_in, bias_in};
      f32x4 ahn = {bias_hn, bias_hn, bias_hn, bias_hn};
      const float* xr = Xt + (rt * 16 + m) * HS + 4 * q;
      const float* hr = Hp + (rt * 16 + m) * HS + 4 * q;
      const int r0 = rt * 16 + 4 * q;
      if (XS) {
        if (xread) { ar = gB[0]; az = gB[1]; ain = gB[2]; }
        // the tile after this one: the same step's next tile of this team, or the team's first tile of the next step (the
        // stored step index is one ahead of this unroll's; the last ones issued are never consumed)
        const bool same = rt + 2 < RTW;
        const int nts = same ? t + 1 : t + 2;
        gissue(nts < a.T ? nts : a.T - 1, same ? rt + 2 : team);
      }
      // input-side products first, hidden-side products after them (the SAME order in every GRU kernel: the input-side
      // sums bias + x W_ih of one unroll can then stand in for another unroll's - the XS variant)
      // (the candidate's hidden-side product h W_hn has an accumulator of its own and rides along with the input-side
      // products: four independent accumulation chains in the first loop as before the reordering, two in the second)
      if (!xread && AC == 2) {           // register-tight variant: the hidden-side operands are live in one loop only
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          f32x4 ax = *reinterpret_cast<const f32x4*>(xr + 16 * c);
          mfma16x4_il3(ax, wih[0][c], ar, ax, wih[1][c], az, ax, wih[2][c], ain);
        }
      } else if (!xread) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          f32x4 ax = *reinterpret_cast<const f32x4*>(xr + 16 * c);
          f32x4 ah = *reinterpret_cast<const f32x4*>(hr + 16 * c);
          mfma16x4_il4(ax, wih[0][c], ar, ax, wih[1][c], az, ax, wih[2][c], ain, ah, whh[2][c], ahn);
        }
      }
      // (XS, step read: the candidate's hidden-side product joins the other two below - three chains instead of one)
      if (SAVE && a.gi_out) {
        float* const gp = a.gi_out + sv_off(svt + rt, 3, 0, ws, lane);
        *reinterpret_cast<f32x4*>(gp) = ar;
        *reinterpret_cast<f32x4*>(gp + 1024) = az;
        *reinterpret_cast<f32x4*>(gp + 2 * 1024) = ain;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        f32x4 ah = *reinterpret_cast<const f32x4*>(hr + 16 * c);
        if (xread || AC == 2) mfma16x4_il3(ah, whh[2][c], ahn, ah, whh[0][c], ar, ah, whh[1][c], az);
        else mfma16x4_il2(ah, whh[0][c], ar, ah, whh[1][c], az);
      }
      f32x4 vhp, vr, vz, vn, vh;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        vhp[i] = Hp[(r0 + i) * HS + j];
        float r_, z_, n_, h_;
        if (AC == 2) gru_point(ar[i], az[i], ain[i], ahn[i], vhp[i], r_, z_, n_, h_);
        else gru_point_plain(ar[i], az[i], ain[i], ahn[i], vhp[i], r_, z_, n_, h_);
        vr[i] = r_; vz[i] = z_; vn[i] = n_; vh[i] = h_;
        Hn[(r0 + i) * HS + j] = vh[i];
      }
      if (a.hs) {
        const i32x4 ri = *reinterpret_cast<const i32x4*>(rowidx + r0);
        st32(a.hs, ((unsigned)ri.x + trow) * 256u + jb, vh[0]); st32(a.hs, ((unsigned)ri.y + trow) * 256u + jb, vh[1]);
        st32(a.hs, ((unsigned)ri.z + trow) * 256u + jb, vh[2]); st32(a.hs, ((unsigned)ri.w + trow) * 256u + jb, vh[3]);
      }
      const i32x4 rr = *reinterpret_cast<const i32x4*>(rowrho + r0);
      if (SAVE) {
        float* const sp = a.saved + sv_off(svt + rt, 6, 0, ws, lane);       // plane k at sp + 1024 k
        *reinterpret_cast<f32x4*>(sp) = vhp;
        *reinterpret_cast<f32x4*>(sp + 2 * 1024) = vr;
        *reinterpret_cast<f32x4*>(sp + 3 * 1024) = vz;
        *reinterpret_cast<f32x4*>(sp + 4 * 1024) = vn;
        *reinterpret_cast<f32x4*>(sp + 5 * 1024) = AC == 2 ? ahn * MARL_INV_2LOG2E : ahn;      // BPTT wants W_hn h + b_hn itself
        // the hidden state AFTER the last step, where the backward pass looks for h(t): plane 0 of step t+1
        if (t == a.T - 1) *reinterpret_cast<f32x4*>(sp + NTILES * (6 * 1024)) = vh;
      }
      if (last) {
        st32(a.h_last, (unsigned)rr.x * 256u + jb, vh[0]); st32(a.h_last, (unsigned)rr.y * 256u + jb, vh[1]);
        st32(a.h_last, (unsigned)rr.z * 256u + jb, vh[2]); st32(a.h_last, (unsigned)rr.w * 256u + jb, vh[3]);
      }
    }
    if (HALF) {
      commit(1);                                  // right half of step t+1's input
      issue(t + 2 < a.T ? t + 2 : a.T - 1, 0);
    }
    if (DMA && team == 1) __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's DMAs have landed (and its stores)
    ST_MARK(3);
    WG_BARRIER();
    ST_MARK(4);
    // ---------------- phase 3: q = fc2(h')   (row tiles dealt round-robin to the 8 waves)
    for (int rt = wave; rt < RTW; rt += 8) {
      f32x4 acc[AC];
#pragma unroll
      for (int ac = 0; ac < AC; ++ac) acc[ac] = (f32x4){bias2[ac], bias2[ac], bias2[ac], bias2[ac]};
      const float* hr = Hn + (rt * 16 + m) * HS + 4 * q;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        f32x4 ah = *reinterpret_cast<const f32x4*>(hr + 16 * c);
#pragma unroll
        for (int ac = 0; ac < AC; ++ac)
          acc[ac] = mfma16x4(ah, W2L ? *reinterpret_cast<const f32x4*>(W2s + ((ac * 4 + c) * 64 + lane) * 4) : w2[ac][c], acc[ac]);
      }
      const i32x4 ri = *reinterpret_cast<const i32x4*>(rowidx + rt * 16 + 4 * q);
      const unsigned A4 = (unsigned)a.A * 4u;
#pragma unroll
      for (int ac = 0; ac < AC; ++ac) {
        const int col = 16 * ac + m;
        if (col < a.A) {
          const unsigned cb = (unsigned)col * 4u;
          st32(a.q, ((unsigned)ri.x + trow) * A4 + cb, acc[ac][0]);
          st32(a.q, ((unsigned)ri.y + trow) * A4 + cb, acc[ac][1]);
          st32(a.q, ((unsigned)ri.z + trow) * A4 + cb, acc[ac][2]);
          st32(a.q, ((unsigned)ri.w + trow) * A4 + cb, acc[ac][3]);
        }
      }
    }
    float* tmp = Hp; Hp = Hn; Hn = tmp;
    ST_MARK(5);
    // no barrier here: phase 1 of t+1 reads In (refilled before the 2nd barrier above) and writes Xt
    // (last read before it); Hn of step t is only re-written in phase 2 of t+2, two barriers later.
  }
  ST_DUMP(6);
}

// ---------------------------------------------------------------------------------------------
// Software-pipelined variant for ONE row tile per workgroup (the 512-env shards of the 8-GPU strong-scaling run):
// the step is a chain  h(t-1) -> GRU(t) -> h(t)  and with one or two tiles per workgroup nothing hides the LDS
// round trips and barrier skew around its three phases.  Only the GRU is on that chain: fc1(t+1) needs just the
// observation and fc2(t-1) just h(t-1), so here ONE barrier separates the steps and inside a step
//     team 0 : GRU(t) row tiles
//     team 1 : fc1(t+1) -> Xt[(t+1)&1],  q(t-1) = fc2(h(t-1)),  then joins the GRU tiles (per-slice counter)
// with the input tile and the fc1 output double-buffered in LDS (which is why it needs RT <= 3-4).
// Results are bit-identical to agent_fwd_kernel (same MFMA sequences per output element).
// XS (as in agent_fwd_kernel): the input-side work of steps 0..T-2 is read from what an earlier unroll stored (gi_in); fc1 runs
// only for the last step and for the steps flagged in xmask (a row with ep_len - 1 == t; the table is built at the start - team 1
// computes fc1 a step early); observation loads and input-tile refills run for those steps alone.  Team 0 alone walks the row tiles then (static order, so the sums can be prefetched).
template <int AC, bool SAVE, bool XS = false>
__global__ __launch_bounds__(FNT, 2) void agent_fwd_pipe_kernel(FwdArgs a) {
  static_assert(!XS || !SAVE, "XS: no saving");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave >> 2, ws = wave & 3;
  const int q = lane >> 4, m = lane & 15;
  const long NTILES = (a.R + 15) >> 4;               // global 16-row tiles (saved-activation layout)
  const int RTW = (int)((NTILES - (long)blockIdx.x * a.RT) < a.RT ? (NTILES - (long)blockIdx.x * a.RT) : a.RT);   // REAL row tiles of this workgroup: the last one may hold fewer (whole tiles past the batch are not processed)
  const int rows = a.RT * 16;
  const int KP = a.KC * 16, KS = KP + MARL_PAD_K;
  float* W1s = smem;                                  // [4][KC][64] f32x4
  float* In0 = W1s + 4 * a.KC * 64 * 4;               // [2][rows][KS]
  float* Xt0 = In0 + 2 * rows * KS;                   // [2][rows][HS]
  float* Ha = Xt0 + 2 * rows * HS;                    // [rows][HS] x2
  float* Hb = Ha + rows * HS;
  long* rowobs = reinterpret_cast<long*>(Hb + rows * HS);
  long* rowu = rowobs + rows;
  int* rowidx = reinterpret_cast<int*>(rowu + rows);
  int* rown = rowidx + rows;
  int* rowlen = rown + rows;
  int* rowrho = rowlen + rows;
  int* tilecnt = rowrho + rows;                       // [2][4]: next GRU tile of each hidden-unit slice, by step parity
  int* xmask = tilecnt + 8;                           // [T] (XS): step t is computed in full - the last step, or some row of this workgroup has ep_len - 1 == t

  const long row0 = (long)blockIdx.x * rows;
  for (int r = tid; r < rows; r += FNT) {
    long rho = row0 + r;
    if (rho > a.R - 1) rho = a.R - 1;                 // clamped duplicates (see agent_fwd_kernel)
    const long b = rho / a.N;
    const int n = (int)(rho % a.N);
    rowidx[r] = (int)(b * a.T * a.N + n);
    rowobs[r] = ((a.ep_map ? (long)a.ep_map[b] : b) * a.obs_bs + n) * a.O;
    rowu[r] = b * a.u_bs + n;
    rown[r] = n;
    rowlen[r] = a.ep_len ? a.ep_len[b] : 0x7fffffff;
    rowrho[r] = (int)rho;
  }
  if (tid < 8) tilecnt[tid] = 0;
  if (XS) for (int e = tid; e < a.T; e += FNT) xmask[e] = e == a.T - 1 ? 1 : 0;
  __syncthreads();
  for (int e = tid; e < rows * H; e += FNT) {
    int r = e / H, k = e % H;
    Ha[r * HS + k] = a.h0 ? a.h0[(long)rowrho[r] * H + k] : 0.f;
  }
  const int O = a.O;
  const int O4 = O >> 2, n4 = rows * O4;
  const float invO4 = 1.0f / (float)(O4 > 0 ? O4 : 1);
  f32x4 pf[NLDW];
  int pt = 0, pu = -1;
  int pu_lds0 = -1, pu_lds1 = -1;     // one-hot column currently set in each input buffer
  long goff[NLDW]; int loff[NLDW], plen[NLDW];
#pragma unroll
  for (int i = 0; i < NLDW; ++i) {
    int e = tid + FNT * i;
    if (e > n4 - 1) e = n4 - 1;
    if (e < 0) e = 0;
    const int r = (int)(((float)e + 0.5f) * invO4);
    const int k4 = e - r * O4;
    loff[i] = r * KS + 4 * k4;
    goff[i] = rowobs[r] + 4 * k4;
    plen[i] = rowlen[r];
  }
  const long urow = tid < rows ? rowu[tid] : 0;
  auto issue = [&](int t) {
    const long toff = (long)(t + a.obs_t0) * a.N * O;
#pragma unroll
    for (int i = 0; i < NLDW; ++i) pf[i] = *reinterpret_cast<const f32x4*>(a.obs + goff[i] + toff);
    pt = t;
    int u = -1;
    if (tid < rows && a.ufed && t + a.u_t0 >= 0) u = a.ufed[urow + (long)(t + a.u_t0) * a.N];
    pu = u;
  };
  auto commit = [&](float* In, int& pu_lds) {
#pragma unroll
    for (int i = 0; i < NLDW; ++i)
      *reinterpret_cast<f32x4*>(In + loff[i]) = pt < plen[i] ? pf[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.has_act && tid < rows) {
      const int pn = (pu >= 0 && pu < a.A) ? pu : -1;
      if (pn != pu_lds) {
        if (pu_lds >= 0) In[tid * KS + O + pu_lds] = 0.f;
        if (pn >= 0) In[tid * KS + O + pn] = 1.f;
        pu_lds = pn;
      }
    }
  };
  for (int e = tid; e < 2 * rows * (KP - O); e += FNT) {        // constant columns of both input buffers
    const int bsel = e / (rows * (KP - O)), e2 = e - bsel * rows * (KP - O);
    const int r = e2 / (KP - O), k = O + e2 % (KP - O);
    float v = 0.f;
    if (a.has_id && k >= a.I - a.N && k < a.I) v = (rown[r] == k - (a.I - a.N)) ? 1.f : 0.f;
    In0[bsel * rows * KS + r * KS + k] = v;
  }
  __syncthreads();
  const int Tm1 = a.T - 1;
  issue(0); commit(In0, pu_lds0);
  issue(Tm1 < 1 ? Tm1 : 1); commit(In0 + rows * KS, pu_lds1);
  issue(Tm1 < 2 ? Tm1 : 2);
  // XS: stored input-side sums of the next row tile team 0 will process (accumulator layout), a tile ahead; step flags
  f32x4 gB[3];
  auto gissue = [&](int ts, int rt) {
    const float* gp = a.gi_in + sv_off((long)ts * NTILES + (long)blockIdx.x * a.RT + rt, 3, 0, wave & 3, lane);
    gB[0] = *reinterpret_cast<const f32x4*>(gp);
    gB[1] = *reinterpret_cast<const f32x4*>(gp + 1024);
    gB[2] = *reinterpret_cast<const f32x4*>(gp + 2 * 1024);
  };
  const int mylen = tid < rows ? rowlen[tid] : 0x7fffffff;
  if (XS) {
    if (team == 0) gissue(Tm1 < 1 ? Tm1 : 1, 0);
    if (mylen >= 1 && mylen - 1 < a.T) xmask[mylen - 1] = 1;      // published by the barrier below (staged weights)
  }

  f32x4 wih[3][4], whh[3][4], w2[AC][4];
  float bias_r, bias_z, bias_in, bias_hn, bias1, bias2[AC];
  const int j = 16 * ws + m;
  {
    if (team == 0) {
      for (int c = 0; c < a.KC; ++c) {
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int k = 16 * c + 4 * q + i;
          v[i] = k < a.I ? a.W1[(long)j * a.I + k] : 0.f;
        }
        *reinterpret_cast<f32x4*>(W1s + ((ws * a.KC + c) * 64 + lane) * 4) = v;
      }
    }
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        wih[g][c] = *reinterpret_cast<const f32x4*>(a.Wih + (long)(g * H + j) * H + 16 * c + 4 * q);
        whh[g][c] = *reinterpret_cast<const f32x4*>(a.Whh + (long)(g * H + j) * H + 16 * c + 4 * q);
      }
#pragma unroll
    for (int ac = 0; ac < AC; ++ac) {
      int arow = 16 * ac + m; if (arow >= a.A) arow = a.A - 1;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        w2[ac][c] = *reinterpret_cast<const f32x4*>(a.W2 + (long)arow * H + 16 * c + 4 * q);
      bias2[ac] = a.b2[arow];
    }
    bias1 = a.b1[j];
    bias_r = a.bih[j] + a.bhh[j];
    bias_z = a.bih[H + j] + a.bhh[H + j];
    bias_in = a.bih[2 * H + j];
    bias_hn = a.bhh[2 * H + j];
    if (AC == 2) gru_prescale(wih, whh, bias_r, bias_z, bias_in, bias_hn);      // gate math on pre-scaled accumulators: see GRU_PRE in common.h
  }
  WG_BARRIER();
  const unsigned jb = (unsigned)j * 4u;

  // x(ts) = relu(fc1(in)) for row tiles rt, rt+stride.. of this wave's slice (pairs share the B fragments)
  auto fc1 = [&](const float* In, float* Xt, int ts, int rt_first, int rt_stride) __attribute__((always_inline)) {
    const long svt = SAVE ? (long)ts * NTILES + (long)blockIdx.x * a.RT : 0;
    for (int rt = rt_first; rt < RTW; rt += 2 * rt_stride) {
      const bool two = rt + rt_stride < RTW;
      f32x4 acc0 = {bias1, bias1, bias1, bias1}, acc1 = acc0;
      const float* in0 = In + (rt * 16 + m) * KS + 4 * q;
      const float* in1 = in0 + rt_stride * 16 * KS;
      const float* wf = W1s + (ws * a.KC * 64 + lane) * 4;
      for (int c = 0; c < a.KC; ++c) {
        f32x4 bv = *reinterpret_cast<const f32x4*>(wf + c * 256);
        f32x4 a0 = *reinterpret_cast<const f32x4*>(in0 + 16 * c);
        if (two) {
          f32x4 a1 = *reinterpret_cast<const f32x4*>(in1 + 16 * c);
          mfma16x4_il2(a0, bv, acc0, a1, bv, acc1);
        } else {
          acc0 = mfma16x4(a0, bv, acc0);
        }
      }
      const int r0 = rt * 16 + 4 * q;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc0[i] = fmaxf(acc0[i], 0.f);
        Xt[(r0 + i) * HS + j] = acc0[i];
      }
      if (SAVE) *reinterpret_cast<f32x4*>(a.saved + sv_off(svt + rt, 6, 1, ws, lane)) = acc0;
      if (two) {
        const int r1 = r0 + rt_stride * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc1[i] = fmaxf(acc1[i], 0.f);
          Xt[(r1 + i) * HS + j] = acc1[i];
        }
        if (SAVE) *reinterpret_cast<f32x4*>(a.saved + sv_off(svt + rt + rt_stride, 6, 1, ws, lane)) = acc1;
      }
    }
  };
  // q(ts) = fc2(h) for row tile rt (whole K = 64 in one wave)
  auto fc2 = [&](const float* Hs, int ts, int rt) __attribute__((always_inline)) {
    f32x4 acc[AC];
#pragma unroll
    for (int ac = 0; ac < AC; ++ac) acc[ac] = (f32x4){bias2[ac], bias2[ac], bias2[ac], bias2[ac]};
    const float* hr = Hs + (rt * 16 + m) * HS + 4 * q;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f32x4 ah = *reinterpret_cast<const f32x4*>(hr + 16 * c);
#pragma unroll
      for (int ac = 0; ac < AC; ++ac) acc[ac] = mfma16x4(ah, w2[ac][c], acc[ac]);
    }
    const int4 ri = *reinterpret_cast<const int4*>(rowidx + rt * 16 + 4 * q);
    const unsigned A4 = (unsigned)a.A * 4u, trow = (unsigned)ts * (unsigned)a.N;
#pragma unroll
    for (int ac = 0; ac < AC; ++ac) {
      const int col = 16 * ac + m;
      if (col < a.A) {
        const unsigned cb = (unsigned)col * 4u;
        st32(a.q, ((unsigned)ri.x + trow) * A4 + cb, acc[ac][0]);
        st32(a.q, ((unsigned)ri.y + trow) * A4 + cb, acc[ac][1]);
        st32(a.q, ((unsigned)ri.z + trow) * A4 + cb, acc[ac][2]);
        st32(a.q, ((unsigned)ri.w + trow) * A4 + cb, acc[ac][3]);
      }
    }
  };

  // prologue: x(0) by both teams (tiles alternate between them); XS: only when step 0 is computed in full
  // (the barrier above - staged weights - also published the step flags)
  if (!XS || xmask[0]) fc1(In0, Xt0, 0, team, 2);
  WG_BARRIER();

  float* Hp = Ha;
  float* Hn = Hb;
  ST_DECL(6);
  for (int t = 0; t < a.T; ++t) {
    const int par = t & 1;
    const unsigned trow = (unsigned)t * (unsigned)a.N;
    const long svt = SAVE ? (long)t * NTILES + (long)blockIdx.x * a.RT : 0;      // (step, first row tile of this workgroup)
    float* Xc = Xt0 + par * rows * HS;                 // x(t), written in the previous step
    float* Inn = In0 + (par ^ 1) * rows * KS;           // input of step t+1 (committed during step t-1)
    const bool xread = XS && xmask[t] == 0;            // this step's input-side work is read, not computed
    if (team == 1) {
      if (t + 1 < a.T && (!XS || xmask[t + 1])) fc1(Inn, Xt0 + (par ^ 1) * rows * HS, t + 1, 0, 1);
      if (t > 0)
        for (int rt = ws; rt < RTW; rt += 4) fc2(Hp, t - 1, rt);
    }
    ST_MARK(0);
    if (tid < 4) tilecnt[(par ^ 1) * 4 + tid] = 0;      // counters of the NEXT step (nobody grabs them before the barrier)
    // ---------------- GRU row tiles of this slice, shared by the two waves of the SIMD through a counter
    const bool last = (t == a.T - 1) && a.h_last;
    auto grab = [&]() {
      int v = 0;
      if (lane == 0) v = atomicAdd(&tilecnt[par * 4 + ws], 1);
      return __builtin_amdgcn_readfirstlane(v);
    };
    int rt_next = XS ? (team == 0 ? 0 : RTW) : grab();
    while (rt_next < RTW) {
      const int rt = rt_next;
      rt_next = XS ? rt + 1 : grab();
      f32x4 ar = {bias_r, bias_r, bias_r, bias_r};
      f32x4 az = {bias_z, bias_z, bias_z, bias_z};
      f32x4 ain = {bias_in, bias_in, bias_in, bias_in};
      f32x4 ahn = {bias_hn, bias_hn, bias_hn, bias_hn};
      const float* xr = Xc + (rt * 16 + m) * HS + 4 * q;
      const float* hr = Hp + (rt * 16 + m) * HS + 4 * q;
      if (XS) {
        if (xread) { ar = gB[0]; az = gB[1]; ain = gB[2]; }
        const bool same = rt + 1 < RTW;              // the next tile: this step's, or tile 0 of the next step (stored step + 1)
        const int nts = same ? t + 1 : t + 2;
        gissue(nts < a.T ? nts : Tm1, same ? rt + 1 : 0);
      }
      // input-side products first, hidden-side products after them (the SAME order in every GRU kernel: the input-side
      // sums bias + x W_ih of one unroll can then stand in for another unroll's - see the GI variants of agent_fwd_kernel)
      if (!xread) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          f32x4 ax = *reinterpret_cast<const f32x4*>(xr + 16 * c);
          mfma16x4_il3(ax, wih[0][c], ar, ax, wih[1][c], az, ax, wih[2][c], ain);
        }
      }
      if (SAVE && a.gi_out) {
        float* const gp = a.gi_out + sv_off(svt + rt, 3, 0, ws, lane);
        *reinterpret_cast<f32x4*>(gp) = ar;
        *reinterpret_cast<f32x4*>(gp + 1024) = az;
        *reinterpret_cast<f32x4*>(gp + 2 * 1024) = ain;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        f32x4 ah = *reinterpret_cast<const f32x4*>(hr + 16 * c);
        mfma16x4_il3(ah, whh[2][c], ahn, ah, whh[0][c], ar, ah, whh[1][c], az);
      }
      const int r0 = rt * 16 + 4 * q;
      f32x4 vhp, vr, vz, vn, vh;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        vhp[i] = Hp[(r0 + i) * HS + j];
        float r_, z_, n_, h_;
        if (AC == 2) gru_point(ar[i], az[i], ain[i], ahn[i], vhp[i], r_, z_, n_, h_);
        else gru_point_plain(ar[i], az[i], ain[i], ahn[i], vhp[i], r_, z_, n_, h_);
        vr[i] = r_; vz[i] = z_; vn[i] = n_; vh[i] = h_;
        Hn[(r0 + i) * HS + j] = vh[i];
      }
      if (a.hs) {
        const int4 ri = *reinterpret_cast<const int4*>(rowidx + r0);
        st32(a.hs, ((unsigned)ri.x + trow) * 256u + jb, vh[0]); st32(a.hs, ((unsigned)ri.y + trow) * 256u + jb, vh[1]);
        st32(a.hs, ((unsigned)ri.z + trow) * 256u + jb, vh[2]); st32(a.hs, ((unsigned)ri.w + trow) * 256u + jb, vh[3]);
      }
      const int4 rr = *reinterpret_cast<const int4*>(rowrho + r0);
      if (SAVE) {
        float* const sp = a.saved + sv_off(svt + rt, 6, 0, ws, lane);       // plane k at sp + 1024 k
        *reinterpret_cast<f32x4*>(sp) = vhp;
        *reinterpret_cast<f32x4*>(sp + 2 * 1024) = vr;
        *reinterpret_cast<f32x4*>(sp + 3 * 1024) = vz;
        *reinterpret_cast<f32x4*>(sp + 4 * 1024) = vn;
        *reinterpret_cast<f32x4*>(sp + 5 * 1024) = AC == 2 ? ahn * MARL_INV_2LOG2E : ahn;      // BPTT wants W_hn h + b_hn itself
        // the hidden state AFTER the last step, where the backward pass looks for h(t): plane 0 of step t+1
        if (t == a.T - 1) *reinterpret_cast<f32x4*>(sp + NTILES * (6 * 1024)) = vh;
      }
      if (last) {
        st32(a.h_last, (unsigned)rr.x * 256u + jb, vh[0]); st32(a.h_last, (unsigned)rr.y * 256u + jb, vh[1]);
        st32(a.h_last, (unsigned)rr.z * 256u + jb, vh[2]); st32(a.h_last, (unsigned)rr.w * 256u + jb, vh[3]);
      }
    }
    ST_MARK(1);
    // input tile of step t+2 -> the buffer fc1 finished with in the previous step; start the loads of step t+3
    // (XS: only the steps computed in full read an input tile - refill and observation loads run for those alone)
    if (!XS || (t + 2 < a.T && xmask[t + 2])) {
      if (par) commit(In0 + rows * KS, pu_lds1);
      else commit(In0, pu_lds0);
    }
    if (!XS) issue(t + 3 < a.T ? t + 3 : Tm1);
    else if (t + 3 < a.T && xmask[t + 3]) issue(t + 3);
    ST_MARK(2);
    WG_BARRIER();
    ST_MARK(3);
    float* tmp = Hp; Hp = Hn; Hn = tmp;
  }
  ST_DUMP(6);
  for (int rt = wave; rt < RTW; rt += 8) fc2(Hp, a.T - 1, rt);      // q of the last step
}

// ---------------------------------------------------------------------------------------------
// Backward through time, fully fused: per step the delta pass (dh carry, gate gradients, dx) AND
// the weight-gradient reductions of W_ih, W_hh, W_2 and their biases.  4 waves, one per SIMD
// (up to 512 VGPRs): wave w keeps its B-fragments of W_ih^T / W_hh^T / W_2^T and 100 accumulator
// registers of dW for the whole kernel.  The accumulator (D) layout of two tiles over the same 16
// rows IS the (A^T, B) operand pair of v_mfma_f32_16x16x4, so dW += G^T X needs no LDS at all; the
// saved activations are software-pipelined one row tile ahead straight into registers.
struct BwdArgs {
  const float *Wih, *Whh, *W2;
  const float* dq;        // (B,T,N,A) dense gradient on q, or (when dq_idx != null) unused
  const int* dq_idx;      // (B,T,N) sparse form: the only non-zero of row (b,t,n) is column dq_idx with value dq_val
  const float* dq_val;    // (B,T,N) / gdiv  (the Q-learning losses touch one action per row, q_learner.py:93)
  const int* dq_idx2;     // optional second (column, value) pair per row (QTRAN: taken action AND greedy action,
  const float* dq_val2;   //   qtran_learner.py:139,145); values of equal columns add
  int dq_gdiv;            // value index = row index / dq_gdiv (N: one value per (episode, step) shared by its agents)
  const float* dhs;       // (B,T,N,64) external gradient on hs[t], or null
  const float* saved;     // [T][B*N][6][64]
  const float* hs;        // unused since the forward pass stores h(t) with the saved planes (kept in the ABI)
  float* dxp;             // (B,T,N,64): gradient at fc1 pre-activation
  float* dh0;             // (B*N,64) gradient wrt the initial hidden state, or null
  float* ws;              // [n_wg][slab]: dW_ih | dW_hh | dW_2 | db_ih | db_hh | db_2 partials
  int B, T, N, A, RT;
  long R;
};

constexpr int DGS = 256 + MARL_PAD_G;
constexpr int BNT = 512;      // 8 waves: two per SIMD
constexpr int NQ = 4;         // dq prefetch registers per thread

__host__ __device__ inline long bwd_slab_floats(int A) { return 2L * 192 * 64 + (long)A * 64 + 2 * 192 + A; }

__device__ __forceinline__ long b_of(long rho, int N) { return rho / N; }
__device__ __forceinline__ int n_of(long rho, int N) { return (int)(rho % N); }

// 8 waves = 2 teams x 4 hidden-unit slices, two waves per SIMD (<= 256 registers each), so that one wave's
// loads / pointwise math / LDS traffic overlap the other's MFMAs (the previous 4-wave, 457-register version
// kept the matrix pipe 53 % busy).
//   phase B (row tiles split between the teams): dh = carry + dhs + dq W2^T, gate gradients -> LDS (DG, CAR)
//   phase C (every wave walks ALL row tiles, the teams split the PRODUCTS):
//       team 0 ("ih"): dx = [drp|dzp|dnp] W_ih -> relu gate -> dxp ;  dW_ih += [drp|dzp|dnp]^T x
//       team 1 ("hh"): dh_prev = carry + [drp|dzp|dhn] W_hh          ;  dW_hh += [drp|dzp|dhn]^T h_prev ; dW_2 += dq^T h
//   Both roles run the SAME code on the same register arrays (wT, accW, cur/nxt); only base pointers and
//   LDS column offsets differ, so the register allocation is that of one role.
// SPQ: 0 dense dq tile, 1 / 2 sparse (column, value) pairs per row
template <int AC, bool DHS, int SPQ>
__global__ __launch_bounds__(BNT, 2) void agent_bwd_kernel(BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave >> 2, ws = wave & 3;      // wave-uniform (scalar registers)
  const int q = lane >> 4, m = lane & 15;
  const int rows = a.RT * 16;
  constexpr int QP = AC * 16, QS = QP + 4;
  float* DG = smem;                         // [rows][DGS]  drp|dzp|dnp|dhn
  float* DQ0 = DG + rows * DGS;             // [rows][QS] x2
  float* DQ1 = DQ0 + rows * QS;
  float* CAR = DQ1 + rows * QS;             // [rows][HS] carried dh
  float* RED = CAR + rows * HS;             // [4][64] bias partial sums of team 1 (combined at the end)
  int* rowidx = reinterpret_cast<int*>(RED + 4 * 64);   // [rows]: output row b*T*N + n
  int* rowrho = rowidx + rows;                          // [rows]: b*N + n
  float* rowok = reinterpret_cast<float*>(rowrho + rows);  // [rows]: 1 real row, 0 row past the batch

  // rows past the batch are clamped to the last valid row for LOADS; their incoming gradients (dq, dhs) are
  // zeroed, which makes every quantity they contribute (gate gradients, dW, bias sums) exactly zero
  const long row0 = (long)blockIdx.x * rows;
  for (int r = tid; r < rows; r += BNT) {
    long rho = row0 + r;
    const float ok = rho < a.R ? 1.f : 0.f;
    if (rho > a.R - 1) rho = a.R - 1;
    const long b = rho / a.N;
    const int n = (int)(rho % a.N);
    rowidx[r] = (int)(b * a.T * a.N + n);
    rowrho[r] = (int)rho;
    rowok[r] = ok;
  }
  const long tstride = a.N;
  const long NTILES = (a.R + 15) >> 4, tile0 = (long)blockIdx.x * a.RT;      // global 16-row tiles (saved-activation layout)
  const int RTW = (int)((NTILES - tile0) < a.RT ? (NTILES - tile0) : a.RT);     // REAL row tiles of this workgroup (the last one may hold fewer)
  // carried dh of the last step: zero, plus the external gradient on hs[T-1] when there is one.  The external gradient of
  // the earlier steps is added where the carry is written (team 1, procC): it never occupies a slot of the prefetch sets.
  for (int e = tid; e < rows * HS; e += BNT) {
    const int r = e / HS, k = e - r * HS;
    float v = 0.f;
    if (DHS && k < H && row0 + r < a.R) v = a.dhs[((long)(b_of(row0 + r, a.N) * a.T + a.T - 1) * a.N + n_of(row0 + r, a.N)) * H + k];
    CAR[e] = v;
  }
  const int j = 16 * ws + m;

  // role weights: B-fragments of the transposed product, lane (q,m) holds W[k = 16c+4q+i][col 16ws+m]
  const float* Wrole = team ? a.Whh : a.Wih;
  f32x4 wT[12], w2T[AC];
#pragma unroll
  for (int c = 0; c < 12; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) wT[c][i] = Wrole[(long)(16 * c + 4 * q + i) * H + j];
#pragma unroll
  for (int ac = 0; ac < AC; ++ac)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = 16 * ac + 4 * q + i;
      w2T[ac][i] = k < a.A ? a.W2[(long)k * H + j] : 0.f;
    }
  f32x4 accW[3][4], accW2[AC];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int c = 0; c < 4; ++c) accW[g][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ac = 0; ac < AC; ++ac) accW2[ac] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float sb_r = 0.f, sb_z = 0.f, sb_n = 0.f, sb_hn = 0.f, sb2[AC];
#pragma unroll
  for (int ac = 0; ac < AC; ++ac) sb2[ac] = 0.f;
  __syncthreads();

  // ---- prefetch sets (named registers; cur = tile being processed, nxt = tile in flight)
  //   B set: [0]=h_prev(own cols) [1]=r [2]=z [3]=n [4]=hn
  //   C set: [0..3]=x (team 0) / h_prev (team 1), all 64 columns ; [4]=x own cols (team 0) / h_t own cols (team 1)
  // (every load is unconditional: at t = 0 the "next step" loads re-read step 0 and are never consumed - a
  // predicated load makes the compiler zero the register first and that write waits for older loads)
#define LOAD_B(P, tt, rr, en)                                                                            \
  {                                                                                                      \
    const float* sp_ = a.saved + sv_off((long)(tt) * NTILES + tile0 + (rr), 6, 0, ws, lane);             \
    P[0] = *reinterpret_cast<const f32x4*>(sp_);                                                         \
    P[1] = *reinterpret_cast<const f32x4*>(sp_ + 2 * 1024);                                              \
    P[2] = *reinterpret_cast<const f32x4*>(sp_ + 3 * 1024);                                              \
    P[3] = *reinterpret_cast<const f32x4*>(sp_ + 4 * 1024);                                              \
    P[4] = *reinterpret_cast<const f32x4*>(sp_ + 5 * 1024);                                              \
  }
#define LOAD_C(P, tt, rr, en)                                                                            \
  {                                                                                                      \
    const float* sp_ = a.saved + sv_off((long)(tt) * NTILES + tile0 + (rr), 6, team ? 0 : 1, 0, lane);   \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) P[c] = *reinterpret_cast<const f32x4*>(sp_ + 256 * c); \
    /* team 1: h(tt) = the hidden state fed into step tt+1 = plane 0 of that step (the forward pass also */ \
    /* writes a plane 0 for step T) */                                                                   \
    P[4] = *reinterpret_cast<const f32x4*>(sp_ + (team ? NTILES * (6 * 1024) : 0) + 256 * ws);           \
  }
  auto dq_elem = [&](int t, int e) -> float {
    const int r = e / QP, k = e - r * QP;
    return (k < a.A) ? a.dq[((long)rowidx[r] + (long)t * tstride) * a.A + k] * rowok[r] : 0.f;
  };

  // sparse form: the dq tile is never materialised - per row one (column, value) pair, the MFMA operand
  // fragments are synthesised from it in registers
  auto sp_load = [&](int t, int r, int& u, float& g, int& u2, float& g2) {
    const long o = (long)rowidx[r] + (long)t * tstride;
    const long ov = a.dq_gdiv > 1 ? o / a.dq_gdiv : o;
    u = a.dq_idx[o];
    g = a.dq_val[ov] * rowok[r];
    if (SPQ == 2) { u2 = a.dq_idx2[o]; g2 = a.dq_val2[ov] * rowok[r]; }
  };
  if (SPQ) {
    for (int r = tid; r < rows; r += BNT) {
      int u, u2 = -1; float g, g2 = 0.f;
      sp_load(a.T - 1, r, u, g, u2, g2);
      reinterpret_cast<int*>(DQ0)[r] = u;
      DQ0[rows + r] = g;
      if (SPQ == 2) { reinterpret_cast<int*>(DQ0)[2 * rows + r] = u2; DQ0[3 * rows + r] = g2; }
    }
  } else {
    for (int e = tid; e < rows * QP; e += BNT) DQ0[(e / QP) * QS + (e % QP)] = dq_elem(a.T - 1, e);
  }
  const bool hasB = team < RTW;                   // this team owns at least one row tile in phase B
  const bool full_wg = row0 + rows <= a.R;         // no rows past the batch in this workgroup
  // Register sets of the software pipeline.  NO set is ever copied into another inside the step loop: a copy
  // needs the loaded values and makes the compiler drain vmcnt right where the prefetch was issued.
  //   both phases alternate their row tiles between sA and sB
  f32x4 sA[5], sB[5];
  // item 1 of a step (the team's 2nd phase-B tile, or phase-C tile 0 when it has only one) is issued into sB as
  // soon as the previous step's last phase-C tile has released that set - before the end-of-step barrier
#define LOAD_ITEM1(tt, en)                                         \
  {                                                                \
    if (team + 2 < RTW) LOAD_B(sB, tt, team + 2, en)              \
    else LOAD_C(sB, tt, 0, en)                                     \
  }
  if (hasB) LOAD_B(sA, a.T - 1, team, true)
  else LOAD_C(sA, a.T - 1, 0, true)
  if (hasB) LOAD_ITEM1(a.T - 1, true)
  __syncthreads();

  // ---- phase B body for one row tile: dh = carry + dhs + dq W2^T ; gate gradients -> DG, carry*z -> CAR
  auto procB = [&](const f32x4 (&P)[5], int rt, const float* DQ) __attribute__((always_inline)) {
    f32x4 dh;
#pragma unroll
    for (int i = 0; i < 4; ++i) dh[i] = CAR[(rt * 16 + 4 * q + i) * HS + j];
    const float* dqr = DQ + (rt * 16 + m) * QS + 4 * q;
    const int su = SPQ ? reinterpret_cast<const int*>(DQ)[rt * 16 + m] : 0;
    const float sg = SPQ ? DQ[rows + rt * 16 + m] : 0.f;
    const int su2 = SPQ == 2 ? reinterpret_cast<const int*>(DQ)[2 * rows + rt * 16 + m] : -1;
    const float sg2 = SPQ == 2 ? DQ[3 * rows + rt * 16 + m] : 0.f;
#pragma unroll
    for (int ac = 0; ac < AC; ++ac) {
      f32x4 av;
      if (SPQ) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          av[i] = (su == 16 * ac + 4 * q + i) ? sg : 0.f;
          if (SPQ == 2) av[i] += (su2 == 16 * ac + 4 * q + i) ? sg2 : 0.f;
        }
      } else {
        av = *reinterpret_cast<const f32x4*>(dqr + 16 * ac);
      }
      dh = mfma16x4(av, w2T[ac], dh);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = rt * 16 + 4 * q + i;
      const float d = dh[i];                        // (an external gradient on hs[t] is already part of the carry)
      const float rg = P[1][i], zg = P[2][i], ng = P[3][i];
      const float dn = d * (1.f - zg);
      const float dz = d * (P[0][i] - ng);
      const float dnp = dn * (1.f - ng * ng);
      const float dzp = dz * zg * (1.f - zg);
      const float drp = dnp * P[4][i] * rg * (1.f - rg);
      const float dhn = dnp * rg;
      float* l = DG + r * DGS + j;
      l[0] = drp; l[64] = dzp; l[128] = dnp; l[192] = dhn;
      CAR[r * HS + j] = d * zg;
      sb_r += drp; sb_z += dzp; sb_n += dnp; sb_hn += dhn;
    }
  };
  // ---- phase C body for one row tile: this role's products
  auto procC = [&](const f32x4 (&P)[5], int rt, int t, const float* DQ) __attribute__((always_inline)) {
    const int r0 = rt * 16 + 4 * q;
    // external gradient on hs[t-1] (QTRAN heads): joins the carry this tile writes for the previous step
    f32x4 dext = {0.f, 0.f, 0.f, 0.f};
    if (DHS && team && t > 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) dext[i] = a.dhs[((long)rowidx[r0 + i] + (long)(t - 1) * tstride) * H + j];
    }
    f32x4 main;                                   // dx (team 0)  /  dh_prev (team 1)
#pragma unroll
    for (int i = 0; i < 4; ++i) main[i] = team ? CAR[(r0 + i) * HS + j] : 0.f;
    const float* gr = DG + (rt * 16 + m) * DGS + 4 * q;
    // three independent accumulation chains (one per gate block) instead of 48 dependent MFMAs: with few row
    // tiles per workgroup (small shards) nothing else hides the dependent-issue latency
    f32x4 m1 = {0.f, 0.f, 0.f, 0.f}, m2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int c2off = team ? 192 + 16 * c : 128 + 16 * c;                  // team 1 reads dhn instead of dnp
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(gr + 16 * c);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(gr + 64 + 16 * c);
      const f32x4 a2 = *reinterpret_cast<const f32x4*>(gr + c2off);
      mfma16x4_il3(a0, wT[c], main, a1, wT[4 + c], m1, a2, wT[8 + c], m2);
    }
    main += m1 + m2;
    // gate-gradient tiles of this wave's 16 columns in accumulator layout (they ARE the A^T fragments)
    f32x4 g0, g1, g2;
    const int g2off = team ? 192 : 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* l = DG + (r0 + i) * DGS + j;
      g0[i] = l[0]; g1[i] = l[64]; g2[i] = l[g2off];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      mfma16x4_il3(g0, P[c], accW[0][c], g1, P[c], accW[1][c], g2, P[c], accW[2][c]);
    }
    if (team) {
#pragma unroll
      for (int i = 0; i < 4; ++i) CAR[(r0 + i) * HS + j] = DHS ? main[i] + dext[i] * rowok[r0 + i] : main[i];
      int4 su4 = {0, 0, 0, 0}, tu4 = {-1, -1, -1, -1};
      f32x4 sg4 = {0.f, 0.f, 0.f, 0.f}, tg4 = {0.f, 0.f, 0.f, 0.f};
      if (SPQ) {                                   // the 4 rows' (action, gradient) pairs: two 16-byte LDS reads
        su4 = *reinterpret_cast<const int4*>(reinterpret_cast<const int*>(DQ) + r0);
        sg4 = *reinterpret_cast<const f32x4*>(DQ + rows + r0);
      }
      if (SPQ == 2) {
        tu4 = *reinterpret_cast<const int4*>(reinterpret_cast<const int*>(DQ) + 2 * rows + r0);
        tg4 = *reinterpret_cast<const f32x4*>(DQ + 3 * rows + r0);
      }
#pragma unroll
      for (int ac = 0; ac < AC; ++ac) {
        f32x4 dqf;
        if (SPQ) {
          const int col = 16 * ac + m;
          dqf[0] = su4.x == col ? sg4[0] : 0.f; dqf[1] = su4.y == col ? sg4[1] : 0.f;
          dqf[2] = su4.z == col ? sg4[2] : 0.f; dqf[3] = su4.w == col ? sg4[3] : 0.f;
          if (SPQ == 2) {
            dqf[0] += tu4.x == col ? tg4[0] : 0.f; dqf[1] += tu4.y == col ? tg4[1] : 0.f;
            dqf[2] += tu4.z == col ? tg4[2] : 0.f; dqf[3] += tu4.w == col ? tg4[3] : 0.f;
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) dqf[i] = DQ[(r0 + i) * QS + 16 * ac + m];
        }
        accW2[ac] = mfma16x4(dqf, P[4], accW2[ac]);
        sb2[ac] += dqf[0] + dqf[1] + dqf[2] + dqf[3];
      }
    } else {
      // uniform base + 32-bit byte offsets (B*T*N*64*4 < 4 GB is checked on the host); only the last workgroup
      // can hold rows past the batch and needs the per-row predicate
      const int4 ri = *reinterpret_cast<const int4*>(rowidx + r0);
      const unsigned trow = (unsigned)t * (unsigned)a.N, jb = (unsigned)j * 4u;
      const unsigned o0 = ((unsigned)ri.x + trow) * 256u + jb, o1 = ((unsigned)ri.y + trow) * 256u + jb;
      const unsigned o2 = ((unsigned)ri.z + trow) * 256u + jb, o3 = ((unsigned)ri.w + trow) * 256u + jb;
      const float v0 = P[4][0] > 0.f ? main[0] : 0.f, v1 = P[4][1] > 0.f ? main[1] : 0.f;
      const float v2 = P[4][2] > 0.f ? main[2] : 0.f, v3 = P[4][3] > 0.f ? main[3] : 0.f;
      if (full_wg) {
        st32(a.dxp, o0, v0); st32(a.dxp, o1, v1); st32(a.dxp, o2, v2); st32(a.dxp, o3, v3);
      } else {
        if (rowok[r0] != 0.f) st32(a.dxp, o0, v0);
        if (rowok[r0 + 1] != 0.f) st32(a.dxp, o1, v1);
        if (rowok[r0 + 2] != 0.f) st32(a.dxp, o2, v2);
        if (rowok[r0 + 3] != 0.f) st32(a.dxp, o3, v3);
      }
    }
  };
  // loads of the first item of the NEXT step (its phase-B tile, or phase-C tile 0 for a team without one)
#define LOAD_NEXT_STEP(P, t)                                  \
  {                                                           \
    if (hasB) LOAD_B(P, ((t) > 0 ? (t) - 1 : 0), team, (t) > 0) \
    else LOAD_C(P, ((t) > 0 ? (t) - 1 : 0), 0, (t) > 0)       \
  }

  int par = 0;
  bool c0A_carry = true;             // where a team WITHOUT phase-B tiles finds phase-C tile 0 of the coming step
#ifdef MARL_PRIO_YOUNG
  if (wave >= 4) __builtin_amdgcn_s_setprio(MARL_PRIO_YOUNG);      // A/B: static priority for the younger half of the workgroup
#endif
  ST_DECL(5);
  for (int t = a.T - 1; t >= 0; --t, par ^= 1) {
    float* DQ = par ? DQ1 : DQ0;
    float* DQn = par ? DQ0 : DQ1;
    float dqpre[NQ];
    int spu = -1, spu2 = -1; float spg = 0.f, spg2 = 0.f;
    if (SPQ) {
      if (t > 0 && tid < rows) sp_load(t - 1, tid, spu, spg, spu2, spg2);
    } else {
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int e = tid + BNT * i;
        dqpre[i] = (t > 0 && e < rows * QP) ? dq_elem(t - 1, e) : 0.f;
      }
    }
    // ---------------- phase B.  Entering: sA = the team's first tile (or, without one, phase-C tile 0).
    // Tiles alternate between sA and sB; the item after the last tile is phase-C tile 0, which must end up in sA.
    bool c0A = c0A_carry;            // phase-C tile 0 ends up in sA (else sB)
    if (hasB) {
      c0A = true;
      int rt = team;
      bool pre = true;               // sB already holds (in flight) the item after the first tile
      while (true) {
        bool more = rt + 2 < RTW;
        if (!pre) {
          if (more) LOAD_B(sB, t, rt + 2, true)
          else LOAD_C(sB, t, 0, true)
        }
        pre = false;
        procB(sA, rt, DQ);
        if (!more) { c0A = false; break; }
        rt += 2;
        more = rt + 2 < RTW;
        if (more) LOAD_B(sA, t, rt + 2, true)
        else LOAD_C(sA, t, 0, true)
        procB(sB, rt, DQ);
        if (!more) break;
        rt += 2;
      }
    }
    ST_MARK(0);
    WG_BARRIER();
    ST_MARK(1);
    // ---------------- phase C: every wave, all row tiles, alternating sA / sB
    // tiles alternate between the sets; when tile 0 sits in sB (odd number of phase-B tiles) a prologue processes
    // it from there and the main loop starts at tile 1.  Item 0 of the next step is loaded while the last tile is
    // processed and must end up in sA.
    int rt0 = 0;
    bool in_a = true;                 // next-step item 0 landed in sA
    if (!c0A) {
      if (1 < RTW) LOAD_C(sA, t, 1, true)
      else LOAD_NEXT_STEP(sA, t)
      procC(sB, 0, t, DQ);
      rt0 = 1;
    }
    for (int rt = rt0; rt < RTW; rt += 2) {
      if (rt + 1 < RTW) LOAD_C(sB, t, rt + 1, true)
      else { LOAD_NEXT_STEP(sB, t) in_a = false; }
      procC(sA, rt, t, DQ);
      if (rt + 1 < RTW) {
        if (rt + 2 < RTW) LOAD_C(sA, t, rt + 2, true)
        else LOAD_NEXT_STEP(sA, t)
        procC(sB, rt + 1, t, DQ);
      }
    }
    c0A_carry = true;
    if (!in_a) {
      if (hasB) {                     // phase B wants its first tile in sA: one copy per step, a whole tile after the issue
#pragma unroll
        for (int k = 0; k < 5; ++k) sA[k] = sB[k];
      } else {
        c0A_carry = false;            // a team without phase-B tiles (one row tile per workgroup) just starts from sB
      }
    }
    if (hasB && t > 0) LOAD_ITEM1(t - 1, true)
    ST_MARK(2);
    if (t > 0) {
      if (SPQ) {
        if (tid < rows) {
          reinterpret_cast<int*>(DQn)[tid] = spu; DQn[rows + tid] = spg;
          if (SPQ == 2) { reinterpret_cast<int*>(DQn)[2 * rows + tid] = spu2; DQn[3 * rows + tid] = spg2; }
        }
      } else {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          const int e = tid + BNT * i;
          if (e < rows * QP) DQn[(e / QP) * QS + (e % QP)] = dqpre[i];
        }
      }
    }
    ST_MARK(3);
    WG_BARRIER();
    ST_MARK(4);
  }
  ST_DUMP_AT(5, 8);     // columns 8.. of the stamp rows (the forward kernel uses 0..5)
  if (a.dh0 && team) {
    for (int r = 4 * q; r < rows; r += 16)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (rowok[r + i] != 0.f) a.dh0[(row0 + r + i) * H + j] = CAR[(r + i) * HS + j];
  }
  // ---- this workgroup's partial weight gradients -> slab (summed in fixed order by the reduce kernel)
  float* slab = a.ws + (long)blockIdx.x * bwd_slab_floats(a.A);
  float* s_role = slab + (team ? 192 * 64 : 0);          // dW_ih (team 0) | dW_hh (team 1)
  float* s_w2 = slab + 2 * 192 * 64;
  float* s_bih = s_w2 + (long)a.A * 64;
  float* s_bhh = s_bih + 192;
  float* s_b2 = s_bhh + 192;
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        s_role[(g * 64 + 16 * ws + 4 * q + i) * 64 + 16 * c + m] = accW[g][c][i];
  auto red4 = [&](float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; };
  sb_r = red4(sb_r); sb_z = red4(sb_z); sb_n = red4(sb_n); sb_hn = red4(sb_hn);
  if (team) {
#pragma unroll
    for (int ac = 0; ac < AC; ++ac)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int arow = 16 * ac + 4 * q + i;
        if (arow < a.A) s_w2[arow * 64 + j] = accW2[ac][i];
      }
#pragma unroll
    for (int ac = 0; ac < AC; ++ac) {
      const float v = red4(sb2[ac]);
      if (ws == 0 && q == 0 && 16 * ac + m < a.A) s_b2[16 * ac + m] = v;
    }
    if (q == 0) { RED[j] = sb_r; RED[64 + j] = sb_z; RED[128 + j] = sb_n; RED[192 + j] = sb_hn; }
  }
  __syncthreads();
  if (!team && q == 0) {       // both teams saw disjoint row tiles in phase B: add the two partial bias sums
    const float br = sb_r + RED[j], bz = sb_z + RED[64 + j], bn = sb_n + RED[128 + j], bh = sb_hn + RED[192 + j];
    s_bih[j] = br; s_bih[64 + j] = bz; s_bih[128 + j] = bn;
    s_bhh[j] = br; s_bhh[64 + j] = bz; s_bhh[128 + j] = bh;
  }
}

#undef LOAD_B
#undef LOAD_C
#undef LOAD_ITEM1
#undef LOAD_NEXT_STEP

// ---------------------------------------------------------------------------------------------
// Software-pipelined BPTT for few row tiles per workgroup (RT <= 4: the small shards of the multi-GPU runs).
// With one or two row tiles nothing hides the pointwise phase B of agent_bwd_kernel (16 % of a step with the matrix
// pipe idle, half of the waves parked at its barrier), and only a quarter of a step's multiplies are on the dependent
// chain   carry(t) -> gate gradients(t) -> [drp|dzp|dhn](t) W_hh -> carry(t-1).   Here ONE barrier separates the steps:
//     team 1 ("hh"): per row tile  dh_prev = carry z + [drp|dzp|dhn](t) W_hh   -> straight on (same lanes, no barrier)
//                    the gate gradients of step t-1 -> the OTHER gate-gradient buffer;  then the products that are off
//                    the chain: dW_hh += [..](t)^T h_prev, dW_2 += dq(t)^T h
//     team 0 ("ih"): dx = [drp|dzp|dnp](t) W_ih -> relu gate -> dxp ;  dW_ih += [..](t)^T x
// so the pointwise math of one wave runs under the multiplies of its SIMD partner.  The gate-gradient tile is double
// buffered in LDS (which is what limits RT), the sparse dq pairs triple buffered (steps t, t-1 in use, t-2 arriving).
// One prefetch set per kind (saved planes of the tile's step / of the step before), re-issued as soon as its
// consumer is done: the other half of the item hides the latency.  Same MFMA sequences per output element as
// agent_bwd_kernel -> bitwise the same dxp / dh0; the weight-gradient slabs too (same per-workgroup accumulation order).
template <int AC, bool DHS, int SPQ>
__global__ __launch_bounds__(BNT, 2) void agent_bwd_pipe_kernel(BwdArgs a) {
  static_assert(SPQ == 1 || SPQ == 2, "sparse dq only");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave >> 2, ws = wave & 3;
  const int q = lane >> 4, m = lane & 15;
  const int rows = a.RT * 16;
  float* DGb = smem;                        // [2][rows][DGS]  drp|dzp|dnp|dhn of step parity
  float* CAR = DGb + 2 * rows * DGS;        // [rows][HS] carried dh (touched by team 1 only, each lane its own cells)
  float* DQt = CAR + rows * HS;             // [3][4][rows]: (column, value[, column2, value2]) per row, by step % 3
  float* RED = DQt + 12 * rows;             // [4][64]
  int* rowidx = reinterpret_cast<int*>(RED + 4 * 64);
  int* rowrho = rowidx + rows;
  float* rowok = reinterpret_cast<float*>(rowrho + rows);

  const long row0 = (long)blockIdx.x * rows;
  for (int r = tid; r < rows; r += BNT) {
    long rho = row0 + r;
    const float ok = rho < a.R ? 1.f : 0.f;
    if (rho > a.R - 1) rho = a.R - 1;
    const long b = rho / a.N;
    const int n = (int)(rho % a.N);
    rowidx[r] = (int)(b * a.T * a.N + n);
    rowrho[r] = (int)rho;
    rowok[r] = ok;
  }
  const long tstride = a.N;
  const long NTILES = (a.R + 15) >> 4, tile0 = (long)blockIdx.x * a.RT;      // global 16-row tiles (saved-activation layout)
  const int RTW = (int)((NTILES - tile0) < a.RT ? (NTILES - tile0) : a.RT);     // REAL row tiles of this workgroup (the last one may hold fewer)
  for (int e = tid; e < rows * HS; e += BNT) {
    const int r = e / HS, k = e - r * HS;
    float v = 0.f;
    if (DHS && k < H && row0 + r < a.R) v = a.dhs[((long)(b_of(row0 + r, a.N) * a.T + a.T - 1) * a.N + n_of(row0 + r, a.N)) * H + k];
    CAR[e] = v;
  }
  const int j = 16 * ws + m;
  const float* Wrole = team ? a.Whh : a.Wih;
  f32x4 wT[12], w2T[AC];
#pragma unroll
  for (int c = 0; c < 12; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) wT[c][i] = Wrole[(long)(16 * c + 4 * q + i) * H + j];
#pragma unroll
  for (int ac = 0; ac < AC; ++ac)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = 16 * ac + 4 * q + i;
      w2T[ac][i] = k < a.A ? a.W2[(long)k * H + j] : 0.f;
    }
  f32x4 accW[3][4], accW2[AC];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int c = 0; c < 4; ++c) accW[g][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ac = 0; ac < AC; ++ac) accW2[ac] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float sb_r = 0.f, sb_z = 0.f, sb_n = 0.f, sb_hn = 0.f, sb2[AC];
#pragma unroll
  for (int ac = 0; ac < AC; ++ac) sb2[ac] = 0.f;
  __syncthreads();                          // row tables

  auto sp_load = [&](int t, int r, int& u, float& g, int& u2, float& g2) {
    const long o = (long)rowidx[r] + (long)t * tstride;
    const long ov = a.dq_gdiv > 1 ? o / a.dq_gdiv : o;
    u = a.dq_idx[o];
    g = a.dq_val[ov] * rowok[r];
    if (SPQ == 2) { u2 = a.dq_idx2[o]; g2 = a.dq_val2[ov] * rowok[r]; }
  };
  auto sp_store = [&](int t, int r, int u, float g, int u2, float g2) {
    float* D = DQt + (t % 3) * 4 * rows;
    reinterpret_cast<int*>(D)[r] = u; D[rows + r] = g;
    if (SPQ == 2) { reinterpret_cast<int*>(D)[2 * rows + r] = u2; D[3 * rows + r] = g2; }
  };
  for (int s = 0; s < 2; ++s) {
    const int tt = a.T - 1 - s;
    if (tt >= 0)
      for (int r = tid; r < rows; r += BNT) {
        int u, u2 = -1; float g, g2 = 0.f;
        sp_load(tt, r, u, g, u2, g2);
        sp_store(tt, r, u, g, u2, g2);
      }
  }
  __syncthreads();

  // saved planes of (step tt, row tile rr), this wave's 16 columns: [0]=h_prev [1]=r [2]=z [3]=n [4]=hn
  auto load_b = [&](f32x4 (&P)[5], int tt, int rr) __attribute__((always_inline)) {
    const float* sp = a.saved + sv_off((long)tt * NTILES + tile0 + rr, 6, 0, ws, lane);
    P[0] = *reinterpret_cast<const f32x4*>(sp);
    P[1] = *reinterpret_cast<const f32x4*>(sp + 2 * 1024);
    P[2] = *reinterpret_cast<const f32x4*>(sp + 3 * 1024);
    P[3] = *reinterpret_cast<const f32x4*>(sp + 4 * 1024);
    P[4] = *reinterpret_cast<const f32x4*>(sp + 5 * 1024);
  };
  // [0..3] = x (team 0) / h_prev (team 1) of step tt, all 64 columns ; [4] = x own columns (team 0) / h(tt) own columns (team 1)
  auto load_c = [&](f32x4 (&P)[5], int tt, int rr) __attribute__((always_inline)) {
    const float* sp = a.saved + sv_off((long)tt * NTILES + tile0 + rr, 6, team ? 0 : 1, 0, lane);
#pragma unroll
    for (int c = 0; c < 4; ++c) P[c] = *reinterpret_cast<const f32x4*>(sp + 256 * c);
    // team 1: h(tt) = the hidden state fed into step tt+1 = plane 0 of that step (the forward pass also writes one for step T)
    P[4] = *reinterpret_cast<const f32x4*>(sp + (team ? NTILES * (6 * 1024) : 0) + 256 * ws);
  };
  // dh = carry + dq W2^T ; gate gradients -> DG, carry z -> CAR   (agent_bwd_kernel's phase B for one row tile)
  auto proc_b = [&](const f32x4 (&P)[5], int rt, const float* DQ, float* DG) __attribute__((always_inline)) {
    f32x4 dh;
#pragma unroll
    for (int i = 0; i < 4; ++i) dh[i] = CAR[(rt * 16 + 4 * q + i) * HS + j];
    const int su = reinterpret_cast<const int*>(DQ)[rt * 16 + m];
    const float sg = DQ[rows + rt * 16 + m];
    const int su2 = SPQ == 2 ? reinterpret_cast<const int*>(DQ)[2 * rows + rt * 16 + m] : -1;
    const float sg2 = SPQ == 2 ? DQ[3 * rows + rt * 16 + m] : 0.f;
#pragma unroll
    for (int ac = 0; ac < AC; ++ac) {
      f32x4 av;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        av[i] = (su == 16 * ac + 4 * q + i) ? sg : 0.f;
        if (SPQ == 2) av[i] += (su2 == 16 * ac + 4 * q + i) ? sg2 : 0.f;
      }
      dh = mfma16x4(av, w2T[ac], dh);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = rt * 16 + 4 * q + i;
      const float d = dh[i];
      const float rg = P[1][i], zg = P[2][i], ng = P[3][i];
      const float dn = d * (1.f - zg);
      const float dz = d * (P[0][i] - ng);
      const float dnp = dn * (1.f - ng * ng);
      const float dzp = dz * zg * (1.f - zg);
      const float drp = dnp * P[4][i] * rg * (1.f - rg);
      const float dhn = dnp * rg;
      float* l = DG + r * DGS + j;
      l[0] = drp; l[64] = dzp; l[128] = dnp; l[192] = dhn;
      CAR[r * HS + j] = d * zg;
      sb_r += drp; sb_z += dzp; sb_n += dnp; sb_hn += dhn;
    }
  };
  // this role's transposed product over one row tile: three independent chains (one per gate block)
  auto product = [&](int rt, const float* DG, f32x4 init) __attribute__((always_inline)) -> f32x4 {
    const float* gr = DG + (rt * 16 + m) * DGS + 4 * q;
    f32x4 main = init, m1 = {0.f, 0.f, 0.f, 0.f}, m2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int c2off = team ? 192 + 16 * c : 128 + 16 * c;                  // team 1 reads dhn instead of dnp
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(gr + 16 * c);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(gr + 64 + 16 * c);
      const f32x4 a2 = *reinterpret_cast<const f32x4*>(gr + c2off);
      mfma16x4_il3(a0, wT[c], main, a1, wT[4 + c], m1, a2, wT[8 + c], m2);
    }
    main += m1 + m2;
    return main;
  };
  // dW_role += [gate gradients of this wave's 16 columns]^T P   (accumulator layout = the A^T fragments)
  auto accum = [&](const f32x4 (&P)[5], int rt, const float* DG) __attribute__((always_inline)) {
    const int r0 = rt * 16 + 4 * q;
    f32x4 g0, g1, g2;
    const int g2off = team ? 192 : 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* l = DG + (r0 + i) * DGS + j;
      g0[i] = l[0]; g1[i] = l[64]; g2[i] = l[g2off];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      mfma16x4_il3(g0, P[c], accW[0][c], g1, P[c], accW[1][c], g2, P[c], accW[2][c]);
    }
  };
  const bool full_wg = row0 + rows <= a.R;
  const int T = a.T, RT = RTW;

  if (team) {
    // the chain wave wins the matrix pipe against its SIMD partner (which has 44 % of a step to spare): at equal
    // priority the two products run interleaved and the chain takes twice its pipe time
    __builtin_amdgcn_s_setprio(3);
    f32x4 bS[5], cS[5];
    // gate gradients of the last step (its carry is the external gradient on hs[T-1], or zero)
    for (int k = 0; k < RT; ++k) {
      load_b(bS, T - 1, k);
      proc_b(bS, k, DQt + ((T - 1) % 3) * 4 * rows, DGb + ((T - 1) & 1) * rows * DGS);
    }
    load_b(bS, T > 1 ? T - 2 : 0, 0);
    load_c(cS, T - 1, 0);
    WG_BARRIER();
    ST_DECL(4);
    for (int t = T - 1; t >= 0; --t) {
      const float* DGc = DGb + (t & 1) * rows * DGS;
      float* DGn = DGb + ((t & 1) ^ 1) * rows * DGS;
      const float* DQc = DQt + (t % 3) * 4 * rows;
      const float* DQp = DQt + ((t + 2) % 3) * 4 * rows;        // step t-1
      for (int k = 0; k < RT; ++k) {
        const int r0 = k * 16 + 4 * q;
        // ---- on the chain: dh_prev, then the gate gradients of step t-1
        f32x4 dext = {0.f, 0.f, 0.f, 0.f};
        if (DHS && t > 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) dext[i] = a.dhs[((long)rowidx[r0 + i] + (long)(t - 1) * tstride) * H + j];
        }
        f32x4 car;
#pragma unroll
        for (int i = 0; i < 4; ++i) car[i] = CAR[(r0 + i) * HS + j];
        const f32x4 main = product(k, DGc, car);
#pragma unroll
        for (int i = 0; i < 4; ++i) CAR[(r0 + i) * HS + j] = DHS ? main[i] + dext[i] * rowok[r0 + i] : main[i];
        ST_MARK(0);
        if (t > 0) proc_b(bS, k, DQp, DGn);
        ST_MARK(1);
        // the next item's step-before planes: the rest of this item and the next product hide the latency
        const int nk = k + 1 < RT ? k + 1 : 0, nt = k + 1 < RT ? t : t - 1;
        load_b(bS, nt > 0 ? nt - 1 : 0, nk);
        // ---- off the chain
        accum(cS, k, DGc);
        {
          const int4 su4 = *reinterpret_cast<const int4*>(reinterpret_cast<const int*>(DQc) + r0);
          const f32x4 sg4 = *reinterpret_cast<const f32x4*>(DQc + rows + r0);
          int4 tu4 = {-1, -1, -1, -1};
          f32x4 tg4 = {0.f, 0.f, 0.f, 0.f};
          if (SPQ == 2) {
            tu4 = *reinterpret_cast<const int4*>(reinterpret_cast<const int*>(DQc) + 2 * rows + r0);
            tg4 = *reinterpret_cast<const f32x4*>(DQc + 3 * rows + r0);
          }
#pragma unroll
          for (int ac = 0; ac < AC; ++ac) {
            f32x4 dqf;
            const int col = 16 * ac + m;
            dqf[0] = su4.x == col ? sg4[0] : 0.f; dqf[1] = su4.y == col ? sg4[1] : 0.f;
            dqf[2] = su4.z == col ? sg4[2] : 0.f; dqf[3] = su4.w == col ? sg4[3] : 0.f;
            if (SPQ == 2) {
              dqf[0] += tu4.x == col ? tg4[0] : 0.f; dqf[1] += tu4.y == col ? tg4[1] : 0.f;
              dqf[2] += tu4.z == col ? tg4[2] : 0.f; dqf[3] += tu4.w == col ? tg4[3] : 0.f;
            }
            accW2[ac] = mfma16x4(dqf, cS[4], accW2[ac]);
            sb2[ac] += dqf[0] + dqf[1] + dqf[2] + dqf[3];
          }
        }
        load_c(cS, nt > 0 ? nt : 0, nk);
        ST_MARK(2);
      }
      WG_BARRIER();
      ST_MARK(3);
    }
    ST_DUMP_AT(4, 8);
  } else {
    f32x4 cS[5];
    load_c(cS, T - 1, 0);
    WG_BARRIER();
    ST_DECL(4);
    for (int t = T - 1; t >= 0; --t) {
      const float* DGc = DGb + (t & 1) * rows * DGS;
      // the dq pairs of step t-2 travel through this team (it has time to spare; hipcc waits for ALL outstanding loads
      // where the registers are stored, which on the chain team exposed a full memory latency per step)
      int spu = -1, spu2 = -1; float spg = 0.f, spg2 = 0.f;
      if (t >= 2 && tid < rows) sp_load(t - 2, tid, spu, spg, spu2, spg2);
      for (int k = 0; k < RT; ++k) {
        const int r0 = k * 16 + 4 * q;
        const f32x4 main = product(k, DGc, (f32x4){0.f, 0.f, 0.f, 0.f});
        ST_MARK(0);
        const int4 ri = *reinterpret_cast<const int4*>(rowidx + r0);
        const unsigned trow = (unsigned)t * (unsigned)a.N, jb = (unsigned)j * 4u;
        const unsigned o0 = ((unsigned)ri.x + trow) * 256u + jb, o1 = ((unsigned)ri.y + trow) * 256u + jb;
        const unsigned o2 = ((unsigned)ri.z + trow) * 256u + jb, o3 = ((unsigned)ri.w + trow) * 256u + jb;
        const float v0 = cS[4][0] > 0.f ? main[0] : 0.f, v1 = cS[4][1] > 0.f ? main[1] : 0.f;
        const float v2 = cS[4][2] > 0.f ? main[2] : 0.f, v3 = cS[4][3] > 0.f ? main[3] : 0.f;
        if (full_wg) {
          st32(a.dxp, o0, v0); st32(a.dxp, o1, v1); st32(a.dxp, o2, v2); st32(a.dxp, o3, v3);
        } else {
          if (rowok[r0] != 0.f) st32(a.dxp, o0, v0);
          if (rowok[r0 + 1] != 0.f) st32(a.dxp, o1, v1);
          if (rowok[r0 + 2] != 0.f) st32(a.dxp, o2, v2);
          if (rowok[r0 + 3] != 0.f) st32(a.dxp, o3, v3);
        }
        ST_MARK(1);
        accum(cS, k, DGc);
        const int nk = k + 1 < RT ? k + 1 : 0, nt = k + 1 < RT ? t : t - 1;
        load_c(cS, nt > 0 ? nt : 0, nk);
        ST_MARK(2);
      }
      if (t >= 2 && tid < rows) sp_store(t - 2, tid, spu, spg, spu2, spg2);
      WG_BARRIER();
      ST_MARK(3);
    }
    ST_DUMP_AT(4, 8);
  }

  if (a.dh0 && team) {
    for (int r = 4 * q; r < rows; r += 16)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (rowok[r + i] != 0.f) a.dh0[(row0 + r + i) * H + j] = CAR[(r + i) * HS + j];
  }
  float* slab = a.ws + (long)blockIdx.x * bwd_slab_floats(a.A);
  float* s_role = slab + (team ? 192 * 64 : 0);
  float* s_w2 = slab + 2 * 192 * 64;
  float* s_bih = s_w2 + (long)a.A * 64;
  float* s_bhh = s_bih + 192;
  float* s_b2 = s_bhh + 192;
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        s_role[(g * 64 + 16 * ws + 4 * q + i) * 64 + 16 * c + m] = accW[g][c][i];
  auto red4 = [&](float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; };
  sb_r = red4(sb_r); sb_z = red4(sb_z); sb_n = red4(sb_n); sb_hn = red4(sb_hn);
  if (team) {
#pragma unroll
    for (int ac = 0; ac < AC; ++ac)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int arow = 16 * ac + 4 * q + i;
        if (arow < a.A) s_w2[arow * 64 + j] = accW2[ac][i];
      }
#pragma unroll
    for (int ac = 0; ac < AC; ++ac) {
      const float v = red4(sb2[ac]);
      if (ws == 0 && q == 0 && 16 * ac + m < a.A) s_b2[16 * ac + m] = v;
    }
    if (q == 0) {            // every gate-gradient tile went through team 1
      s_bih[j] = sb_r; s_bih[64 + j] = sb_z; s_bih[128 + j] = sb_n;
      s_bhh[j] = sb_r; s_bhh[64 + j] = sb_z; s_bhh[128 + j] = sb_hn;
    }
  }
}

struct BwdRedArgs {
  const float* ws; int nwg; int A;
  float *dWih, *dWhh, *dW2, *dbih, *dbhh, *db2;
};

__global__ __launch_bounds__(256) void agent_bwd_reduce_kernel(BwdRedArgs a) {
  __shared__ float part[4][64];
  const long slab = bwd_slab_floats(a.A);
  const int el = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const long e = (long)blockIdx.x * 64 + el;
  float s = 0.f;
  if (e < slab) s = slab_sum(a.ws + e, slab, sg, 4, a.nwg);
  part[sg][el] = s;
  __syncthreads();
  if (sg != 0 || e >= slab) return;
  s = ((part[0][el] + part[1][el]) + part[2][el]) + part[3][el];
  long k = e;
  if (k < 192 * 64) { a.dWih[k] += s; return; }
  k -= 192 * 64;
  if (k < 192 * 64) { a.dWhh[k] += s; return; }
  k -= 192 * 64;
  if (k < (long)a.A * 64) { a.dW2[k] += s; return; }
  k -= (long)a.A * 64;
  if (k < 192) { a.dbih[k] += s; return; }
  k -= 192;
  if (k < 192) { a.dbhh[k] += s; return; }
  k -= 192;
  a.db2[k] += s;
}

static int marl_fwd_pipe_max_rt = 1;   // row tiles per workgroup up to which the pipelined unroll is used (measured: -6 % per update at 1 tile, neutral at 2-3; 0 = never)
extern "C" void marl_debug_set_pipe_max_rt(int v) { marl_fwd_pipe_max_rt = v; }
static int marl_fwd_rt_single = 8;   // measured: 1/2/3/5 tiles per workgroup -> 12.1/9.4/8.1/6.5 ms per 120-step rollout
extern "C" void marl_debug_set_rt_single(int v) { marl_fwd_rt_single = v < 1 ? 1 : v; }

ST_DEFINE_SETTER(marl_debug_stamps_fwd)
#ifdef MARL_STAMPS
extern "C" int marl_debug_stamps_bwd(void* p) { return marl_debug_stamps_fwd(p); }
#endif

// choose row tiles per workgroup: fill the CU budget, keep LDS within budget
inline int pick_rt(long R, size_t bytes_per_row, size_t fixed_bytes, int rt_cap, int cus = 256) {
  const long tiles = (R + 15) / 16;
  const size_t budget = 160 * 1024;
  int rt_max = (int)((budget - fixed_bytes) / (bytes_per_row * 16));
  if (rt_max > rt_cap) rt_max = rt_cap;
  if (rt_max < 1) rt_max = 1;
  int rt = (int)((tiles + cus - 1) / cus);
  if (rt < 1) rt = 1;
  if (rt > rt_max) rt = rt_max;
  return rt;
}

}  // namespace

// 1 when a T-step unroll of these dimensions over cu_budget CUs runs a kernel that stores (gi_out) / reads (gi_in) the
// input-side gate sums: the multi-tile unroll or the software-pipelined kernel (one row tile per workgroup, T >= 4).
// (Alignment of the actual pointers is checked at launch; a launch that cannot reuse computes.)
extern "C" int marl_agent_unroll_reuse_supported(int B, int T, int N, int O, int A, int cu_budget) {
  if (B <= 0 || T < 2 || A > 32 || A < 1 || cu_budget < 0 || cu_budget > 256 || (O % 4) != 0 || O < 4) return 0;
  if (!marl_switches()->fwd_xs) return 0;
  if (cu_budget == 0) cu_budget = 256;
  const long tiles = ((long)B * N + 15) / 16;
  const int want = (int)((tiles + cu_budget - 1) / cu_budget);
  const int cap2 = (NLDW * FNT) / (16 * (O / 4));
  (void)want;
  if (cap2 < 1) return 0;                                        // no vector path
  // (wide observations too: the reading launch is not held to the rows its prefetch registers cover, see commit())
  return 1;
}

extern "C" int marl_agent_unroll_fwd(const marl_agent_weights_t* w, const float* obs, long obs_bs, int obs_t0,
                                     const int* ufed, long u_bs, int u_t0, const int* ep_len, const int* ep_map,
                                     const float* h0, float* q, float* hs, float* h_last, float* saved, int B,
                                     int T, int N, int O, int A, int last_action, int reuse_network, int cu_budget,
                                     float* gi_out, const float* gi_in, void* stream) {
  if (B <= 0 || T <= 0) return 0;
  if (w->H != H || A > 32 || A < 1 || cu_budget < 0 || cu_budget > 256) return (int)hipErrorInvalidValue;
  if (cu_budget == 0) cu_budget = 256;      // CUs this launch may occupy: 128 lets two independent unrolls run side by side
  FwdArgs a;
  a.W1 = w->fc1_w; a.b1 = w->fc1_b; a.Wih = w->w_ih; a.Whh = w->w_hh; a.bih = w->b_ih; a.bhh = w->b_hh;
  a.W2 = w->fc2_w; a.b2 = w->fc2_b;
  a.obs = obs; a.obs_bs = obs_bs; a.obs_t0 = obs_t0; a.ufed = ufed; a.u_bs = u_bs; a.u_t0 = u_t0; a.ep_len = ep_len; a.ep_map = ep_map; a.h0 = h0; a.q = q; a.hs = hs; a.h_last = h_last; a.saved = saved;
  a.B = B; a.T = T; a.N = N; a.O = O; a.A = A; a.gi_out = saved ? gi_out : nullptr; a.gi_in = gi_in;
  a.has_act = last_action ? 1 : 0; a.has_id = reuse_network ? 1 : 0;
  a.I = O + (last_action ? A : 0) + (reuse_network ? N : 0);
  a.KC = (a.I + 15) / 16;
  a.R = (long)B * N;
  const int KS = a.KC * 16 + MARL_PAD_K;
  const size_t per_row = (size_t)(KS + 3 * HS) * 4 + 32 + 8;   // + row tables: 2 long + 4 int, + 2 int of the DMA kernels' action table
  const size_t fixed = (size_t)4 * a.KC * 64 * 16 + 16 + (((size_t)T * 4 + 15) & ~(size_t)15);   // fc1 fragments + step flags of the x-reusing variants (pipelined: 4; else one per step)
  a.vload = (O % 4 == 0) && ((reinterpret_cast<uintptr_t>(obs) & 15) == 0) && O >= 4;
  int rt_cap = 8;
  bool half = false, dma = false, w2l = false;
  const bool xs_off = !marl_switches()->fwd_xs;      // A/B switch for measurements (common.h: MarlSwitches)
  // switch fwd_dma = 1: the activation-saving unroll fills its observation tile by LDS-DMA (DMA kernels).  OFF by
  // default - measured slower (2s3z / 4096 envs 1.75 vs 1.62 ms; MMM2 / 1024 envs 2.98 ms at three row tiles per workgroup vs
  // 2.24 ms at two through registers): the issuing waves spend a third of every step in the nine to eleven DMA issues
  // (profiles/archive/r03_stamps_dma.txt), although a wave alone issues such a DMA every ~100 cycles (profiles/archive/r03_dma_probe.txt)
  const int dma_mode = marl_switches()->fwd_dma;
  const bool xs_req = a.vload && gi_in && !saved && T >= 2 && !xs_off;      // the launch reads stored input-side sums
  if (a.vload && !xs_req) {   // the workgroup keeps one step's obs tile (rows * O/4 float4) in NLDW * 512 registers
    int cap2 = (NLDW * FNT) / (16 * (O / 4));
    const long tiles = (a.R + 15) / 16;
    const int cus = T > 1 ? cu_budget : 256;
    const int want = (int)((tiles + cus - 1) / cus);                 // row tiles per workgroup that fill the CUs in one round
    // activation-saving unroll of wide observations: the observation tile goes through LDS-DMA, the workgroup is not held
    // to the rows its prefetch registers would cover (DMA kernels)
    if (saved && T > 1 && dma_mode == 1) { dma = true; cap2 = 8; }
    // activation-saving unroll, wide observations, two action tiles: six prefetch registers, fc2 fragments in LDS (W2L kernels)
    const bool w2l_off = !marl_switches()->fwd_w2l;      // A/B switch
    // (only where it makes the launch a single round of workgroups: beside the target unroll under the pair schedule -
    // cu_budget 128 - three tiles per workgroup were SLOWER than two, 2.57 vs 2.24 ms at MMM2 / 1024 envs; alone on the chip
    // 1.25 vs 1.75 ms, profiles/archive/r03_mmm2_schedules.txt)
    const int cap6 = (6 * FNT) / (16 * (O / 4));
    if (saved && !dma && T > 1 && A > 16 && cap2 < want && cap2 < 8 && want <= cap6 && !w2l_off) { w2l = true; cap2 = cap6; }
    if (cap2 < want && cap2 < 8 && T > 1 && O % 8 == 0 && !saved) {  // wide observations: the registers hold one column
      half = true;                                                   // half of the tile at a time (HALF kernels).  Not the
      cap2 = (NLDW * FNT) / (16 * (O / 8));                          // activation-saving variant: hipcc cannot count its
    }                                                                // stores across the tile loop and waits vmcnt(0) for
                                                                     // the right half at the end of every gate phase
                                                                     // (measured 2.2 -> 3.2 ms at MMM2 / 1024 envs)
    if (cap2 < 1) { a.vload = 0; half = false; } else if (cap2 < rt_cap) rt_cap = cap2;
  }
  // a single step (rollout) is latency-bound: many small workgroups overlap their prologues better than
  // 256 big ones; a long unroll amortises the prologue and wants one workgroup per CU
  if (T == 1 && rt_cap > marl_fwd_rt_single) rt_cap = marl_fwd_rt_single;
  const size_t fixed_k = fixed + (w2l ? (size_t)2 * 4 * 64 * 16 : 0);
  a.RT = pick_rt(a.R, per_row, fixed_k, rt_cap, T > 1 ? cu_budget : 256);
  const size_t lds = fixed_k + per_row * a.RT * 16;
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  const long rows = a.RT * 16;
  dim3 grid((unsigned)((a.R + rows - 1) / rows)), block(FNT);
  hipStream_t s = (hipStream_t)stream;
  // per-step outputs are addressed with 32-bit byte offsets
  if ((double)B * T * N * H * 4.0 >= 4294967296.0) return (int)hipErrorInvalidValue;
  hipError_t e;
  // few row tiles per workgroup and a long unroll: the software-pipelined variant (one barrier per step)
  if (a.vload && !dma && !w2l && a.RT <= marl_fwd_pipe_max_rt && T >= 4 && (long)a.RT * 16 * (O / 4) <= (long)NLDW * FNT) {
    const size_t per_row_p = (size_t)(2 * KS + 4 * HS) * 4 + 32;
    const size_t lds_p = fixed + per_row_p * a.RT * 16 + 64;
    if (lds_p <= 160 * 1024) {
      const void* fp;
      const bool xs = gi_in && !saved && !xs_off;
      if (A <= 16) fp = saved ? (const void*)agent_fwd_pipe_kernel<1, true> : xs ? (const void*)agent_fwd_pipe_kernel<1, false, true> : (const void*)agent_fwd_pipe_kernel<1, false>;
      else fp = saved ? (const void*)agent_fwd_pipe_kernel<2, true> : xs ? (const void*)agent_fwd_pipe_kernel<2, false, true> : (const void*)agent_fwd_pipe_kernel<2, false>;
      e = hipFuncSetAttribute(fp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p);
      if (e != hipSuccess) return (int)e;
      void* kp[] = {(void*)&a};
      e = hipLaunchKernel(fp, grid, block, kp, lds_p, s);
      if (e != hipSuccess) return (int)e;
      MARL_CHECK_LAUNCH();
      return 0;
    }
  }
  const void* fn;
#define FWD_PICK(AC_, SV_, VL_) (const void*)agent_fwd_kernel<AC_, SV_, VL_>
  const bool sv = saved != nullptr, vl = a.vload != 0;
  if (xs_req) {
    fn = A <= 16 ? (const void*)agent_fwd_kernel<1, false, true, NLDW, true> : (const void*)agent_fwd_kernel<2, false, true, NLDW, true>;
  } else if (w2l) {
    fn = (const void*)agent_fwd_kernel<2, true, true, 6, false, false, false, true>;
  } else if (dma) {
    fn = A <= 16 ? (const void*)agent_fwd_kernel<1, true, true, NLDW, false, false, true> : (const void*)agent_fwd_kernel<2, true, true, NLDW, false, false, true>;
  } else if (vl && half) {
    fn = A <= 16 ? (const void*)agent_fwd_kernel<1, false, true, NLDW, false, true> : (const void*)agent_fwd_kernel<2, false, true, NLDW, false, true>;
  } else
  if (A <= 16) fn = sv ? (vl ? FWD_PICK(1, true, true) : FWD_PICK(1, true, false)) : (vl ? FWD_PICK(1, false, true) : FWD_PICK(1, false, false));
  else fn = sv ? (vl ? FWD_PICK(2, true, true) : FWD_PICK(2, true, false)) : (vl ? FWD_PICK(2, false, true) : FWD_PICK(2, false, false));
#undef FWD_PICK
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  void* kargs[] = {(void*)&a};
  e = hipLaunchKernel(fn, grid, block, kargs, lds, s);
  if (e != hipSuccess) return (int)e;
  MARL_CHECK_LAUNCH();
  return 0;
}

static int bwd_rt(long R, int A) {
  const int AC = A <= 16 ? 1 : 2;
  const int QS = AC * 16 + 4;
  const size_t per_row = (size_t)(DGS + 2 * QS + HS) * 4 + 12;
  int cap = (NQ * BNT) / (16 * AC * 16);      // dq prefetch registers cover rows*QP elements
  if (cap > 8) cap = 8;
  return pick_rt(R, per_row, 4 * 64 * 4, cap);
}

extern "C" size_t marl_agent_bwd_workspace(int B, int N, int A) {
  const long R = (long)B * N;
  const int rt = bwd_rt(R, A);
  const long nwg = (R + rt * 16 - 1) / (rt * 16);
  return (size_t)nwg * bwd_slab_floats(A) * sizeof(float);
}

extern "C" int marl_agent_unroll_bwd(const marl_agent_weights_t* w, const float* dq, const int* dq_idx,
                                     const float* dq_val, const int* dq_idx2, const float* dq_val2, int dq_gdiv,
                                     const float* dhs,
                                     const float* saved, const float* hs, float* dxp, float* dh0,
                                     const marl_agent_grads_t* g, float* ws, size_t ws_bytes,
                                     int B, int T, int N, int A, void* stream) {
  if (B <= 0 || T <= 0) return 0;
  if (w->H != H || A > 32 || A < 1) return (int)hipErrorInvalidValue;
  if (ws_bytes < marl_agent_bwd_workspace(B, N, A)) return (int)hipErrorInvalidValue;
  if ((double)B * T * N * H * 4.0 >= 4294967296.0) return (int)hipErrorInvalidValue;   // 32-bit byte offsets into dxp
  BwdArgs a;
  a.Wih = w->w_ih; a.Whh = w->w_hh; a.W2 = w->fc2_w;
  a.dq = dq; a.dq_idx = dq_idx; a.dq_val = dq_val; a.dq_idx2 = dq_idx2; a.dq_val2 = dq_val2; a.dq_gdiv = dq_gdiv > 1 ? dq_gdiv : 1;
  a.dhs = dhs; a.saved = saved; a.hs = hs; a.dxp = dxp; a.dh0 = dh0; a.ws = ws;
  a.B = B; a.T = T; a.N = N; a.A = A; a.R = (long)B * N;
  const int AC = A <= 16 ? 1 : 2;
  const int QS = AC * 16 + 4;
  const size_t per_row = (size_t)(DGS + 2 * QS + HS) * 4 + 12;
  a.RT = bwd_rt(a.R, A);
  const size_t lds = per_row * a.RT * 16 + 4 * 64 * 4;
  const long rows = a.RT * 16;
  const unsigned nwg = (unsigned)((a.R + rows - 1) / rows);
  dim3 grid(nwg), block(BNT);
  hipStream_t s = (hipStream_t)stream;
  hipError_t e;
  const void* fn;
  const bool sp = dq_idx != nullptr, sp2 = sp && dq_idx2 != nullptr;
  if (sp && !dq_val) return (int)hipErrorInvalidValue;
  if (!sp && (!dq || dq_idx2)) return (int)hipErrorInvalidValue;
  if (sp2 && !dq_val2) return (int)hipErrorInvalidValue;
#define BWD_PICK(AC_) (dhs ? (sp2 ? (const void*)agent_bwd_kernel<AC_, true, 2> : sp ? (const void*)agent_bwd_kernel<AC_, true, 1> : (const void*)agent_bwd_kernel<AC_, true, 0>) \
                           : (sp2 ? (const void*)agent_bwd_kernel<AC_, false, 2> : sp ? (const void*)agent_bwd_kernel<AC_, false, 1> : (const void*)agent_bwd_kernel<AC_, false, 0>))
  if (AC == 1) fn = BWD_PICK(1);
  else fn = BWD_PICK(2);
#undef BWD_PICK
  size_t lds_used = lds;
  // few row tiles per workgroup (small shards), sparse dq: the one-barrier pipelined variant
  const int pipe_max_rt = marl_switches()->bwd_pipe_max_rt;   // A/B switch for measurements (helps at every RT its LDS allows: +15 % per update at 1 tile, +1 % at 4)
  if (sp && a.RT <= pipe_max_rt && T >= 2) {
    const size_t lds_p = ((size_t)(2 * DGS + HS + 12) * 4 + 12) * rows + 4 * 64 * 4;
    if (lds_p <= 160 * 1024) {
#define BWDP_PICK(AC_) (dhs ? (sp2 ? (const void*)agent_bwd_pipe_kernel<AC_, true, 2> : (const void*)agent_bwd_pipe_kernel<AC_, true, 1>) \
                            : (sp2 ? (const void*)agent_bwd_pipe_kernel<AC_, false, 2> : (const void*)agent_bwd_pipe_kernel<AC_, false, 1>))
      fn = AC == 1 ? BWDP_PICK(1) : BWDP_PICK(2);
#undef BWDP_PICK
      lds_used = lds_p;
    }
  }
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_used);
  if (e != hipSuccess) return (int)e;
  void* kargs[] = {(void*)&a};
  e = hipLaunchKernel(fn, grid, block, kargs, lds_used, s);
  if (e != hipSuccess) return (int)e;
  MARL_CHECK_LAUNCH();
  BwdRedArgs r;
  r.ws = ws; r.nwg = (int)nwg; r.A = A;
  r.dWih = g->w_ih; r.dWhh = g->w_hh; r.dW2 = g->fc2_w; r.dbih = g->b_ih; r.dbhh = g->b_hh; r.db2 = g->fc2_b;
  const long slab = bwd_slab_floats(A);
  hipLaunchKernelGGL(agent_bwd_reduce_kernel, dim3((unsigned)((slab + 63) / 64)), dim3(256), 0, s, r);
  MARL_CHECK_LAUNCH();
  return 0;
}
