// GRU-agent unroll with every fp32 product as six bf16 MFMA products (x6.h) - opt-in args.gemm_mode = "bf16x6": the three forward
// unrolls of a Q-learning update (reference controller/share_params.py:125-168, network/q_network.py:16-21; q_learner.py:96-117) -
// the eval pass that saves activations for BPTT, the target pass, the double-Q continuation that reads the eval pass's input-side
// gate sums.  The default path is agent.hip on v_mfma_f32_16x16x4_f32.
//
// Why another decomposition: as bf16 triples the weight fragments of a hidden-unit slice no longer fit one wave
// beside its working set (W_ih + W_hh + fc1 + fc2 slices = 204 registers), and 190 KB of pre-split weights do not fit LDS.  So the
// two teams of the workgroup hold DIFFERENT weights instead of the same ones:
//   team I (waves 4-7, slice s): fc1 and W_ih fragments (up to 108 registers) - what depends only on a step's INPUT:
//          x(t+2) = relu(fc1(in(t+2))) and the input-side gate sums gi(t+1) = bias + x(t+1) W_ih, one / two steps ahead of the chain;
//   team R (waves 0-3, slice s): W_hh fragments (72 registers) - the recurrent part of step t: accumulators start from the handed
//          gi(t), += h W_hh, gate math, h' (kept in fp32 registers for the blend of the next step);
//   q(t-1) = fc2(h), the observation prefetch and the candidate gate's x W_in go to whichever team has the time (see the kernel).
// Products are out[row][unit] = act[row][k] W[unit][k]: activations are the MFMA A operand, read as 16 bytes per lane (row m, 8
// consecutive k) from bf16 PLANES in LDS - every activation element is split once, where it is produced, and shared by the four
// slice waves that consume it (5.5 vector instructions per element, once); weights are the B operand, pre-split in registers.  The
// accumulator layout (lane (q, m): rows 4q + r of unit 16s + m) is agent.hip's, so saved activations and gate sums keep their formats.
// One workgroup barrier per step; every LDS buffer is double-buffered by step parity:
//   In[b]  input tile of step t+2 (planes)     Xp[b]  x(t+1) (planes)     Hp[b]  h fed into step t (planes)     GI[b]  gi(t) (fp32, accumulator layout)
// ~70 KB of LDS per 16-row tile: one or two row tiles per workgroup, as many workgroups as that takes (they run in rounds; the
// last round of a launch holds one tile per workgroup).
#include "x6.h"
#include <cstdlib>
#include "../../include/marl_hip.h"

// the plain unroll in the round-6 decomposition (agent_x6p.hip): row tiles per workgroup (0: not its launch), and its launch
int marl_agent_x6p_tiles(int B, int T, int N, int O, int A, int last_action, int reuse_network, int cu_budget);
int marl_agent_x6p_launch(const marl_agent_weights_t* w, const float* obs, long obs_bs, int obs_t0, const int* ufed, long u_bs, int u_t0,
                          const int* ep_len, const int* ep_map, const float* h0, float* q, float* h_last, int B, int T, int N, int O, int A,
                          int last_action, int reuse_network, int tpw, void* stream);

namespace {

constexpr int H = 64;
constexpr int XNT = 512;          // 8 waves
constexpr int HP = 72;            // pitch (bf16) of the 64-wide planes: 144-byte rows spread a 16-lane group over all banks
constexpr int NLD = 4;            // float4 prefetch registers per thread (the next-but-two step's observation tile)

struct X6Args {
  const float *W1, *b1, *Wih, *Whh, *bih, *bhh, *W2, *b2;
  const float* obs; long obs_bs; int obs_t0;
  const int* ufed; long u_bs; int u_t0;
  const int* ep_len; const int* ep_map;
  const float* h0;
  float *q, *hs, *h_last;
  float* saved;            // SAVE: (T+1) * R16 * 6 * 64 floats in agent.hip's tile layout
  float* gi_out;           // SAVE: input-side gate sums for a later XS launch, or null
  const float* gi_in;      // XS: gi_out of the eval unroll (its step t + 1 is this unroll's step t)
  int B, T, N, O, A, I, KI;        // KI: input width rounded up to 32
  int has_act, has_id, RT;
  int n_full;              // workgroups that hold RT row tiles; the rest hold one
  long R;
};

#define WG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// the six products of a split multiply, smallest first (x6.h: mm6), for schedules that interleave several accumulators
#define X6_TERMS(OP) OP(m, m) OP(h, l) OP(l, h) OP(h, m) OP(m, h) OP(h, h)

// A fragment of W (row-major, ldw floats per row): A[i][slot j] = W[row0 + i][32 c + 8g + j]  (rows >= rows_valid and columns >= K: 0)
__device__ __forceinline__ F3 wfrag(const float* W, int ldw, int row0, int rows_valid, int K, int c, int lane) {
  const int i = lane & 15, g = lane >> 4, row = row0 + i;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 32 * c + 8 * g + j;
    v[j] = (row < rows_valid && k < K) ? W[(long)row * ldw + k] : 0.f;
  }
  return split8((f32x4){v[0], v[1], v[2], v[3]}, (f32x4){v[4], v[5], v[6], v[7]});
}
// B fragment from a plane tile (hi plane at pl, the others ps elements further): lane (g, m) reads row m, columns 32 c + 8g .. + 7
__device__ __forceinline__ F3 bfrag(const short* pl, int pitch, int ps, int c, int lane) {
  const int m = lane & 15, g = lane >> 4;
  const short* p = pl + m * pitch + 32 * c + 8 * g;
  F3 f;
  f.h = *reinterpret_cast<const i32x4*>(p);
  f.m = *reinterpret_cast<const i32x4*>(p + ps);
  f.l = *reinterpret_cast<const i32x4*>(p + 2 * ps);
  return f;
}
// accumulator tile (rows row0 + r, r = 0..3, of column col) -> planes: one 2-byte write per row and plane
__device__ __forceinline__ void put4(short* pl, int pitch, int ps, int row0, int col, const f32x4& v) {
  const F3h f = split4(v);
  short* p = pl + row0 * pitch + col;
  const int d[3][2] = {{f.h[0], f.h[1]}, {f.m[0], f.m[1]}, {f.l[0], f.l[1]}};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    p[k * ps] = (short)d[k][0];
    p[k * ps + pitch] = (short)((unsigned)d[k][0] >> 16);
    p[k * ps + 2 * pitch] = (short)d[k][1];
    p[k * ps + 3 * pitch] = (short)((unsigned)d[k][1] >> 16);
  }
}
__device__ __forceinline__ f32x4 relu4x(const f32x4& v) { return (f32x4){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)}; }
__device__ __forceinline__ f32x4 splat(float v) { return (f32x4){v, v, v, v}; }
// GRU gate math (common.h: gru_point_plain) with every fused / unfused operation spelled out: all instantiations round the same
// way, so the launch that reads stored gate sums == the one that computes them, bit for bit, whatever the compiler would contract
__device__ __forceinline__ void gru_point_x6(float ar, float az, float ain, float ahn, float hp, float& r, float& z, float& n, float& h) {
  r = __builtin_amdgcn_rcpf(__fadd_rn(1.0f, __builtin_amdgcn_exp2f(__fmul_rn(ar, -1.4426950408889634f))));
  z = __builtin_amdgcn_rcpf(__fadd_rn(1.0f, __builtin_amdgcn_exp2f(__fmul_rn(az, -1.4426950408889634f))));
  const float e = __builtin_amdgcn_exp2f(__fmul_rn(__fmaf_rn(r, ahn, ain), 2.8853900817779268f));
  n = __fmaf_rn(-2.0f, __builtin_amdgcn_rcpf(__fadd_rn(e, 1.0f)), 1.0f);
  h = __fmaf_rn(z, hp, __fmul_rn(__fsub_rn(1.0f, z), n));
}
// offset (floats) into the tile layout of agent.hip's saved activations / gate sums: [t][row tile][plane][column tile c][lane][4]
__device__ __forceinline__ long sv_off(long tile_t, int planes, int plane, int c, int lane) {
  return ((tile_t * planes + plane) * 4 + c) * 256 + lane * 4;
}

// Products are  D[row][unit] = act[row][k] W[unit][k]  (activations = A operand, weights = B operand): the accumulator layout -
// lane (q, m): rows 4q + r of unit 16s + m - is the layout of agent.hip, so the saved activations (SAVE: six planes per row-step for
// the fp32 BPTT kernel), the input-side gate sums an unroll stores (gi_out) and a later one reads (XS: gi_in) keep their formats.
// SAVE: the eval network's pass.  XS: the double-Q continuation - team I computes only the steps flagged in xmask (the last step and
// steps at which a row has ep_len - 1 == t), the other steps' sums come from gi_in (the eval pass's step t + 1).
//
// The two teams run DIFFERENT loops (same barrier sequence): each keeps only its own weights and working set in registers.  Three
// pieces of a step belong to neither team by nature and are dealt by variant (ALLP / FC2I / NGR below), by two rules: the longer team
// gives work away (stamps: tools/stamps_unroll_x6.py), and no wave may end up waiting for a STORE to complete in the steady state
// (loads and stores share one in-order counter; a wait for a load is a counted vmcnt only while the stores issued after it are a
// fixed number):
//   the observation prefetch (loads a step ahead, split into the input planes): team R's; team I's in the two-tile saving variant
//          (team R's store count per step varies there) - team I then stores a FIXED number of planes per step (steps past the end and
//          row tiles past the batch recompute the last valid one and store the same values again) - and in XS (a few steps only);
//   fc2 / q: team I's wherever team I has no loads left (and in XS, where team R has the gate-sum loads); team R's otherwise;
//   the candidate gate's input-side product x W_in: team R's, a step ahead and off its chain, in the variants with one action tile.
// NK1: 32-wide k chunks of fc1 whose weight fragments the kernel holds (3: inputs up to 96 wide - 2s3z; 5: up to 160 - 3s5z; 7: up to
// 224 - MMM2; beyond 3 one row tile per workgroup only: the input planes grow with the width).  AC: 16-wide action tiles of fc2
template <int RTC, bool SAVE, bool XS, bool GIO = false, int NK1 = 3, int AC = 1>
__global__ __launch_bounds__(XNT, 2) void agent_fwd_x6_kernel(X6Args a) {
  static_assert(!(SAVE && XS) && (SAVE || !GIO), "XS: no saving; GIO: the saving pass also stores its input-side gate sums");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave >> 2, s = wave & 3;      // team 0 = R (recurrent), team 1 = I (input side); hidden-unit slice s
  const int q = lane >> 4, m = lane & 15;
  const int rows = RTC * 16;
  const int IP = a.KI + 8;                        // pitch of the input planes
  const int IN_B = 3 * rows * IP * 2, XP_B = 3 * rows * HP * 2, GI_B = RTC * 4 * 3 * 1024;
  short* In0 = reinterpret_cast<short*>(smem);                          // [2][3][rows][IP]
  short* Xp0 = reinterpret_cast<short*>(smem + 2 * IN_B);               // [2][3][rows][HP]
  short* Hp0 = reinterpret_cast<short*>(smem + 2 * IN_B + 2 * XP_B);    // [2][3][rows][HP]
  float* GI0 = reinterpret_cast<float*>(smem + 2 * IN_B + 4 * XP_B);    // [2][RTC][4][3][64] f32x4
  long* rowobs = reinterpret_cast<long*>(smem + 2 * IN_B + 4 * XP_B + 2 * GI_B);
  long* rowu = rowobs + rows;
  int* rowidx = reinterpret_cast<int*>(rowu + rows);
  int* rown = rowidx + rows;
  int* rowlen = rown + rows;
  int* rowrho = rowlen + rows;
  int* xmask = rowrho + rows;                     // [T] (XS): step t is computed in full
  constexpr int NK1R = NK1 > 5 ? 5 : NK1;         // fc1 chunks whose fragments stay in registers; the others as ready fragments in LDS
  char* WL1 = reinterpret_cast<char*>(xmask + ((a.T + 3) & ~3));      // [4 slices][NK1 - NK1R][3 planes][64 lanes] 16 bytes
  auto inp = [&](int b) { return In0 + b * 3 * rows * IP; };
  auto xpp = [&](int b) { return Xp0 + b * 3 * rows * HP; };
  auto hpp = [&](int b) { return Hp0 + b * 3 * rows * HP; };

  // workgroups [0, n_full) hold RTC row tiles, the ones after them one tile each (the last round of a launch that runs in rounds)
  const long NTILES = (a.R + 15) >> 4;
  const long bx = blockIdx.x;
  const long tile0 = bx < a.n_full ? bx * RTC : (long)a.n_full * RTC + (bx - a.n_full);
  const long row0 = tile0 * 16;
  const int cap = bx < a.n_full ? RTC : 1;
  const int RTW = RTC == 1 ? 1 : (int)((NTILES - tile0) < cap ? (NTILES - tile0) : cap);      // (one tile per workgroup: it exists)
  for (int r = tid; r < rows; r += XNT) {
    long rho = row0 + r;
    if (rho > a.R - 1) rho = a.R - 1;             // clamped duplicates: same loads, same values, same stores
    const long b = rho / a.N;
    const int n = (int)(rho % a.N);
    rowidx[r] = (int)(b * a.T * a.N + n);
    rowobs[r] = ((a.ep_map ? (long)a.ep_map[b] : b) * a.obs_bs + n) * a.O;
    rowu[r] = b * a.u_bs + n;
    rown[r] = n;
    rowlen[r] = a.ep_len ? a.ep_len[b] : 0x7fffffff;
    rowrho[r] = (int)rho;
  }
  if (XS) for (int e = tid; e < a.T; e += XNT) xmask[e] = e == a.T - 1 ? 1 : 0;
  __syncthreads();
  if (XS && tid < RTW * 16) { const int L = rowlen[tid]; if (L >= 1 && L - 1 < a.T) xmask[L - 1] = 1; }
  // constant columns of both input buffers: empty one-hot, agent id, zero pad - all three planes
  const int O = a.O;
  for (int e = tid; e < 2 * 3 * rows * (a.KI - O); e += XNT) {
    const int w = a.KI - O, rr = e / w, k = O + e % w;        // rr = (b * 3 + plane) * rows + row
    const int plane = (rr / rows) % 3, r = rr % rows;
    short v = 0;
    if (plane == 0 && a.has_id && k >= a.I - a.N && k < a.I && rown[r] == k - (a.I - a.N)) v = (short)0x3F80;
    In0[rr * IP + k] = v;
  }
  const int Tm1 = a.T - 1;
  const int KC1 = a.KI >> 5;
  const int u = 16 * s + m;
  auto full = [&](int t) { return !XS || xmask[t < Tm1 ? t : Tm1] != 0; };      // step t's input side is computed here
  ST_DECL(4);

  // ---- observation prefetch: thread -> (row, 4-column group) of the tile, the same every step.  ALLP: the whole prefetch (loads a step
  // ahead, split into the input planes) is TEAM R's - its waves wait a third of a step at the barrier, team I is the longer team and
  // keeps none of it - except where team R would then wait for its own stores: the saving variant with two row tiles (its store count
  // per step is not a constant) and XS (which loads an input tile for a few steps only, slot by slot); there it is team I's.
  // NGR: the candidate gate's input-side product x W_in runs in team R (a step ahead, off its chain) - team I is the longer team, and
  // team R waits a third of a step at the barrier.  Not in XS, where team I is idle anyway.  (Same products in the same order on
  // either team: the gate sums stay bit-identical between the variants.)
  constexpr bool NGR = !XS && AC == 1;      // (two action tiles: team R has no registers to spare)
  constexpr bool ALLP = !XS && !(SAVE && RTC == 2);
  // FC2I: fc2 / q in team I - in XS (team R then has loads only) and wherever team I has no loads to count stores against (ALLP): the
  // one wave per row tile that carries fc2 would otherwise be team R's longest, on the chain
  // (round 5, again: fc2 of the plain unroll in team R - stamps at one tile show team I's fc2 wave 450 cycles behind its three
  // partners and team R waiting a third of the step - measured 0.183 -> 0.200 ms at 512 envs, 1.018 -> 1.039 at 4096: not kept)
  constexpr bool FC2I = XS || (ALLP && AC == 1);
  constexpr int NLDK = ALLP ? 3 : NLD;
  constexpr int PT = 256;                          // participating threads: one team
  const int pti = ALLP ? tid : tid - 256;          // this thread's index among them
  const bool pfw = ALLP ? team == 0 : team == 1;
  const int O4 = O >> 2, n4 = rows * O4;
  const float invO4 = 1.0f / (float)(O4 > 0 ? O4 : 1);
  const int NS = (n4 + PT - 1) / PT;               // prefetch slots (float4 per thread) the tile fills
  f32x4 pf[NLDK];
  long goff[NLDK]; int prk[NLDK];                   // (row << 16) | first column of the thread's float4
  int pt = 0, pu = -1, pu_lds0 = -1, pu_lds1 = -1;      // one-hot column currently set in each input buffer
  auto slot = [&](int i, int& prk_, long& goff_) {
    int e = PT * i + ((pti - 64 * i) & (PT - 1));      // (slot i rotated by i waves: a partly filled last slot is not wave 0's, which has fc2)
    if (e > n4 - 1) e = n4 - 1;
    if (e < 0) e = 0;
    const int r = (int)(((float)e + 0.5f) * invO4);
    const int k4 = e - r * O4;
    prk_ = (r << 16) | (4 * k4);
    goff_ = rowobs[r] + 4 * k4;
  };
  if (pfw) {
#pragma unroll
    for (int i = 0; i < NLDK; ++i) slot(i, prk[i], goff[i]);
  }
  const long urow = pfw && pti < rows ? rowu[pti] : 0;
  // (a wave whose 64 elements of a slot all lie past the tile skips the slot - wave-uniform)
  auto wave_has = [&](int i) { return i < NS && PT * i + (((pti & ~63) - 64 * i) & (PT - 1)) < n4; };
  auto issue = [&](int t) {
    const long toff = (long)(t + a.obs_t0) * a.N * O;
#pragma unroll
    for (int i = 0; i < NLDK; ++i)
      if (wave_has(i)) pf[i] = *reinterpret_cast<const f32x4*>(a.obs + goff[i] + toff);
    pt = t;
    int uu = -1;
    if (pti < rows && a.ufed && t + a.u_t0 >= 0) uu = a.ufed[urow + (long)(t + a.u_t0) * a.N];
    pu = uu;
  };
  auto commit = [&](int b, int& pu_lds) {          // prefetch registers -> input planes of buffer b (split once, here)
    short* P = inp(b);
#pragma unroll
    for (int i = 0; i < NLDK; ++i) {
      if (!wave_has(i)) break;
      const int r = prk[i] >> 16, lo = r * IP + (prk[i] & 0xffff);
      const f32x4 v = pt < rowlen[r] ? pf[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
      const F3h f = split4(v);
      *reinterpret_cast<i32x2*>(P + lo) = f.h;
      *reinterpret_cast<i32x2*>(P + rows * IP + lo) = f.m;
      *reinterpret_cast<i32x2*>(P + 2 * rows * IP + lo) = f.l;
    }
    if (a.has_act && pti < rows) {                 // one-hot(last action): bf16 1.0 in the hi plane, flipped in place
      const int pn = (pu >= 0 && pu < a.A) ? pu : -1;
      if (pn != pu_lds) {
        if (pu_lds >= 0) P[pti * IP + O + pu_lds] = 0;
        if (pn >= 0) P[pti * IP + O + pn] = (short)0x3F80;
        pu_lds = pn;
      }
    }
  };

  if (team == 1) {
    // =============================== team I: everything that depends only on a step's input ===============================
    F3 w1[NK1R], wi[6], w2[AC][2];                 // pre-split B fragments (lane (g, j): W[unit j][k = 32 c + 8g ..])
#pragma unroll
    for (int c = 0; c < NK1R; ++c) w1[c] = !(XS && NK1 > 3) && c < KC1 ? wfrag(a.W1, a.I, 16 * s, H, a.I, c, lane) : F3{};      // (XS with wide inputs: loaded where used, see fc1)
    if (!XS) {
#pragma unroll
      for (int c = NK1R; c < NK1; ++c) {
        const F3 f = c < KC1 ? wfrag(a.W1, a.I, 16 * s, H, a.I, c, lane) : F3{};
        i32x4* p = reinterpret_cast<i32x4*>(WL1) + (s * (NK1 - NK1R) + (c - NK1R)) * 192 + lane;
        p[0] = f.h; p[64] = f.m; p[128] = f.l;
      }
    }
    auto w1f = [&](int c) __attribute__((always_inline)) {      // (compile-time c after unrolling)
      if (c < NK1R) return w1[c < NK1R ? c : 0];
      const i32x4* p = reinterpret_cast<const i32x4*>(WL1) + (s * (NK1 - NK1R) + (c - NK1R)) * 192 + lane;
      F3 f;
      f.h = p[0]; f.m = p[64]; f.l = p[128];
      return f;
    };
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int c = 0; c < 2; ++c) wi[2 * g + c] = (NGR && g == 2) ? F3{} : wfrag(a.Wih, H, g * H + 16 * s, 3 * H, H, c, lane);
    float bias_2[AC];
#pragma unroll
    for (int ac = 0; ac < AC; ++ac) {
      if (FC2I) { w2[ac][0] = wfrag(a.W2, H, 16 * ac, a.A, H, 0, lane); w2[ac][1] = wfrag(a.W2, H, 16 * ac, a.A, H, 1, lane); }
      bias_2[ac] = FC2I && 16 * ac + m < a.A ? a.b2[16 * ac + m] : 0.f;
    }
    const float bias_1 = a.b1[u], bias_r = a.bih[u] + a.bhh[u], bias_z = a.bih[H + u] + a.bhh[H + u], bias_n = a.bih[2 * H + u];

    // x(ts) = relu(fc1(in)) of every row tile: this wave's 16 units -> planes Xp[bx] (and plane 1 of the saved activations).
    // Row tiles past the batch recompute the last valid one (a fixed number of stores per call, see above).
    auto fc1 = [&](int bin, int bxx, int ts) __attribute__((always_inline)) {
#pragma unroll
      for (int rt = 0; rt < RTC; ++rt) {
        const int rr = rt < RTW ? rt : RTW - 1;
        // three accumulators, chunk c on accumulator c % 3: independent chains of six products, issued round robin, three chunks'
        // fragments in registers at a time; chunks past the input width multiply chunk 0 by zero weights
        // (XS: the few steps computed in full - one chunk at a time, fewer registers; the same sums in the same order)
        f32x4 acc[3] = {splat(bias_1), splat(0.f), splat(0.f)};
        if (XS) {
#pragma unroll
          for (int c = 0; c < NK1; ++c) {
            const F3 wc = NK1 > 3 ? (c < KC1 ? wfrag(a.W1, a.I, 16 * s, H, a.I, c, lane) : F3{}) : w1[c < NK1R ? c : 0];      // (wide inputs: not kept in registers in this variant)
            mm6(bfrag(inp(bin) + rr * 16 * IP, IP, rows * IP, c < KC1 ? c : 0, lane), wc, acc[c % 3]);
          }
        } else {
#pragma unroll
          for (int c0 = 0; c0 < NK1; c0 += 3) {
            F3 xi[3], wf[3];
#pragma unroll
            for (int c = 0; c < 3; ++c)
              if (c0 + c < NK1) { xi[c] = bfrag(inp(bin) + rr * 16 * IP, IP, rows * IP, c0 + c < KC1 ? c0 + c : 0, lane); wf[c] = w1f(c0 + c); }
#define OP(p_, q_) _Pragma("unroll") for (int c = 0; c < 3; ++c) if (c0 + c < NK1) acc[c] = mm(xi[c].p_, wf[c].q_, acc[c]);
            X6_TERMS(OP)
#undef OP
          }
        }
        const f32x4 x = relu4x((acc[0] + acc[1]) + acc[2]);
        put4(xpp(bxx), HP, rows * HP, rr * 16 + 4 * q, u, x);
        if (SAVE) *reinterpret_cast<f32x4*>(a.saved + sv_off((long)ts * NTILES + tile0 + rr, 6, 1, s, lane)) = x;
      }
    };
    // gi(ts) = bias + x W_ih of every row tile -> GI[bg] (for the R wave of this slice) and gi_out
    auto gih = [&](int bxx, int bg, int ts) __attribute__((always_inline)) {
#pragma unroll
      for (int rt = 0; rt < RTC; ++rt) {
        const int rr = rt < RTW ? rt : RTW - 1;
        constexpr int NG = NGR ? 2 : 3;              // gates computed here = independent chains, round robin
        f32x4 ag[3] = {splat(bias_r), splat(bias_z), splat(bias_n)};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const F3 xb = bfrag(xpp(bxx) + rr * 16 * HP, HP, rows * HP, c, lane);
#define OP(p_, q_) _Pragma("unroll") for (int g = 0; g < NG; ++g) ag[g] = mm(xb.p_, wi[2 * g + c].q_, ag[g]);
          X6_TERMS(OP)
#undef OP
        }
        f32x4* gp = reinterpret_cast<f32x4*>(GI0) + ((bg * RTC + rr) * 4 + s) * 192 + lane;
        gp[0] = ag[0]; gp[64] = ag[1];
        if (!NGR) gp[128] = ag[2];
        // (GIO: the sums are stored to gi_out by the team R wave that consumes them - this team is the longer one in the saving
        // variants, by the stamps: gate sums + their stores 4 400 of 8 900 cycles per two-tile step)
      }
    };
    // XS: q(ts) = fc2(h) of row tile rt from planes Hp[bh]
    auto fc2 = [&](int bh, int ts, int rt) __attribute__((always_inline)) {
      F3 hb[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) hb[c] = bfrag(hpp(bh) + rt * 16 * HP, HP, rows * HP, c, lane);
#pragma unroll
      for (int at = 0; at < AC; ++at) {              // action tiles of 16
        f32x4 ac[2] = {splat(bias_2[at]), splat(0.f)};     // one chain per k chunk
#define OP(p_, q_) _Pragma("unroll") for (int c = 0; c < 2; ++c) ac[c] = mm(hb[c].p_, w2[at][c].q_, ac[c]);
        X6_TERMS(OP)
#undef OP
        const f32x4 acc = ac[0] + ac[1];
        if (16 * at + m < a.A) {
#pragma unroll
          for (int r = 0; r < 4; ++r) a.q[(unsigned)(rowidx[rt * 16 + 4 * q + r] + ts * a.N) * (unsigned)a.A + (unsigned)(16 * at + m)] = acc[r];      // (32-bit element offsets: the host checks B T N H 4 < 2^32)
        }
      }
    };

    __syncthreads();                               // constant columns, h0 planes (team R), step flags
    if (!ALLP) {
      issue(0); commit(0, pu_lds0);
      issue(Tm1 < 1 ? Tm1 : 1); commit(1, pu_lds1);
      issue(Tm1 < 2 ? Tm1 : 2);
    }
    WG_BARRIER();                                  // A: input planes of steps 0 and 1
    if (full(0)) fc1(0, 0, 0);
    if (full(1)) fc1(1, 1, 1);
    WG_BARRIER();                                  // B: x(0), x(1)
    if (!ALLP) commit(0, pu_lds0);                 // input(2)
    if (!XS && !ALLP) issue(Tm1 < 3 ? Tm1 : 3);
    if (full(0)) gih(0, 0, 0);
    WG_BARRIER();                                  // C: gi(0), input(2)
    for (int t = 0; t < a.T; ++t) {
      const int par = t & 1;
      if (!XS) {
        // steps past the end recompute step T - 1 (its input tile is what the clamped loads brought) and store the same values
        gih(par ^ 1, par ^ 1, t + 1 < a.T ? t + 1 : Tm1);        // gi(t+1) from x(t+1)
        ST_MARK(0);
        fc1(par, par, t + 2 < a.T ? t + 2 : Tm1);                // x(t+2) from input(t+2) -> the buffer x(t) has left
        if (FC2I && t > 0 && s < RTW) fc2(par, t - 1, s);        // q(t-1) from h fed into step t
        ST_MARK(1);
        // input tile of step t+3 -> the buffer fc1(t+1) finished with in the previous step; start the loads of step t+4
        if (!ALLP) {
          if (par) commit(0, pu_lds0); else commit(1, pu_lds1);
          issue(t + 4 < a.T ? t + 4 : Tm1);
        }
        ST_MARK(2);
      } else {
        if (t + 1 < a.T && full(t + 1)) gih(par ^ 1, par ^ 1, t + 1);
        if (t + 2 < a.T && full(t + 2)) fc1(par, par, t + 2);
        if (t > 0 && s < RTW) fc2(par, t - 1, s);                // q(t-1) from h fed into step t (RTC <= 4 tiles: one per wave)
        if (t + 3 < a.T && xmask[t + 3]) {         // rare: the load is not hidden behind a step
          issue(t + 3);
          if (par) commit(0, pu_lds0); else commit(1, pu_lds1);
        }
      }
      WG_BARRIER();
      ST_MARK(3);
    }
    if (FC2I && s < RTW) fc2(a.T & 1, a.T - 1, s);   // q of the last step
    ST_DUMP(4);
  } else {
    // =============================== team R: the recurrent part of a step ===============================
    F3 wh[6], w2[AC][2];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int c = 0; c < 2; ++c) wh[2 * g + c] = wfrag(a.Whh, H, g * H + 16 * s, 3 * H, H, c, lane);
    float bias_2[AC];
#pragma unroll
    for (int ac = 0; ac < AC; ++ac) {
      if (!FC2I) { w2[ac][0] = wfrag(a.W2, H, 16 * ac, a.A, H, 0, lane); w2[ac][1] = wfrag(a.W2, H, 16 * ac, a.A, H, 1, lane); }
      bias_2[ac] = !FC2I && 16 * ac + m < a.A ? a.b2[16 * ac + m] : 0.f;
    }
    const float bias_hn = a.bhh[2 * H + u];
    F3 win[2];                                      // NGR: W_in fragments (the n rows of W_ih), this wave's 16 units
    if (NGR) { win[0] = wfrag(a.Wih, H, 2 * H + 16 * s, 3 * H, H, 0, lane); win[1] = wfrag(a.Wih, H, 2 * H + 16 * s, 3 * H, H, 1, lane); }
    const float bias_nin = a.bih[2 * H + u];
    f32x4 anx[RTC];                                 // NGR: bias + x W_in of the step to come (computed during the step before it)
    // bias + x(ts) W_in of every row tile from planes Xp[bxx] (+ the n plane of gi_out): the products of team I's gih, same order
    auto nin = [&](int bxx, int ts) __attribute__((always_inline)) {
#pragma unroll
      for (int rt = 0; rt < RTC; ++rt) {
        if (rt >= RTW) break;
        f32x4 an = splat(bias_nin);
#pragma unroll
        for (int c = 0; c < 2; ++c) mm6(bfrag(xpp(bxx) + rt * 16 * HP, HP, rows * HP, c, lane), win[c], an);
        anx[rt] = an;
        if (GIO) *reinterpret_cast<f32x4*>(a.gi_out + sv_off((long)ts * NTILES + tile0 + rt, 3, 2, s, lane)) = an;
      }
    };
    // initial hidden state: fp32 registers (rows 4q + r of unit 16s + m) and planes Hp[0]
    f32x4 hreg[RTC];
#pragma unroll
    for (int rt = 0; rt < RTC; ++rt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) hreg[rt][r] = a.h0 ? a.h0[(long)rowrho[rt * 16 + 4 * q + r] * H + u] : 0.f;
      put4(hpp(0), HP, rows * HP, rt * 16 + 4 * q, u, hreg[rt]);
    }
    auto fc2 = [&](int bh, int ts, int rt) __attribute__((always_inline)) {
      F3 hb[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) hb[c] = bfrag(hpp(bh) + rt * 16 * HP, HP, rows * HP, c, lane);
#pragma unroll
      for (int at = 0; at < AC; ++at) {              // action tiles of 16
        f32x4 ac[2] = {splat(bias_2[at]), splat(0.f)};     // one chain per k chunk
#define OP(p_, q_) _Pragma("unroll") for (int c = 0; c < 2; ++c) ac[c] = mm(hb[c].p_, w2[at][c].q_, ac[c]);
        X6_TERMS(OP)
#undef OP
        const f32x4 acc = ac[0] + ac[1];
        if (16 * at + m < a.A) {
#pragma unroll
          for (int r = 0; r < 4; ++r) a.q[(unsigned)(rowidx[rt * 16 + 4 * q + r] + ts * a.N) * (unsigned)a.A + (unsigned)(16 * at + m)] = acc[r];      // (32-bit element offsets: the host checks B T N H 4 < 2^32)
        }
      }
    };
    // XS: stored input-side sums of the NEXT row tile (accumulator layout = the storing kernel's), a tile ahead
    f32x4 gB[3];
    auto gissue = [&](int ts, int rt) {
      if (XS) {
        const float* gp = a.gi_in + sv_off((long)ts * NTILES + tile0 + rt, 3, 0, s, lane);
        gB[0] = *reinterpret_cast<const f32x4*>(gp);
        gB[1] = *reinterpret_cast<const f32x4*>(gp + 1024);
        gB[2] = *reinterpret_cast<const f32x4*>(gp + 2048);
      }
    };
    __syncthreads();
    if (XS) gissue(Tm1 < 1 ? Tm1 : 1, 0);          // (step 0's sums = the storing unroll's step 1)
    if (ALLP) {                                    // the input tiles are this team's
      issue(0); commit(0, pu_lds0);
      issue(Tm1 < 1 ? Tm1 : 1); commit(1, pu_lds1);
      issue(Tm1 < 2 ? Tm1 : 2);
    }
    WG_BARRIER();                                  // A
    WG_BARRIER();                                  // B
    if (ALLP) { commit(0, pu_lds0); issue(Tm1 < 3 ? Tm1 : 3); }
    if (NGR) nin(0, 0);                            // x(0) is in Xp[0] since barrier B
    WG_BARRIER();                                  // C
    for (int t = 0; t < a.T; ++t) {
      const int par = t & 1;
      const bool fl = full(t);
#pragma unroll
      for (int rt = 0; rt < RTC; ++rt) {
        if (rt >= RTW) break;
        const f32x4* gp = reinterpret_cast<const f32x4*>(GI0) + ((par * RTC + rt) * 4 + s) * 192 + lane;
        f32x4 ah[3], an;                             // r, z (on top of the input-side sums) and the candidate's hidden side
        ah[2] = splat(bias_hn);
        if (fl) { ah[0] = gp[0]; ah[1] = gp[64]; an = NGR ? anx[rt] : gp[128]; }
        else { ah[0] = gB[0]; ah[1] = gB[1]; an = gB[2]; }
        if (GIO) {                                   // the input-side gate sums of step t for a later XS launch (the n plane of the NGR variants: nin())
          float* const go = a.gi_out + sv_off((long)t * NTILES + tile0 + rt, 3, 0, s, lane);
          *reinterpret_cast<f32x4*>(go) = ah[0];
          *reinterpret_cast<f32x4*>(go + 1024) = ah[1];
          if (!NGR) *reinterpret_cast<f32x4*>(go + 2048) = an;
        }
        if (XS) {                                    // the next tile: this step's, or tile 0 of the next step (stored step + 1)
          const bool same = rt + 1 < RTW;
          const int nts = same ? t + 1 : t + 2;
          if (nts < a.T) gissue(nts, same ? rt + 1 : 0);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const F3 hb = bfrag(hpp(par) + rt * 16 * HP, HP, rows * HP, c, lane);
#define OP(p_, q_) _Pragma("unroll") for (int g = 0; g < 3; ++g) ah[g] = mm(hb.p_, wh[2 * g + c].q_, ah[g]);
          X6_TERMS(OP)
#undef OP
        }
        const f32x4 ar = ah[0], az = ah[1], ahn = ah[2];
        ST_MARK(0);
        f32x4 vr, vz, vn, hn;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float r_, z_, n_, h_;
          gru_point_x6(ar[r], az[r], an[r], ahn[r], hreg[rt][r], r_, z_, n_, h_);
          vr[r] = r_; vz[r] = z_; vn[r] = n_; hn[r] = h_;
        }
        put4(hpp(par ^ 1), HP, rows * HP, rt * 16 + 4 * q, u, hn);
        ST_MARK(1);
        if (SAVE) {
          float* const sp = a.saved + sv_off((long)t * NTILES + tile0 + rt, 6, 0, s, lane);      // plane k at sp + 1024 k
          *reinterpret_cast<f32x4*>(sp) = hreg[rt];
          *reinterpret_cast<f32x4*>(sp + 2 * 1024) = vr;
          *reinterpret_cast<f32x4*>(sp + 3 * 1024) = vz;
          *reinterpret_cast<f32x4*>(sp + 4 * 1024) = vn;
          *reinterpret_cast<f32x4*>(sp + 5 * 1024) = ahn;
          if (t == a.T - 1) *reinterpret_cast<f32x4*>(sp + NTILES * (6 * 1024)) = hn;      // h after the last step: plane 0 of step T
        }
        hreg[rt] = hn;
        if (a.hs) {
#pragma unroll
          for (int r = 0; r < 4; ++r) a.hs[(unsigned)(rowidx[rt * 16 + 4 * q + r] + t * a.N) * (unsigned)H + (unsigned)u] = hn[r];
        }
        if (t == a.T - 1 && a.h_last) {
#pragma unroll
          for (int r = 0; r < 4; ++r) a.h_last[(unsigned)rowrho[rt * 16 + 4 * q + r] * (unsigned)H + (unsigned)u] = hn[r];
        }
      }
      if (!FC2I && t > 0 && s < RTW) fc2(par, t - 1, s);         // q(t-1) from h fed into step t
      if (NGR && t + 1 < a.T) nin(par ^ 1, t + 1);               // off the chain: the next step's candidate input side (x(t+1) is in Xp[par ^ 1])
      ST_MARK(2);
      if (ALLP) {                                  // input tile of step t+3 -> the buffer fc1(t+1) finished with in the previous step; loads of step t+4
        if (par) commit(0, pu_lds0); else commit(1, pu_lds1);
        issue(t + 4 < a.T ? t + 4 : Tm1);
      }
      WG_BARRIER();
      ST_MARK(3);
    }
    if (!FC2I && s < RTW) fc2(a.T & 1, a.T - 1, s);              // q of the last step
    ST_DUMP(4);
  }
}

}  // namespace

// shapes the split unroll covers: H = 64, observation width a multiple of 8, input width <= 224 (<= 16 actions up to 160, <= 32
// beyond), T >= 4, rows
// addressed with 32-bit offsets
extern "C" int marl_agent_unroll_x6_supported(int B, int T, int N, int O, int A, int last_action, int reuse_network) {
  const int I = O + (last_action ? A : 0) + (reuse_network ? N : 0);
  if (A < 1 || O < 8 || (O & 7) || I > 224 || T < 4 || B < 1) return 0;
  if (A > (I > 160 ? 32 : 16)) return 0;               // (two action tiles of fc2 only in the widest instantiation)
  if (16 * (O / 4) > 3 * 256) return 0;                // (a row tile's observations fit the prefetch registers of one team)
  if ((double)B * T * N * H * 4.0 >= 4294967296.0) return 0;
  return 1;
}

// 1 when a NON-SAVING split unroll of this batch (no activations kept, no gate sums read, no hidden states written) runs on the
// round-6 decomposition (agent_x6p.hip): the caller then keeps no input-side gate sums for a double-Q continuation to read - that
// pass recomputes its input side faster than the hand-over kernel reads it, and the saving pass stores a third less
extern "C" int marl_agent_unroll_x6_plain_r6(int B, int T, int N, int O, int A, int last_action, int reuse_network, int cu_budget) {
  return marl_agent_unroll_x6_supported(B, T, N, O, A, last_action, reuse_network) &&
         marl_agent_x6p_tiles(B, T, N, O, A, last_action, reuse_network, cu_budget) ? 1 : 0;
}

extern "C" int marl_agent_unroll_fwd_x6(const marl_agent_weights_t* w, const float* obs, long obs_bs, int obs_t0,
                                        const int* ufed, long u_bs, int u_t0, const int* ep_len, const int* ep_map,
                                        const float* h0, float* q, float* hs, float* h_last, float* saved, int B, int T, int N,
                                        int O, int A, int last_action, int reuse_network, int cu_budget, float* gi_out,
                                        const float* gi_in, void* stream) {
  if (B <= 0 || T <= 0) return 0;
  if (w->H != H || cu_budget < 0 || cu_budget > 256 || !marl_agent_unroll_x6_supported(B, T, N, O, A, last_action, reuse_network))
    return (int)hipErrorInvalidValue;
  if (saved && gi_in) return (int)hipErrorInvalidValue;      // a launch stores the input-side sums or reads them
  if (!saved && !gi_in && !hs) {                             // a plain unroll of a large batch: the round-6 decomposition (agent_x6p.hip)
    const int tpw = marl_agent_x6p_tiles(B, T, N, O, A, last_action, reuse_network, cu_budget);
    if (tpw) return marl_agent_x6p_launch(w, obs, obs_bs, obs_t0, ufed, u_bs, u_t0, ep_len, ep_map, h0, q, h_last, B, T, N, O, A, last_action,
                                          reuse_network, tpw, stream);
  }
  if ((reinterpret_cast<uintptr_t>(obs) & 15) || (h0 && (reinterpret_cast<uintptr_t>(h0) & 3)) || (saved && (reinterpret_cast<uintptr_t>(saved) & 15)) ||
      (gi_out && (reinterpret_cast<uintptr_t>(gi_out) & 15)) || (gi_in && (reinterpret_cast<uintptr_t>(gi_in) & 15)))
    return (int)hipErrorInvalidValue;
  if (cu_budget == 0) cu_budget = 256;
  X6Args a;
  a.W1 = w->fc1_w; a.b1 = w->fc1_b; a.Wih = w->w_ih; a.Whh = w->w_hh; a.bih = w->b_ih; a.bhh = w->b_hh; a.W2 = w->fc2_w; a.b2 = w->fc2_b;
  a.obs = obs; a.obs_bs = obs_bs; a.obs_t0 = obs_t0; a.ufed = ufed; a.u_bs = u_bs; a.u_t0 = u_t0; a.ep_len = ep_len; a.ep_map = ep_map;
  a.h0 = h0; a.q = q; a.hs = hs; a.h_last = h_last; a.saved = saved; a.gi_out = saved ? gi_out : nullptr; a.gi_in = gi_in;
  a.B = B; a.T = T; a.N = N; a.O = O; a.A = A;
  a.has_act = last_action ? 1 : 0; a.has_id = reuse_network ? 1 : 0;
  a.I = O + (last_action ? A : 0) + (reuse_network ? N : 0);
  a.KI = (a.I + 31) / 32 * 32;
  a.R = (long)B * N;
  const long tiles = (a.R + 15) / 16;
  // two row tiles per workgroup once there are more tiles than CUs this launch may occupy (more than two do not fit LDS: larger
  // batches run in rounds of workgroups, and when the last round is at most one tile per CU its workgroups hold one tile each)
  const int wide = a.KI > 160 ? 2 : a.KI > 96 ? 1 : 0;      // five / seven fc1 chunks: one row tile per workgroup (LDS), any number of rounds
  const int rt = tiles > cu_budget && !wide ? 2 : 1;
  if (rt * 16 * (O / 4) > 3 * 256) return (int)hipErrorInvalidValue;        // (the prefetch registers: three float4 per thread of one team)
  a.RT = rt;
  long n_wg = (tiles + rt - 1) / rt;
  a.n_full = (int)n_wg;
  if (rt == 2) {
    const long k = (tiles + 2 * cu_budget - 1) / (2 * cu_budget), rem = tiles - 2L * cu_budget * (k - 1);
    if (rem <= cu_budget) { a.n_full = (int)(cu_budget * (k - 1)); n_wg = a.n_full + rem; }
  }
  const int rows = rt * 16, IP = a.KI + 8;
  const size_t lds = (size_t)2 * 3 * rows * IP * 2 + (size_t)4 * 3 * rows * HP * 2 + (size_t)2 * rt * 4 * 3 * 1024 + (size_t)rows * (2 * 8 + 4 * 4) +
                     (((size_t)T * 4 + 15) & ~(size_t)15) + (wide == 2 ? (size_t)4 * 2 * 3 * 1024 : 0);      // (+ the fc1 fragments of the widest instantiation that live in LDS)
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  const void* fn;
#define X6_PICKF(SAVE_, XS_, GIO_) (wide == 2 ? (const void*)agent_fwd_x6_kernel<1, SAVE_, XS_, GIO_, 7, 2>                         \
                                    : wide == 1 ? (const void*)agent_fwd_x6_kernel<1, SAVE_, XS_, GIO_, 5, 1>                       \
                                    : rt == 2 ? (const void*)agent_fwd_x6_kernel<2, SAVE_, XS_, GIO_, 3, 1> : (const void*)agent_fwd_x6_kernel<1, SAVE_, XS_, GIO_, 3, 1>)
  if (saved && a.gi_out) fn = X6_PICKF(true, false, true);
  else if (saved) fn = X6_PICKF(true, false, false);
  else if (gi_in) fn = X6_PICKF(false, true, false);
  else fn = X6_PICKF(false, false, false);
#undef X6_PICKF
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  dim3 grid((unsigned)n_wg), block(XNT);
  void* kargs[] = {(void*)&a};
  e = hipLaunchKernel(fn, grid, block, kargs, lds, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  MARL_CHECK_LAUNCH();
  return 0;
}

ST_DEFINE_SETTER(marl_debug_stamps_agent_x6)
