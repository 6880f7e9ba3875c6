// GRU-agent unroll with every fp32 product as six bf16 MFMA products (x6.h) - opt-in args.gemm_mode = "bf16x6", forward only, no
// saved activations: the TARGET network's unroll of a Q-learning update (reference controller/share_params.py:147-168,
// network/q_network.py:16-21).  The default path is agent.hip on v_mfma_f32_16x16x4_f32.
//
// Why another decomposition: as bf16 triples the weight fragments of a hidden-unit slice no longer fit one wave
// beside its working set (W_ih + W_hh + fc1 + fc2 slices = 204 registers), and 190 KB of pre-split weights do not fit LDS.  So the
// two teams of the workgroup hold DIFFERENT weights instead of the same ones:
//   team I (waves 4-7, slice s): fc1 and W_ih fragments (108 registers) - everything that depends only on a step's INPUT:
//          x(t+2) = relu(fc1(in(t+2))) and the input-side gate sums gi(t+1) = bias + x(t+1) W_ih, one / two steps ahead of the chain;
//   team R (waves 0-3, slice s): W_hh and fc2 fragments (96 registers) - the recurrent part of step t: accumulators start from the
//          handed gi(t), += h W_hh, gate math, h' (kept in fp32 registers for the blend of the next step), q(t-1) = fc2(h).
// Transposed formulation (mlp3_x6.hip): out^T[unit][row] = W[unit][k] in^T[k][row] - weights are the MFMA A operand, activations the
// B operand, read as 16 bytes per lane (row m, 8 consecutive k) from bf16 PLANES in LDS: every activation element is split once,
// where it is produced, and shared by the four slice waves that consume it (5.5 vector instructions per element, once).
// One workgroup barrier per step; every LDS buffer is double-buffered by step parity:
//   In[b]  input tile of step t+2 (planes)     Xp[b]  x(t+1) (planes)     Hp[b]  h fed into step t (planes)     GI[b]  gi(t) (fp32, accumulator layout)
// ~70 KB of LDS per 16-row tile: one or two row tiles per workgroup, as many workgroups as that takes (they run in rounds).
#include "x6.h"
#include <cstdlib>
#include "../../include/marl_hip.h"

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
  int B, T, N, O, A, I, KI;        // KI: input width rounded up to 32
  int has_act, has_id, RT;
  long R;
};

#define WG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

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
// accumulator tile (units col + r of row m, r = 0..3) -> planes: four consecutive columns of row m, 8 bytes per plane
__device__ __forceinline__ void put4(short* pl, int pitch, int ps, int col, int m, const f32x4& v) {
  const F3h f = split4(v);
  short* p = pl + m * pitch + col;
  *reinterpret_cast<i32x2*>(p) = f.h;
  *reinterpret_cast<i32x2*>(p + ps) = f.m;
  *reinterpret_cast<i32x2*>(p + 2 * ps) = f.l;
}
__device__ __forceinline__ f32x4 relu4x(const f32x4& v) { return (f32x4){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)}; }

template <int RTC>
__global__ __launch_bounds__(XNT, 2) void agent_fwd_x6_kernel(X6Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave >> 2, s = wave & 3;      // team 0 = R (recurrent), team 1 = I (input side); hidden-unit slice s
  const int q = lane >> 4, m = lane & 15;
  const int rows = RTC * 16;
  const int IP = a.KI + 8;                        // pitch of the input planes
  // LDS carve (bytes): per buffer parity b
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
  // plane p of buffer b of a [2][3][rows][pitch] array: base + (b * 3 + p) * rows * pitch
  auto inp = [&](int b) { return In0 + b * 3 * rows * IP; };
  auto xpp = [&](int b) { return Xp0 + b * 3 * rows * HP; };
  auto hpp = [&](int b) { return Hp0 + b * 3 * rows * HP; };

  const long NTILES = (a.R + 15) >> 4;
  const long row0 = (long)blockIdx.x * rows;
  const int RTW = (int)((NTILES - (long)blockIdx.x * RTC) < RTC ? (NTILES - (long)blockIdx.x * RTC) : RTC);
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
  __syncthreads();

  // ---- weights: registers, as pre-split A fragments
  F3 wA[9];                                        // team I: fc1 (3 or 4 chunks... <= 3 here) + W_ih (3 gates x 2 chunks); team R: W_hh (6) + fc2 (2)
  const int KC1 = a.KI >> 5;                       // k-chunks of fc1 (<= 3: I <= 96)
  f32x4 bias_a, bias_r, bias_z, bias_n;            // team I: b1 | b_ir + b_hr, b_iz + b_hz, b_in ;  team R: b2 | b_hn in bias_n
  {
    const int u0 = 16 * s + 4 * q;
    if (team == 1) {
#pragma unroll
      for (int c = 0; c < 3; ++c) wA[c] = c < KC1 ? wfrag(a.W1, a.I, 16 * s, H, a.I, c, lane) : F3{};
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int c = 0; c < 2; ++c) wA[3 + 2 * g + c] = wfrag(a.Wih, H, g * H + 16 * s, 3 * H, H, c, lane);
      bias_a = *reinterpret_cast<const f32x4*>(a.b1 + u0);
      bias_r = *reinterpret_cast<const f32x4*>(a.bih + u0) + *reinterpret_cast<const f32x4*>(a.bhh + u0);
      bias_z = *reinterpret_cast<const f32x4*>(a.bih + H + u0) + *reinterpret_cast<const f32x4*>(a.bhh + H + u0);
      bias_n = *reinterpret_cast<const f32x4*>(a.bih + 2 * H + u0);
    } else {
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int c = 0; c < 2; ++c) wA[2 * g + c] = wfrag(a.Whh, H, g * H + 16 * s, 3 * H, H, c, lane);
#pragma unroll
      for (int c = 0; c < 2; ++c) wA[6 + c] = wfrag(a.W2, H, 0, a.A, H, c, lane);
      wA[8] = F3{};
#pragma unroll
      for (int r = 0; r < 4; ++r) bias_a[r] = 4 * q + r < a.A ? a.b2[4 * q + r] : 0.f;
      bias_n = *reinterpret_cast<const f32x4*>(a.bhh + 2 * H + u0);
      bias_r = bias_z = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }

  // ---- observation prefetch: thread -> (row, 4-column group) of the tile, the same every step
  const int O = a.O, O4 = O >> 2, n4 = rows * O4;
  const float invO4 = 1.0f / (float)(O4 > 0 ? O4 : 1);
  f32x4 pf[NLD];
  long goff[NLD]; int loff[NLD], plen[NLD];
  int pt = 0, pu = -1, pu_lds0 = -1, pu_lds1 = -1;      // one-hot column currently set in each input buffer
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    int e = tid + XNT * i;
    if (e > n4 - 1) e = n4 - 1;
    const int r = (int)(((float)e + 0.5f) * invO4);
    const int k4 = e - r * O4;
    loff[i] = r * IP + 4 * k4;
    goff[i] = rowobs[r] + 4 * k4;
    plen[i] = rowlen[r];
  }
  const long urow = tid < rows ? rowu[tid] : 0;
  auto issue = [&](int t) {
    const long toff = (long)(t + a.obs_t0) * a.N * O;
#pragma unroll
    for (int i = 0; i < NLD; ++i) pf[i] = *reinterpret_cast<const f32x4*>(a.obs + goff[i] + toff);
    pt = t;
    int u = -1;
    if (tid < rows && a.ufed && t + a.u_t0 >= 0) u = a.ufed[urow + (long)(t + a.u_t0) * a.N];
    pu = u;
  };
  auto commit = [&](int b, int& pu_lds) {          // prefetch registers -> input planes of buffer b (split once, here)
    short* P = inp(b);
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const f32x4 v = pt < plen[i] ? pf[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
      const F3h f = split4(v);
      *reinterpret_cast<i32x2*>(P + loff[i]) = f.h;
      *reinterpret_cast<i32x2*>(P + rows * IP + loff[i]) = f.m;
      *reinterpret_cast<i32x2*>(P + 2 * rows * IP + loff[i]) = f.l;
    }
    if (a.has_act && tid < rows) {                 // one-hot(last action): bf16 1.0 in the hi plane, flipped in place
      const int pn = (pu >= 0 && pu < a.A) ? pu : -1;
      if (pn != pu_lds) {
        if (pu_lds >= 0) P[tid * IP + O + pu_lds] = 0;
        if (pn >= 0) P[tid * IP + O + pn] = (short)0x3F80;
        pu_lds = pn;
      }
    }
  };
  // constant columns of both input buffers: empty one-hot, agent id, zero pad - all three planes
  for (int e = tid; e < 2 * 3 * rows * (a.KI - O); e += XNT) {
    const int w = a.KI - O, rr = e / w, k = O + e % w;        // rr = (b * 3 + plane) * rows + row
    const int plane = (rr / rows) % 3, r = rr % rows;
    short v = 0;
    if (plane == 0 && a.has_id && k >= a.I - a.N && k < a.I && rown[r] == k - (a.I - a.N)) v = (short)0x3F80;
    In0[rr * IP + k] = v;
  }
  // initial hidden state: fp32 registers of team R (units 16s + 4q + r of row m) and planes Hp[0]
  f32x4 hreg[RTC];
#pragma unroll
  for (int rt = 0; rt < RTC; ++rt) {
    hreg[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.h0) hreg[rt] = *reinterpret_cast<const f32x4*>(a.h0 + (long)rowrho[rt * 16 + m] * H + 16 * s + 4 * q);
    if (team == 0) put4(hpp(0) + rt * 16 * HP, HP, rows * HP, 16 * s + 4 * q, m, hreg[rt]);
  }
  __syncthreads();
  const int Tm1 = a.T - 1;
  issue(0); commit(0, pu_lds0);
  issue(Tm1 < 1 ? Tm1 : 1); commit(1, pu_lds1);
  issue(Tm1 < 2 ? Tm1 : 2);
  WG_BARRIER();

  // x(ts) = relu(fc1(in)) of every row tile: this wave's 16 units -> planes Xp[bx]
  auto fc1 = [&](int bin, int bx) __attribute__((always_inline)) {
#pragma unroll
    for (int rt = 0; rt < RTC; ++rt) {
      if (rt >= RTW) break;
      f32x4 acc = bias_a;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        if (c < KC1) mm6(wA[c], bfrag(inp(bin) + rt * 16 * IP, IP, rows * IP, c, lane), acc);
      put4(xpp(bx) + rt * 16 * HP, HP, rows * HP, 16 * s + 4 * q, m, relu4x(acc));
    }
  };
  // gi(ts) = bias + x W_ih of every row tile -> GI[bg] (accumulator layout, for the R wave of this slice)
  auto gih = [&](int bx, int bg) __attribute__((always_inline)) {
#pragma unroll
    for (int rt = 0; rt < RTC; ++rt) {
      if (rt >= RTW) break;
      f32x4 ar = bias_r, az = bias_z, an = bias_n;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const F3 xb = bfrag(xpp(bx) + rt * 16 * HP, HP, rows * HP, c, lane);
        mm6(wA[3 + c], xb, ar); mm6(wA[5 + c], xb, az); mm6(wA[7 + c], xb, an);
      }
      f32x4* gp = reinterpret_cast<f32x4*>(GI0) + ((bg * RTC + rt) * 4 + s) * 192 + lane;
      gp[0] = ar; gp[64] = az; gp[128] = an;
    }
  };
  // q(ts) = fc2(h) of row tile rt from planes Hp[bh]
  auto fc2 = [&](int bh, int ts, int rt) __attribute__((always_inline)) {
    f32x4 acc = bias_a;
#pragma unroll
    for (int c = 0; c < 2; ++c) mm6(wA[6 + c], bfrag(hpp(bh) + rt * 16 * HP, HP, rows * HP, c, lane), acc);
    const int ri = rowidx[rt * 16 + m];
    float* qp = a.q + ((long)ri + (long)ts * a.N) * a.A + 4 * q;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * q + r < a.A) qp[r] = acc[r];
  };

  // prologue: x(0), x(1) by team I; then input(2) -> In[0], gi(0)
  if (team == 1) { fc1(0, 0); if (a.T > 1) fc1(1, 1); }
  WG_BARRIER();
  commit(0, pu_lds0);
  issue(Tm1 < 3 ? Tm1 : 3);
  if (team == 1) gih(0, 0);
  WG_BARRIER();

  for (int t = 0; t < a.T; ++t) {
    const int par = t & 1;
    if (team == 1) {
      if (t + 1 < a.T) gih(par ^ 1, par ^ 1);                 // gi(t+1) from x(t+1)
      if (t + 2 < a.T) fc1(par, par);                         // x(t+2) from input(t+2) -> the buffer x(t) has left
    } else {
#pragma unroll
      for (int rt = 0; rt < RTC; ++rt) {
        if (rt >= RTW) break;
        const f32x4* gp = reinterpret_cast<const f32x4*>(GI0) + ((par * RTC + rt) * 4 + s) * 192 + lane;
        f32x4 ar = gp[0], az = gp[64], an = gp[128], ahn = bias_n;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const F3 hb = bfrag(hpp(par) + rt * 16 * HP, HP, rows * HP, c, lane);
          mm6(wA[c], hb, ar); mm6(wA[2 + c], hb, az); mm6(wA[4 + c], hb, ahn);
        }
        f32x4 hn;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float r_, z_, n_, h_;
          gru_point_plain(ar[r], az[r], an[r], ahn[r], hreg[rt][r], r_, z_, n_, h_);
          hn[r] = h_;
        }
        hreg[rt] = hn;
        put4(hpp(par ^ 1) + rt * 16 * HP, HP, rows * HP, 16 * s + 4 * q, m, hn);
        const int ri = rowidx[rt * 16 + m];
        if (a.hs) *reinterpret_cast<f32x4*>(a.hs + ((long)ri + (long)t * a.N) * H + 16 * s + 4 * q) = hn;
        if (t == a.T - 1 && a.h_last) *reinterpret_cast<f32x4*>(a.h_last + (long)rowrho[rt * 16 + m] * H + 16 * s + 4 * q) = hn;
      }
      if (t > 0)
        for (int rt = s; rt < RTW; rt += 4) fc2(par, t - 1, rt);     // q(t-1) from h fed into step t
    }
    // input tile of step t+3 -> the buffer fc1(t+1) finished with in the previous step; start the loads of step t+4
    if (par) commit(0, pu_lds0); else commit(1, pu_lds1);
    issue(t + 4 < a.T ? t + 4 : Tm1);
    WG_BARRIER();
  }
  if (team == 0)
    for (int rt = s; rt < RTW; rt += 4) fc2(a.T & 1, a.T - 1, rt);   // q of the last step
}

}  // namespace

// shapes the split unroll covers: H = 64, <= 16 actions, observation width a multiple of 8, input width <= 96, T >= 4, rows
// addressed with 32-bit offsets
extern "C" int marl_agent_unroll_x6_supported(int B, int T, int N, int O, int A, int last_action, int reuse_network) {
  const int I = O + (last_action ? A : 0) + (reuse_network ? N : 0);
  if (A < 1 || A > 16 || O < 8 || (O & 7) || I > 96 || T < 4 || B < 1) return 0;
  if ((double)B * T * N * H * 4.0 >= 4294967296.0) return 0;
  return 1;
}

extern "C" int marl_agent_unroll_fwd_x6(const marl_agent_weights_t* w, const float* obs, long obs_bs, int obs_t0,
                                        const int* ufed, long u_bs, int u_t0, const int* ep_len, const int* ep_map,
                                        const float* h0, float* q, float* hs, float* h_last, int B, int T, int N, int O, int A,
                                        int last_action, int reuse_network, void* stream) {
  if (B <= 0 || T <= 0) return 0;
  if (w->H != H || !marl_agent_unroll_x6_supported(B, T, N, O, A, last_action, reuse_network)) return (int)hipErrorInvalidValue;
  if ((reinterpret_cast<uintptr_t>(obs) & 15) || (reinterpret_cast<uintptr_t>(w->b_ih) & 15) || (reinterpret_cast<uintptr_t>(w->b_hh) & 15) ||
      (reinterpret_cast<uintptr_t>(w->fc1_b) & 15) || (h0 && (reinterpret_cast<uintptr_t>(h0) & 15)) || (hs && (reinterpret_cast<uintptr_t>(hs) & 15)) ||
      (h_last && (reinterpret_cast<uintptr_t>(h_last) & 15)))
    return (int)hipErrorInvalidValue;
  X6Args a;
  a.W1 = w->fc1_w; a.b1 = w->fc1_b; a.Wih = w->w_ih; a.Whh = w->w_hh; a.bih = w->b_ih; a.bhh = w->b_hh; a.W2 = w->fc2_w; a.b2 = w->fc2_b;
  a.obs = obs; a.obs_bs = obs_bs; a.obs_t0 = obs_t0; a.ufed = ufed; a.u_bs = u_bs; a.u_t0 = u_t0; a.ep_len = ep_len; a.ep_map = ep_map;
  a.h0 = h0; a.q = q; a.hs = hs; a.h_last = h_last;
  a.B = B; a.T = T; a.N = N; a.O = O; a.A = A;
  a.has_act = last_action ? 1 : 0; a.has_id = reuse_network ? 1 : 0;
  a.I = O + (last_action ? A : 0) + (reuse_network ? N : 0);
  a.KI = (a.I + 31) / 32 * 32;
  a.R = (long)B * N;
  const long tiles = (a.R + 15) / 16;
  // two row tiles per workgroup once there are more tiles than CUs (the prefetch registers cover NLD * 512 float4 = 2 tiles of O <= 256)
  const int rt = tiles > 256 && 2 * 16 * (O / 4) <= NLD * XNT ? 2 : 1;
  if (rt * 16 * (O / 4) > NLD * XNT) return (int)hipErrorInvalidValue;
  a.RT = rt;
  const int rows = rt * 16, IP = a.KI + 8;
  const size_t lds = (size_t)2 * 3 * rows * IP * 2 + (size_t)4 * 3 * rows * HP * 2 + (size_t)2 * rt * 4 * 3 * 1024 + (size_t)rows * (2 * 8 + 4 * 4);
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  const void* fn = rt == 2 ? (const void*)agent_fwd_x6_kernel<2> : (const void*)agent_fwd_x6_kernel<1>;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  dim3 grid((unsigned)((tiles + rt - 1) / rt)), block(XNT);
  void* kargs[] = {(void*)&a};
  e = hipLaunchKernel(fn, grid, block, kargs, lds, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  MARL_CHECK_LAUNCH();
  return 0;
}
