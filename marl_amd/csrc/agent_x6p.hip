// The PLAIN T-step unroll of the GRU agent (no activations kept: the target network's pass and the double-Q continuation of the eval
// pass, reference q_learner.py:104-110 / controller/share_params.py:148-168) on bf16x6 split products, in the decomposition of the
// round-6 whole-rollout kernel (rollout_x6.hip): the recurrent team holds BOTH gate matrices and runs x W_ih and h W_hh down one
// accumulator chain per gate, so no input-side gate sums travel through LDS - 30 KB of LDS per 16-row tile instead of agent_x6.hip's
// 70 KB, up to FIVE row tiles per workgroup: a 4096-env batch (1280 tiles) is ONE round of 256 workgroups where agent_x6.hip runs
// three rounds of one- and two-tile workgroups.  The actions an unroll feeds are known (the record's), so nothing but the recurrence
// is a dependent chain: TWO barriers per step.
//   team R (waves 0-3, hidden-unit slice s): W_ih and W_hh fragments, the hidden state in fp32 registers; per step
//          gates(t) = bias + x(t) W_ih + h(t-1) W_hh, gate math, h(t) -> planes (software-pipelined over the row tiles, transposed
//          products: a lane holds four consecutive columns of one row)
//   team I (waves 4-7): fc1, fc2 and the input stream:
//      phase 1 (beside the recurrence)  pre(t+1) = bias + W1[:, obs | id] in(t+1);  q(t-1) = fc2(h(t-1)) -> HBM (a row tile per wave)
//      phase 2                          x(t+1) = relu(pre + W1[:, O + u]) -> planes;  observations(t+2) -> input planes (split once);
//                                       the loads of step t+3's observations and of the action fed at step t+2
// fc1's one-hot(last action) block is one fp32 column of W1 per row, added when x is formed (a table in LDS) - the arithmetic of the
// rollout kernels; agent_x6.hip carries the one-hot inside the product, so the two unrolls agree to rounding, not bit for bit.
// Launched by marl_agent_unroll_fwd_x6 (agent_x6.hip) for non-saving launches over the whole chip of more than 512 row tiles of
// 2s3z-sized agents; experiments: unroll_r6 = 0 keeps agent_x6.hip everywhere.
#include "x6.h"
#include "../../include/marl_hip.h"

namespace {

constexpr int H = 64;
constexpr int PNT = 512;
constexpr int HP = 72;            // pitch (bf16) of the 64-wide planes
constexpr int NPF = 7;            // float4 prefetch registers per thread of team I (a step's observations: rows x O / 4 <= 256 NPF)

#define WG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define X6_TERMS(OP) OP(m, m) OP(h, l) OP(l, h) OP(h, m) OP(m, h) OP(h, h)

// fragment of W (row-major, ldw floats per row): lane (g, j): W[row0 + j][32 c + 8g .. + 7]  (rows >= rows_valid and columns >= K: 0)
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
// fragment from a plane tile (hi plane at pl, the others ps elements further): lane (g, m) reads row m, columns 32 c + 8g .. + 7
__device__ __forceinline__ F3 bfrag(const short* pl, int pitch, int ps, int c, int lane) {
  const int m = lane & 15, g = lane >> 4;
  const short* p = pl + m * pitch + 32 * c + 8 * g;
  F3 f;
  f.h = *reinterpret_cast<const i32x4*>(p);
  f.m = *reinterpret_cast<const i32x4*>(p + ps);
  f.l = *reinterpret_cast<const i32x4*>(p + 2 * ps);
  return f;
}
// accumulator tile of a transposed product (lane (q, m): columns col0 .. col0 + 3 of row `row`) -> planes: one 8-byte write per plane
__device__ __forceinline__ void put4t(short* pl, int pitch, int ps, int row, int col0, const f32x4& v) {
  const F3h f = split4(v);
  short* p = pl + row * pitch + col0;
  *reinterpret_cast<i32x2*>(p) = f.h;
  *reinterpret_cast<i32x2*>(p + ps) = f.m;
  *reinterpret_cast<i32x2*>(p + 2 * ps) = f.l;
}
__device__ __forceinline__ f32x4 splat(float v) { return (f32x4){v, v, v, v}; }
// the gate math of agent_x6.hip (every fused / unfused operation spelled out)
__device__ __forceinline__ float gru_h_x6(float ar, float az, float ain, float ahn, float hp) {
  const float r = __builtin_amdgcn_rcpf(__fadd_rn(1.0f, __builtin_amdgcn_exp2f(__fmul_rn(ar, -1.4426950408889634f))));
  const float z = __builtin_amdgcn_rcpf(__fadd_rn(1.0f, __builtin_amdgcn_exp2f(__fmul_rn(az, -1.4426950408889634f))));
  const float e = __builtin_amdgcn_exp2f(__fmul_rn(__fmaf_rn(r, ahn, ain), 2.8853900817779268f));
  const float n = __fmaf_rn(-2.0f, __builtin_amdgcn_rcpf(__fadd_rn(e, 1.0f)), 1.0f);
  return __fmaf_rn(z, hp, __fmul_rn(__fsub_rn(1.0f, z), n));
}

struct PX6Args {
  const float *W1, *b1, *Wih, *Whh, *bih, *bhh, *W2, *b2;
  const float* obs; long obs_bs; int obs_t0;
  const int* ufed; long u_bs; int u_t0;
  const int* ep_len; const int* ep_map;
  const float* h0;
  float *q, *h_last;
  int B, T, N, O, A, I, KI;
  int has_act, has_id;
  long R;
};

template <int RTC>
__global__ __launch_bounds__(PNT, 2) void agent_fwd_x6p_kernel(PX6Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave >> 2, s = wave & 3;
  const int q = lane >> 4, m = lane & 15, u4 = 16 * s + 4 * q;
  constexpr int rows = 16 * RTC;
  const int T = a.T, N = a.N, O = a.O, A = a.A;
  const int IP = a.KI + 8;
  const int IN_E = 3 * rows * IP, XP_E = 3 * rows * HP;
  short* In0 = reinterpret_cast<short*>(smem);                             // [3][rows][IP]   input planes of the step fc1 reads next
  short* Xp0 = In0 + IN_E;                                                 // [3][rows][HP]   x(t+1)
  short* Hp0 = Xp0 + XP_E;                                                 // [2][3][rows][HP] h by step parity
  float* W1a = reinterpret_cast<float*>(Hp0 + 2 * XP_E);                   // [A + 1][64]: fc1 columns of the one-hot(last action) block; row A = zeros
  unsigned* rowobs = reinterpret_cast<unsigned*>(W1a + (A + 1) * H);       // [rows] BYTE offset of the row's observation at time index 0 (32 bits: a uniform
  unsigned* rowu = rowobs + rows;                                          // [rows] ... of its fed action      base + a 32-bit lane offset is one instruction's addressing)
  int* rowidx = reinterpret_cast<int*>(rowu + rows);                       // [rows] b T N + n: the row's place in q (element offset / A at step 0)
  int* rowlen = rowidx + rows;                                             // [rows] episode length (steps from it on feed zeros)
  // fc2 fragments, the same for every wave, of the A action lanes only (the other lanes of a fragment are zero): [2 k chunks][3 planes][4 g][A]
  i32x4* W2f = reinterpret_cast<i32x4*>(rowlen + rows);
  auto hpp = [&](int b) { return Hp0 + b * XP_E; };

  const long row0 = (long)blockIdx.x * rows;
  for (int r = tid; r < rows; r += PNT) {
    long rho = row0 + r;
    if (rho > a.R - 1) rho = a.R - 1;             // clamped duplicates: same loads, same values, same stores
    const long b = rho / N;
    const int n = (int)(rho % N);
    rowidx[r] = (int)(b * T * N + n);
    rowobs[r] = (unsigned)((((a.ep_map ? (long)a.ep_map[b] : b) * a.obs_bs + n) * O) * 4);
    rowu[r] = (unsigned)((b * a.u_bs + n) * 4);
    rowlen[r] = a.ep_len ? a.ep_len[b] : 0x7fffffff;
  }
  // planes: zero everywhere (pad columns, the one-hot block: its contribution comes from the W1a table)
  for (int e = tid; e < (IN_E + 3 * XP_E) / 2; e += PNT) reinterpret_cast<int*>(In0)[e] = 0;
  for (int e = tid; e < (A + 1) * H; e += PNT) {
    const int aa = e / H, j = e % H;
    W1a[e] = (a.has_act && aa < A) ? a.W1[(long)j * a.I + O + aa] : 0.f;
  }
  __syncthreads();
  if (a.has_id)
    for (int r = tid; r < rows; r += PNT) {
      long rho = row0 + r; if (rho > a.R - 1) rho = a.R - 1;
      In0[r * IP + (a.I - N) + (int)(rho % N)] = (short)0x3F80;        // agent id: bf16 1.0 in the hi plane
    }
  const int KC1 = a.KI >> 5;

  if (team == 0) {
    // =============================== team R: the recurrence ===============================
    F3 wi[6], wh[6];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        wi[2 * g + c] = wfrag(a.Wih, H, g * H + 16 * s, 3 * H, H, c, lane);
        wh[2 * g + c] = wfrag(a.Whh, H, g * H + 16 * s, 3 * H, H, c, lane);
      }
    f32x4 bias_r, bias_z, bias_n, bias_hn;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      bias_r[r] = a.bih[u4 + r] + a.bhh[u4 + r]; bias_z[r] = a.bih[H + u4 + r] + a.bhh[H + u4 + r];
      bias_n[r] = a.bih[2 * H + u4 + r]; bias_hn[r] = a.bhh[2 * H + u4 + r];
    }
    f32x4 hreg[RTC];
#pragma unroll
    for (int rt = 0; rt < RTC; ++rt) {              // h(-1): h0 or zero - registers and planes (lane (q, m): units u4 .. u4 + 3 of row m)
      long rho = row0 + rt * 16 + m; if (rho > a.R - 1) rho = a.R - 1;
      hreg[rt] = a.h0 ? *reinterpret_cast<const f32x4*>(a.h0 + rho * H + u4) : splat(0.f);
      put4t(hpp(0), HP, rows * HP, rt * 16 + m, u4, hreg[rt]);
    }
    WG_BARRIER();                                  // P0: step 0's input planes (team I: fc1, x(0)), h(-1)
    WG_BARRIER();                                  // P1: x(0)
    WG_BARRIER();                                  // P2: step 1's input planes
    ST_DECL(4);
    for (int t = 0; t < T; ++t) {
      const int par = t & 1;
      f32x4 G[2][4];
      F3 fa = bfrag(Xp0, HP, rows * HP, 0, lane);
#pragma unroll
      for (int rt = 0; rt <= RTC; ++rt) {
        if (rt < RTC) {
          f32x4* g = G[rt & 1];
          g[0] = bias_r; g[1] = bias_z; g[2] = bias_n; g[3] = bias_hn;
          const F3 fb = bfrag(Xp0 + rt * 16 * HP, HP, rows * HP, 1, lane);
#define OP(p_, q_) _Pragma("unroll") for (int k = 0; k < 3; ++k) g[k] = mm(wi[2 * k].q_, fa.p_, g[k]);
          X6_TERMS(OP)
#undef OP
          const F3 fc = bfrag(hpp(par) + rt * 16 * HP, HP, rows * HP, 0, lane);
#define OP(p_, q_) _Pragma("unroll") for (int k = 0; k < 3; ++k) g[k] = mm(wi[2 * k + 1].q_, fb.p_, g[k]);
          X6_TERMS(OP)
#undef OP
          const F3 fd = bfrag(hpp(par) + rt * 16 * HP, HP, rows * HP, 1, lane);
#define OP(p_, q_) g[0] = mm(wh[0].q_, fc.p_, g[0]); g[1] = mm(wh[2].q_, fc.p_, g[1]); g[3] = mm(wh[4].q_, fc.p_, g[3]);
          X6_TERMS(OP)
#undef OP
          if (rt + 1 < RTC) fa = bfrag(Xp0 + (rt + 1) * 16 * HP, HP, rows * HP, 0, lane);
#define OP(p_, q_) g[0] = mm(wh[1].q_, fd.p_, g[0]); g[1] = mm(wh[3].q_, fd.p_, g[1]); g[3] = mm(wh[5].q_, fd.p_, g[3]);
          X6_TERMS(OP)
#undef OP
        }
        if (rt > 0) {
          const f32x4* g = G[(rt - 1) & 1];
          f32x4 hn;
#pragma unroll
          for (int r = 0; r < 4; ++r) hn[r] = gru_h_x6(g[0][r], g[1][r], g[2][r], g[3][r], hreg[rt - 1][r]);
          put4t(hpp(par ^ 1), HP, rows * HP, (rt - 1) * 16 + m, u4, hn);
          hreg[rt - 1] = hn;
        }
        if (rt > 0 && rt < RTC) {      // the products of tile rt with the gate math of tile rt-1 in their gaps
#pragma unroll
          for (int i = 0; i < 72; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
          }
        }
      }
      ST_MARK(0);
      WG_BARRIER();                                // X: h(t) planes | pre(t+1) taken, q(t-1) read
      ST_MARK(1);
      WG_BARRIER();                                // Y: x(t+1) planes, step t+2's input planes
      ST_MARK(2);
    }
    ST_DUMP(4);
    if (a.h_last) {
#pragma unroll
      for (int rt = 0; rt < RTC; ++rt) {
        const long rho = row0 + rt * 16 + m;
        if (rho < a.R) *reinterpret_cast<f32x4*>(a.h_last + rho * H + u4) = hreg[rt];
      }
    }
  } else {
    // =============================== team I: the input stream, fc1, x, fc2 ===============================
    const int ti = tid - PNT / 2;
    F3 w1[3], w2[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) w1[c] = c < KC1 ? wfrag(a.W1, a.I, 16 * s, H, a.I, c, lane) : F3{};
    constexpr bool W2R = RTC < 5;                  // (five row tiles: the fc2 fragments are read from LDS at every use - this team's registers hold the prefetch)
    if constexpr (W2R) {
#pragma unroll
      for (int c = 0; c < 2; ++c) w2[c] = wfrag(a.W2, H, 0, A, H, c, lane);
    } else if (wave < 6) {                         // waves 4, 5: chunk 0, 1 -> LDS (visible behind P0)
      const F3 f = wfrag(a.W2, H, 0, A, H, wave - 4, lane);
      if (m < A) {
        i32x4* d = W2f + ((wave - 4) * 3 * 4 + q) * A + m;
        d[0] = f.h; d[4 * A] = f.m; d[8 * A] = f.l;
      }
    }
    const f32x4 bias_1 = {a.b1[u4], a.b1[u4 + 1], a.b1[u4 + 2], a.b1[u4 + 3]};
    const float bias_2 = m < A ? a.b2[m] : 0.f;
    f32x4 pre[RTC];
    // ---- the observation stream: thread -> (row, 4-column group) items of the workgroup's rows, the same every step
    const int O4 = O >> 2, n4 = rows * O4;
    const float invO4 = 1.0f / (float)(O4 > 0 ? O4 : 1);
    // (five row tiles: this team's registers do not hold the items' metadata beside the prefetch - it is recomputed from the row tables)
    constexpr bool META = RTC < 5;
    f32x4 pf[NPF];
    unsigned goff[META ? NPF : 1]; int pmeta[META ? NPF : 1];      // BYTE offset of the item at time index 0; its place in the input planes
                                                                   // (low 16 bits) | its row's episode length + 1 (0: no item: nothing loaded / written)
    // Branch-free: a thread whose slot j lies past the workgroup's items takes the LAST item again (same load, same values written to the
    // same place), time indices past the unroll are clamped - every load of this team is unconditional, so the compiler counts them
    // (a load behind a lane branch made it wait for everything in flight at the next join, fresh loads included).
    auto item = [&](int j, unsigned& go, int& pm) __attribute__((always_inline)) {
      int e = ti + (PNT / 2) * j;
      e = e < n4 ? e : n4 - 1;
      const int r = (int)(((float)e + 0.5f) * invO4), k4 = e - r * O4;
      const int L = rowlen[r] < 32766 ? rowlen[r] : 32766;
      go = rowobs[r] + 16u * (unsigned)k4; pm = (r * IP + 4 * k4) | (L << 16);
    };
    if constexpr (META) {
#pragma unroll
      for (int j = 0; j < NPF; ++j) item(j, goff[j], pmeta[j]);
    }
    const int Tm1 = T - 1;
    auto issue = [&](int t) __attribute__((always_inline)) {
      const char* ob = reinterpret_cast<const char*>(a.obs + (long)((t < Tm1 ? t : Tm1) + a.obs_t0) * N * O);
#pragma unroll
      for (int j = 0; j < NPF; ++j) {
        unsigned go; int pm;
        if constexpr (META) { go = goff[j]; pm = pmeta[j]; } else item(j, go, pm);
        pf[j] = *reinterpret_cast<const f32x4*>(ob + (size_t)go);
      }
    };
    auto commit = [&](int t) __attribute__((always_inline)) {      // prefetch registers -> input planes (split once, here); steps past a row's episode feed zeros
#pragma unroll
      for (int j = 0; j < NPF; ++j) {
        unsigned go; int pm;
        if constexpr (META) pm = pmeta[j]; else item(j, go, pm);
        const bool live = t < (pm >> 16);
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = live ? pf[j][i] : 0.f;
        const F3h f = split4(v);
        short* p = In0 + (pm & 0xffff);
        *reinterpret_cast<i32x2*>(p) = f.h;
        *reinterpret_cast<i32x2*>(p + rows * IP) = f.m;
        *reinterpret_cast<i32x2*>(p + 2 * rows * IP) = f.l;
      }
    };
    // the action fed at step t to this lane's row of each tile (transposed layout: lane (q, m) = row m); the RAW loaded value is kept
    // (turned into a table row where x is formed: a load is never consumed in the phase that issues it)
    int uw[RTC];
    const int* const ufd = a.ufed ? a.ufed : reinterpret_cast<const int*>(a.obs);      // (no fed actions: any readable address - the value is not used)
    auto ufetch = [&](int t) __attribute__((always_inline)) {
      int tu = (t < Tm1 ? t : Tm1) + a.u_t0;
      tu = tu > 0 ? tu : 0;
      const char* ub = reinterpret_cast<const char*>(ufd + (long)tu * N);
#pragma unroll
      for (int rt = 0; rt < RTC; ++rt) uw[rt] = *reinterpret_cast<const int*>(ub + (size_t)(a.ufed ? rowu[rt * 16 + m] : 0u));
    };
    // (the action is "none" - the table's zero row - without fed actions and before the first fed step)
    auto urow_of = [&](int t, int v) { return (a.ufed && t + a.u_t0 >= 0 && v >= 0 && v < A) ? v : A; };
    // pre = bias + W1[:, obs | id] in  of every row tile from the input planes (three accumulator chains, chunk c on chain c)
    auto fc1 = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int rt = 0; rt < RTC; ++rt) {
        f32x4 acc[3] = {bias_1, splat(0.f), splat(0.f)};
        if constexpr (RTC >= 5) {                  // (five tiles: this team's registers hold the prefetch too - one input chunk at a time)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const F3 xi = bfrag(In0 + rt * 16 * IP, IP, rows * IP, c < KC1 ? c : 0, lane);
#define OP(p_, q_) acc[c] = mm(w1[c].q_, xi.p_, acc[c]);
            X6_TERMS(OP)
#undef OP
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          F3 xi[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) xi[c] = bfrag(In0 + rt * 16 * IP, IP, rows * IP, c < KC1 ? c : 0, lane);
#define OP(p_, q_) _Pragma("unroll") for (int c = 0; c < 3; ++c) acc[c] = mm(w1[c].q_, xi[c].p_, acc[c]);
          X6_TERMS(OP)
#undef OP
          if constexpr (RTC >= 4) __builtin_amdgcn_sched_barrier(0);      // (one tile's input fragments at a time)
        }
        pre[rt] = (acc[0] + acc[1]) + acc[2];
      }
    };
    auto xput = [&](int t) __attribute__((always_inline)) {      // x(t): the action fed at step t
      f32x4 wv[RTC];
#pragma unroll
      for (int rt = 0; rt < RTC; ++rt) wv[rt] = *reinterpret_cast<const f32x4*>(W1a + urow_of(t, uw[rt]) * H + u4);
#pragma unroll
      for (int rt = 0; rt < RTC; ++rt) {
        f32x4 x;
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = fmaxf(__fadd_rn(pre[rt][r], wv[rt][r]), 0.f);
        put4t(Xp0, HP, rows * HP, rt * 16 + m, u4, x);
      }
    };
    // q(ts) = fc2(h(ts)) of the row tiles this wave carries (tile rt: wave rt % 4), from the planes of buffer bh -> HBM.  Loads and
    // stores share one in-order counter: the loads in flight (next step's observations and fed action, issued at the end of the previous
    // phase 2) are waited for BEFORE the first store goes out (`landed`), so no later use of them ever waits for a store to complete.
    constexpr int QT = (RTC + 3) / 4;
    auto landed = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < NPF; ++j) asm volatile("" : "+v"(pf[j]));
#pragma unroll
      for (int rt = 0; rt < RTC; ++rt) asm volatile("" : "+v"(uw[rt]));
    };
    auto fc2 = [&](int bh, int ts, bool wait_loads) __attribute__((always_inline)) {
      if constexpr (!W2R) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const i32x4* d = W2f + (c * 3 * 4 + q) * A + (m < A ? m : 0);
          const i32x4 z = {0, 0, 0, 0};
          w2[c].h = m < A ? d[0] : z; w2[c].m = m < A ? d[4 * A] : z; w2[c].l = m < A ? d[8 * A] : z;
        }
      }
      f32x4 qa[QT];
#pragma unroll
      for (int k = 0; k < QT; ++k) {
        const int rt = 4 * k + s;
        qa[k] = splat(0.f);
        if (rt < RTC) {
          F3 hb[2];
#pragma unroll
          for (int c = 0; c < 2; ++c) hb[c] = bfrag(hpp(bh) + rt * 16 * HP, HP, rows * HP, c, lane);
          f32x4 ac[2] = {splat(bias_2), splat(0.f)};
#define OP(p_, q_) _Pragma("unroll") for (int c = 0; c < 2; ++c) ac[c] = mm(hb[c].p_, w2[c].q_, ac[c]);
          X6_TERMS(OP)
#undef OP
          qa[k] = ac[0] + ac[1];
        }
      }
      if (wait_loads) landed();
#pragma unroll
      for (int k = 0; k < QT; ++k) {
        const int rt = 4 * k + s;
        if (rt < RTC && m < A) {
#pragma unroll
          for (int r = 0; r < 4; ++r) a.q[(unsigned)(rowidx[rt * 16 + 4 * q + r] + ts * N) * (unsigned)A + (unsigned)m] = qa[k][r];      // (32-bit element offsets: the host checks B T N H 4 < 2^32)
        }
      }
    };
    issue(0);
    ufetch(0);
    commit(0);
    WG_BARRIER();                                  // P0: step 0's input planes
    fc1();
    xput(0);                                       // x(0)
    issue(1); ufetch(1);
    WG_BARRIER();                                  // P1: x(0); pre(0) taken: the input planes are free
    commit(1);
    issue(2);
    WG_BARRIER();                                  // P2: step 1's input planes
    ST_DECL(4);
    for (int t = 0; t < T; ++t) {
      // ---- phase 1 (beside the recurrence): pre(t+1); q(t-1) -> HBM
      if (t + 1 < T) fc1();
      if (t >= 1) fc2(t & 1, t - 1, true);         // h(t-1): the buffer the recurrence reads in this phase
      else landed();
      ST_MARK(0);
      WG_BARRIER();                                // X
      ST_MARK(1);
      // ---- phase 2: x(t+1) -> planes (the recurrence has read x(t)); step t+2 -> input planes (fc1 has read step t+1's).  The loads of
      // the NEXT phase 2 (observations of step t+3, the action fed at step t+2) go out last: every wait of this team then stands a whole
      // step behind the loads it waits for (loads and stores share one in-order counter: nothing fresh is ever waited for)
      if (t + 1 < T) xput(t + 1);
      commit(t + 2);                               // (past the end: the clamped last step again - nothing reads it)
      issue(t + 3);
      ufetch(t + 2);
      ST_MARK(2);
      WG_BARRIER();                                // Y
      ST_MARK(3);
    }
    ST_DUMP(4);
    fc2(T & 1, T - 1, false);                      // q of the last step
  }
}

static size_t px6_lds(int rtc, int KI, int A) {
  const size_t rows = 16 * (size_t)rtc, IP = KI + 8;
  return 3 * rows * IP * 2 + 3 * 3 * rows * HP * 2 + (size_t)(A + 1) * H * 4 + rows * (4 + 4 + 4 + 4) + (rtc >= 5 ? (size_t)2 * 3 * 4 * A * 16 : 0);
}

}  // namespace

ST_DEFINE_SETTER(marl_debug_stamps_agent_x6p)

// row tiles per workgroup of a launch on this kernel, or 0 when agent_x6.hip keeps it: non-saving launches over the whole chip of more
// than 512 row tiles (agent_x6.hip: one round of two-tile workgroups up to there), 2s3z-sized inputs (three fc1 chunks), one action tile
__attribute__((visibility("hidden"))) int marl_agent_x6p_tiles(int B, int T, int N, int O, int A, int last_action, int reuse_network, int cu_budget) {
  if (!marl_switches()->unroll_r6) return 0;
  const int I = O + (last_action ? A : 0) + (reuse_network ? N : 0);
  if (I > 96 || A < 1 || A > 16 || O < 8 || (O & 3) || T < 2 || B < 1) return 0;
  if (cu_budget != 0 && cu_budget != 256) return 0;
  const long tiles = ((long)B * N + 15) / 16;
  if (tiles <= 512) return 0;
  if ((long)16 * 5 * (O / 4) > 256L * NPF) return 0;                               // the prefetch registers hold a step's observations
  if (T > 32000) return 0;
  if ((double)B * T * N * H * 4.0 >= 4294967296.0) return 0;
  int tpw = (int)((tiles + 255) / 256);
  if (tpw > 5) {                                    // rounds of workgroups, evenly filled
    const long rounds = (tiles + 256 * 5 - 1) / (256 * 5);
    tpw = (int)((tiles + 256 * rounds - 1) / (256 * rounds));
    if (tpw > 5) tpw = 5;
  }
  if (tpw < 3) tpw = 3;
  return px6_lds(tpw, (I + 31) / 32 * 32, A) <= 160 * 1024 ? tpw : 0;
}

__attribute__((visibility("hidden"))) int marl_agent_x6p_launch(const marl_agent_weights_t* w, const float* obs, long obs_bs, int obs_t0,
                                                                 const int* ufed, long u_bs, int u_t0, const int* ep_len, const int* ep_map,
                                                                 const float* h0, float* q, float* h_last, int B, int T, int N, int O, int A,
                                                                 int last_action, int reuse_network, int tpw, void* stream) {
  if ((reinterpret_cast<uintptr_t>(obs) & 15) || (h0 && (reinterpret_cast<uintptr_t>(h0) & 15)) || (h_last && (reinterpret_cast<uintptr_t>(h_last) & 15)))
    return (int)hipErrorInvalidValue;
  PX6Args a;
  a.W1 = w->fc1_w; a.b1 = w->fc1_b; a.Wih = w->w_ih; a.Whh = w->w_hh; a.bih = w->b_ih; a.bhh = w->b_hh; a.W2 = w->fc2_w; a.b2 = w->fc2_b;
  a.obs = obs; a.obs_bs = obs_bs; a.obs_t0 = obs_t0; a.ufed = ufed; a.u_bs = u_bs; a.u_t0 = u_t0; a.ep_len = ep_len; a.ep_map = ep_map;
  a.h0 = h0; a.q = q; a.h_last = h_last;
  a.B = B; a.T = T; a.N = N; a.O = O; a.A = A;
  a.has_act = last_action ? 1 : 0; a.has_id = reuse_network ? 1 : 0;
  a.I = O + (last_action ? A : 0) + (reuse_network ? N : 0);
  a.KI = (a.I + 31) / 32 * 32;
  a.R = (long)B * N;
  // 32-bit element offsets of the observation items and fed actions (ep_map may point anywhere in a ring of obs_bs-sized episodes: the
  // caller's storage is one allocation of at most 2^31 elements - checked by the host wrapper through the record's size)
  const long tiles = (a.R + 15) / 16;
  const size_t lds = px6_lds(tpw, a.KI, A);
  const void* fn = tpw == 3 ? (const void*)agent_fwd_x6p_kernel<3> : tpw == 4 ? (const void*)agent_fwd_x6p_kernel<4> : (const void*)agent_fwd_x6p_kernel<5>;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  dim3 grid((unsigned)((tiles + tpw - 1) / tpw)), block(PNT);
  void* kargs[] = {(void*)&a};
  e = hipLaunchKernel(fn, grid, block, kargs, lds, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  MARL_CHECK_LAUNCH();
  return 0;
}
