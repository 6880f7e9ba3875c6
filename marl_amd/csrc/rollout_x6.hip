// Whole-rollout persistent kernel with the agent step as bf16x6 split products (x6.h; args.gemm_mode = "bf16x6"): ONE launch plays all
// T lock-steps of E synthetic SMAC-shaped environments (reference rollout.py:60-101 + controller/share_params.py:37-72, vectorised).
// Same environment (synth_hash.h), same epsilon-greedy choice, same episode record as rollout_fused.hip (the fp32 MFMA kernel); the
// agent's products (network/q_network.py:16-21) are six bf16 MFMA products each, fp32 accumulate.
//
// Round 6 decomposition (the round-5 one lives on in rollout_x6_v1.hip as this kernel's bitwise twin).  What bounded the first one:
// its input-side gate sums travelled team I -> LDS (12 KB of fp32 per row tile) -> team R, which capped a workgroup at three row
// tiles (4096 envs x 5 agents = five tiles per CU -> TWO rounds of three-tile workgroups, one tile-slot in six empty) and cost a
// fourth barrier per lock-step.  Here the recurrent team holds BOTH gate matrices and runs x W_ih and h W_hh down ONE accumulator
// chain (bias, the x chunks, then the h chunks: the very order of additions the hand-over had, so the two kernels agree bit for bit):
//   team R (waves 0-3, hidden-unit slice s): W_ih and W_hh fragments (144 registers), the hidden state in fp32 registers
//   team I (waves 4-7, slice s):             fc1 (observation part) and fc2 fragments, the epsilon-greedy choice, the env step
// No gate-sum buffer: 30 KB of LDS per row tile -> up to FIVE tiles per workgroup, the whole share of a CU in one round.
// A lock-step, three LDS-only barriers; the dependent chain is  rec (R) | choice (I) | x (I), everything else rides beside it:
//   A  R: gates(t) = bias + x(t) W_ih + h(t-1) W_hh, gate math, h(t) -> planes (software-pipelined over the row tiles: a tile's gate
//         math sits in the gaps of the next tile's products)            I: pre(t+1) = fc1 of the slot-(t+1) planes; state + availability of slot t+1
//                                                                          (record, bit masks)
//   B  I: q(t) = fc2(h(t)) and the epsilon-greedy choice IN REGISTERS (a row's 16 actions on a DPP row: max + three ballots), every
//         stage for all row tiles side by side                          R: observations of slot t+2 (record + input planes), part 1
//   C  I: x(t+1) = relu(pre + W1[:, O + u(t)]) -> planes; env step      R: part 2; the hashes of the steps to come (one prefix per
//                                                                          environment, stream and time)
// fc1 = (bias + W1[:, obs | id] in) - on the matrix cores - + W1[:, O + u], one column of fp32 weights added per row once u is known
// (a table in LDS).  Availability lives in LDS as one bit mask per row and slot (a ring of four slots: no hazards).
#include "x6.h"
#include "synth_hash.h"
#include "../../include/marl_hip.h"

// the round-5 kernel (rollout_x6_v1.hip)
int marl_rollout_x6_v1_supported(int N, int O, int A);
int marl_rollout_x6_v1(const marl_agent_weights_t* w, unsigned seed, unsigned rseed, int env0, int episode, int fixed_len, const float* eps,
                       float* obs, float* state, long state_ld, float* avail, int* u, float* r, float* term, float* padded, int* length,
                       int* won, float* h_out, float* stats, double eps0, double eps_anneal, double eps_min, int E, int T, int N, int O,
                       int S, int A, int last_action, int reuse_network, void* stream);

namespace {

constexpr int H = 64;
constexpr int RNT = 512;
constexpr int HP = 72;            // pitch (bf16) of the 64-wide planes

#define WG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#ifdef MARL_STAMPS
#define ST_RESET() ST_NOW(st_prev_)
#else
#define ST_RESET()
#endif
#ifndef OI_CUT_16
#define OI_CUT_16 9      // sixteenths of a slot's observation items team R generates before the choice barrier (A/B builds)
#endif
#define X6_TERMS(OP) OP(m, m) OP(h, l) OP(l, h) OP(h, m) OP(m, h) OP(h, h)

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float group_max16(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));     // quad_perm [1,0,3,2]
  v = fmaxf(v, dpp_mov<0x4E>(v));     // quad_perm [2,3,0,1]
  v = fmaxf(v, dpp_mov<0x141>(v));    // row_half_mirror
  v = fmaxf(v, dpp_mov<0x140>(v));    // row_mirror
  return v;
}
// B fragment of W (row-major, ldw floats per row): lane (g, j): W[row0 + j][32 c + 8g .. + 7]  (rows >= rows_valid and columns >= K: 0)
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
// A fragment from a plane tile (hi plane at pl, the others ps elements further): lane (g, m) reads row m, columns 32 c + 8g .. + 7
__device__ __forceinline__ F3 bfrag(const short* pl, int pitch, int ps, int c, int lane) {
  const int m = lane & 15, g = lane >> 4;
  const short* p = pl + m * pitch + 32 * c + 8 * g;
  F3 f;
  f.h = *reinterpret_cast<const i32x4*>(p);
  f.m = *reinterpret_cast<const i32x4*>(p + ps);
  f.l = *reinterpret_cast<const i32x4*>(p + 2 * ps);
  return f;
}
// accumulator tile (rows row0 + r, r = 0..3, of column col) -> planes
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
// accumulator tile of a TRANSPOSED product (weights as the A operand: lane (q, m) holds columns col0 .. col0 + 3 of row `row`) -> planes:
// one 8-byte write per plane (the row-major products' put4 above needs four 2-byte writes per plane)
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

struct RX6Args {
  const float *W1, *b1, *Wih, *Whh, *bih, *bhh, *W2, *b2;
  const float* eps;       // [T] epsilon of each lock-step (device), or null: the schedule below
  double eps0, eps_anneal, eps_min;
  float* stats;           // [3][E] or null: per episode  sum_t r | won | length
  float *obs, *state, *avail;   // (E,T+1,N,O) (E,T+1,SL >= S) (E,T+1,N,A)
  long SL;
  int* u;                 // (E,T,N)
  float *r, *term, *padded;     // (E,T)
  int *length, *won;      // (E)
  float* h_out;           // (E*N,64) final hidden state or null
  unsigned seed, rseed;
  int env0, episode, fixed_len;
  int E, T, N, O, S, A, I, KI;
  int EPW;                // whole environments per workgroup
  int has_act, has_id;
  long R;
};

template <int RTC, int NK1, int AC = 1>      // row tiles per workgroup, fc1 chunks of 32 input columns, action tiles of 16 (fc2 / the choice)
__global__ __launch_bounds__(RNT, 2) void synth_rollout_x6_kernel(RX6Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave >> 2, s = wave & 3;      // team 0 = R (the recurrence; off the chain: slot generation), team 1 = I (fc1, fc2 + choice, x, env step)
  const int q = lane >> 4, m = lane & 15, u = 16 * s + m;
  constexpr int rows = 16 * RTC;
  const int T = a.T, N = a.N, O = a.O, S = a.S, A = a.A;
  const int IP = a.KI + 8;
  const int IN_E = 3 * rows * IP, XP_E = 3 * rows * HP;
  short* In0 = reinterpret_cast<short*>(smem);                             // [3][rows][IP]   input planes of the slot fc1 reads next
  short* Xp0 = In0 + IN_E;                                                 // [3][rows][HP]   x(t+1)
  short* Hp0 = Xp0 + XP_E;                                                 // [2][3][rows][HP] h by step parity
  float* W1a = reinterpret_cast<float*>(Hp0 + 2 * XP_E);                   // [A + 1][64]: fc1 columns of the one-hot(last action) block; row A = zeros
  unsigned* avm = reinterpret_cast<unsigned*>(W1a + (A + 1) * H);          // [4 slots][rows] availability bit masks (bit k = action k)
  int* act = reinterpret_cast<int*>(avm + 4 * rows);                       // [rows] chosen action (-1: none)
  float* uex = reinterpret_cast<float*>(act + rows);                       // [2 step parities][2][rows] explore / pick uniforms of a step's choice
  int4* rmeta = reinterpret_cast<int4*>(uex + 4 * rows);                   // [rows] {obs offset of (b,0,n,0), avail offset, episode length, n | local env << 16}
  int4* emeta = rmeta + rows;                                              // [EPW] {state offset of (b,0,0), episode length, env in range, -}
  // hash prefixes over (seed, stream, env, time): ONE per environment, stream and time - [6 streams][2 parities][EPW]:
  // observations / availability / state of a slot (parity of the slot), reward / explore / pick of a step (parity of the step)
  unsigned* P = reinterpret_cast<unsigned*>(emeta + a.EPW);
  // (two action tiles: the fc2 fragments live in LDS - [2 tiles x 2 k chunks][3 planes][64 lanes] 16 bytes, the same for every wave -
  // team I's registers hold seven fc1 chunks there)
  i32x4* W2f = reinterpret_cast<i32x4*>(reinterpret_cast<char*>(P) + ((12 * a.EPW * 4 + 15) & ~15));
  enum { K_OBS = 0, K_AVAIL, K_STATE, K_REWARD, K_EXPLORE, K_PICK };
  auto hpp = [&](int b) { return Hp0 + b * XP_E; };

  const int nenv_wg = a.EPW;
  const int vrows = nenv_wg * N;                         // valid rows; rows vrows .. 16 RTC - 1 are padding
  // (the host sizes RTC = ceil(EPW N / 16): every row tile holds valid rows)
  const int b0 = blockIdx.x * nenv_wg;
  const long row0 = (long)b0 * N;
  const int lmin = T / 2 > 1 ? T / 2 : 1;
  auto ep_len = [&](int b) {
    const int L = lmin + (int)(hkey(a.seed, ST_LEN, (unsigned)(a.env0 + b), (unsigned)a.episode, 0u) % (unsigned)(T - lmin + 1));
    return a.fixed_len ? T : L;
  };
  for (int r = tid; r < rows; r += RNT) {
    long rho = row0 + (r < vrows ? r : vrows - 1);
    if (rho > a.R - 1) rho = a.R - 1;                    // clamp: duplicates of the last row
    const int b = (int)(rho / N), n = (int)(rho % N);
    const int bn = b * (T + 1) * N + n;
    rmeta[r] = make_int4(bn * O, bn * A, ep_len(b), n | ((b - b0) << 16));
    act[r] = -1;
  }
  for (int r = tid; r < 4 * rows; r += RNT) avm[r] = 0u;
  for (int e = tid; e < nenv_wg; e += RNT) {
    int b = b0 + e; if (b > a.E - 1) b = a.E - 1;
    const int L = ep_len(b);
    a.length[b] = L;
    const int won_ = (int)(hkey(a.seed, ST_WON, (unsigned)(a.env0 + b), (unsigned)a.episode, 0u) & 1u);
    a.won[b] = won_;
    if (a.stats && b0 + e < a.E) { a.stats[a.E + b] = (float)won_; a.stats[2L * a.E + b] = (float)L; }
    emeta[e] = make_int4((b0 + e) * (T + 1) * (int)a.SL, L, b0 + e < a.E ? 1 : 0, 0);
  }
  // planes: zero everywhere (padding rows, pad columns, the one-hot block: its contribution comes from the W1a table), h(-1) = 0
  for (int e = tid; e < (IN_E + 3 * XP_E) / 2; e += RNT) reinterpret_cast<int*>(In0)[e] = 0;
  for (int e = tid; e < (A + 1) * H; e += RNT) {
    const int aa = e / H, j = e % H;
    W1a[e] = (a.has_act && aa < A) ? a.W1[(long)j * a.I + O + aa] : 0.f;
  }
  __syncthreads();
  if (a.has_id)
    for (int r = tid; r < vrows; r += RNT) In0[r * IP + (a.I - N) + (rmeta[r].w & 0xffff)] = (short)0x3F80;      // agent id: bf16 1.0 in the hi plane

  // ---- the environment's slot t -> record (+ input planes / availability masks); flattened over (row, 4-column group) items,
  // branch-free (rollout_fused.hip: gen_slot); `tl` / `nthr`: the threads that share the work.  The slot's hash prefixes sit in the
  // entry t & 1 of the prefix arrays (the other entry is being written for slot t + 1 meanwhile).
  const int O4 = O >> 2, S4 = (S + 3) >> 2;
  const float invO4 = 1.0f / (float)(O4 > 0 ? O4 : 1), invA = 1.0f / (float)A;
  const float invS4 = 1.0f / (float)(S4 > 0 ? S4 : 1), invS = 1.0f / (float)S;
  const bool svec = (a.SL & 3) == 0 && a.SL >= 4 * S4 && (reinterpret_cast<uintptr_t>(a.state) & 15) == 0;
  auto bits = [](float v) { return __builtin_bit_cast(unsigned, v); };
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  auto Pp = [&](int kind, int par) { return P + (kind * 2 + par) * nenv_wg; };
  auto kstream = [](int kind) { return kind == K_OBS ? ST_OBS : kind == K_AVAIL ? ST_AVAIL : kind == K_STATE ? ST_STATE : kind == K_REWARD ? ST_REWARD : kind == K_EXPLORE ? ST_EXPLORE : ST_PICK; };
  // prefix of (kind, local env el) at time index tt -> entry par
  auto put_prefix = [&](int kind, int el, int tt, int par) {
    const unsigned tg = (unsigned)(a.episode * (T + 1) + tt);
    Pp(kind, par)[el] = hprefix(kind >= K_EXPLORE ? a.rseed : a.seed, (unsigned)kstream(kind), (unsigned)(a.env0 + b0 + el), tg);
  };
  const float inv_nE = 1.0f / (float)nenv_wg, inv_rows = 1.0f / (float)rows;
  // observation items e_lo <= e < e_hi of slot t (item = (row, 4-column group)), thread tl of nthr
  auto gen_obs = [&](int t, bool to_lds, int e_lo, int e_hi, int tl, int nthr) __attribute__((always_inline)) {
    const int tNO = t * N * O;
    const unsigned* pO = Pp(K_OBS, t & 1);
    for (int e = e_lo + tl; e < e_hi; e += nthr) {
      const int r = (int)(((float)e + 0.5f) * invO4);
      const int k = 4 * (e - r * O4);
      const int4 mt = rmeta[r];
      const unsigned po = pO[mt.w >> 16];
      const unsigned lm = t <= mt.z ? 0xffffffffu : 0u, fm = t < mt.z ? 0xffffffffu : 0u;
      const unsigned idx = (unsigned)((mt.w & 0xffff) * O + k);
      u32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = bits(2.0f * u01(hfin(po, idx + (unsigned)i)) - 1.0f) & lm;
      *reinterpret_cast<u32x4*>(a.obs + (long)mt.x + tNO + k) = v;
      if (to_lds) {                              // padded steps feed zeros (rollout.py:122-133); split once, here
        f32x4 w;
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = __builtin_bit_cast(float, v[i] & fm);
        const F3h f = split4(w);
        short* p = In0 + r * IP + k;
        *reinterpret_cast<i32x2*>(p) = f.h;
        *reinterpret_cast<i32x2*>(p + rows * IP) = f.m;
        *reinterpret_cast<i32x2*>(p + 2 * rows * IP) = f.l;
      }
    }
  };
  // state + availability of slot t: record, and the availability bit masks of the slot (ring entry t & 3: zero when this starts - the
  // entry of slot t + 1 is cleared here for the next call)
  auto gen_rest = [&](int t, int tl, int nthr, bool do_avail = true, bool do_state = true) __attribute__((always_inline)) {
    const int tNA = t * N * A, tS = t * (int)a.SL;
    unsigned* am = avm + (t & 3) * rows;
    const unsigned* pA = Pp(K_AVAIL, t & 1);
    const unsigned* pS = Pp(K_STATE, t & 1);
    if (do_avail) for (int r = tl; r < rows; r += nthr) avm[((t + 1) & 3) * rows + r] = 0u;
    if (do_avail) for (int e = tl; e < vrows * A; e += nthr) {
      const int r = (int)(((float)e + 0.5f) * invA);
      const int k = e - r * A;
      const int4 mt = rmeta[r];
      const float uu = u01(hfin(pA[mt.w >> 16], (unsigned)((mt.w & 0xffff) * A + k)));
      const bool on = (t <= mt.z) & ((k == 0) | (uu < 0.7f));
      a.avail[(long)mt.y + tNA + k] = on ? 1.f : 0.f;
      if (on) atomicOr(am + r, 1u << k);
    }
    if (!do_state) return;
    if (svec) {
      for (int e = tl; e < nenv_wg * S4; e += nthr) {
        const int el = (int)(((float)e + 0.5f) * invS4);
        const int k = 4 * (e - el * S4);
        const int4 mt = emeta[el];
        const unsigned ps = pS[el];
        const unsigned lm = t <= mt.y ? 0xffffffffu : 0u;
        u32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = bits(2.0f * u01(hfin(ps, (unsigned)(k + i))) - 1.0f) & (k + i < S ? lm : 0u);
        if (mt.z) *reinterpret_cast<u32x4*>(a.state + (long)mt.x + tS + k) = v;
      }
    } else {
      for (int e = tl; e < nenv_wg * S; e += nthr) {
        const int el = (int)(((float)e + 0.5f) * invS);
        const int k = e - el * S;
        const int4 mt = emeta[el];
        const unsigned lm = t <= mt.y ? 0xffffffffu : 0u;
        const unsigned v = bits(2.0f * u01(hfin(pS[el], (unsigned)k)) - 1.0f) & lm;
        if (mt.z) reinterpret_cast<unsigned*>(a.state)[(long)mt.x + tS + k] = v;
      }
    }
  };
  // ---- prologue: slot 0 (record, planes, masks) -> x(0) (team I) -> slot 1 takes the input planes' place; the uniforms of the first choice
  const int n_oi = vrows * O4;                   // observation items of a slot
  __syncthreads();
  for (int x = tid; x < 10 * nenv_wg; x += RNT) {      // slots 0, 1 (obs / avail / state) and steps 0, 1 (explore / pick)
    const int j = (int)(((float)x + 0.5f) * inv_nE), el = x - j * nenv_wg;
    if (j < 6) put_prefix(j % 3, el, j / 3, j / 3);
    else put_prefix(K_EXPLORE + (j - 6) % 2, el, (j - 6) / 2, (j - 6) / 2);
  }
  __syncthreads();
  for (int x = tid; x < 2 * rows; x += RNT) {           // the uniforms of the first choice
    const int which = x >= rows ? 1 : 0, r = x - which * rows;
    const int w_ = rmeta[r].w;
    uex[which * rows + r] = u01(hfin(Pp(K_EXPLORE + which, 0)[w_ >> 16], (unsigned)(w_ & 0xffff)));
  }
  gen_obs(0, true, 0, n_oi, tid, RNT);
  gen_rest(0, tid, RNT);
  auto prologue_slot1 = [&]() __attribute__((always_inline)) {      // (behind P1: x(0) made, the input planes and slot 0's prefixes are free)
    gen_obs(1, 1 < T, 0, n_oi, tid, RNT);
    gen_rest(1, tid, RNT);
    for (int x = tid; x < 2 * nenv_wg; x += RNT)        // slot 2's observations and state (the entries slot 0 left)
      put_prefix(x < nenv_wg ? K_OBS : K_STATE, x < nenv_wg ? x : x - nenv_wg, 2, 0);
  };
  // the hashes of the steps to come, ONE item per thread of a 256-thread team at the headline shape (6 EPW prefixes + 2 x rows uniforms):
  // prefixes of slot t+3's observations and state (team R generates them in B / C of step t+1), of slot t+2's availability (team I, A
  // of step t+1), of step t+1's reward and step t+2's explore / pick draws; the uniforms of step t+1's choice from the explore / pick
  // prefixes the previous call left
  auto hashes = [&](int t, int tl, int nthr) __attribute__((always_inline)) {
    const int nP = 6 * nenv_wg;
    for (int x = tl; x < nP + 2 * rows; x += nthr) {
      if (x < nP) {
        const int kind = (int)(((float)x + 0.5f) * inv_nE), el = x - kind * nenv_wg;
        if (kind == K_OBS || kind == K_STATE) put_prefix(kind, el, t + 3, (t + 1) & 1);
        else if (kind == K_AVAIL) put_prefix(kind, el, t + 2, t & 1);
        else if (kind == K_REWARD) put_prefix(kind, el, t + 1, (t + 1) & 1);
        else put_prefix(kind, el, t + 2, t & 1);
      } else {
        const int j = x - nP, which = j >= rows ? 1 : 0, r = j - which * rows;
        const int w_ = rmeta[r].w;
        uex[((t + 1) & 1) * 2 * rows + which * rows + r] = u01(hfin(Pp(K_EXPLORE + which, (t + 1) & 1)[w_ >> 16], (unsigned)(w_ & 0xffff)));
      }
    }
  };
  // env step (reward / terminated / padded; fixed-order fp32 sum over agents): lane -> (env, agent); the N rows of an environment sit in
  // ONE wave, environments dealt to the four slice waves of a team in turn.  Team I's up from three row tiles per workgroup (beside its
  // x, while team R generates observations); team R's at one or two tiles, where team R has nothing else left in phase C and the step
  // is a chain of latencies (stamps at one tile: the two waves that carry the environments were the last at the barrier by 1 400 cycles)
  constexpr bool ENV_R = RTC <= 2;
  const int es_epw = 64 / N, es_er = lane / N;
  const int es_n = lane - es_er * N, es_l0 = es_er * N;
  const int es_el = 4 * es_er + s;
  const bool es_has = es_er < es_epw && es_el < nenv_wg && b0 + es_el < a.E;
  const int es_L = es_has ? emeta[es_el].y : 0;
  float ep_r = 0.f;
  auto env_step = [&](int t) __attribute__((always_inline)) {
    if (es_has) {
      const unsigned tg = (unsigned)(a.episode * (T + 1) + t);
      const int L = es_L;
      const bool live = t < L;
      float term = 0.f;
      if (live) {
        const unsigned pre_ = t > 0 ? Pp(K_REWARD, t & 1)[es_el] : hprefix(a.seed, ST_REWARD, (unsigned)(a.env0 + b0 + es_el), tg);
        term = u01(hfin(pre_, (unsigned)(es_n * A + act[es_el * N + es_n]))) - 0.5f;
      }
      float acc = 0.f;
      for (int n0 = 0; n0 < N; n0 += 4) {          // four shuffles in flight; the sum stays in agent order
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = __shfl(term, (es_l0 + n0 + k) & 63, 64);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = n0 + k < N ? acc + v[k] : acc;
      }
      if (es_n == 0) {
        const long o = (long)(b0 + es_el) * T + t;
        const float rew = live ? acc * (1.0f / (float)N) : 0.f;
        ep_r = ep_r + rew;
        a.r[o] = rew;
        a.term[o] = live ? (t + 1 >= L ? 1.f : 0.f) : 1.f;
        a.padded[o] = live ? 0.f : 1.f;
      }
    }
  };
  const int KC1 = a.KI >> 5;
  // epsilon of step t: a device vector, or the reference's per-step anneal (rollout.py:100-101) evaluated here in fp64
  float eps_next = a.eps ? a.eps[0] : (float)a.eps0;
  double eps_d = a.eps0;

  if (AC == 2 && wave < 4) {
    const F3 f = wfrag(a.W2, H, 16 * (wave >> 1), A, H, wave & 1, lane);
    W2f[(wave * 3 + 0) * 64 + lane] = f.h; W2f[(wave * 3 + 1) * 64 + lane] = f.m; W2f[(wave * 3 + 2) * 64 + lane] = f.l;
  }
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
    // the products run TRANSPOSED (weights as the A operand): lane (q, m) holds hidden units 16 s + 4 q + r (r = 0..3) of row m - four
    // consecutive columns of one row of the planes
    const int u4 = 16 * s + 4 * q;
    f32x4 bias_r, bias_z, bias_n, bias_hn;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      bias_r[r] = a.bih[u4 + r] + a.bhh[u4 + r]; bias_z[r] = a.bih[H + u4 + r] + a.bhh[H + u4 + r];
      bias_n[r] = a.bih[2 * H + u4 + r]; bias_hn[r] = a.bhh[2 * H + u4 + r];
    }
    f32x4 hreg[RTC];
#pragma unroll
    for (int rt = 0; rt < RTC; ++rt) hreg[rt] = splat(0.f);
    // R's share of a slot's observation items before the choice barrier (team I needs ~the choice's time to get there)
    const int oi_cut = (n_oi * OI_CUT_16) >> 4;
    auto rot = [&](int w) { return (tid + 64 * w) & (RNT / 2 - 1); };      // this thread's index with the team's waves rotated by w
    WG_BARRIER();                                  // P0: slot 0 in the planes, tables (team I: fc1, x(0))
    WG_BARRIER();                                  // P1: x(0), pre(0) taken: the input planes are free
    prologue_slot1();
    WG_BARRIER();                                  // P2
    ST_DECL(6);
    for (int t = 0; t < T; ++t) {
      const int par = t & 1;
      // ---- A: gates(t) = bias + x(t) W_ih + h(t-1) W_hh (one accumulator chain per gate: bias, x chunks, h chunks), gate math, h(t).
      // Software pipeline over the row tiles: the 72 products of tile rt are issued with the gate math of tile rt-1 between them
      // (two accumulator sets; an operand chunk is read from LDS while the previous one multiplies) - written tile by tile the matrix
      // pipe idled through every tile's gate math and the gate math waited for every tile's last product.
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
      WG_BARRIER();                                // B1: h(t) planes | pre(t+1) taken: the input planes are free; prefixes of slot t+2
      ST_MARK(1);
      // ---- B, C (beside team I's choice and x): slot t+2 -> record; its observations -> input planes (fc1 reads them in A of step
      // t+1), its availability -> bit masks (the choice of step t+2 reads them)
      // (the item loops of the different jobs start at different waves - rot(): with few items, one or two row tiles per workgroup,
      // every job would otherwise land on wave 0, which shares its SIMD with team I's first wave)
      if (t + 2 <= T) {
        gen_rest(t + 2, rot(3), RNT / 2, false, true);
        gen_obs(t + 2, t + 2 < T, 0, oi_cut, tid, RNT / 2);
      }
      ST_MARK(2);
      WG_BARRIER();                                // B2: act(t)
      ST_MARK(3);
      if (ENV_R) env_step(t);
      if (t + 2 <= T) gen_obs(t + 2, t + 2 < T, oi_cut, n_oi, rot(1), RNT / 2);
      hashes(t, rot(2), RNT / 2);
      ST_MARK(4);
      WG_BARRIER();                                // B3: x(t+1) planes | slot t+2
      ST_MARK(5);
    }
    ST_DUMP(6);
    if (ENV_R && a.stats && es_has && es_n == 0) a.stats[b0 + es_el] = ep_r;
    if (a.h_out) {
#pragma unroll
      for (int rt = 0; rt < RTC; ++rt) {
        const int row = rt * 16 + m;
        const long rho = row0 + row;
        if (row < vrows && rho < a.R) *reinterpret_cast<f32x4*>(a.h_out + rho * H + u4) = hreg[rt];
      }
    }
  } else {
    // =============================== team I: the slots to come, fc1, fc2 + the choice, x, the environment's bookkeeping ===============================
    const int ti = tid - RNT / 2;
    ST_DECL(8);
    F3 w1[NK1], w2[1][2];      // (w2: one action tile; two tiles read their fragments from LDS)
#pragma unroll
    for (int c = 0; c < NK1; ++c) w1[c] = c < KC1 ? wfrag(a.W1, a.I, 16 * s, H, a.I, c, lane) : F3{};
#pragma unroll
    for (int c = 0; c < 2; ++c)
      if constexpr (AC == 1) w2[0][c] = wfrag(a.W2, H, 0, A, H, c, lane);
    const int u4 = 16 * s + 4 * q;                 // fc1 runs transposed like the recurrence: lane (q, m) = units u4 .. u4 + 3 of row m
    const f32x4 bias_1 = {a.b1[u4], a.b1[u4 + 1], a.b1[u4 + 2], a.b1[u4 + 3]};
    float bias_2[AC];
#pragma unroll
    for (int at = 0; at < AC; ++at) bias_2[at] = 16 * at + m < A ? a.b2[16 * at + m] : 0.f;
    f32x4 pre[RTC];
    // pre = bias + W1[:, obs | id] in  of every row tile from the input planes (three accumulator chains, chunk c on chain c % 3)
    auto fc1 = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int rt = 0; rt < RTC; ++rt) {
        f32x4 acc[3] = {bias_1, splat(0.f), splat(0.f)};
#pragma unroll
        for (int c0 = 0; c0 < NK1; c0 += 3) {
          F3 xi[3];
#pragma unroll
          for (int c = 0; c < 3; ++c)
            if (c0 + c < NK1) xi[c] = bfrag(In0 + rt * 16 * IP, IP, rows * IP, c0 + c < KC1 ? c0 + c : 0, lane);
#define OP(p_, q_) _Pragma("unroll") for (int c = 0; c < 3; ++c) if (c0 + c < NK1) acc[c] = mm(w1[c0 + c].q_, xi[c].p_, acc[c]);
          X6_TERMS(OP)
#undef OP
        }
        pre[rt] = (acc[0] + acc[1]) + acc[2];
        if constexpr (NK1 == 7) __builtin_amdgcn_sched_barrier(0);      // (84 registers of fc1 fragments: one tile's input fragments at a time)
      }
    };
    // x = relu(pre + W1[:, O + last action]) -> planes (rows whose last action is "none": the zero row of the table); stage by stage
    // over the row tiles: every LDS read of a stage is in flight before the first result is used
    auto xput = [&]() __attribute__((always_inline)) {
      int aa[RTC];
#pragma unroll
      for (int rt = 0; rt < RTC; ++rt) aa[rt] = act[rt * 16 + m];
      f32x4 wv[RTC];
#pragma unroll
      for (int rt = 0; rt < RTC; ++rt) wv[rt] = *reinterpret_cast<const f32x4*>(W1a + (aa[rt] < 0 ? A : aa[rt]) * H + u4);
#pragma unroll
      for (int rt = 0; rt < RTC; ++rt) {
        f32x4 x;
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = fmaxf(__fadd_rn(pre[rt][r], wv[rt][r]), 0.f);
        put4t(Xp0, HP, rows * HP, rt * 16 + m, u4, x);
      }
    };
    // ---- q = fc2(h) and the epsilon-greedy choice (share_params.py:66-70), in registers: lane (q, m) of the accumulator tile holds
    // rows 4q + r of action m - a row's 16 actions sit on the 16 lanes of a DPP row.  Every wave of the team computes q of every tile
    // (12 products) and makes the choice of its rows 4q + s (four rows in ONE pass of ballots).  Stage by stage over the row tiles (the
    // tiles' chains - products, DPP maxima, ballots - are independent: written side by side they hide each other's latencies).
    int c_rr[RTC], c_len[RTC], c_uoff[RTC];      // this lane's row of each tile: its index (clamped), episode length (-1: not a row), u offset
#pragma unroll
    for (int rt = 0; rt < RTC; ++rt) {
      const int row = rt * 16 + 4 * q + s;
      const bool valid = row < vrows;
      const int rr = valid ? row : vrows - 1;
      const int4 mt = rmeta[rr];
      c_rr[rt] = rr; c_len[rt] = valid ? mt.z : -1;
      c_uoff[rt] = (b0 + (mt.w >> 16)) * T * N + (mt.w & 0xffff);
    }
    auto choose_all = [&](int t, float eps) __attribute__((always_inline)) {
      const int par = t & 1;
      unsigned avw[RTC]; float ue[RTC], up[RTC], qsel[RTC][AC];
#pragma unroll
      for (int rt = 0; rt < RTC; ++rt) {
        avw[rt] = avm[(t & 3) * rows + c_rr[rt]];
        ue[rt] = uex[par * 2 * rows + c_rr[rt]];
        up[rt] = uex[par * 2 * rows + rows + c_rr[rt]];
      }
#pragma unroll
      for (int rt = 0; rt < RTC; ++rt) {
        F3 hb[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) hb[c] = bfrag(hpp(par ^ 1) + rt * 16 * HP, HP, rows * HP, c, lane);
#pragma unroll
        for (int at = 0; at < AC; ++at) {
          f32x4 ac[2] = {splat(bias_2[at]), splat(0.f)};
          F3 wq[2];
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            if constexpr (AC == 1) wq[c] = w2[0][c];
            else { const i32x4* f = W2f + (at * 2 + c) * 3 * 64 + lane; wq[c].h = f[0]; wq[c].m = f[64]; wq[c].l = f[128]; }
          }
#define OP(p_, q_) _Pragma("unroll") for (int c = 0; c < 2; ++c) ac[c] = mm(hb[c].p_, wq[c].q_, ac[c]);
          X6_TERMS(OP)
#undef OP
          const f32x4 qv = ac[0] + ac[1];
          qsel[rt][at] = s == 0 ? qv[0] : s == 1 ? qv[1] : s == 2 ? qv[2] : qv[3];
          if constexpr (AC == 2) __builtin_amdgcn_sched_barrier(0);      // (seven fc1 chunks live in this team's registers: one tile's fragments at a time)
        }
      }
      ST_MARK(6);
      const int sh = 16 * q;
#pragma unroll
      for (int rt = 0; rt < RTC; ++rt) {
        // a lane holds the row's actions m (and 16 + m with two action tiles); masks over the row's actions are 16 bits per tile
        bool on[AC]; float v[AC], mx = -3.0e38f;
        unsigned am = 0u, em = 0u;
#pragma unroll
        for (int at = 0; at < AC; ++at) {
          on[at] = 16 * at + m < A && ((avw[rt] >> (16 * at + m)) & 1u) != 0u;
          v[at] = on[at] ? qsel[rt][at] : -3.0e38f;
          mx = fmaxf(mx, group_max16(v[at]));
          am |= ((unsigned)(__ballot(on[at]) >> sh) & 0xffffu) << (16 * at);
        }
#pragma unroll
        for (int at = 0; at < AC; ++at) em |= ((unsigned)(__ballot(on[at] && v[at] == mx) >> sh) & 0xffffu) << (16 * at);
        const int navail = __popc(am);
        int arg = em ? __ffs(em) - 1 : (am ? __ffs(am) - 1 : 0);
        const bool explore = ue[rt] < eps;
        int kk = (int)floorf(up[rt] * (float)navail);
        if (kk > navail - 1) kk = navail - 1;
        unsigned sm = 0u;
#pragma unroll
        for (int at = 0; at < AC; ++at) {
          const bool sel = explore && on[at] && __popc(am & ((1u << (16 * at + m)) - 1u)) == kk;
          sm |= ((unsigned)(__ballot(sel) >> sh) & 0xffffu) << (16 * at);
        }
        if (sm) arg = __ffs(sm) - 1;
        if (!(t < c_len[rt])) arg = -1;
        if (c_len[rt] >= 0 && m == 0) {
          act[rt * 16 + 4 * q + s] = arg;
          a.u[c_uoff[rt] + t * N] = arg;
        }
      }
    };
    WG_BARRIER();                                  // P0: slot 0 in the planes, tables
    fc1();
    xput();                                        // x(0): no last action
    WG_BARRIER();                                  // P1: x(0); the input planes are free
    prologue_slot1();
    WG_BARRIER();                                  // P2
    ST_RESET();
    for (int t = 0; t < T; ++t) {
      const float eps = eps_next;
      if (a.eps) { if (t + 1 < T) eps_next = a.eps[t + 1]; }
      else { eps_d = eps_d > a.eps_min ? eps_d - a.eps_anneal : eps_d; eps_next = (float)eps_d; }
      const unsigned tg = (unsigned)(a.episode * (T + 1) + t);
      // ---- A (beside the recurrence): pre(t+1) from the planes of slot t+1; state + availability of slot t+1 (record, bit masks; slot 1's
      // were made in the prologue)
      if (t + 1 < T) fc1();
      if (t >= 1) gen_rest(t + 1, ti, RNT / 2, true, false);
      ST_MARK(0);
      WG_BARRIER();                                // B1: h(t)
      ST_MARK(1);
      // ---- B: q(t) and the choice
      choose_all(t, eps);
      ST_MARK(2);
      WG_BARRIER();                                // B2: act(t)
      ST_MARK(3);
      // ---- C: x(t+1); env step (reward / terminated / padded; fixed-order fp32 sum over agents)
      if (t + 1 < T) xput();
      ST_MARK(7);
      if (!ENV_R) env_step(t);
      ST_MARK(4);
      WG_BARRIER();                                // B3: x(t+1)
      ST_MARK(5);
    }
    ST_DUMP(8);
    if (!ENV_R && a.stats && es_has && es_n == 0) a.stats[b0 + es_el] = ep_r;
  }
}

// LDS bytes of a workgroup of RTC row tiles holding EPW environments
static size_t rx6_lds(int rtc, int KI, int A, int epw) {
  const size_t rows = 16 * (size_t)rtc, IP = KI + 8;
  return 3 * rows * IP * 2 + 3 * 3 * rows * HP * 2 + (size_t)(A + 1) * H * 4 + 4 * rows * 4 /* avm */ + rows * (4 /* act */ + 4 * 4 /* uex */) +
         rows * 16 /* rmeta */ + (size_t)epw * (16 /* emeta */ + 12 * 4 /* P */) + 16 + (KI > 160 ? 12 * 1024 : 0) /* W2f */;
}
// row tiles a workgroup may hold: five / seven fc1 chunks (wide inputs) cost registers and LDS
static int rx6_max_tiles(int KI) { return KI > 160 ? 3 : KI > 96 ? 4 : 5; }

// environments per workgroup: one workgroup per CU while the batch fits one round (small batches spread over all CUs with partly
// filled tiles); beyond that as few FULL rounds of 256 workgroups as five row tiles per workgroup allow, evenly filled (0: none fits)
static int rx6_epw(int E, int N, int KI, int A) {
  int epw_max = 16 * rx6_max_tiles(KI) / N;
  while (epw_max > 1 && rx6_lds((epw_max * N + 15) / 16, KI, A, epw_max) > 160 * 1024) --epw_max;
  if (epw_max < 1) return 0;
  int epw = (E + 255) / 256;
  if (epw > epw_max) {
    const int rounds = (E + 256 * epw_max - 1) / (256 * epw_max);
    epw = (E + 256 * rounds - 1) / (256 * rounds);
    if (epw > epw_max) epw = epw_max;
  }
  return epw;
}
// which decomposition runs a batch: the round-5 kernel where it holds ONE row tile per workgroup (0.46 against 0.47 ms at 512 envs: a
// lock-step is a chain of latencies there, and its four short phases carry them better), this file's kernel from two tiles on (1024
// envs 0.67 / 0.68, 2048 envs 0.91 / 0.95, 4096 envs 1.44 / 1.90 ms).  experiments: rollout_v1 = 1 / 2 forces one of them
static bool rx6_use_v1(int E, int N, int O, int A, int last_action, int reuse_network) {
  const int sw = marl_switches()->rollout_v1;
  const int v1_epw_one_tile = 16 / (N > 0 ? N : 1);
  return marl_rollout_x6_v1_supported(N, O, A) && (sw == 1 || (sw == 0 && v1_epw_one_tile >= 1 && (long)E <= 256L * v1_epw_one_tile));
}

}  // namespace

ST_DEFINE_SETTER(marl_debug_stamps_rollout_x6)

// shapes the split rollout covers: H = 64, observation width a multiple of 4, input width <= 224 (three / five / seven fc1 chunks:
// 2s3z-, 3s5z- and MMM2-sized agents), <= 16 actions up to 160 input columns and <= 32 beyond (two action tiles of fc2 and of the
// choice), whole environments in at most five (wide inputs: four / three) row tiles; an environment's agents in one wave
extern "C" int marl_synth_rollout_x6_supported(int N, int O, int A) {
  if (A < 1 || N < 1 || O < 4 || (O & 3)) return 0;
  const int I = O + A + N;
  if (I > 224 || A > (I > 160 ? 32 : 16)) return 0;
  const int KI = (I + 31) / 32 * 32;
  const int mt = rx6_max_tiles(KI);
  if (N > 16 * mt || N > 64) return 0;                   // a whole environment in one workgroup; its agents on one wave (env step)
  return rx6_lds((N + 15) / 16, KI, A, 1) <= 160 * 1024 ? 1 : 0;
}

// how marl_synth_rollout_x6 would run a batch (what bench.py's roofline model counts products by): plan[0] = decomposition (1: round 5,
// rollout_x6_v1.hip; 2: round 6, this file), plan[1] = workgroups, plan[2] = row tiles of 16 (episode, agent) rows per workgroup,
// plan[3] = environments per workgroup, plan[4] = fc1 chunks of 32 input columns.  Returns 0, or hipErrorInvalidValue for unsupported shapes
extern "C" int marl_synth_rollout_x6_plan(int E, int N, int O, int A, int last_action, int reuse_network, int* plan) {
  if (!plan || E <= 0 || !marl_synth_rollout_x6_supported(N, O, A)) return (int)hipErrorInvalidValue;
  const int I = O + (last_action ? A : 0) + (reuse_network ? N : 0), KI = (I + 31) / 32 * 32;
  int epw;
  if (rx6_use_v1(E, N, O, A, last_action, reuse_network)) {
    const int epw_max = (KI > 96 ? 32 : 48) / N;
    epw = (E + 255) / 256;
    if (epw > epw_max) {
      const int rounds = (E + 256 * epw_max - 1) / (256 * epw_max);
      epw = (E + 256 * rounds - 1) / (256 * rounds);
      if (epw > epw_max) epw = epw_max;
    }
    plan[0] = 1;
  } else {
    epw = rx6_epw(E, N, KI, A);
    if (epw < 1) return (int)hipErrorInvalidValue;
    plan[0] = 2;
  }
  plan[1] = (E + epw - 1) / epw; plan[2] = (epw * N + 15) / 16; plan[3] = epw; plan[4] = KI > 160 ? 7 : KI > 96 ? 5 : 3;
  return 0;
}

extern "C" int marl_synth_rollout_x6(const marl_agent_weights_t* w, unsigned seed, unsigned rseed, int env0, int episode,
                                     int fixed_len, const float* eps, float* obs, float* state, long state_ld, float* avail, int* u,
                                     float* r, float* term, float* padded, int* length, int* won, float* h_out,
                                     float* stats, double eps0, double eps_anneal, double eps_min, int E, int T, int N,
                                     int O, int S, int A, int last_action, int reuse_network, void* stream) {
  const bool use_v1 = rx6_use_v1(E, N, O, A, last_action, reuse_network);
  if (use_v1)
    return marl_rollout_x6_v1(w, seed, rseed, env0, episode, fixed_len, eps, obs, state, state_ld, avail, u, r, term, padded, length, won,
                              h_out, stats, eps0, eps_anneal, eps_min, E, T, N, O, S, A, last_action, reuse_network, stream);
  if (E <= 0 || T <= 0) return 0;
  if (w->H != H || state_ld < S || !marl_synth_rollout_x6_supported(N, O, A)) return (int)hipErrorInvalidValue;
  if (reinterpret_cast<uintptr_t>(obs) & 15) return (int)hipErrorInvalidValue;
  RX6Args a;
  a.W1 = w->fc1_w; a.b1 = w->fc1_b; a.Wih = w->w_ih; a.Whh = w->w_hh; a.bih = w->b_ih; a.bhh = w->b_hh;
  a.W2 = w->fc2_w; a.b2 = w->fc2_b;
  a.eps = eps; a.eps0 = eps0; a.eps_anneal = eps_anneal; a.eps_min = eps_min; a.obs = obs; a.state = state; a.SL = state_ld; a.avail = avail;
  a.u = u; a.r = r; a.term = term; a.padded = padded; a.length = length; a.won = won; a.h_out = h_out; a.stats = stats;
  a.seed = seed; a.rseed = rseed; a.env0 = env0; a.episode = episode; a.fixed_len = fixed_len;
  a.E = E; a.T = T; a.N = N; a.O = O; a.S = S; a.A = A;
  a.has_act = last_action ? 1 : 0; a.has_id = reuse_network ? 1 : 0;
  a.I = O + (last_action ? A : 0) + (reuse_network ? N : 0);
  a.KI = (a.I + 31) / 32 * 32;
  a.R = (long)E * N;
  // record offsets are 32-bit element offsets inside the kernel
  if ((double)E * (T + 1) * N * (O > A ? O : A) >= 2147483648.0 || (double)E * (T + 1) * state_ld >= 2147483648.0)
    return (int)hipErrorInvalidValue;
  const int nk1 = a.KI > 160 ? 7 : a.KI > 96 ? 5 : 3;
  const int epw = rx6_epw(E, N, a.KI, A);
  if (epw < 1) return (int)hipErrorInvalidValue;
  a.EPW = epw;
  const int rtc = (epw * N + 15) / 16;
  const size_t lds = rx6_lds(rtc, a.KI, A, epw);
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  dim3 grid((unsigned)((E + epw - 1) / epw)), block(RNT);
  const void* fn = nullptr;
#define RX6_PICK(NK_) (rtc == 1 ? (const void*)synth_rollout_x6_kernel<1, NK_> : rtc == 2 ? (const void*)synth_rollout_x6_kernel<2, NK_> \
                       : rtc == 3 ? (const void*)synth_rollout_x6_kernel<3, NK_> : (const void*)synth_rollout_x6_kernel<4, NK_>)
  if (nk1 == 3) fn = rtc == 5 ? (const void*)synth_rollout_x6_kernel<5, 3> : RX6_PICK(3);
  else if (nk1 == 5) fn = RX6_PICK(5);
  else fn = rtc == 1 ? (const void*)synth_rollout_x6_kernel<1, 7, 2> : rtc == 2 ? (const void*)synth_rollout_x6_kernel<2, 7, 2> : (const void*)synth_rollout_x6_kernel<3, 7, 2>;
#undef RX6_PICK
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  void* kargs[] = {(void*)&a};
  e = hipLaunchKernel(fn, grid, block, kargs, lds, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  MARL_CHECK_LAUNCH();
  return 0;
}
