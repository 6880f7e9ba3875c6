// Whole-rollout persistent kernel for the synthetic SMAC-shaped environment: ONE launch plays all T
// lock-steps of E environments (reference rollout.py:60-101 + share_params.py:37-72, vectorised).
//
// A workgroup owns rows = 16*RT (episode, agent) rows, a multiple of N, i.e. whole environments, for
// the entire episode: agent weights stay in registers / LDS (they were re-staged 120x by the
// launch-per-step path), the hidden state never leaves LDS, and because the workgroup holds every
// agent of its environments the epsilon-greedy choice, the env step (reward / terminated / padding)
// and the next observation are computed in place - no inter-workgroup communication at all.
// The MFMA phases are those of agent_fwd_kernel (agent.hip); the environment is the counter-hash
// env of synth_hash.h, so the episode record is bit-identical to the launch-per-step path and to the
// numpy oracle.
#include "common.h"
#include "synth_hash.h"
#include "../../include/marl_hip.h"

namespace {

constexpr int H = 64;
constexpr int HS = H + 4;
constexpr int RNT = 512;

#define WG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// max over each aligned group of 16 lanes with DPP moves (no LDS crossbar round trips)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float group_max16(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));     // quad_perm [1,0,3,2]
  v = fmaxf(v, dpp_mov<0x4E>(v));     // quad_perm [2,3,0,1]
  v = fmaxf(v, dpp_mov<0x141>(v));    // row_half_mirror: the other quad of each 8 lanes
  v = fmaxf(v, dpp_mov<0x140>(v));    // row_mirror: the other half of the row
  return v;
}

struct RollArgs {
  const float *W1, *b1, *Wih, *Whh, *bih, *bhh, *W2, *b2;
  const float* eps;       // [T] epsilon of each lock-step (device), or null: the schedule below
  double eps0, eps_anneal, eps_min;   // eps(0) = eps0, eps(t+1) = eps(t) > eps_min ? eps(t) - eps_anneal : eps(t), evaluated in
                          // fp64 like the host loop of rollout.py:100-101 and rounded to fp32 per step
  float* stats;           // [3][E] or null: per episode  sum_t r (fp32, in step order) | won | length  (the host's rollout statistics)
  float *obs, *state, *avail;   // (E,T+1,N,O) (E,T+1,SL >= S) (E,T+1,N,A)
  long SL;                // row stride of the state storage (a multiple of 4 gives 16-byte state rows for any S)
  int* u;                 // (E,T,N)
  float *r, *term, *padded;     // (E,T)
  int *length, *won;      // (E)
  float* h_out;           // (E*N,64) final hidden state or null
  unsigned seed, rseed;
  int env0, episode, fixed_len;
  int E, T, N, O, S, A, I, KC, RT;
  int EPW;                // whole environments per workgroup: EPW*N valid rows in 16*RT tile rows
  int has_act, has_id;
  long R;
};

template <int AC>
__global__ __launch_bounds__(RNT, 2) void synth_rollout_kernel(RollArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int team = wave >> 2, ws = wave & 3;
  const int q = lane >> 4, m = lane & 15;
  const int rows = a.RT * 16;
  const int KP = a.KC * 16, KS = KP + 4;
  const int AS = a.A + 1;
  float* W1s = smem;                                  // [4][KC][64] f32x4
  float* In = W1s + 4 * a.KC * 64 * 4;                // [rows][KS]
  float* Xt = In + rows * KS;                         // [rows][HS]
  float* Ha = Xt + rows * HS;                         // [rows][HS] x2
  float* Hb = Ha + rows * HS;
  float* Qs = Hb + rows * HS;                         // [rows][AS]  q of the current step
  float* Av0 = Qs + rows * AS;                        // [rows][A] x2  availability of step t / t+1
  float* Av1 = Av0 + rows * a.A;
  int* act = reinterpret_cast<int*>(Av1 + rows * a.A);   // [rows]
  int* rowe = act + rows;                                // [rows]: local env index (b - b0), clamped
  int* rown = rowe + rows;                               // [rows]: n
  int* elen = rown + rows;                               // [rows/N]: episode length of the local env
  unsigned* pfx = reinterpret_cast<unsigned*>(elen + rows);   // [3][rows]: hash prefixes (obs, avail per row; state per env) of the slot being generated
  int4* rmeta = reinterpret_cast<int4*>(pfx + 3 * rows);      // [rows]: {obs offset of (b,0,n,0), avail offset, episode length, n}
  int4* emeta = rmeta + rows;                                 // [rows/N..]: {state offset of (b,0,0), episode length, env in range, -}
  float* uex = reinterpret_cast<float*>(emeta + rows);        // [2][rows]: explore / pick uniforms of the coming choice
  int* tilecnt = reinterpret_cast<int*>(uex + 2 * rows);      // [4]: next GRU row tile of each hidden-unit slice

  const int T = a.T, N = a.N, O = a.O, S = a.S, A = a.A;
  const int nenv_wg = a.EPW;
  const int vrows = nenv_wg * N;                       // valid rows; rows vrows..16*RT-1 are padding (zero input)
  const int b0 = blockIdx.x * nenv_wg;
  const long row0 = (long)b0 * N;
  for (int r = tid; r < rows; r += RNT) {
    long rho = row0 + (r < vrows ? r : vrows - 1);
    if (rho > a.R - 1) rho = a.R - 1;                   // clamp: duplicates of the last row (same values, same addresses)
    rowe[r] = (int)(rho / N) - b0;
    rown[r] = (int)(rho % N);
  }
  // env step: lane -> (env, agent); the N rows of an environment sit in ONE wave (64 / N environments per wave)
  const int es_epw = 64 / N, es_er = lane / N;
  const int es_n = lane - es_er * N, es_l0 = es_er * N;
  const int es_el = ((wave + 4) & 7) * es_epw + es_er;     // team 1's waves first: team 0 goes straight on to fc1 of the next step
  const bool es_has = es_er < es_epw && es_el < nenv_wg && b0 + es_el < a.E;
  const int lmin = T / 2 > 1 ? T / 2 : 1;
  for (int e = tid; e < nenv_wg; e += RNT) {
    int b = b0 + e; if (b > a.E - 1) b = a.E - 1;
    const unsigned env = (unsigned)(a.env0 + b);
    int L = lmin + (int)(hkey(a.seed, ST_LEN, env, (unsigned)a.episode, 0u) % (unsigned)(T - lmin + 1));
    if (a.fixed_len) L = T;
    elen[e] = L;
    a.length[b] = L;
    const int won_ = (int)(hkey(a.seed, ST_WON, env, (unsigned)a.episode, 0u) & 1u);
    a.won[b] = won_;
    if (a.stats && b0 + e < a.E) { a.stats[a.E + b] = (float)won_; a.stats[2L * a.E + b] = (float)L; }
  }
  for (int e = tid; e < rows * H; e += RNT) Ha[(e / H) * HS + (e % H)] = 0.f;     // init_hidden: zeros
  __syncthreads();
  for (int r = tid; r < vrows; r += RNT) {
    const int el = rowe[r], n = rown[r];
    const int bn = (b0 + el) * (T + 1) * N + n;
    rmeta[r] = make_int4(bn * O, bn * A, elen[el], n);
  }
  for (int el = tid; el < nenv_wg; el += RNT)
    emeta[el] = make_int4((b0 + el) * (T + 1) * (int)a.SL, elen[el], b0 + el < a.E ? 1 : 0, 0);

  // ---- environment observation of slot t -> record (+ LDS input tile / availability when wanted)
  // prefixes of slot t (3 of the 4 hash rounds depend only on (stream, env, t)): one thread per row
  auto gen_prefix = [&](int t) {
    if (tid < rows) {
      const unsigned tg = (unsigned)(a.episode * (T + 1) + t);
      const unsigned env = (unsigned)(a.env0 + b0 + rowe[tid]);
      pfx[tid] = hprefix(a.seed, ST_OBS, env, tg);
      pfx[rows + tid] = hprefix(a.seed, ST_AVAIL, env, tg);
      if (tid < nenv_wg) pfx[2 * rows + tid] = hprefix(a.seed, ST_STATE, (unsigned)(a.env0 + b0 + tid), tg);
    }
  };
  // Flattened over the slot's elements: every thread handles independent (row, column group) items - the row
  // lookup is one 16-byte LDS read of the metadata table, so nothing serialises on per-row dependent chains and
  // the stores are 16 bytes per lane (the row-per-wave form of this took 30-55 % of a lock-step).
  const int O4 = O >> 2, S4 = (S + 3) >> 2;           // the last state group may run into the row padding (zeros)
  const float invO4 = 1.0f / (float)(O4 > 0 ? O4 : 1), invO = 1.0f / (float)O, invA = 1.0f / (float)A;
  const float invS4 = 1.0f / (float)(S4 > 0 ? S4 : 1), invS = 1.0f / (float)S;
  const bool ovec = (O & 3) == 0, svec = (a.SL & 3) == 0 && a.SL >= 4 * S4 && (reinterpret_cast<uintptr_t>(a.state) & 15) == 0;
  // Branch-free: every element is hashed and then masked (bitwise AND with all-ones / zero keeps +0.0 exactly) - with
  // `live ? hash : 0` hipcc emits a branch per ELEMENT, each re-reading its row metadata from LDS and waiting for it
  // (four basic blocks with three waits each per float4; the generation then takes as long as the gate MFMAs beside it).
  auto bits = [](float v) { return __builtin_bit_cast(unsigned, v); };
  auto gen_slot = [&](int t, bool to_lds, float* Av, int first, int nthr) {
    if (tid < first || tid >= first + nthr) return;
    const int tl = tid - first;
    const int tNO = t * N * O, tNA = t * N * A, tS = t * (int)a.SL;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    if (ovec) {
      for (int e = tl; e < vrows * O4; e += nthr) {
        const int r = (int)(((float)e + 0.5f) * invO4);
        const int k = 4 * (e - r * O4);
        const int4 mt = rmeta[r];
        const unsigned po = pfx[r];
        const unsigned lm = t <= mt.z ? 0xffffffffu : 0u, fm = t < mt.z ? 0xffffffffu : 0u;
        const unsigned idx = (unsigned)(mt.w * O + k);
        u32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = bits(2.0f * u01(hfin(po, idx + (unsigned)i)) - 1.0f) & lm;
        *reinterpret_cast<u32x4*>(a.obs + (long)mt.x + tNO + k) = v;
        if (to_lds) {                              // padded steps feed zeros (rollout.py:122-133)
          u32x4 w;
#pragma unroll
          for (int i = 0; i < 4; ++i) w[i] = v[i] & fm;
          *reinterpret_cast<u32x4*>(In + r * KS + k) = w;
        }
      }
    } else {
      for (int e = tl; e < vrows * O; e += nthr) {
        const int r = (int)(((float)e + 0.5f) * invO);
        const int k = e - r * O;
        const int4 mt = rmeta[r];
        const unsigned lm = t <= mt.z ? 0xffffffffu : 0u, fm = t < mt.z ? 0xffffffffu : 0u;
        const unsigned v = bits(2.0f * u01(hfin(pfx[r], (unsigned)(mt.w * O + k))) - 1.0f) & lm;
        reinterpret_cast<unsigned*>(a.obs)[(long)mt.x + tNO + k] = v;
        if (to_lds) reinterpret_cast<unsigned*>(In)[r * KS + k] = v & fm;
      }
    }
    for (int e = tl; e < vrows * A; e += nthr) {
      const int r = (int)(((float)e + 0.5f) * invA);
      const int k = e - r * A;
      const int4 mt = rmeta[r];
      const float uu = u01(hfin(pfx[rows + r], (unsigned)(mt.w * A + k)));
      const bool on = (t <= mt.z) & ((k == 0) | (uu < 0.7f));
      const float v = on ? 1.f : 0.f;
      a.avail[(long)mt.y + tNA + k] = v;
      if (Av) Av[r * A + k] = v;
    }
    if (svec) {
      for (int e = tl; e < nenv_wg * S4; e += nthr) {
        const int el = (int)(((float)e + 0.5f) * invS4);
        const int k = 4 * (e - el * S4);
        const int4 mt = emeta[el];
        const unsigned ps = pfx[2 * rows + el];
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
        const unsigned v = bits(2.0f * u01(hfin(pfx[2 * rows + el], (unsigned)k)) - 1.0f) & lm;
        if (mt.z) reinterpret_cast<unsigned*>(a.state)[(long)mt.x + tS + k] = v;
      }
    }
  };
  // constant columns of the input tile: one-hot(last action) starts empty, agent id, zero pad
  for (int e = tid; e < rows * (KP - O); e += RNT) {
    const int r = e / (KP - O), k = O + e % (KP - O);
    float v = 0.f;
    if (a.has_id && k >= a.I - N && k < a.I) v = (rown[r] == k - (a.I - N)) ? 1.f : 0.f;
    In[r * KS + k] = v;
  }
  for (int e = tid; e < (rows - vrows) * O; e += RNT) In[(vrows + e / O) * KS + e % O] = 0.f;   // padding rows
  gen_prefix(0);
  __syncthreads();
  gen_slot(0, true, Av0, 0, RNT);
  __syncthreads();
  gen_prefix(1);          // consumed by gen_slot(1) after the first barrier of step 0
  if (tid < rows) {       // uniforms of the first choice; tile counters
    const unsigned env = (unsigned)(a.env0 + b0 + rowe[tid]), tg0 = (unsigned)(a.episode * (T + 1));
    uex[tid] = u01(hkey(a.rseed, ST_EXPLORE, env, tg0, (unsigned)rown[tid]));
    uex[rows + tid] = u01(hkey(a.rseed, ST_PICK, env, tg0, (unsigned)rown[tid]));
    if (tid < 4) tilecnt[tid] = 0;
  }

  // ---- weights (as in agent_fwd_kernel)
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
      int arow = 16 * ac + m; if (arow >= A) arow = A - 1;
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
  }
  WG_BARRIER();

  float* Hp = Ha;
  float* Hn = Hb;
  float* AvC = Av0;
  float* AvN = Av1;
  ST_DECL(10);
  float ep_r = 0.f;                                // episode reward of this lane's environment (lanes es_n == 0)
  float eps_next = a.eps ? a.eps[0] : (float)a.eps0;
  double eps_d = a.eps0;
  for (int t = 0; t < T; ++t) {
    const float eps = eps_next;                    // scalar load issued a step ahead
    if (a.eps) { if (t + 1 < T) eps_next = a.eps[t + 1]; }
    else { eps_d = eps_d > a.eps_min ? eps_d - a.eps_anneal : eps_d; eps_next = (float)eps_d; }
    // ---------------- phase 1: x = relu(fc1(in))
    for (int rt = team; rt < a.RT; rt += 4) {
      const bool two = rt + 2 < a.RT;
      f32x4 acc0 = {bias1, bias1, bias1, bias1}, acc1 = acc0;
      const float* in0 = In + (rt * 16 + m) * KS + 4 * q;
      const float* in1 = in0 + 32 * KS;
      const float* wf = W1s + (ws * a.KC * 64 + lane) * 4;
      // the k-chunk loop has a runtime trip count: two named operand sets, the loads of chunk c+1 issued before the
      // multiplies of chunk c (as written before - load, wait, four dependent MFMAs - every chunk paid an LDS round trip)
#define FC1_LD(B_, A0_, A1_, c_)                                                  \
      B_ = *reinterpret_cast<const f32x4*>(wf + (c_) * 256);                      \
      A0_ = *reinterpret_cast<const f32x4*>(in0 + 16 * (c_));                     \
      if (two) A1_ = *reinterpret_cast<const f32x4*>(in1 + 16 * (c_));
      f32x4 bA, aA, a1A = acc0, bB, aB, a1B = acc0;
      FC1_LD(bA, aA, a1A, 0)
      for (int c = 0; c < a.KC; c += 2) {
        const bool odd = c + 1 < a.KC;
        const int cB = odd ? c + 1 : c, cA = c + 2 < a.KC ? c + 2 : c;
        FC1_LD(bB, aB, a1B, cB)
        acc0 = mfma16x4(aA, bA, acc0);
        if (two) acc1 = mfma16x4(a1A, bA, acc1);
        FC1_LD(bA, aA, a1A, cA)
        if (odd) {
          acc0 = mfma16x4(aB, bB, acc0);
          if (two) acc1 = mfma16x4(a1B, bB, acc1);
        }
      }
#undef FC1_LD
      const int r0 = rt * 16 + 4 * q;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        Xt[(r0 + i) * HS + j] = fmaxf(acc0[i], 0.f);
        if (two) Xt[(r0 + 32 + i) * HS + j] = fmaxf(acc1[i], 0.f);
      }
    }
    ST_MARK(0);
    WG_BARRIER();
    ST_MARK(1);
    // the input tile is consumed: the env already knows the next observation (it does not depend on the
    // actions), so slot t+1 is generated now, under the shadow of the gate MFMAs of the other waves
    // The two waves of a SIMD (same hidden-unit slice, one per team) share that slice's row tiles through a
    // counter: team 1 first generates the whole slot (VALU + stores, overlapping team 0's MFMAs), then joins.
    // (waves 4-7 are the younger half and lose the VALU arbitration against their partners' MFMA streams: raise
    // their priority while they generate)
    if (team == 1) {
      __builtin_amdgcn_s_setprio(2);
      gen_slot(t + 1, t + 1 < T, AvN, RNT / 2, RNT / 2);
      __builtin_amdgcn_s_setprio(0);
    }
    ST_MARK(2);
    // ---------------- phase 2: GRU
    auto grab = [&]() {
      int v = 0;
      if (lane == 0) v = atomicAdd(&tilecnt[ws], 1);
      return __builtin_amdgcn_readfirstlane(v);
    };
    int rt_next = grab();
    while (rt_next < a.RT) {
      const int rt = rt_next;
      rt_next = grab();
      f32x4 ar = {bias_r, bias_r, bias_r, bias_r};
      f32x4 az = {bias_z, bias_z, bias_z, bias_z};
      f32x4 ain = {bias_in, bias_in, bias_in, bias_in};
      f32x4 ahn = {bias_hn, bias_hn, bias_hn, bias_hn};
      const float* xr = Xt + (rt * 16 + m) * HS + 4 * q;
      const float* hr = Hp + (rt * 16 + m) * HS + 4 * q;
      // input-side products first, hidden-side products after them (the SAME order in every GRU kernel: the input-side
      // sums bias + x W_ih of one unroll can then stand in for another unroll's - see the GI variants of agent_fwd_kernel)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        f32x4 ax = *reinterpret_cast<const f32x4*>(xr + 16 * c);
        ar = mfma16x4(ax, wih[0][c], ar);
        az = mfma16x4(ax, wih[1][c], az);
        ain = mfma16x4(ax, wih[2][c], ain);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        f32x4 ah = *reinterpret_cast<const f32x4*>(hr + 16 * c);
        ahn = mfma16x4(ah, whh[2][c], ahn);
        ar = mfma16x4(ah, whh[0][c], ar);
        az = mfma16x4(ah, whh[1][c], az);
      }
      const int r0 = rt * 16 + 4 * q;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float hp = Hp[(r0 + i) * HS + j];
        const float rg = sigmoidf_(ar[i]);
        const float zg = sigmoidf_(az[i]);
        const float ng = tanhf_(ain[i] + rg * ahn[i]);
        Hn[(r0 + i) * HS + j] = (1.f - zg) * ng + zg * hp;
      }
    }
    ST_MARK(3);
    WG_BARRIER();
    ST_MARK(4);
    // ---------------- phase 3: q = fc2(h') -> LDS; the two last waves (no fc2 tile unless RT > 6) hash the prefixes
    // of slot t+2, whose observation is generated during the gates of step t+1
    if (tid >= RNT - 128 && tid - (RNT - 128) < rows) {
      const int r = tid - (RNT - 128);
      const unsigned tg2 = (unsigned)(a.episode * (T + 1) + t) + 2u;
      const unsigned env = (unsigned)(a.env0 + b0 + rowe[r]);
      pfx[r] = hprefix(a.seed, ST_OBS, env, tg2);
      pfx[rows + r] = hprefix(a.seed, ST_AVAIL, env, tg2);
      if (r < nenv_wg) pfx[2 * rows + r] = hprefix(a.seed, ST_STATE, (unsigned)(a.env0 + b0 + r), tg2);
      if (r < 4) tilecnt[r] = 0;                  // all tiles of this step were grabbed before the barrier above
    }
    for (int rt = wave; rt < a.RT; rt += 8) {
      f32x4 acc[AC];
#pragma unroll
      for (int ac = 0; ac < AC; ++ac) acc[ac] = (f32x4){bias2[ac], bias2[ac], bias2[ac], bias2[ac]};
      const float* hr = Hn + (rt * 16 + m) * HS + 4 * q;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        f32x4 ah = *reinterpret_cast<const f32x4*>(hr + 16 * c);
#pragma unroll
        for (int ac = 0; ac < AC; ++ac) acc[ac] = mfma16x4(ah, w2[ac][c], acc[ac]);
      }
#pragma unroll
      for (int ac = 0; ac < AC; ++ac) {
        const int col = 16 * ac + m;
        if (col < A) {
#pragma unroll
          for (int i = 0; i < 4; ++i) Qs[(rt * 16 + 4 * q + i) * AS + col] = acc[ac][i];
        }
      }
    }
    ST_MARK(5);
    WG_BARRIER();
    ST_MARK(6);
    // ---------------- epsilon-greedy choice (share_params.py:66-70): 16*AC lanes per (env, agent) row, lane = action;
    // first-index argmax over the available actions by a lane-group max + ballot, the explored action is the
    // kk-th set bit of the availability mask (the serial per-thread form of this took 13-35 % of a lock-step)
    const unsigned tg = (unsigned)(a.episode * (T + 1) + t);
    {
      constexpr int LG = 16 * AC;
      constexpr unsigned GM = LG == 32 ? 0xffffffffu : ((1u << LG) - 1u);
      const int gl = lane & (LG - 1), sh = lane & ~(LG - 1);
      for (int base = wave * (64 / LG); base < vrows; base += (RNT / 64) * (64 / LG)) {
        const int r = base + lane / LG;
        const bool valid = r < vrows;
        const int rr = valid ? r : vrows - 1;
        const int4 mt = rmeta[rr];
        const int n = mt.w;
        const bool on = gl < A && AvC[rr * A + (gl < A ? gl : 0)] != 0.f;
        const float v = on ? Qs[rr * AS + gl] : -3.0e38f;
        float mx = group_max16(v);
        if (LG == 32) mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        const unsigned am = (unsigned)(__ballot(on) >> sh) & GM;
        const unsigned em = (unsigned)(__ballot(on && v == mx) >> sh) & GM;
        const int navail = __popc(am);
        int arg = em ? __ffs(em) - 1 : (am ? __ffs(am) - 1 : 0);
        const bool explore = uex[rr] < eps;
        int kk = (int)floorf(uex[rows + rr] * (float)navail);
        if (kk > navail - 1) kk = navail - 1;
        const bool sel = explore && on && __popc(am & ((1u << gl) - 1u)) == kk;
        const unsigned sm = (unsigned)(__ballot(sel) >> sh) & GM;
        if (sm) arg = __ffs(sm) - 1;
        if (!(t < mt.z)) arg = -1;
        if (valid) {
          if (gl == 0) {
            act[r] = arg;
            a.u[((long)(b0 + rowe[rr]) * T + t) * N + n] = arg;
          }
          if (a.has_act && gl < A) In[r * KS + O + gl] = (gl == arg) ? 1.f : 0.f;      // one-hot fed to step t+1
        }
      }
    }
    ST_MARK(7);
    WG_BARRIER();
    ST_MARK(8);
    // ---------------- env step: reward / terminated / padded (fixed-order fp32 sum over agents); the two last
    // waves hash the uniforms of the NEXT step's epsilon-greedy choice (they do not depend on q)
    if (tid >= RNT - 128 && tid - (RNT - 128) < rows) {
      const int r = tid - (RNT - 128);
      const unsigned env = (unsigned)(a.env0 + b0 + rowe[r]), nn_ = (unsigned)rown[r];
      uex[r] = u01(hkey(a.rseed, ST_EXPLORE, env, tg + 1u, nn_));
      uex[rows + r] = u01(hkey(a.rseed, ST_PICK, env, tg + 1u, nn_));
    }
    // one lane per (env, agent) hashes its reward term; the env's lane sums them in agent order (the serial form -
    // one thread per env looping over the agents' hashes - was 4-12 % of a lock-step on the critical path)
    if (es_has) {
      const int L = elen[es_el];
      const bool live = t < L;
      float term = 0.f;
      if (live) {
        const unsigned pre = hprefix(a.seed, ST_REWARD, (unsigned)(a.env0 + b0 + es_el), tg);
        term = u01(hfin(pre, (unsigned)(es_n * A + act[es_el * N + es_n]))) - 0.5f;
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
    float* tmp = Hp; Hp = Hn; Hn = tmp;
    tmp = AvC; AvC = AvN; AvN = tmp;
    ST_MARK(9);
    // no barrier: `act` is next written after three more barriers; In/Xt/H hazards as in agent_fwd_kernel
  }
  ST_DUMP(10);
  if (a.stats && es_has && es_n == 0) a.stats[b0 + es_el] = ep_r;
  if (a.h_out) {
    WG_BARRIER();
    for (int e = tid; e < rows * H; e += RNT) {
      const int r = e / H, k = e % H;
      const long rho = row0 + r;
      if (r < vrows && rho < a.R) a.h_out[rho * H + k] = Hp[r * HS + k];
    }
  }
}

}  // namespace

ST_DEFINE_SETTER(marl_debug_stamps_rollout)

// a workgroup holds whole environments (EPW*N rows padded to 16*RT, RT <= 8); the fc1 slice + the row state
// must fit the 160 KB LDS (widest input assumed: last action and agent id appended)
static int max_rt(int I, int A) {
  const int KC = (I + 15) / 16, KS = KC * 16 + 4;
  const size_t fixed = (size_t)4 * KC * 64 * 16 + 16;
  const size_t per_row = (size_t)(KS + 3 * HS + (A + 1) + 2 * A) * 4 + 16 + 12 + 32 + 8;
  int rt = 0;
  for (int c = 1; c <= 8; ++c) if (fixed + per_row * 16 * c <= 160 * 1024) rt = c;
  return rt;
}

extern "C" int marl_synth_rollout_supported(int N, int O, int A) {
  if (A > 32 || A < 1 || N < 1) return 0;
  return 16 * max_rt(O + A + N, A) >= N ? 1 : 0;
}

extern "C" int marl_synth_rollout(const marl_agent_weights_t* w, unsigned seed, unsigned rseed, int env0, int episode,
                                  int fixed_len, const float* eps, float* obs, float* state, long state_ld, float* avail, int* u,
                                  float* r, float* term, float* padded, int* length, int* won, float* h_out,
                                  float* stats, double eps0, double eps_anneal, double eps_min, int E, int T, int N,
                                  int O, int S, int A, int last_action, int reuse_network, void* stream) {
  if (E <= 0 || T <= 0) return 0;
  if (w->H != H || A > 32 || A < 1 || state_ld < S) return (int)hipErrorInvalidValue;
  RollArgs a;
  a.W1 = w->fc1_w; a.b1 = w->fc1_b; a.Wih = w->w_ih; a.Whh = w->w_hh; a.bih = w->b_ih; a.bhh = w->b_hh;
  a.W2 = w->fc2_w; a.b2 = w->fc2_b;
  a.eps = eps; a.eps0 = eps0; a.eps_anneal = eps_anneal; a.eps_min = eps_min; a.obs = obs; a.state = state; a.SL = state_ld; a.avail = avail; a.u = u; a.r = r; a.term = term; a.padded = padded;
  a.length = length; a.won = won; a.h_out = h_out; a.stats = stats;
  a.seed = seed; a.rseed = rseed; a.env0 = env0; a.episode = episode; a.fixed_len = fixed_len;
  a.E = E; a.T = T; a.N = N; a.O = O; a.S = S; a.A = A;
  a.has_act = last_action ? 1 : 0; a.has_id = reuse_network ? 1 : 0;
  a.I = O + (last_action ? A : 0) + (reuse_network ? N : 0);
  a.KC = (a.I + 15) / 16;
  a.R = (long)E * N;
  const int KS = a.KC * 16 + 4;
  const size_t fixed = (size_t)4 * a.KC * 64 * 16 + 16;
  const size_t per_row = (size_t)(KS + 3 * HS + (A + 1) + 2 * A) * 4 + 16 + 12 + 32 + 8;
  // environments per workgroup: one workgroup per CU when the batch allows it (a lock-step is latency
  // bound, so small batches spread over all CUs with partly filled tiles), capped by the LDS budget
  // record offsets are 32-bit element offsets inside the kernel
  if ((double)E * (T + 1) * N * (O > A ? O : A) >= 2147483648.0 || (double)E * (T + 1) * state_ld >= 2147483648.0)
    return (int)hipErrorInvalidValue;
  const int rt_max = max_rt(a.I, A);
  if (16 * rt_max < N) return (int)hipErrorInvalidValue;
  int epw = (E + 255) / 256;
  if (epw * N > 16 * rt_max) epw = (16 * rt_max) / N;
  a.EPW = epw;
  a.RT = (epw * N + 15) / 16;
  const size_t lds = fixed + per_row * a.RT * 16;
  dim3 grid((unsigned)((E + epw - 1) / epw)), block(RNT);
  hipStream_t s = (hipStream_t)stream;
  hipError_t e;
  if (A <= 16) {
    e = hipFuncSetAttribute((const void*)synth_rollout_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((synth_rollout_kernel<1>), grid, block, lds, s, a);
  } else {
    e = hipFuncSetAttribute((const void*)synth_rollout_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((synth_rollout_kernel<2>), grid, block, lds, s, a);
  }
  MARL_CHECK_LAUNCH();
  return 0;
}
