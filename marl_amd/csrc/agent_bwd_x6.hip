// Backward through time of the GRU-agent unroll with every fp32 product as six bf16 MFMA products (x6.h) - opt-in
// args.gemm_mode = "bf16x6"; the default is agent_bwd_kernel / agent_bwd_pipe_kernel (agent.hip) on v_mfma_f32_16x16x4_f32.
// Autograd of controller/share_params.py:125-146 + network/q_network.py:16-21 (reference q_learner.py:171 loss.backward()):
// per step the delta pass (dh carry, gate gradients, dx) AND the weight-gradient reductions of W_ih, W_hh, W_2 and their biases.
//
// One workgroup = 32 rows (two 16-row tiles of the saved-activation layout) = one k block of v_mfma_f32_16x16x32_bf16 for the
// reductions over rows (NT = 2; batches of up to 256 row tiles: one tile per workgroup, NT = 1, reductions on the 16-deep MFMA); 8 waves = 2 teams x 4 hidden-unit slices (wave s of a team: units 16s .. 16s+15), two barriers per step
// (image filled | image read):
//   team R (waves 0-3), the dependent chain:  dh = carry + dhs + dq W_2 -> gate gradients (fp32, accumulator layout: lane (q, m) =
//          rows 4q + r of unit 16s + m, the layout of the saved planes) -> G image in LDS | barrier |
//          carry' = dh z + [drp|dzp|dhn] W_hh  and  dx = [drp|dzp|dnp] W_ih -> relu gate -> dxp (the contractions over gate columns:
//          both sets of weight fragments live in this team's registers);  bias sums (exact fp32)
//   team I (waves 4-7), off the chain, no weights, 100 accumulator registers:  dW_ih += [drp|dzp|dnp]^T x ;  dW_hh += [drp|dzp|dhn]^T h_prev ;
//          dW_2 += dq^T h ;  it also loads x / h_prev, splits them once into ready operand fragments and publishes relu'(x)
// The gate-gradient image G^T[gate column 0..255 = drp|dzp|dnp|dhn][row 0..31] holds three bf16 planes (each element split once,
// where it is produced): a product that CONTRACTS over gate columns (dx, carry') reads it transposed (ds_read_b64_tr_b16: lane i =
// row), a product with gate columns as OUTPUT rows (the weight gradients, contraction over the 32 rows) reads it plainly - the k
// slots of a lane are rows 4g .. 4g+3 of tile 0 then of tile 1, which is exactly how a lane of the saved tile layout holds h_prev
// and x: those operands are split8(tile 0 value, tile 1 value) of what the owning wave loaded anyway, passed on as ready fragments.
// Image addressing: byte(plane, column k, piece p of 8 bytes = 4 rows) = plane * 16384 + k * 64 + 8 * (p ^ 2 ((k >> 2) & 3)),
// p = 4 tile + (row >> 2) & 3: writes (lane (q, m): column m, piece q), transposed reads (lane (g; qq, p): column 8g + .. + qq)
// and plain reads (lane (g, i): column i, piece g) all spread a 32-lane pass over the 64 banks.
#include "x6.h"
#include <cstdlib>
#include "../../include/marl_hip.h"

#ifndef DWHH_HEAD
#define DWHH_HEAD -1      // A/B builds: chunks of dW_hh before the second barrier (-1: by tile count)
#endif
namespace {

constexpr int H = 64;
constexpr int BNT = 512;
constexpr int GPL = 256 * 64;       // bytes per plane of the gate-gradient image
constexpr int GBUF = 3 * GPL;       // one image (three planes)
constexpr int FRB = 4 * 3 * 1024;   // ready-made operand fragments of the four column tiles of h_prev (or x): [tile][plane][lane] 16 bytes

struct BX6Args {
  const float *Wih, *Whh, *W2;
  const int* dq_idx; const float* dq_val;       // (B,T,N): row (b,t,n) has the non-zero dq_val[row / gdiv] in column dq_idx
  const int* dq_idx2; const float* dq_val2;     // optional second pair per row
  int dq_gdiv;
  const float* dhs;        // (B,T,N,64) external gradient on hs[t], or null
  const float* saved;      // [T+1][row tile][6][4][64][4]
  float* dxp;              // (B,T,N,64)
  float* dh0;              // (B*N,64) or null
  float* ws;               // [n_wg][slab]
  int B, T, N, A;
  long R;
  long tile_base;          // first row tile of this launch (a batch may run as two launches: full rounds of two-tile workgroups, then
  int slab_base;           // one round of one-tile workgroups) and its first slab
};

__host__ __device__ inline long slab_floats(int A) { return 2L * 192 * 64 + (long)A * 64 + 2 * 192 + A; }

#define WG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define X6_TERMS(OP) OP(m, m) OP(h, l) OP(l, h) OP(h, m) OP(m, h) OP(h, h)
// two split multiplies (A0 B0, A1 B1) on four accumulators: the small products on c0 / c1, the large ones on c2 / c3
#define X6_TERMS4(A0, B0, A1, B1, c0, c1, c2, c3)                                                                   \
  c0 = mm(A0.m, B0.m, c0); c1 = mm(A1.m, B1.m, c1); c2 = mm(A0.h, B0.m, c2); c3 = mm(A1.h, B1.m, c3);               \
  c0 = mm(A0.h, B0.l, c0); c1 = mm(A1.h, B1.l, c1); c2 = mm(A0.m, B0.h, c2); c3 = mm(A1.m, B1.h, c3);               \
  c0 = mm(A0.l, B0.h, c0); c1 = mm(A1.l, B1.h, c1); c2 = mm(A0.h, B0.h, c2); c3 = mm(A1.h, B1.h, c3);

__device__ __forceinline__ long sv_off(long tile_t, int plane, int c, int lane) { return ((tile_t * 6 + plane) * 4 + c) * 256 + lane * 4; }
__device__ __forceinline__ f32x4 splat(float v) { return (f32x4){v, v, v, v}; }

// B fragment of W^T for  out[row][unit] = sum_k G[row][k] W[k][unit]:  lane (g, j): W[32 c + 8g + (0..7)][u0 + j]  (W: [192][64])
__device__ __forceinline__ F3 wTfrag(const float* W, int c, int u0, int lane) {
  const int j = lane & 15, g = lane >> 4;
  float v[8];
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) v[jj] = W[(long)(32 * c + 8 * g + jj) * H + u0 + j];
  return split8((f32x4){v[0], v[1], v[2], v[3]}, (f32x4){v[4], v[5], v[6], v[7]});
}
// A fragment (lane i = row of tile `tile`, k slots = gate columns k0 + 8g + (0..7)) from the image at `img`: transposed reads
// tb0 / tb1: the lane's bases for the column quads 8g + (0..3) / 8g + (4..7)
__device__ __forceinline__ i32x4 g_tr(const char* img, int tb0, int tb1, int off, int tile) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + off + (tb0 ^ (32 * tile))));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + off + (tb1 ^ (32 * tile))));
  return __builtin_bit_cast(i32x4, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ F3 g_tr3(const char* img, int tb0, int tb1, int k0, int tile) {
  F3 f;
  f.h = g_tr(img, tb0, tb1, k0 * 64, tile); f.m = g_tr(img, tb0, tb1, k0 * 64 + GPL, tile); f.l = g_tr(img, tb0, tb1, k0 * 64 + 2 * GPL, tile);
  return f;
}
// A fragment with gate columns as the OUTPUT rows (lane i = column k0 + i, k slots = rows 4g .. 4g+3 of tile 0, then of tile 1)
__device__ __forceinline__ F3 g_cols(const char* img, int rb, int k0) {
  F3 f;
  const char* p = img + k0 * 64;
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    const i32x2 t0 = *reinterpret_cast<const i32x2*>(p + pl * GPL + rb);
    const i32x2 t1 = *reinterpret_cast<const i32x2*>(p + pl * GPL + (rb ^ 32));
    const i32x4 v = (i32x4){t0[0], t0[1], t1[0], t1[1]};
    if (pl == 0) f.h = v; else if (pl == 1) f.m = v; else f.l = v;
  }
  return f;
}
// the four rows 4q .. 4q+3 (tile `tile`) of column k0 + m -> the image
__device__ __forceinline__ void g_put(char* img, int wb, int k0, int tile, const F3h& f) {
  char* p = img + k0 * 64 + (wb ^ (32 * tile));
  *reinterpret_cast<i32x2*>(p) = f.h;
  *reinterpret_cast<i32x2*>(p + GPL) = f.m;
  *reinterpret_cast<i32x2*>(p + 2 * GPL) = f.l;
}
__device__ __forceinline__ void fr_put(char* fr, int tile, int lane, const F3& f) {
  i32x4* p = reinterpret_cast<i32x4*>(fr) + tile * 192 + lane;
  p[0] = f.h; p[64] = f.m; p[128] = f.l;
}
__device__ __forceinline__ F3 fr_get(const char* fr, int tile, int lane) {
  const i32x4* p = reinterpret_cast<const i32x4*>(fr) + tile * 192 + lane;
  F3 f;
  f.h = p[0]; f.m = p[64]; f.l = p[128];
  return f;
}

// one row tile per workgroup (NT = 1, the small shards): the reductions over rows run on v_mfma_f32_16x16x16_bf16 - k slots = rows
// 4g .. 4g+3 of the one tile, fragments of 8 bytes per lane and plane
template <int NT> struct FW;
template <> struct FW<2> { typedef F3 T; };
template <> struct FW<1> { typedef F3h T; };
__device__ __forceinline__ f32x4 mmx(const i32x4& a, const i32x4& b, f32x4 c) { return mm(a, b, c); }
__device__ __forceinline__ f32x4 mmx(const i32x2& a, const i32x2& b, f32x4 c) { return mmh(a, b, c); }
__device__ __forceinline__ void mm6x(const F3& a, const F3& b, f32x4& c) { mm6(a, b, c); }
__device__ __forceinline__ void mm6x(const F3h& a, const F3h& b, f32x4& c) { mm6h(a, b, c); }
__device__ __forceinline__ void fr_put(char* fr, int tile, int lane, const F3h& f) {
  i32x2* p = reinterpret_cast<i32x2*>(fr) + tile * 192 + lane;
  p[0] = f.h; p[64] = f.m; p[128] = f.l;
}
template <int NT> __device__ __forceinline__ typename FW<NT>::T fr_getn(const char* fr, int tile, int lane);
template <> __device__ __forceinline__ F3 fr_getn<2>(const char* fr, int tile, int lane) { return fr_get(fr, tile, lane); }
template <> __device__ __forceinline__ F3h fr_getn<1>(const char* fr, int tile, int lane) {
  const i32x2* p = reinterpret_cast<const i32x2*>(fr) + tile * 192 + lane;
  F3h f;
  f.h = p[0]; f.m = p[64]; f.l = p[128];
  return f;
}
template <int NT> __device__ __forceinline__ typename FW<NT>::T g_colsn(const char* img, int rb, int k0);
template <> __device__ __forceinline__ F3 g_colsn<2>(const char* img, int rb, int k0) { return g_cols(img, rb, k0); }
template <> __device__ __forceinline__ F3h g_colsn<1>(const char* img, int rb, int k0) {
  const char* p = img + k0 * 64 + rb;
  F3h f;
  f.h = *reinterpret_cast<const i32x2*>(p); f.m = *reinterpret_cast<const i32x2*>(p + GPL); f.l = *reinterpret_cast<const i32x2*>(p + 2 * GPL);
  return f;
}
__device__ __forceinline__ F3 splitn(const f32x4 (&x)[2]) { return split8(x[0], x[1]); }
__device__ __forceinline__ F3h splitn(const f32x4 (&x)[1]) { return split4(x[0]); }

template <bool DHS, bool TWO, int NT>
__global__ __launch_bounds__(BNT, 2) void agent_bwd_x6_kernel(BX6Args a) {
  typedef typename FW<NT>::T FWT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave >> 2, s = wave & 3;
  const int q = lane >> 4, m = lane & 15, u = 16 * s + m;
  char* Gt = smem;                                   // [3][256][64 B]: one image, freed for the next step by the second barrier
  char* WL = Gt + GBUF;                              // [4 waves][4] weight fragments of team R that do not fit its registers
  char* HB = WL + 4 * FRB;                           // [2] fragments of h_prev(t) column tiles
  char* XB = HB + 2 * FRB;                           // [2] fragments of x(t) column tiles
  float* W2s = reinterpret_cast<float*>(XB + 2 * FRB);      // [32][64]
  int4* QT = reinterpret_cast<int4*>(W2s + 32 * H);   // [3][32]: (idx, val, idx2, val2) of a row at a step
  int* rowidx = reinterpret_cast<int*>(QT + 3 * 32);  // [32]: index of (b, 0, n) in (B,T,N) or -1
  int* rowrho = rowidx + 32;                          // [32]
  unsigned* XM = reinterpret_cast<unsigned*>(rowrho + 32);      // [2][4][64]: relu'(x) bits of a lane's 8 elements, by step parity
  float* BS = reinterpret_cast<float*>(XM + 512);               // [4][256]: team R's bias sums, one slot per thread

  const long NTILES = (a.R + 15) >> 4;
  const long tile0 = (long)NT * blockIdx.x + a.tile_base;
  const long tl[2] = {tile0, tile0 + 1 < NTILES ? tile0 + 1 : tile0};      // (a missing second tile reads the first: its gradients are zero)
  if (tid < 16 * NT) {
    const long rho = tile0 * 16 + tid;
    const bool ok = rho < a.R;
    rowidx[tid] = ok ? (int)((rho / a.N) * a.T * a.N + rho % a.N) : -1;
    rowrho[tid] = ok ? (int)rho : -1;
  }
  for (int e = tid; e < 32 * H; e += BNT) W2s[e] = e < a.A * H ? a.W2[e] : 0.f;
  __syncthreads();
  const int T = a.T;
  // (idx, val, idx2, val2) of row r at step t
  // The pairs are READ a step before they are handed over (qraw: index and value unconditionally, nothing looks at them) and
  // range-checked when they are written to the table (qfix): no global round trip inside a step of the weight-gradient team
  // (stamps at one tile: the wave that does this was 1 600 cycles behind the other three at the second barrier, 21 % of the step).
  // Raw: (idx, val bits, idx2, val2 bits), idx = -1 where the row / step does not exist.
  auto qraw = [&](int r, int t) {
    int4 e = make_int4(-1, 0, -1, 0);
    const int ri = rowidx[r];
    if (ri >= 0 && t >= 0) {
      const unsigned k = (unsigned)ri + (unsigned)t * (unsigned)a.N;      // (B T N < 2^24: the host checks B T N H 4 < 2^32)
      const unsigned kv = a.dq_gdiv == 1 ? k : k / (unsigned)a.dq_gdiv;    // (a 64-bit '/' is a few hundred instructions here)
      e.x = a.dq_idx[k];
      e.y = __float_as_int(a.dq_val[kv]);
      if (a.dq_idx2) { e.z = a.dq_idx2[k]; e.w = __float_as_int(a.dq_val2[kv]); }
    }
    return e;
  };
  // an absent pair is (column 0, value 0): no branches where it is used
  // Who hands the pairs of step t - 2 over during step t (slot (t - 2) % 3 = (t + 1) % 3 was last read by team R's gate gradients of
  // step t + 1 and by the dW_2 product of step t + 2, both before the first barrier of step t + 1).  One tile: team R's wave 0, at
  // the END of its gate phase - it waits ~700 cycles at the first barrier for the weight-gradient team, whose wave that did this
  // was the last at the second barrier by ~950 cycles (stamps; moving the duty to another wave of that team moved the lag with
  // it).  Two tiles: team R runs at the register ceiling, the weight-gradient team's first wave keeps the duty behind the barrier.
  constexpr bool QDR = NT == 1;
  auto qfix = [&](int4 e) {
    const bool ok1 = e.x >= 0 && e.x < a.A, ok2 = e.z >= 0 && e.z < a.A;
    return make_int4(ok1 ? e.x : 0, ok1 ? e.y : 0, ok2 ? e.z : 0, ok2 ? e.w : 0);
  };
  // per-lane address parts of the image - worked out again in every step from the lane number (a dozen integer instructions) rather
  // than kept in registers across the loop: both teams run at the 256-register ceiling
  auto lane_parts = [&](int& wb_, int& tb0_, int& tb1_) {
    int l = lane;
    asm volatile("" : "+v"(l));                      // (opaque: not hoisted out of the loop)
    const int q_ = l >> 4, m_ = l & 15, qq = m_ >> 2, p4 = m_ & 3;
    wb_ = m_ * 64 + 8 * (q_ ^ (2 * ((m_ >> 2) & 3)));                              // put / plain read: column (k0 + m), piece q (tile 0; ^32: tile 1)
    tb0_ = (8 * q_ + qq) * 64 + 8 * (p4 ^ (2 * ((2 * q_) & 3)));                    // transposed read: columns 8g + qq (g = q), half 0
    tb1_ = (8 * q_ + 4 + qq) * 64 + 8 * (p4 ^ (2 * ((2 * q_ + 1) & 3)));            // half 1
  };
  ST_DECL(4);
  if (team == 0) {
    // =============================== team R: gate gradients, the carry chain, dx; the bias sums ===============================
    // k chunks of [drp | dzp | dhn] <-> rows of W_hh (r | z | n);  of [drp | dzp | dnp] <-> rows of W_ih.  Chunks 0..3 of both in
    // registers (96), chunks 4, 5 (the n rows) as ready fragments in LDS: with all twelve the saved planes in flight would spill
    F3 whT[4], wiT[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { whT[c] = wTfrag(a.Whh, c, 16 * s, lane); wiT[c] = wTfrag(a.Wih, c, 16 * s, lane); }
    char* wl = WL + s * FRB;
    fr_put(wl, 0, lane, wTfrag(a.Whh, 4, 16 * s, lane)); fr_put(wl, 1, lane, wTfrag(a.Whh, 5, 16 * s, lane));
    fr_put(wl, 2, lane, wTfrag(a.Wih, 4, 16 * s, lane)); fr_put(wl, 3, lane, wTfrag(a.Wih, 5, 16 * s, lane));
    f32x4 carry[NT];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) carry[tt] = splat(0.f);
    // column sums of drp, dzp, dnp, dhn over this lane's rows: thread-private fp32 slots in LDS, not registers (this team runs at
    // the register ceiling, and a spilled register's reload waits for every prefetched plane in flight)
    // (the variants that fit keep them in registers)
    constexpr bool BSL = DHS || NT == 1;
    float* bsl = BS + tid;
    float bsr[4] = {0.f, 0.f, 0.f, 0.f};
    if (BSL) {
#pragma unroll
      for (int k = 0; k < 4; ++k) bsl[256 * k] = 0.f;
    }
    f32x4 sv[NT][5];                                    // saved planes of the step: h_prev, r, z, n, W_hn h + b_hn
    f32x4 dhsv[NT];
    auto svload = [&](int t, int tt) {
      const float* sp = a.saved + sv_off((long)t * NTILES + tl[tt], 0, s, lane);
      sv[tt][0] = *reinterpret_cast<const f32x4*>(sp);
      sv[tt][1] = *reinterpret_cast<const f32x4*>(sp + 2 * 1024);
      sv[tt][2] = *reinterpret_cast<const f32x4*>(sp + 3 * 1024);
      sv[tt][3] = *reinterpret_cast<const f32x4*>(sp + 4 * 1024);
      sv[tt][4] = *reinterpret_cast<const f32x4*>(sp + 5 * 1024);
      if (DHS) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ri = rowidx[16 * tt + 4 * q + r];      // (32-bit element offsets: B T N H 4 < 2^32, checked by the host)
          dhsv[tt][r] = ri >= 0 ? a.dhs[(unsigned)(ri + t * a.N) * (unsigned)H + (unsigned)u] : 0.f;
        }
      }
    };
    svload(T - 1, 0);
    if (NT == 2) svload(T - 1, NT - 1);
    if (tid < 16 * NT) { QT[((T - 1) % 3) * 32 + tid] = qfix(qraw(tid, T - 1)); QT[((T - 2 + 3) % 3) * 32 + tid] = qfix(qraw(tid, T - 2)); }
    int4 qnR = make_int4(-1, 0, -1, 0);
    if (QDR && tid < 16 * NT) qnR = qraw(tid, T - 3);
    __syncthreads();
    for (int t = T - 1; t >= 0; --t) {
      const int par = t & 1;
      char* img = Gt;
      const int4* qt = QT + (t % 3) * 32;
      int wb, tb0, tb1;
      lane_parts(wb, tb0, tb1);
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        f32x4 drp, dzp, dnp, dhn;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int4 e = qt[16 * tt + 4 * q + r];
          float dh = carry[tt][r];
          if (DHS) dh += dhsv[tt][r];
          dh = __fmaf_rn(__int_as_float(e.y), W2s[e.x * H + u], dh);
          if (TWO) dh = __fmaf_rn(__int_as_float(e.w), W2s[e.z * H + u], dh);
          const float hp = sv[tt][0][r], rg = sv[tt][1][r], zg = sv[tt][2][r], ng = sv[tt][3][r], hn = sv[tt][4][r];
          const float dn = dh * (1.f - zg);
          const float dz = dh * (hp - ng);
          carry[tt][r] = dh * zg;                    // the direct path h_prev -> h; the W_hh products are added after the barrier
          const float dnp_ = dn * (1.f - ng * ng);
          const float dr = dnp_ * hn;
          dnp[r] = dnp_;
          dhn[r] = dnp_ * rg;
          drp[r] = dr * rg * (1.f - rg);
          dzp[r] = dz * zg * (1.f - zg);
        }
        const float s0 = (drp[0] + drp[1]) + (drp[2] + drp[3]), s1 = (dzp[0] + dzp[1]) + (dzp[2] + dzp[3]);
        const float s2 = (dnp[0] + dnp[1]) + (dnp[2] + dnp[3]), s3 = (dhn[0] + dhn[1]) + (dhn[2] + dhn[3]);
        if (BSL) { bsl[0] += s0; bsl[256] += s1; bsl[512] += s2; bsl[768] += s3; }      // (ds_add_f32 without return was measured: 0.40 -> 0.55 ms at 512 envs - LDS float atomics serialise)
        else { bsr[0] += s0; bsr[1] += s1; bsr[2] += s2; bsr[3] += s3; }
        g_put(img, wb, 0 * 64 + 16 * s, tt, split4(drp));
        g_put(img, wb, 1 * 64 + 16 * s, tt, split4(dzp));
        g_put(img, wb, 2 * 64 + 16 * s, tt, split4(dnp));
        g_put(img, wb, 3 * 64 + 16 * s, tt, split4(dhn));
      }
      if (QDR && tid < 16 * NT) {                    // the dq pairs of the steps to come (see QDR)
        if (t >= 2) QT[((t - 2) % 3) * 32 + tid] = qfix(qnR);
        qnR = qraw(tid, t - 3);
      }
      ST_MARK(0);
      WG_BARRIER();
      ST_MARK(1);
      __builtin_amdgcn_sched_barrier(0);
      if (t > 0) { svload(t - 1, 0); if (NT == 2) svload(t - 1, NT - 1); }      // in flight under the products below
      const unsigned xmask = XM[par * 256 + s * 64 + lane];    // relu'(x(t)) of this lane's 8 elements (team I published it)
      __builtin_amdgcn_sched_barrier(0);
      // carry' = dh z + [drp | dzp | dhn] W_hh  and  dx = [drp | dzp | dnp] W_ih -> relu gate -> dxp : a tile at a time; the fragments
      // of drp and dzp serve both products
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        f32x4 ca = carry[tt], cb = splat(0.f), da = splat(0.f), db = splat(0.f);
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
          const F3 a0 = g_tr3(img, tb0, tb1, 64 * cp, tt), a1 = g_tr3(img, tb0, tb1, 64 * cp + 32, tt);
#define OP(p_, q_) ca = mm(a0.p_, whT[2 * cp].q_, ca); cb = mm(a1.p_, whT[2 * cp + 1].q_, cb); da = mm(a0.p_, wiT[2 * cp].q_, da); db = mm(a1.p_, wiT[2 * cp + 1].q_, db);
          X6_TERMS(OP)
#undef OP
          __builtin_amdgcn_sched_barrier(0);          // (keeps the next chunk's fragments from being fetched early: registers)
        }
        {
          const F3 a0 = g_tr3(img, tb0, tb1, 192, tt), a1 = g_tr3(img, tb0, tb1, 192 + 32, tt);
          const F3 w4 = fr_get(wl, 0, lane), w5 = fr_get(wl, 1, lane);
#define OP(p_, q_) ca = mm(a0.p_, w4.q_, ca); cb = mm(a1.p_, w5.q_, cb);
          X6_TERMS(OP)
#undef OP
        }
        carry[tt] = ca + cb;
        __builtin_amdgcn_sched_barrier(0);
        {
          const F3 a0 = g_tr3(img, tb0, tb1, 128, tt), a1 = g_tr3(img, tb0, tb1, 128 + 32, tt);
          const F3 w4 = fr_get(wl, 2, lane), w5 = fr_get(wl, 3, lane);
#define OP(p_, q_) da = mm(a0.p_, w4.q_, da); db = mm(a1.p_, w5.q_, db);
          X6_TERMS(OP)
#undef OP
        }
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 dx = da + db;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ri = rowidx[16 * tt + 4 * q + r];
          if (ri >= 0) a.dxp[(unsigned)(ri + t * a.N) * (unsigned)H + (unsigned)u] = (xmask >> (4 * tt + r)) & 1u ? dx[r] : 0.f;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      ST_MARK(2);
      WG_BARRIER();                                  // second barrier of the step: everybody has read the image, the next step may fill it
      ST_MARK(3);
    }
    ST_DUMP(4);
    // ---- epilogue: dh0, the bias sums of this workgroup's slab
    if (a.dh0) {
#pragma unroll
      for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rho = rowrho[16 * tt + 4 * q + r];
          if (rho >= 0) a.dh0[(long)rho * H + u] = carry[tt][r];
        }
    }
    float* slab = a.ws + ((long)blockIdx.x + a.slab_base) * slab_floats(a.A);
    float bs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      bs[k] = BSL ? bsl[256 * k] : bsr[k];
      bs[k] += __shfl_xor(bs[k], 16);
      bs[k] += __shfl_xor(bs[k], 32);
    }
    if (q == 0) {
      float* db = slab + 2 * 192 * 64 + (long)a.A * H;
      db[0 * 64 + u] = bs[0]; db[1 * 64 + u] = bs[1]; db[2 * 64 + u] = bs[2];                     // db_ih: r | z | n (input side)
      db[192 + 0 * 64 + u] = bs[0]; db[192 + 1 * 64 + u] = bs[1]; db[192 + 2 * 64 + u] = bs[3];   // db_hh: r | z | n (hidden side)
    }
  } else {
    // =============================== team I: the weight gradients dW_ih, dW_hh, dW_2 ===============================
    const int ti = tid - 256;
    f32x4 accI[3][4], accH[3][4], acc2[2] = {splat(0.f), splat(0.f)};      // (acc2[1]: actions 16 .. 31)
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c) { accI[b][c] = splat(0.f); accH[b][c] = splat(0.f); }
    float bs2[2] = {0.f, 0.f};
    f32x4 xn[NT], hn[NT];                             // x and h_prev of the step to come, column tile s (in flight across the barrier)
    auto xhload = [&](int t) {
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const float* sp = a.saved + sv_off((long)t * NTILES + tl[tt], 0, s, lane);
        hn[tt] = *reinterpret_cast<const f32x4*>(sp);
        xn[tt] = *reinterpret_cast<const f32x4*>(sp + 1024);
      }
    };
    // x(t), h_prev(t) of column tile s -> ready B fragments of the weight gradients for the whole team; relu'(x) for team R's dx
    auto publish = [&](int t) {
      fr_put(XB + (t & 1) * FRB, s, lane, splitn(xn));
      fr_put(HB + (t & 1) * FRB, s, lane, splitn(hn));
      unsigned mk = 0;
#pragma unroll
      for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mk |= xn[tt][r] > 0.f ? 1u << (4 * tt + r) : 0u;
      XM[(t & 1) * 256 + s * 64 + lane] = mk;
    };
    auto dw2 = [&](const int4* qt, const FWT& hb) {
#pragma unroll
      for (int ac = 0; ac < 2; ++ac) {
        if (ac == 1 && a.A <= 16) break;
        const int col = 16 * ac + m;
        f32x4 d[NT];
        float sum = 0.f;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int4 e = qt[16 * tt + 4 * q + r];
            d[tt][r] = (e.x == col ? __int_as_float(e.y) : 0.f) + (TWO && e.z == col ? __int_as_float(e.w) : 0.f);
          }
          sum += (d[tt][0] + d[tt][1]) + (d[tt][2] + d[tt][3]);
        }
        bs2[ac] += sum;
        mm6x(splitn(d), hb, acc2[ac]);
      }
    };
    xhload(T - 1);
    publish(T - 1);
    xhload(T - 2);
    int4 qn = make_int4(-1, 0, -1, 0);
    if (!QDR && ti < 16 * NT) qn = qraw(ti, T - 3);
    __syncthreads();
    {      // the last step's dq meets h(T-1), the hidden state after the last step (plane 0 of step T)
      f32x4 hT[NT];
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) hT[tt] = *reinterpret_cast<const f32x4*>(a.saved + sv_off((long)T * NTILES + tl[tt], 0, s, lane));
      dw2(QT + ((T - 1) % 3) * 32, splitn(hT));
    }
    for (int t = T - 1; t >= 0; --t) {
      const int par = t & 1;
      const char* img = Gt;
      WG_BARRIER();                                  // first barrier of the step: the image holds the gate gradients of step t
      ST_MARK(0);
      // the steps to come: the dq pairs of step t-2 handed over (their slot's previous content - step t+1 - was last read before
      // this barrier: team R's gate gradients of step t+1 and this team's dW_2 product of step t+2)
      if (!QDR && ti < 16 * NT) {
        if (t >= 2) QT[((t - 2) % 3) * 32 + ti] = qfix(qn);
        qn = qraw(ti, t - 3);
      }
      int rb, tb0, tb1;
      lane_parts(rb, tb0, tb1);
      // ALL of this step's image fragments go to registers first (48 at NT = 2), so that the second barrier - "the image has been
      // read" - can come BEFORE this team's 150 products: they then run while team R computes the gate gradients of step t-1 (vector
      // work that used to leave the matrix pipe idle with this team parked at the barrier) and refills the image
      const FWT ar = g_colsn<NT>(img, rb, 0 * 64 + 16 * s), az = g_colsn<NT>(img, rb, 1 * 64 + 16 * s);
      const FWT ani = g_colsn<NT>(img, rb, 2 * 64 + 16 * s), anh = g_colsn<NT>(img, rb, 3 * 64 + 16 * s);
      if (t > 0) publish(t - 1);                     // (loaded during the previous step; the buffers were last read before this barrier)
      if (t > 1) xhload(t - 2);
      __builtin_amdgcn_sched_barrier(0);
      // dW_ih[this wave's 16 columns of r | z | n][all 64] += [drp | dzp | dnp]^T x - BEFORE the second barrier: these products fill
      // the matrix pipe's bubbles while team R (the other wave of this SIMD) waits for its transposed image reads
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const FWT xb = fr_getn<NT>(XB + par * FRB, c, lane);
#define OP(p_, q_) accI[0][c] = mmx(ar.p_, xb.q_, accI[0][c]); accI[1][c] = mmx(az.p_, xb.q_, accI[1][c]); accI[2][c] = mmx(ani.p_, xb.q_, accI[2][c]);
        X6_TERMS(OP)
#undef OP
      }
      // dW_hh[...] += [drp | dzp | dhn]^T h_prev: HEAD of its four column chunks before the second barrier, the rest behind it
      // beside team R's gate gradients of the next step.  Two tiles (stamps, 3s5z 1024 envs): the interval before the barrier is
      // bound by team R's products (6 070 cycles against this team's 5 100), the one behind it by THIS team (4 030 against 3 320)
      // -> one chunk moves forward.  One tile: the interval before the barrier is the longer one for this team already.
      constexpr int HEAD = DWHH_HEAD >= 0 ? DWHH_HEAD : (NT == 2 ? 1 : 0);
#pragma unroll
      for (int c = 0; c < HEAD; ++c) {
        const FWT hb = fr_getn<NT>(HB + par * FRB, c, lane);
#define OP(p_, q_) accH[0][c] = mmx(ar.p_, hb.q_, accH[0][c]); accH[1][c] = mmx(az.p_, hb.q_, accH[1][c]); accH[2][c] = mmx(anh.p_, hb.q_, accH[2][c]);
        X6_TERMS(OP)
#undef OP
      }
      __builtin_amdgcn_sched_barrier(0);
      ST_MARK(1);
      WG_BARRIER();                                  // second barrier of the step (the image fragments are in registers: the image may be refilled)
      ST_MARK(2);
#pragma unroll
      for (int c = HEAD; c < 4; ++c) {
        const FWT hb = fr_getn<NT>(HB + par * FRB, c, lane);
#define OP(p_, q_) accH[0][c] = mmx(ar.p_, hb.q_, accH[0][c]); accH[1][c] = mmx(az.p_, hb.q_, accH[1][c]); accH[2][c] = mmx(anh.p_, hb.q_, accH[2][c]);
        X6_TERMS(OP)
#undef OP
      }
      __builtin_amdgcn_sched_barrier(0);
      // dW_2[action][this wave's 16 columns] += dq(t-1)^T h(t-1): h(t-1) = h_prev of THIS step (HB[par], stable until the barrier after
      // the next).  Lane (g, i): action i, k slots = rows 4g .. 4g+3 of tile 0, then of tile 1
      if (t > 0) dw2(QT + ((t - 1) % 3) * 32, fr_getn<NT>(HB + par * FRB, s, lane));
      ST_MARK(3);
    }
    ST_DUMP(4);
    float* slab = a.ws + ((long)blockIdx.x + a.slab_base) * slab_floats(a.A);
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          slab[(long)(64 * b + 16 * s + 4 * q + r) * H + 16 * c + m] = accI[b][c][r];
          slab[192 * 64 + (long)(64 * b + 16 * s + 4 * q + r) * H + 16 * c + m] = accH[b][c][r];
        }
#pragma unroll
    for (int ac = 0; ac < 2; ++ac) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * ac + 4 * q + r < a.A) slab[2 * 192 * 64 + (long)(16 * ac + 4 * q + r) * H + u] = acc2[ac][r];
      float b = bs2[ac];
      b += __shfl_xor(b, 16);
      b += __shfl_xor(b, 32);
      if (s == 0 && q == 0 && 16 * ac + m < a.A) slab[2 * 192 * 64 + (long)a.A * H + 2 * 192 + 16 * ac + m] = b;
    }
  }
}

struct RedArgs {
  const float* ws; int nwg; int A;
  float *dWih, *dWhh, *dW2, *dbih, *dbhh, *db2;
};
// fixed-order sum of the workgroups' slabs into the gradient tensors (accumulated into): 16 partial sums per element (768 slabs of
// 100 KB at the headline shape: four partial sums per element left the reduction latency-bound, 73 us)
constexpr int BRSG = 16;
__global__ __launch_bounds__(64 * BRSG) void agent_bwd_x6_reduce_kernel(RedArgs a) {
  __shared__ float part[BRSG][64];
  const long slab = slab_floats(a.A);
  const int el = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const long e = (long)blockIdx.x * 64 + el;
  float s = 0.f;
  if (e < slab) s = slab_sum(a.ws + e, slab, sg, BRSG, a.nwg);
  part[sg][el] = s;
  __syncthreads();
  if (sg != 0 || e >= slab) return;
  s = 0.f;
#pragma unroll
  for (int g = 0; g < BRSG; ++g) s += part[g][el];
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

}  // namespace

// shapes the split BPTT covers: H = 64, <= 32 actions, a sparse dq (one or two (column, value) pairs per row), T >= 3
extern "C" int marl_agent_unroll_bwd_x6_supported(int B, int T, int N, int A, int sparse_dq) {
  if (B < 1 || T < 3 || N < 1 || A < 1 || A > 32 || !sparse_dq) return 0;
  if ((double)B * T * N * H * 4.0 >= 4294967296.0) return 0;
  return 1;
}

// Workgroups of a batch of R rows: up to 256 row tiles run one tile per workgroup (the small shards: one round of workgroups);
// beyond that two tiles per workgroup in FULL rounds of 256 workgroups, and when what is left fits one round of one-tile workgroups
// (at most 256 tiles) it runs as such - a second launch of the one-tile instantiation, whose step is ~0.65 of the two-tile one's
// (4096 envs x 5 agents = 1280 tiles: 512 two-tile workgroups + 256 one-tile ones instead of 640 = 2.5 rounds of two-tile ones)
static void bx6_plan(long R, long& n2, long& n1) {
  const long tiles = (R + 15) / 16;
  if (tiles <= 256) { n2 = 0; n1 = tiles; return; }
  n2 = tiles / 512 * 256;
  const long rem = tiles - 2 * n2;
  if (n2 > 0 && rem > 0 && rem <= 256) { n1 = rem; return; }
  n2 = (tiles + 1) / 2; n1 = 0;
}

extern "C" size_t marl_agent_bwd_x6_workspace(int B, int N, int A) {
  long n2, n1;
  bx6_plan((long)B * N, n2, n1);
  return (size_t)(n2 + n1) * slab_floats(A) * sizeof(float);
}

extern "C" int marl_agent_unroll_bwd_x6(const marl_agent_weights_t* w, const int* dq_idx, const float* dq_val, const int* dq_idx2,
                                        const float* dq_val2, int dq_gdiv, const float* dhs, const float* saved, float* dxp,
                                        float* dh0, const marl_agent_grads_t* g, float* ws, size_t ws_bytes, int B, int T, int N,
                                        int A, void* stream) {
  if (B <= 0 || T <= 0) return 0;
  if (w->H != H || !marl_agent_unroll_bwd_x6_supported(B, T, N, A, dq_idx != nullptr) || !dq_val || (dq_idx2 && !dq_val2))
    return (int)hipErrorInvalidValue;
  if (ws_bytes < marl_agent_bwd_x6_workspace(B, N, A) || (reinterpret_cast<uintptr_t>(saved) & 15)) return (int)hipErrorInvalidValue;
  BX6Args a;
  a.Wih = w->w_ih; a.Whh = w->w_hh; a.W2 = w->fc2_w;
  a.dq_idx = dq_idx; a.dq_val = dq_val; a.dq_idx2 = dq_idx2; a.dq_val2 = dq_val2; a.dq_gdiv = dq_gdiv > 1 ? dq_gdiv : 1;
  a.dhs = dhs; a.saved = saved; a.dxp = dxp; a.dh0 = dh0; a.ws = ws;
  a.B = B; a.T = T; a.N = N; a.A = A; a.R = (long)B * N;
  long n2, n1;
  bx6_plan(a.R, n2, n1);
  const size_t lds = (size_t)GBUF + 8 * FRB + 32 * H * 4 + 3 * 32 * 16 + 2 * 32 * 4 + 2 * 256 * 4 + 4 * 256 * 4;
#define BX6_PICK(NT_) (dhs ? (dq_idx2 ? (const void*)agent_bwd_x6_kernel<true, true, NT_> : (const void*)agent_bwd_x6_kernel<true, false, NT_>) \
                           : (dq_idx2 ? (const void*)agent_bwd_x6_kernel<false, true, NT_> : (const void*)agent_bwd_x6_kernel<false, false, NT_>))
  hipStream_t st = (hipStream_t)stream;
  for (int pass = 0; pass < 2; ++pass) {
    const long nwg = pass == 0 ? n2 : n1;
    if (nwg == 0) continue;
    const void* fn = pass == 0 ? BX6_PICK(2) : BX6_PICK(1);
    a.tile_base = pass == 0 ? 0 : 2 * n2;
    a.slab_base = pass == 0 ? 0 : (int)n2;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    void* kargs[] = {(void*)&a};
    e = hipLaunchKernel(fn, dim3((unsigned)nwg), dim3(BNT), kargs, lds, st);
    if (e != hipSuccess) return (int)e;
    MARL_CHECK_LAUNCH();
  }
#undef BX6_PICK
  RedArgs r;
  r.ws = ws; r.nwg = (int)(n2 + n1); r.A = A;
  r.dWih = g->w_ih; r.dWhh = g->w_hh; r.dW2 = g->fc2_w; r.dbih = g->b_ih; r.dbhh = g->b_hh; r.db2 = g->fc2_b;
  const long slab = slab_floats(A);
  hipLaunchKernelGGL(agent_bwd_x6_reduce_kernel, dim3((unsigned)((slab + 63) / 64)), dim3(64 * BRSG), 0, st, r);
  MARL_CHECK_LAUNCH();
  return 0;
}

ST_DEFINE_SETTER(marl_debug_stamps_agent_bwd_x6)
