// bf16x6 split arithmetic shared by the split kernels (mlp3_x6.hip, agent_x6.hip): an fp32 operand element is split EXACTLY into
// three bf16 terms, a = hi + mid + lo (round to nearest even each: 8 + 8 + 8 significand bits), and an fp32 product is the six bf16
// products mid.mid, hi.lo, lo.hi, hi.mid, mid.hi, hi.hi accumulated in fp32 by v_mfma_f32_16x16x32_bf16, smallest first (the dropped
// terms mid.lo, lo.mid, lo.lo are <= 2^-24 of a product).  Lane maps: tools/probe/x6_layout_probe.hip.
#pragma once
#include "common.h"

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

struct F3 { i32x4 h, m, l; };       // 8 k-slots per lane, three bf16 planes
struct F3h { i32x2 h, m, l; };      // 4 k-slots per lane (v_mfma_f32_16x16x16_bf16: k = 4g + j)

// two fp32 values -> packed bf16 pairs of their hi / mid / lo terms (low half = first value); plain casts: hipcc emits
// v_cvt_pk_bf16_f32 (round to nearest even, NaN stays NaN)
__device__ __forceinline__ unsigned pk2(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf2));
}
__device__ __forceinline__ void split2(float x0, float x1, int& h, int& m, int& l) {
  const unsigned hp = pk2(x0, x1);
  const float r0 = x0 - __uint_as_float(hp << 16), r1 = x1 - __uint_as_float(hp & 0xffff0000u);      // exact
  const unsigned mp = pk2(r0, r1);
  const float s0 = r0 - __uint_as_float(mp << 16), s1 = r1 - __uint_as_float(mp & 0xffff0000u);      // exact, <= 8 bits left
  h = (int)hp; m = (int)mp; l = (int)pk2(s0, s1);
}
__device__ __forceinline__ F3h split4(const f32x4& a) {
  int h0, m0, l0, h1, m1, l1;
  split2(a[0], a[1], h0, m0, l0);
  split2(a[2], a[3], h1, m1, l1);
  F3h f;
  f.h = (i32x2){h0, h1}; f.m = (i32x2){m0, m1}; f.l = (i32x2){l0, l1};
  return f;
}
// slots 0..3 = a, slots 4..7 = b
__device__ __forceinline__ F3 split8(const f32x4& a, const f32x4& b) {
  const F3h x = split4(a), y = split4(b);
  F3 f;
  f.h = (i32x4){x.h[0], x.h[1], y.h[0], y.h[1]};
  f.m = (i32x4){x.m[0], x.m[1], y.m[0], y.m[1]};
  f.l = (i32x4){x.l[0], x.l[1], y.l[0], y.l[1]};
  return f;
}

__device__ __forceinline__ f32x4 mm(const i32x4& a, const i32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mmh(const i32x2& a, const i32x2& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}
// the six products, smallest first
__device__ __forceinline__ void mm6(const F3& a, const F3& b, f32x4& c) {
  c = mm(a.m, b.m, c); c = mm(a.h, b.l, c); c = mm(a.l, b.h, c);
  c = mm(a.h, b.m, c); c = mm(a.m, b.h, c); c = mm(a.h, b.h, c);
}
// four A fragments against one B fragment, the four accumulators round robin
__device__ __forceinline__ void mm6x4(const F3 (&a)[4], const F3& b, f32x4 (&c)[4]) {
#pragma unroll
  for (int t = 0; t < 4; ++t) c[t] = mm(a[t].m, b.m, c[t]);
#pragma unroll
  for (int t = 0; t < 4; ++t) c[t] = mm(a[t].h, b.l, c[t]);
#pragma unroll
  for (int t = 0; t < 4; ++t) c[t] = mm(a[t].l, b.h, c[t]);
#pragma unroll
  for (int t = 0; t < 4; ++t) c[t] = mm(a[t].h, b.m, c[t]);
#pragma unroll
  for (int t = 0; t < 4; ++t) c[t] = mm(a[t].m, b.h, c[t]);
#pragma unroll
  for (int t = 0; t < 4; ++t) c[t] = mm(a[t].h, b.h, c[t]);
}
__device__ __forceinline__ void mm6h(const F3h& a, const F3h& b, f32x4& c) {
  c = mmh(a.m, b.m, c); c = mmh(a.h, b.l, c); c = mmh(a.l, b.h, c);
  c = mmh(a.h, b.m, c); c = mmh(a.m, b.h, c); c = mmh(a.h, b.h, c);
}

