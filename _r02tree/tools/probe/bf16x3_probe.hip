// What a split-precision (3 x bf16) emulation of the fp32 GEMMs would buy on gfx950 - a MEASUREMENT for DESIGN.md section 8,
// not used by the product (every kernel of the product multiplies in fp32: v_mfma_f32_16x16x4_f32).
//   1. issue cost of v_mfma_f32_16x16x16_bf16 (8192 FLOP) and v_mfma_f32_16x16x32_bf16 (16384 FLOP, new in gfx950) next to
//      v_mfma_f32_16x16x4_f32 (2048 FLOP), whole chip
//   2. does VALU work of the SIMD partner wave overlap bf16 MFMAs (it does not overlap fp32 ones: coissue_probe.hip)
//   3. accuracy: C = A B (K = 64 and 208, the GRU / fc1 depths) with a = a_hi + a_mid + a_lo (bf16 each, exact split of the 24-bit
//      significand), 6 products (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid) accumulated in fp32 by the bf16 MFMA, against the
//      fp32 MFMA result and an fp64 host reference
//   hipcc -O3 --offload-arch=gfx950 -o bf16x3_probe bf16x3_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ s16x4 to_bf16(const f32x4& v) { return __builtin_bit_cast(s16x4, __builtin_convertvector(v, bf16x4_t)); }
__device__ __forceinline__ f32x4 from_bf16(const s16x4& v) { return __builtin_convertvector(__builtin_bit_cast(bf16x4_t, v), f32x4); }

template <int MODE>   // 0: fp32 MFMAs   1: bf16 16x16x16   2: bf16 16x16x32 (gfx950) ; waves 4-7: nv dependent-free v_fma_f32
__global__ __launch_bounds__(512) void rate(int nm, int nv, unsigned long long* out, float* sink) {
  const int wave = threadIdx.x >> 6;
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  float keep = 0.f;
  if (wave < 4) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const float x = (float)threadIdx.x * 1e-6f, y = 1e-6f;
    const s16x4 xb = to_bf16((f32x4){x, x, x, x}), yb = to_bf16((f32x4){y, y, y, y});
    bf16x8_t x8, y8;
    for (int i = 0; i < 8; ++i) { x8[i] = (__bf16)x; y8[i] = (__bf16)y; }
    for (int i = 0; i < nm; i += 4) {
      if (MODE == 0) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
      } else if (MODE == 1) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(xb, yb, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(xb, yb, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(xb, yb, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(xb, yb, a3, 0, 0, 0);
      } else {
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x8, y8, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x8, y8, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x8, y8, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x8, y8, a3, 0, 0, 0);
      }
    }
    keep = a0[0] + a1[1] + a2[2] + a3[3];
  } else {
    float v0 = (float)threadIdx.x, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f;
    for (int i = 0; i < nv; i += 4) {
      v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 1.0001f, 0.5f);
      v2 = __builtin_fmaf(v2, 1.0001f, 0.5f); v3 = __builtin_fmaf(v3, 1.0001f, 0.5f);
    }
    keep = v0 + v1 + v2 + v3;
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  if (keep == 12345.678f) sink[0] = keep;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { out[wave] = t1 - t0; out[8 + wave] = r1 - r0; }
}

// one wave: C (16 x 16) = A (16 x K) B (K x 16), K a multiple of 16; A row-major, B given transposed (16 x K)
__global__ __launch_bounds__(64) void gemm16(const float* A, const float* Bt, int K, float* C32, float* C3, float* C6) {
  const int lane = threadIdx.x, q = lane >> 4, m = lane & 15;
  f32x4 c32 = {0, 0, 0, 0}, c3 = c32, c6 = c32;
  for (int k0 = 0; k0 < K; k0 += 16) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(A + m * K + k0 + 4 * q);      // k-permutation: lane group q feeds k0+4q..+3
    const f32x4 b = *reinterpret_cast<const f32x4*>(Bt + m * K + k0 + 4 * q);
    for (int i = 0; i < 4; ++i) c32 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], c32, 0, 0, 0);
    const s16x4 ah = to_bf16(a), bh = to_bf16(b);
    const f32x4 ar = a - from_bf16(ah), br = b - from_bf16(bh);
    const s16x4 am = to_bf16(ar), bm = to_bf16(br);
    const s16x4 al = to_bf16(ar - from_bf16(am)), bl = to_bf16(br - from_bf16(bm));
    // smallest terms first
    c6 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(am, bm, c6, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bl, c6, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bh, c6, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bm, c6, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(am, bh, c6, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, c6, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bm, c3, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(am, bh, c3, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, c3, 0, 0, 0);
  }
  for (int i = 0; i < 4; ++i) {
    C32[(4 * q + i) * 16 + m] = c32[i]; C3[(4 * q + i) * 16 + m] = c3[i]; C6[(4 * q + i) * 16 + m] = c6[i];
  }
}

int main() {
  unsigned long long* out; float* sink;
  (void)hipMalloc(&out, 128); (void)hipMalloc(&sink, 4);
  unsigned long long h[16];
  const int NM = 8192, NV = 8192;
  printf("issue cost, 256 workgroups x 8 waves (waves 0-3: MFMAs with 4 independent accumulators, waves 4-7: v_fma_f32), s_memtime ticks:\n");
  for (int mode = 0; mode < 3; ++mode) {
    int cfg[3][2] = {{NM, 0}, {0, NV}, {NM, NV}};
    printf("  %-28s", mode == 2 ? "v_mfma_f32_16x16x32_bf16" : mode ? "v_mfma_f32_16x16x16_bf16" : "v_mfma_f32_16x16x4_f32");
    for (auto& c : cfg) {
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 2) rate<2><<<256, 512>>>(c[0], c[1], out, sink);
        else if (mode) rate<1><<<256, 512>>>(c[0], c[1], out, sink);
        else rate<0><<<256, 512>>>(c[0], c[1], out, sink);
      }
      (void)hipMemcpy(h, out, 128, hipMemcpyDeviceToHost);
      printf("  [mfma %d, valu %d] wave0 %llu (%.1f / MFMA) wave4 %llu", c[0], c[1], h[0], c[0] ? (double)h[0] / c[0] : 0.0, h[4]);
    }
    printf("\n");
  }
  printf("accuracy of one 16 x 16 output tile, inputs N(0,1) (a GRU / fc1 dot product), error relative to sqrt(K) (the size of an output):\n");
  for (int K : {64, 208}) {
    std::vector<float> A(16 * K), Bt(16 * K);
    unsigned s = 12345u + K;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xFFFF) / 65536.f + (float)((s >> 4) & 0xFF) / 16777216.f; };
    auto gauss = [&]() { float u1 = rnd() + 1e-7f, u2 = rnd(); return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); };
    for (auto& v : A) v = gauss();
    for (auto& v : Bt) v = gauss();
    float *dA, *dB, *d32, *d3, *d6;
    (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, Bt.size() * 4); (void)hipMalloc(&d32, 1024); (void)hipMalloc(&d3, 1024); (void)hipMalloc(&d6, 1024);
    (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, Bt.data(), Bt.size() * 4, hipMemcpyHostToDevice);
    gemm16<<<1, 64>>>(dA, dB, K, d32, d3, d6);
    float c32[256], c3[256], c6[256];
    (void)hipMemcpy(c32, d32, 1024, hipMemcpyDeviceToHost); (void)hipMemcpy(c3, d3, 1024, hipMemcpyDeviceToHost); (void)hipMemcpy(c6, d6, 1024, hipMemcpyDeviceToHost);
    double e32 = 0, e3 = 0, e6 = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)A[i * K + k] * (double)Bt[j * K + k];
        e32 = fmax(e32, fabs(c32[i * 16 + j] - ref)); e3 = fmax(e3, fabs(c3[i * 16 + j] - ref)); e6 = fmax(e6, fabs(c6[i * 16 + j] - ref));
      }
    const double sc = sqrt((double)K);
    printf("  K = %3d: max |err| / sqrt(K)   fp32 MFMA %.2e   bf16 x 6 products %.2e   bf16 x 3 products %.2e\n", K, e32 / sc, e6 / sc, e3 / sc);
  }
  return 0;
}
