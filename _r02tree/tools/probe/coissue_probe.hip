// Does a wave's VALU work overlap its SIMD partner's fp32 MFMAs on gfx950?  (diagnostic, not part of the product)
// One workgroup of 8 waves per CU (two per SIMD): waves 0-3 issue NM v_mfma_f32_16x16x4_f32 (4 independent accumulators),
// waves 4-7 issue NV VALU ops of a given kind.  Cycles (s_memtime of wave 0 / wave 4) for: MFMA alone, VALU alone, both.
//   hipcc -O3 --offload-arch=gfx950 -o coissue_probe coissue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>   // 0: v_fma_f32   1: v_mul_lo_u32   2: v_exp_f32   3: ds_read_b128 (LDS)
__global__ __launch_bounds__(512) void probe(int nm, int nv, int prio, unsigned long long* out, float* sink) {
  __shared__ f32x4 lds[1024];
  const int wave = threadIdx.x >> 6;
  lds[threadIdx.x] = (f32x4){1.f, 2.f, 3.f, 4.f};
  lds[threadIdx.x + 512] = (f32x4){1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  float keep = 0.f;
  if (wave < 4) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const float x = (float)threadIdx.x, y = 1.0f;
    for (int i = 0; i < nm; i += 4) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
    }
    keep = a0[0] + a1[1] + a2[2] + a3[3];
  } else {
    if (prio) __builtin_amdgcn_s_setprio(3);
    float v0 = (float)threadIdx.x, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    f32x4 l = {0, 0, 0, 0};
    for (int i = 0; i < nv; i += 4) {
      if (KIND == 0) { v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 1.0001f, 0.5f); v2 = __builtin_fmaf(v2, 1.0001f, 0.5f); v3 = __builtin_fmaf(v3, 1.0001f, 0.5f); }
      if (KIND == 1) { u0 *= 0x7FEB352Du; u1 *= 0x7FEB352Du; u2 *= 0x7FEB352Du; u3 *= 0x7FEB352Du; }
      if (KIND == 2) { v0 = __builtin_amdgcn_exp2f(v0); v1 = __builtin_amdgcn_exp2f(v1); v2 = __builtin_amdgcn_exp2f(v2); v3 = __builtin_amdgcn_exp2f(v3); }
      if (KIND == 3) { l += lds[(threadIdx.x + i) & 1023]; l += lds[(threadIdx.x + i + 64) & 1023]; l += lds[(threadIdx.x + i + 128) & 1023]; l += lds[(threadIdx.x + i + 192) & 1023]; }
    }
    keep = v0 + v1 + v2 + v3 + (float)(u0 + u1 + u2 + u3) + l[0] + l[3];
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (keep == 12345.678f) sink[0] = keep;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}

template <int KIND>
void run(const char* name, int prio = 0) {
  unsigned long long* out; float* sink;
  (void)hipMalloc(&out, 64); (void)hipMalloc(&sink, 4);
  unsigned long long h[8];
  const int NM = 4096, NV = 8192;
  int cfg[3][2] = {{NM, 0}, {0, NV}, {NM, NV}};
  printf("%-14s", name);
  for (auto& c : cfg) {
    probe<KIND><<<256, 512>>>(c[0], c[1], prio, out, sink);
    probe<KIND><<<256, 512>>>(c[0], c[1], prio, out, sink);
    (void)hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    printf("  [mfma %4d, valu %4d] wave0 %7llu wave4 %7llu", c[0], c[1], h[0], h[4]);
  }
  printf("\n");
}

// same wave: NV independent VALU ops (of kind K2: 0 fma, 2 exp) spread between its own MFMAs - do they hide under the pipe time?
template <int PER, int K2>
__global__ __launch_bounds__(256) void own(int nm, unsigned long long* out, float* sink) {
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const float x = (float)threadIdx.x, y = 1.0f;
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = x + (float)k;
  auto work = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < PER; ++k) v[k & 7] = K2 == 2 ? __builtin_amdgcn_exp2f(v[k & 7]) : __builtin_fmaf(v[k & 7], 1.0001f, 0.5f);
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int i = 0; i < nm; i += 4) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); work();
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); work();
    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); work();
    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); work();
  }
  float keep = a0[0] + a1[1] + a2[2] + a3[3];
#pragma unroll
  for (int k = 0; k < 8; ++k) keep += v[k];
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (keep == 12345.678f) sink[0] = keep;
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
}
template <int PER, int K2>
void run_own() {
  unsigned long long* out; float* sink;
  (void)hipMalloc(&out, 64); (void)hipMalloc(&sink, 4);
  unsigned long long h;
  own<PER, K2><<<256, 256>>>(4096, out, sink);
  own<PER, K2><<<256, 256>>>(4096, out, sink);
  (void)hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
  printf("  one wave per SIMD, %d %s between consecutive MFMAs: %llu cycles for 4096 MFMAs (%.1f per MFMA)\n", PER, K2 == 2 ? "v_exp_f32" : "v_fma_f32", h, h / 4096.0);
}

int main() {
  printf("cycles (s_memtime ticks) per wave; 4096 MFMAs of 16x16x4 f32 = 131072 pipe cycles if 32 each\n");
  run<0>("v_fma_f32"); run<1>("v_mul_lo_u32"); run<2>("v_exp_f32"); run<3>("ds_read_b128");
  printf("VALU / LDS wave at s_setprio(3):\n");
  run<0>("v_fma_f32", 1); run<2>("v_exp_f32", 1); run<3>("ds_read_b128", 1);
  run_own<0, 0>(); run_own<2, 0>(); run_own<4, 0>(); run_own<6, 0>(); run_own<7, 0>(); run_own<8, 0>(); run_own<12, 0>();
  run_own<1, 2>(); run_own<2, 2>(); run_own<3, 2>();
  return 0;
}
