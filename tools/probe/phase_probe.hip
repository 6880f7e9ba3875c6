// How do the two waves of a SIMD share the vector pipe when each alternates an fp32-MFMA phase with a pointwise phase
// (the GRU row-tile body: 24 v_mfma_f32_16x16x4_f32, then ~40 VALU + 24 transcendental instructions)?  (diagnostic, not
// part of the product)
//   1. issue rate of independent VALU / transcendental instructions with one and with two waves per SIMD;
//   2. the tile body with two waves per SIMD: free running, phases aligned by a barrier, phases staggered by half a body,
//      and with one wave per SIMD - cycles per body.
//   hipcc -O3 --offload-arch=gfx950 -o phase_probe phase_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SB() __builtin_amdgcn_sched_barrier(0)

// NV independent v_fma (16 chains) or NT transcendental ops (16 chains) per iteration
template <int KIND>
__global__ __launch_bounds__(512) void rate(int iters, unsigned long long* out, float* sink) {
  float v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = (float)threadIdx.x * 0.001f + (float)k;
  unsigned long long t0, t1;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (KIND == 0) v[k] = __builtin_fmaf(v[k], 0.9999f, 0.5f);
        if (KIND == 1) v[k] = __builtin_amdgcn_exp2f(v[k]);
        if (KIND == 2) v[k] = __builtin_amdgcn_rcpf(v[k]);
      }
    SB();
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float keep = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) keep += v[k];
  if (keep == 12345.678f) sink[0] = keep;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}

// the tile body.  MODE 0: free running; 1: LDS-only barrier between the MFMA phase and the pointwise phase (aligned);
// 2: waves 4-7 start half a body late (staggered); 3: as 0 but waves 4-7 at s_setprio(1)
template <int MODE, int NMF, int NFMA, int NTR>
__global__ __launch_bounds__(512) void body(int iters, unsigned long long* out, float* sink) {
  const int wave = threadIdx.x >> 6;
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const float x = (float)threadIdx.x * 1e-3f, y = 1.0f;
  float v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = x + (float)k * 0.01f;
  if (MODE == 3 && wave >= 4) __builtin_amdgcn_s_setprio(1);
  __syncthreads();
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  auto mf = [&](int n) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < n; k += 4) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
    }
    SB();
  };
  auto pw = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NFMA; ++k) v[k & 15] = __builtin_fmaf(v[k & 15], 0.9999f, 0.5f);
#pragma unroll
    for (int k = 0; k < NTR; ++k) v[k & 15] = (k & 1) ? __builtin_amdgcn_rcpf(v[k & 15]) : __builtin_amdgcn_exp2f(v[k & 15]);
    SB();
  };
  if (MODE == 2 && wave >= 4) mf(NMF / 2);          // half a body late
  for (int i = 0; i < iters; ++i) {
    mf(NMF);
    if (MODE == 1) asm volatile("s_barrier" ::: "memory");
    pw();
    if (MODE == 1) asm volatile("s_barrier" ::: "memory");
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float keep = a0[0] + a1[1] + a2[2] + a3[3];
#pragma unroll
  for (int k = 0; k < 16; ++k) keep += v[k];
  if (keep == 12345.678f) sink[0] = keep;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}

static unsigned long long* g_out; static float* g_sink;

template <int KIND>
void run_rate(const char* name) {
  unsigned long long h[8];
  const int it = 2000;
  for (int waves = 4; waves <= 8; waves += 4) {
    rate<KIND><<<256, 64 * waves>>>(it, g_out, g_sink);
    rate<KIND><<<256, 64 * waves>>>(it, g_out, g_sink);
    (void)hipMemcpy(h, g_out, 64, hipMemcpyDeviceToHost);
    printf("  %-10s %d wave(s) per SIMD: %.2f cycles per instruction per wave (wave0 %llu cycles for %d instructions)\n", name, waves / 4,
           h[0] / (64.0 * it), h[0], 64 * it);
  }
}

template <int MODE, int NMF, int NFMA, int NTR>
void run_body(const char* name, int threads = 512) {
  unsigned long long h[8];
  const int it = 1000;
  body<MODE, NMF, NFMA, NTR><<<256, threads>>>(it, g_out, g_sink);
  body<MODE, NMF, NFMA, NTR><<<256, threads>>>(it, g_out, g_sink);
  (void)hipMemcpy(h, g_out, 64, hipMemcpyDeviceToHost);
  const double per = (double)h[0] / it;
  const int nw = threads / 256;       // waves per SIMD
  printf("  %-52s wave0 %7.0f wave4 %7.0f cycles per body; MFMA share of the SIMD's time %.3f\n", name, per, threads > 256 ? (double)h[4] / it : 0.0,
         nw * NMF * 32.0 / per);
}

int main() {
  (void)hipMalloc(&g_out, 64); (void)hipMalloc(&g_sink, 4);
  printf("issue rate of independent instructions (16 chains), s_memtime ticks:\n");
  run_rate<0>("v_fma_f32"); run_rate<1>("v_exp_f32"); run_rate<2>("v_rcp_f32");
  printf("tile body = %d fp32 MFMAs (%d cycles of pipe) + 40 v_fma + 24 transcendental, per wave:\n", 24, 24 * 32);
  run_body<0, 24, 40, 24>("one wave per SIMD", 256);
  run_body<0, 24, 40, 24>("two waves per SIMD, free running");
  run_body<1, 24, 40, 24>("two waves per SIMD, barrier-aligned phases");
  run_body<2, 24, 40, 24>("two waves per SIMD, staggered by half a body");
  run_body<3, 24, 40, 24>("two waves per SIMD, waves 4-7 at s_setprio(1)");
  printf("the same with the pointwise part cut to 28 v_fma + 24 transcendental:\n");
  run_body<0, 24, 28, 24>("one wave per SIMD", 256);
  run_body<0, 24, 28, 24>("two waves per SIMD, free running");
  run_body<1, 24, 28, 24>("two waves per SIMD, barrier-aligned phases");
  printf("MFMA phase only / pointwise phase only (two waves per SIMD):\n");
  run_body<0, 24, 0, 0>("24 MFMAs");
  run_body<0, 0, 40, 24>("40 v_fma + 24 transcendental");
  run_body<0, 0, 40, 24>("40 v_fma + 24 transcendental, one wave per SIMD", 256);
  return 0;
}
