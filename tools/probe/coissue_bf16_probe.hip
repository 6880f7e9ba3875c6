// Do vector instructions hide under v_mfma_f32_16x16x32_bf16 on gfx950?  (diagnostic, not part of the product; the fp32-MFMA
// version is coissue_probe.hip)   hipcc -O3 --offload-arch=gfx950 -o coissue_bf16_probe coissue_bf16_probe.hip
//   pair:  waves 0-3 issue NM MFMAs (4 accumulators), their SIMD partners (waves 4-7) NV v_fma_f32: alone / alone / together
//   own<PER, WPS>: every wave issues PER independent v_fma_f32 after each of its MFMAs; WPS waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void pair(int nm, int nv, int prio, unsigned long long* out, float* sink) {
  const int wave = threadIdx.x >> 6;
  unsigned long long t0, t1;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  float keep = 0.f;
  if (wave < 4) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    bf8 x, y;
#pragma unroll
    for (int j = 0; j < 8; ++j) { x[j] = (__bf16)(float)(threadIdx.x + j); y[j] = (__bf16)1.0f; }
    for (int i = 0; i < nm; i += 4) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a3, 0, 0, 0);
    }
    keep = a0[0] + a1[1] + a2[2] + a3[3];
  } else {
    if (prio) __builtin_amdgcn_s_setprio(3);
    float v0 = (float)threadIdx.x, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f;
    for (int i = 0; i < nv; i += 4) {
      v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 1.0001f, 0.5f);
      v2 = __builtin_fmaf(v2, 1.0001f, 0.5f); v3 = __builtin_fmaf(v3, 1.0001f, 0.5f);
    }
    keep = v0 + v1 + v2 + v3;
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (keep == 12345.678f) sink[0] = keep;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}

template <int PER, int WPS>
__global__ __launch_bounds__(256 * WPS) void own(int nm, unsigned long long* out, float* sink) {
  unsigned long long t0, t1;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  bf8 x, y;
#pragma unroll
  for (int j = 0; j < 8; ++j) { x[j] = (__bf16)(float)(threadIdx.x + j); y[j] = (__bf16)1.0f; }
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = (float)threadIdx.x + (float)k;
  auto work = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < PER; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], 1.0001f, 0.5f);
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int i = 0; i < nm; i += 4) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a0, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); work();
    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a1, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); work();
    a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a2, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); work();
    a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a3, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); work();
  }
  float keep = a0[0] + a1[1] + a2[2] + a3[3];
#pragma unroll
  for (int k = 0; k < 8; ++k) keep += v[k];
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (keep == 12345.678f) sink[0] = keep;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}
// blocked: each wave issues BLK MFMAs back to back, then BLK * PER v_fma (the shape of an un-interleaved kernel body)
template <int PER, int WPS, int BLK, int PRIO>
__global__ __launch_bounds__(256 * WPS) void blocked(int nm, unsigned long long* out, float* sink) {
  unsigned long long t0, t1;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  bf8 x, y;
#pragma unroll
  for (int j = 0; j < 8; ++j) { x[j] = (__bf16)(float)(threadIdx.x + j); y[j] = (__bf16)1.0f; }
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = (float)threadIdx.x + (float)k;
  // waves of the second half of the workgroup start with the vector block (staggered by half a body)
  const bool second = WPS > 1 && (threadIdx.x >> 8) & 1;
  for (int i = 0; i < nm; i += BLK) {
    if (!second || i > 0) {
#pragma unroll
      for (int b = 0; b < BLK; b += 4) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a3, 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (PRIO == 1) __builtin_amdgcn_s_setprio(3);
    if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int k = 0; k < BLK * PER; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], 1.0001f, 0.5f);
    if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
    if (PRIO == 2) __builtin_amdgcn_s_setprio(3);
    __builtin_amdgcn_sched_barrier(0);
  }
  float keep = a0[0] + a1[1] + a2[2] + a3[3];
#pragma unroll
  for (int k = 0; k < 8; ++k) keep += v[k];
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (keep == 12345.678f) sink[0] = keep;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}


// the same questions for v_mfma_f32_32x32x16_bf16 (twice the FLOP of a 16x16x32 per instruction, 16 accumulator registers)
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int PER, int WPS>
__global__ __launch_bounds__(256 * WPS) void own32(int nm, unsigned long long* out, float* sink) {
  unsigned long long t0, t1;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  f32x16 a0, a1;
#pragma unroll
  for (int j = 0; j < 16; ++j) { a0[j] = 0.f; a1[j] = 0.f; }
  bf8 x, y;
#pragma unroll
  for (int j = 0; j < 8; ++j) { x[j] = (__bf16)(float)(threadIdx.x + j); y[j] = (__bf16)1.0f; }
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = (float)threadIdx.x + (float)k;
  auto work = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < PER; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], 1.0001f, 0.5f);
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int i = 0; i < nm; i += 2) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); work();
    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); work();
  }
  float keep = a0[0] + a1[1] + a0[15] + a1[14];
#pragma unroll
  for (int k = 0; k < 8; ++k) keep += v[k];
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (keep == 12345.678f) sink[0] = keep;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}
template <int PER, int WPS, int BLK>
__global__ __launch_bounds__(256 * WPS) void blocked32(int nm, unsigned long long* out, float* sink) {
  unsigned long long t0, t1;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  f32x16 a0, a1;
#pragma unroll
  for (int j = 0; j < 16; ++j) { a0[j] = 0.f; a1[j] = 0.f; }
  bf8 x, y;
#pragma unroll
  for (int j = 0; j < 8; ++j) { x[j] = (__bf16)(float)(threadIdx.x + j); y[j] = (__bf16)1.0f; }
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = (float)threadIdx.x + (float)k;
  const bool second = WPS > 1 && (threadIdx.x >> 8) & 1;
  for (int i = 0; i < nm; i += BLK) {
    if (!second || i > 0) {
#pragma unroll
      for (int b = 0; b < BLK; b += 2) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < BLK * PER; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], 1.0001f, 0.5f);
    __builtin_amdgcn_sched_barrier(0);
  }
  float keep = a0[0] + a1[1] + a0[15] + a1[14];
#pragma unroll
  for (int k = 0; k < 8; ++k) keep += v[k];
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (keep == 12345.678f) sink[0] = keep;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}

static unsigned long long* out; static float* sink;
template <int PER, int WPS> void run_own() {
  unsigned long long h[16];
  own<PER, WPS><<<256, 256 * WPS>>>(4096, out, sink);
  own<PER, WPS><<<256, 256 * WPS>>>(4096, out, sink);
  (void)hipMemcpy(h, out, 8 * 4 * WPS, hipMemcpyDeviceToHost);
  unsigned long long mx = 0; for (int i = 0; i < 4 * WPS; ++i) mx = h[i] > mx ? h[i] : mx;
  printf("  interleaved, %d wave(s) per SIMD, %d v_fma_f32 after each MFMA: %.1f cycles per MFMA of the slowest wave (%.1f per SIMD-MFMA)\n",
         WPS, PER, mx / 4096.0, mx / 4096.0 / WPS);
}
template <int PER, int WPS, int BLK, int PRIO = 0> void run_blocked() {
  unsigned long long h[16];
  blocked<PER, WPS, BLK, PRIO><<<256, 256 * WPS>>>(4096, out, sink);
  blocked<PER, WPS, BLK, PRIO><<<256, 256 * WPS>>>(4096, out, sink);
  (void)hipMemcpy(h, out, 8 * 4 * WPS, hipMemcpyDeviceToHost);
  unsigned long long mx = 0; for (int i = 0; i < 4 * WPS; ++i) mx = h[i] > mx ? h[i] : mx;
  printf("  blocks of %d MFMAs then %d v_fma_f32, %d wave(s) per SIMD (second staggered)%s: %.1f cycles per MFMA of the slowest wave (%.1f per SIMD-MFMA)\n",
         BLK, BLK * PER, WPS, PRIO == 1 ? ", vector blocks at s_setprio(3)" : PRIO == 2 ? ", MFMA blocks at s_setprio(3)" : "", mx / 4096.0, mx / 4096.0 / WPS);
}
template <int PER, int WPS> void run_own32() {
  unsigned long long h[16];
  own32<PER, WPS><<<256, 256 * WPS>>>(2048, out, sink);
  own32<PER, WPS><<<256, 256 * WPS>>>(2048, out, sink);
  (void)hipMemcpy(h, out, 8 * 4 * WPS, hipMemcpyDeviceToHost);
  unsigned long long mx = 0; for (int i = 0; i < 4 * WPS; ++i) mx = h[i] > mx ? h[i] : mx;
  printf("  32x32x16: interleaved, %d wave(s) per SIMD, %d v_fma_f32 after each MFMA: %.1f cycles per MFMA of the slowest wave (%.1f per SIMD-MFMA = %.1f per 16x16x32 of work)\n",
         WPS, PER, mx / 2048.0, mx / 2048.0 / WPS, mx / 2048.0 / WPS / 2);
}
template <int PER, int WPS, int BLK> void run_blocked32() {
  unsigned long long h[16];
  blocked32<PER, WPS, BLK><<<256, 256 * WPS>>>(2048, out, sink);
  blocked32<PER, WPS, BLK><<<256, 256 * WPS>>>(2048, out, sink);
  (void)hipMemcpy(h, out, 8 * 4 * WPS, hipMemcpyDeviceToHost);
  unsigned long long mx = 0; for (int i = 0; i < 4 * WPS; ++i) mx = h[i] > mx ? h[i] : mx;
  printf("  32x32x16: blocks of %d MFMAs then %d v_fma_f32, %d wave(s) per SIMD (second staggered): %.1f per SIMD-MFMA = %.1f per 16x16x32 of work\n",
         BLK, BLK * PER, WPS, mx / 2048.0 / WPS, mx / 2048.0 / WPS / 2);
}
int main() {
  (void)hipMalloc(&out, 256); (void)hipMalloc(&sink, 4);
  unsigned long long h[8];
  const int NM = 4096, NV = 16384;
  int cfg[3][2] = {{NM, 0}, {0, NV}, {NM, NV}};
  printf("pair: waves 0-3 %d bf16 MFMAs, SIMD partners %d v_fma_f32 (cycles of wave 0 / wave 4)\n", NM, NV);
  for (int prio = 0; prio < 2; ++prio)
  for (auto& c : cfg) {
    pair<<<256, 512>>>(c[0], c[1], prio, out, sink);
    pair<<<256, 512>>>(c[0], c[1], prio, out, sink);
    (void)hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    printf("  [mfma %5d, valu %5d, vector wave at s_setprio(%d)] wave0 %7llu wave4 %7llu\n", c[0], c[1], prio ? 3 : 0, h[0], h[4]);
  }
  run_own<0, 1>(); run_own<1, 1>(); run_own<2, 1>(); run_own<3, 1>(); run_own<4, 1>(); run_own<6, 1>(); run_own<8, 1>();
  run_own<0, 2>(); run_own<2, 2>(); run_own<3, 2>(); run_own<4, 2>(); run_own<6, 2>();
  run_blocked<4, 1, 24>(); run_blocked<4, 2, 24>(); run_blocked<2, 2, 24>(); run_blocked<4, 2, 96>();
  run_blocked<4, 2, 24, 1>(); run_blocked<4, 2, 24, 2>(); run_blocked<4, 2, 96, 1>(); run_blocked<4, 2, 96, 2>();
  run_blocked<6, 2, 24>(); run_blocked<6, 2, 24, 1>(); run_blocked<6, 2, 24, 2>();
  run_blocked<4, 3, 24>(); run_blocked<4, 4, 24>(); run_blocked<6, 3, 24>(); run_blocked<6, 4, 24>(); run_blocked<2, 4, 24>();
  run_own32<0, 1>(); run_own32<2, 1>(); run_own32<4, 1>(); run_own32<6, 1>(); run_own32<8, 1>(); run_own32<12, 1>();
  run_own32<0, 2>(); run_own32<4, 2>(); run_own32<6, 2>(); run_own32<8, 2>(); run_own32<12, 2>();
  run_blocked32<8, 1, 12>(); run_blocked32<8, 2, 12>(); run_blocked32<4, 2, 12>(); run_blocked32<12, 2, 12>();
  return 0;
}
