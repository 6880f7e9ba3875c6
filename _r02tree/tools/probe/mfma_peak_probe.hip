// Sustained fp32 MFMA rate of the whole chip (calibration for the roofline numbers in DESIGN.md section 5).
// Every SIMD of every CU runs WPS waves issuing independent v_mfma_f32_16x16x4_f32 back to back for ~10 ms; reports the
// HIP-event rate, the s_memtime ticks per MFMA and the shader clock those ticks imply (s_memrealtime = 100 MHz).
//   hipcc -O3 --offload-arch=gfx950 -o mfma_peak_probe mfma_peak_probe.hip && ./mfma_peak_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256, 2) void mfma_loop(int iters, float* out, unsigned long long* ticks) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float a = threadIdx.x * 1e-9f, b = 1e-9f;
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.f) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = r1 - r0; }
}

template <int NACC>
void run(int wgs_per_cu, int iters) {
  float* out; unsigned long long *ticks, h[2];
  hipMalloc(&out, 4); hipMalloc(&ticks, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wgs_per_cu;
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, iters / 8, out, ticks);      // warm-up
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, iters, out, ticks);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
  const double per_wave = (double)iters * 4 * NACC;
  const double flops = per_wave * 2048.0 * 4 * grid;
  printf("  %d workgroup(s) of 4 waves per CU, %d independent accumulators: %.3f ms  %.1f TFLOP/s  | wave 0: %.2f s_memtime ticks per MFMA it issued, shader clock %.0f MHz\n",
         wgs_per_cu, NACC, ms, flops / ms / 1e9, (double)h[0] / per_wave, (double)h[0] / (double)h[1] * 100.0);
  hipFree(out); hipFree(ticks);
}

int main() {
  printf("fp32 MFMA 16x16x4 (2048 FLOP, 32 pipe cycles), all 256 CUs x 4 SIMDs; nominal 157.3 TFLOP/s = 256 FLOP/cycle/CU at 2.4 GHz\n");
  run<4>(1, 40000);
  run<8>(1, 20000);
  run<4>(2, 20000);
  run<8>(2, 10000);
  return 0;
}
