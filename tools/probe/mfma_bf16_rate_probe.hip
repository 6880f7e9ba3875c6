// Wall-clock rate of v_mfma_f32_16x16x32_bf16 with 1 / 2 / 4 waves per SIMD (register operands, 4 accumulators per wave):
// hipEvent time of the launch -> TFLOP/s of the chip, next to the per-wave s_memtime cycles per MFMA.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_bf16_rate_probe mfma_bf16_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
__global__ void rate(int n, unsigned long long* out, float* sink) {
  bf8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)((float)(threadIdx.x + j) * 1e-3f); b[j] = (__bf16)((float)j * 1e-3f); }
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; i += 4) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const float keep = c0[0] + c1[1] + c2[2] + c3[3];
  if (keep == 12345.678f) sink[0] = keep;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}
int main() {
  unsigned long long* out; float* sink; unsigned long long h[16];
  (void)hipMalloc(&out, 128); (void)hipMalloc(&sink, 4);
  const int N = 1 << 18;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int nw = 1; nw <= 4; nw *= 2) {
    rate<<<256, 256 * nw>>>(N, out, sink);
    (void)hipEventRecord(e0);
    rate<<<256, 256 * nw>>>(N, out, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h, out, 128, hipMemcpyDeviceToHost);
    const double flop = 256.0 * 4 * nw * (double)N * 16384.0;
    printf("%d wave(s) per SIMD: %.3f ms  %.0f TFLOP/s chip-wide; s_memtime cycles per MFMA of wave 0: %.1f, of the last wave: %.1f; implied clock %.2f GHz\n",
           nw, ms, flop / ms / 1e9, (double)h[0] / N, (double)h[4 * nw - 1] / N, (double)h[0] / (ms * 1e6));
  }
  return 0;
}
