// HBM read-bandwidth probe (diagnostic, not part of the product): how many workgroups / bytes in flight a streaming
// reduction needs on MI355X, slab-contiguous vs grid-interleaved.   hipcc -O3 --offload-arch=gfx950 -o bw_probe bw_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U, bool SLAB>
__global__ __launch_bounds__(512) void rd(const f32x4* __restrict__ p, long n4, float* out) {
  const long nthr = (long)gridDim.x * blockDim.x;
  f32x4 acc = {0, 0, 0, 0};
  if (SLAB) {
    const long per = n4 / gridDim.x;                       // each block streams its own contiguous slab
    const f32x4* q = p + per * blockIdx.x;
    for (long i = threadIdx.x; i + (U - 1) * (long)blockDim.x < per; i += (long)U * blockDim.x) {
      f32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(q + i + (long)u * blockDim.x);
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u];
    }
  } else {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i + (U - 1) * nthr < n4; i += U * nthr) {
      f32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + i + u * nthr);
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u];
    }
  }
  float s = acc.x + acc.y + acc.z + acc.w;
  if (s == 12345.678f) out[0] = s;
}

template <int U, bool SLAB>
void run(const f32x4* p, long n4, float* out, int grid, int block) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) rd<U, SLAB><<<grid, block>>>(p, n4, out);
  (void)hipEventRecord(a);
  const int R = 5;
  for (int i = 0; i < R; ++i) rd<U, SLAB><<<grid, block>>>(p, n4, out);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  printf("%s U=%d grid=%5d block=%3d : %7.1f us  %6.2f TB/s\n", SLAB ? "slab " : "inter", U, grid, block, ms / R * 1e3,
         n4 * 16.0 / (ms / R * 1e-3) / 1e12);
}

int main() {
  const long bytes = 1440L << 20;                          // ~ the fc1 weight gradient's operands (1.42 GB)
  const long n4 = bytes / 16;
  f32x4* p; float* out;
  (void)hipMalloc(&p, bytes); (void)hipMalloc(&out, 4); (void)hipMemset(p, 0, bytes);
  for (int block : {256, 512})
    for (int grid : {256, 512, 1024, 2048, 8192}) {
      run<1, true>(p, n4, out, grid, block); run<4, true>(p, n4, out, grid, block); run<8, true>(p, n4, out, grid, block);
      run<4, false>(p, n4, out, grid, block); run<8, false>(p, n4, out, grid, block);
    }
  return 0;
}
