// LDS-DMA (global_load_lds_dwordx4) issue behaviour on gfx950 (diagnostic, not part of the product): a wave issues ND
// DMAs of 1 KB each from per-lane source addresses, (a) back to back, (b) with an LDS table read + s_waitcnt lgkmcnt(0)
// between consecutive DMAs (the shape of agent_fwd_kernel's first dma_fill), (c) as (b) with the table read typed so that
// hipcc inserts nothing else; cycles until all are ISSUED and until all have LANDED (s_waitcnt vmcnt(0)).
//   hipcc -O3 --offload-arch=gfx950 -o dma_probe dma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))

template <int MODE>
__global__ __launch_bounds__(512) void k(const float* __restrict__ g, long stride, int nd, unsigned long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* In = smem;                       // nd KB
  int* tab = reinterpret_cast<int*>(smem + 16 * 256);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < 1024; e += 512) tab[e] = e * 4;
  __syncthreads();
  unsigned long long t0, t1, t2;
  const float* base = g + (long)blockIdx.x * stride + (long)wave * 16 * 64 * 4;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  if (wave >= 4) {
    for (int k2 = 0; k2 < nd; ++k2) {
      int off = lane * 4 + k2 * 256;
      if (MODE == 1) { off = tab[(lane + k2 * 64) & 1023]; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
      __builtin_amdgcn_global_load_lds(base + off, LDSP(In + (wave - 4) * 4096 + k2 * 256), 16, 0, 0);
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
  __syncthreads();
  float acc = In[tid] + In[tid + 2048];
  if (acc == 12345.678f) sink[0] = acc;
  if (blockIdx.x == 0 && lane == 0) { out[wave * 2] = t1 - t0; out[wave * 2 + 1] = t2 - t0; }
}

int main() {
  const long stride = 1 << 20;           // floats per workgroup: every DMA misses the caches
  float* g; unsigned long long* out; float* sink;
  (void)hipMalloc(&g, 256 * stride * 4); (void)hipMemset(g, 0, 256 * stride * 4);
  (void)hipMalloc(&out, 256); (void)hipMalloc(&sink, 4);
  unsigned long long h[16];
  for (int mode = 0; mode < 2; ++mode)
    for (int nd : {1, 2, 4, 8, 12}) {
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) k<0><<<256, 512, 80 * 1024>>>(g + rep * 4096, stride, nd, out, sink);
        else k<1><<<256, 512, 80 * 1024>>>(g + rep * 4096 + 2048, stride, nd, out, sink);
      }
      (void)hipMemcpy(h, out, 128, hipMemcpyDeviceToHost);
      printf("%-46s %2d DMAs per wave (4 issuing waves): issued after %6llu cycles, landed after %6llu\n",
             mode ? "table read + lgkmcnt(0) between DMAs:" : "back to back:", nd, h[8], h[9]);
    }
  return 0;
}
