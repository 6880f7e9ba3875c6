// HBM WRITE ceiling on MI355X (diagnostic): what a store-dominated stream like the saving unroll (nine planes of 16-byte stores +
// a read of a seventh of the bytes) can reach.   hipcc -O3 --offload-arch=gfx950 -o wbw_probe wbw_probe.hip
//   contiguous: every workgroup streams its own slab, 16 B per lane;  planes: P planes, a wave writes 1 KiB pieces round robin over
//   the planes (the saving unroll's pattern);  mixed: as planes, plus a 16-byte load per R stores from a separate buffer
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(512) void wr(f32x4* dst, long n16, int planes, long plane_stride16, const f32x4* src, int read_every) {
  // n16: 16-byte elements per plane; the grid covers a plane in pieces of 64 elements (one wave-instruction), grid-stride
  const long waves = (long)gridDim.x * 8, w = (long)blockIdx.x * 8 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  f32x4 v = {1.f, 2.f, 3.f, (float)lane};
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const long per = (n16 / 64 + waves - 1) / waves;           // pieces per wave: a contiguous run (a slab per wave)
  // reads: one 16-byte load per `read_every` stores, issued a whole piece ahead and consumed after the next piece's stores were issued
  // (loads and stores retire through one in-order counter: a load consumed at once would wait for every store before it)
  long k = 0, rp = w * per;
  f32x4 pend = {0.f, 0.f, 0.f, 0.f};
  for (long p = w * per; p < (w + 1) * per && p * 64 < n16; ++p) {
    acc += pend;
    pend = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int pl = 0; pl < planes; ++pl) {
      f32x4* d = dst + pl * plane_stride16 + p * 64 + lane;
      if (NT) __builtin_nontemporal_store(v, d); else *d = v;
      if (read_every && (++k % read_every) == 0) { pend += src[(rp % (n16 / 64)) * 64 + lane]; ++rp; }
    }
  }
  acc += pend;
  if (acc[0] == 12345.f) dst[0] = acc;
}

int main() {
  const long plane_bytes = 512l << 20;                         // 512 MiB per plane
  const int maxp = 9;
  f32x4 *dst, *src;
  if (hipMalloc(&dst, plane_bytes * maxp) != hipSuccess || hipMalloc(&src, plane_bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(src, 0, plane_bytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const long n16 = plane_bytes / 16;
  struct { const char* name; int planes; int read_every; bool nt; int grid; } cfg[] = {
    {"contiguous, 1 plane", 1, 0, false, 256}, {"contiguous, 1 plane", 1, 0, false, 1024}, {"contiguous, 1 plane, nontemporal", 1, 0, true, 256},
    {"9 planes round robin", 9, 0, false, 256}, {"9 planes round robin", 9, 0, false, 768}, {"9 planes round robin, nontemporal", 9, 0, true, 256},
    {"9 planes + a 16 B load per 7 stores", 9, 7, false, 256}, {"9 planes + a 16 B load per 7 stores", 9, 7, false, 768},
  };
  for (auto& c : cfg) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      if (c.nt) wr<true><<<c.grid, 512>>>(dst, n16, c.planes, n16, src, c.read_every);
      else wr<false><<<c.grid, 512>>>(dst, n16, c.planes, n16, src, c.read_every);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    const double wb = (double)plane_bytes * c.planes, rb = c.read_every ? wb / c.read_every : 0.0;
    printf("%-40s grid %4d: %7.3f ms  writes %.2f TB/s  (writes + reads %.2f TB/s)\n", c.name, c.grid, best, wb / best / 1e9, (wb + rb) / best / 1e9);
  }
  return 0;
}
