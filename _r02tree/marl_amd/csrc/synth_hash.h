// Counter-based hash of the synthetic SMAC-shaped environment (shared by rollout.hip and
// rollout_fused.hip; numpy restatement: oracle/rollout.py mix32/key/u01 - bit for bit).
#pragma once
enum { ST_OBS = 0, ST_STATE, ST_AVAIL, ST_REWARD, ST_LEN, ST_WON, ST_EXPLORE, ST_PICK };

__host__ __device__ inline unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
// prefix over (seed, stream, env, t); hfin() adds the element index
__host__ __device__ inline unsigned hprefix(unsigned seed, unsigned stream, unsigned env, unsigned t) {
  unsigned h = mix32(seed + stream * 0x9E3779B1u);
  h = mix32(h + env * 0x85EBCA77u + 1u);
  h = mix32(h + t * 0xC2B2AE3Du + 2u);
  return h;
}
__host__ __device__ inline unsigned hfin(unsigned prefix, unsigned idx) { return mix32(prefix + idx * 0x27D4EB2Fu + 3u); }
__host__ __device__ inline unsigned hkey(unsigned seed, unsigned stream, unsigned env, unsigned t, unsigned idx) {
  return hfin(hprefix(seed, stream, env, t), idx);
}
__host__ __device__ inline float u01(unsigned h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }
