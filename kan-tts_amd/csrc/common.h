// Shared device/host helpers for libkantts_hip.so (gfx950 / CDNA4 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/kantts_hip.h"

#define KANTTS_WAVE 64

#define KANTTS_CHECK_LAUNCH()                         \
  do {                                                \
    hipError_t _e = hipGetLastError();                \
    if (_e != hipSuccess) return (int)_e;             \
    return KANTTS_OK;                                 \
  } while (0)

static inline int kantts_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------
// Counter-based RNG for dropout: one 32-bit hash per (seed, stream, element).  The same triple is
// re-evaluated in backward, so masks are never stored.  (Two rounds of a 64-bit mix, "splitmix"
// finaliser; statistical quality is checked by tests/test_dropout_stats.py on the GPU.)
__device__ __forceinline__ uint32_t kantts_rng_u32(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}
// keep-scale for dropout prob p: returns 0 (dropped) or 1/(1-p)
__device__ __forceinline__ float kantts_dropout_scale(float p, uint64_t seed, uint64_t idx) {
  if (p <= 0.f) return 1.f;
  uint32_t r = kantts_rng_u32(seed, idx);
  // drop iff r < p * 2^32
  uint32_t thr = (uint32_t)fminf(p * 4294967296.0f, 4294967295.0f);
  return (r < thr) ? 0.f : 1.f / (1.f - p);
}

// ---------------------------------------------------------------------------------------------
// Wave / block reductions (64-lane wave).
__device__ __forceinline__ float kantts_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float kantts_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
// Block-wide sum; `red` must hold >= blockDim.x/64 floats of LDS. Result valid in all threads.
__device__ __forceinline__ float kantts_block_sum(float v, float* red) {
  v = kantts_wave_sum(v);
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
