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

// Marks a point where lanes of ONE wave read LDS cells that other lanes of the same wave wrote just before, relying on a
// wave's LDS operations executing in order (no barrier instruction on the device).  Expands to nothing here; the host
// build of the kernel sources used by the CPU tests (tests/hipemu) pre-defines it as a rendezvous of the wave's lanes.
#ifndef KANTTS_WAVE_ORDERED
#define KANTTS_WAVE_ORDERED()
#endif

// Makes the compiler forget what it knows about a vector-register value at this point (no instruction is emitted): used to
// keep loop-invariant operand shuffles from being hoisted into extra registers.  The host build of the kernel sources
// (tests/hipemu) pre-defines it as nothing.
#ifndef KANTTS_OPAQUE_VGPR
#define KANTTS_OPAQUE_VGPR(x) asm volatile("" : "+v"(x))
#endif

static inline int kantts_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------
// Counter-based RNG for dropout.  The same (seed, element) pair is re-evaluated in backward, so masks are never stored.
// One 64-bit hash (two rounds of a "splitmix" finaliser) serves FOUR consecutive elements, 16 bits each (round 2: the
// 64-bit multiplies of one hash per element were 5 us of an 18 us fused feed-forward launch and a third of the ALU work
// of an attention key): element idx takes bits [16*(idx & 3), +16) of mix(seed, idx >> 2) and is dropped iff that value
// is below p * 2^16.  kantts_dropout_scale is the definition; _scale4 / KanttsDropSeq evaluate it for aligned groups /
// sequential walks with one hash per group.  oracle/cabi_numpy.py::dropout_scale is the numpy twin.
__device__ __forceinline__ uint64_t kantts_rng_mix(uint64_t seed, uint64_t blk) {
  uint64_t z = seed + blk * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t kantts_drop_thr(float p) { return (uint32_t)fminf(p * 65536.0f, 65535.0f); }
// keep-scale for dropout prob p: returns 0 (dropped) or 1/(1-p)
__device__ __forceinline__ float kantts_dropout_scale(float p, uint64_t seed, uint64_t idx) {
  if (p <= 0.f) return 1.f;
  const uint32_t r = (uint32_t)(kantts_rng_mix(seed, idx >> 2) >> ((idx & 3) * 16)) & 0xFFFFu;
  return (r < kantts_drop_thr(p)) ? 0.f : 1.f / (1.f - p);
}
// v[0..3] *= keep-scale of elements base .. base + 3; base % 4 == 0
__device__ __forceinline__ void kantts_dropout_scale4(float p, uint64_t seed, uint64_t base, float* v) {
  if (p <= 0.f) return;
  const uint64_t z = kantts_rng_mix(seed, base >> 2);
  const uint32_t thr = kantts_drop_thr(p);
  const float keep = 1.f / (1.f - p);
  const uint32_t lo = (uint32_t)z, hi = (uint32_t)(z >> 32);
  v[0] *= ((lo & 0xFFFFu) < thr) ? 0.f : keep;
  v[1] *= ((lo >> 16) < thr) ? 0.f : keep;
  v[2] *= ((hi & 0xFFFFu) < thr) ? 0.f : keep;
  v[3] *= ((hi >> 16) < thr) ? 0.f : keep;
}
// sequential walk over consecutive elements (attention keys of one query): the hash is recomputed when idx >> 2 changes
struct KanttsDropSeq {
  float p, keep;
  uint32_t thr;
  uint64_t seed, blk, z;
  __device__ __forceinline__ KanttsDropSeq(float p_, uint64_t seed_)
      : p(p_), keep(p_ > 0.f ? 1.f / (1.f - p_) : 1.f), thr(kantts_drop_thr(p_)), seed(seed_), blk(~0ull), z(0ull) {}
  __device__ __forceinline__ float scale(uint64_t idx) {
    if (p <= 0.f) return 1.f;
    const uint64_t b = idx >> 2;
    if (b != blk) {
      blk = b;
      z = kantts_rng_mix(seed, b);
    }
    const uint32_t r = (uint32_t)(z >> ((idx & 3) * 16)) & 0xFFFFu;
    return (r < thr) ? 0.f : keep;
  }
};

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
