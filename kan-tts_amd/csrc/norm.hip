// LayerNorm forward / backward (eps inside the sqrt, biased variance -- torch.nn.LayerNorm).
// Replaces nn.LayerNorm(d, eps=1e-6) at kantts/models/sambert/__init__.py:63,130,198 and
// kantts/models/sambert/kantts_sambert.py:58,128.
//
// HBM-bound streaming kernels: one 64-lane wave per row, the row is held in registers
// (C <= 1024 -> <= 16 values per lane), two-pass mean / variance like ATen's CPU kernel.
// Backward: per-block partial dgamma/dbeta are reduced through LDS, one atomicAdd per column/block.
#include <stdlib.h>

#include "common.h"

#define LN_MAXPL 16  // max elements per lane (C <= 1024)

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ y,
                                                           float* __restrict__ mean, float* __restrict__ rstd, int M,
                                                           int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (long long)row * C;
  float v[LN_MAXPL];
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < LN_MAXPL; ++e) {
    int c = lane + e * 64;
    v[e] = (c < C) ? xr[c] : 0.f;
    s += v[e];
  }
  s = kantts_wave_sum(s);
  const float mu = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < LN_MAXPL; ++e) {
    int c = lane + e * 64;
    float d = (c < C) ? (v[e] - mu) : 0.f;
    q += d * d;
  }
  q = kantts_wave_sum(q);
  const float rs = 1.0f / sqrtf(q / (float)C + eps);
  float* yr = y + (long long)row * C;
#pragma unroll
  for (int e = 0; e < LN_MAXPL; ++e) {
    int c = lane + e * 64;
    if (c < C) yr[c] = (v[e] - mu) * rs * gamma[c] + beta[c];
  }
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
}

// rows_per_block rows are walked by the block's 4 waves; dgamma/dbeta partials live in registers.
#define LN_BWD_WAVES 16
__global__ __launch_bounds__(64 * LN_BWD_WAVES) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, float* __restrict__ dx,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int M,
                                                           int C, int rows_per_block) {
  // 16 waves per block: a block still owns a contiguous slab of rows (one block per CU keeps the 2*C same-address
  // atomics per block rare), but a wave now walks ~2 rows instead of ~7 -- the kernel is a chain of dependent
  // row loads, so its duration follows rows-per-wave
  __shared__ float red[2][LN_BWD_WAVES][64 * LN_MAXPL / 4];  // [dg|db][wave][col chunk]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * rows_per_block;
  float pg[LN_MAXPL], pb[LN_MAXPL], gm[LN_MAXPL];
#pragma unroll
  for (int e = 0; e < LN_MAXPL; ++e) {
    int c = lane + e * 64;
    pg[e] = 0.f;
    pb[e] = 0.f;
    gm[e] = (c < C) ? gamma[c] : 0.f;
  }
  for (int r = wave; r < rows_per_block; r += LN_BWD_WAVES) {
    const int row = row0 + r;
    if (row >= M) break;
    const float mu = mean[row], rs = rstd[row];
    const float* xr = x + (long long)row * C;
    const float* dr = dy + (long long)row * C;
    float xh[LN_MAXPL], g[LN_MAXPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < LN_MAXPL; ++e) {
      int c = lane + e * 64;
      float d = 0.f, xv = 0.f;
      if (c < C) {
        d = dr[c];
        xv = xr[c];
      }
      xh[e] = (c < C) ? (xv - mu) * rs : 0.f;
      g[e] = d * gm[e];
      s1 += g[e];
      s2 += g[e] * xh[e];
      pg[e] += d * xh[e];
      pb[e] += d;
    }
    s1 = kantts_wave_sum(s1) / (float)C;
    s2 = kantts_wave_sum(s2) / (float)C;
    float* dxr = dx + (long long)row * C;
#pragma unroll
    for (int e = 0; e < LN_MAXPL; ++e) {
      int c = lane + e * 64;
      if (c < C) dxr[c] = rs * (g[e] - s1 - xh[e] * s2);
    }
  }
  // cross-wave reduction of the parameter-gradient partials, 4 column-chunks of 256 at a time
  for (int chunk = 0; chunk < LN_MAXPL / 4; ++chunk) {
    if (chunk * 256 >= C) break;
    __syncthreads();
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
      red[0][wave][e4 * 64 + lane] = pg[chunk * 4 + e4];
      red[1][wave][e4 * 64 + lane] = pb[chunk * 4 + e4];
    }
    __syncthreads();
    // thread t handles column chunk*256 + t;  note column = lane + e*64  <->  e4*64 + lane
    int c = chunk * 256 + threadIdx.x;
    if (threadIdx.x < 256 && c < C) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < LN_BWD_WAVES; ++w) {
        a += red[0][w][threadIdx.x];
        b += red[1][w][threadIdx.x];
      }
      atomicAdd(&dgamma[c], a);
      atomicAdd(&dbeta[c], b);
    }
  }
}

extern "C" int kantts_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                    float* rstd, int M, int C, float eps, void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || M < 0 || C < 1) return KANTTS_E_BADARG;
  if (C > 64 * LN_MAXPL) return KANTTS_E_UNSUPPORTED;
  if (M == 0) return KANTTS_OK;
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(kantts_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta,
                     y, mean, rstd, M, C, eps);
  KANTTS_CHECK_LAUNCH();
}

extern "C" int kantts_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                                    const float* rstd, float* dx, float* dgamma_accum, float* dbeta_accum, int M, int C,
                                    void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma_accum || !dbeta_accum || M < 0 || C < 1)
    return KANTTS_E_BADARG;
  if (C > 64 * LN_MAXPL) return KANTTS_E_UNSUPPORTED;
  if (M == 0) return KANTTS_OK;
  // about one block per CU: every block ends with 2*C same-address atomics, so more blocks only add contention
  int rows_per_block = kantts_cdiv(M, 256);
  if (rows_per_block < LN_BWD_WAVES) rows_per_block = LN_BWD_WAVES;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(kantts_cdiv(M, rows_per_block)), dim3(64 * LN_BWD_WAVES), 0, (hipStream_t)stream, dy,
                     x, gamma, mean, rstd, dx, dgamma_accum, dbeta_accum, M, C, rows_per_block);
  KANTTS_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------------
// C = 128 specialisation (every LayerNorm of SAM-BERT except the first encoder layer's 512-wide input), round 2.
// 16 lanes own one row (8 channels each: two float4 loads, one 16-byte bf16 store), a wave normalises 4 rows at once and
// reduces over 16 lanes with 4 xor-shuffles; the generic kernel above spends a whole wave (and 6-step reductions) on a
// 512-byte row.  The output / incoming gradient may be bf16: LayerNorm outputs only feed contractions
// (csrc/gemm_bf16.hip), so the normalised activations never exist in fp32 in HBM.
typedef unsigned int ln_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned ln_pack2(float a, float b) {
  typedef __bf16 v2 __attribute__((ext_vector_type(2)));
  v2 t = {(__bf16)a, (__bf16)b};
  return (unsigned&)t;
}
__device__ __forceinline__ float ln_sum16(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}

template <bool OUT_BF16>
__global__ __launch_bounds__(256) void ln128_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, void* __restrict__ y,
                                                       float* __restrict__ mean, float* __restrict__ rstd, int M, float eps) {
  const int sub = threadIdx.x & 15;
  const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool live = row < M;
  const int c0 = sub * 8;
  float v[8];
  {
    const float4* p = reinterpret_cast<const float4*>(x + (long long)(live ? row : 0) * 128 + c0);
    const float4 a = p[0], b = p[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s += v[e];
  const float mu = ln_sum16(s) * (1.f / 128.f);
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float d = v[e] - mu;
    q += d * d;
  }
  const float rs = 1.0f / sqrtf(ln_sum16(q) * (1.f / 128.f) + eps);
  if (!live) return;
  const float4 g0 = *reinterpret_cast<const float4*>(gamma + c0), g1 = *reinterpret_cast<const float4*>(gamma + c0 + 4);
  const float4 b0 = *reinterpret_cast<const float4*>(beta + c0), b1 = *reinterpret_cast<const float4*>(beta + c0 + 4);
  const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
  const float bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (v[e] - mu) * rs * gm[e] + bt[e];
  if (OUT_BF16) {
    ln_u32x4 w = {ln_pack2(o[0], o[1]), ln_pack2(o[2], o[3]), ln_pack2(o[4], o[5]), ln_pack2(o[6], o[7])};
    *reinterpret_cast<ln_u32x4*>(reinterpret_cast<__bf16*>(y) + (long long)row * 128 + c0) = w;
  } else {
    float4* yp = reinterpret_cast<float4*>(reinterpret_cast<float*>(y) + (long long)row * 128 + c0);
    yp[0] = make_float4(o[0], o[1], o[2], o[3]);
    yp[1] = make_float4(o[4], o[5], o[6], o[7]);
  }
  if (sub == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
}

// backward: grid-stride over 16-row slabs; dgamma / dbeta partials stay in registers, are reduced over the block's 16
// row groups through LDS and leave as 256 atomics per block.
template <bool DY_BF16>
__global__ __launch_bounds__(256) void ln128_bwd_kernel(const void* __restrict__ dy, const float* __restrict__ x,
                                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ dres,
                                                       float* __restrict__ dx, float* __restrict__ dgamma,
                                                       float* __restrict__ dbeta, const unsigned char* __restrict__ zero_rows,
                                                       int M) {
  __shared__ float red[2][16][128];
  const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int c0 = sub * 8;
  const float4 g0 = *reinterpret_cast<const float4*>(gamma + c0), g1 = *reinterpret_cast<const float4*>(gamma + c0 + 4);
  const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
  float pg[8], pb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) pg[e] = pb[e] = 0.f;
  for (int r0 = blockIdx.x * 16; r0 < M; r0 += gridDim.x * 16) {
    const int row = r0 + grp;
    const bool live = row < M;
    const long long off = (long long)(live ? row : 0) * 128 + c0;
    float d[8], xv[8];
    if (DY_BF16) {
      const ln_u32x4 q = *reinterpret_cast<const ln_u32x4*>(reinterpret_cast<const __bf16*>(dy) + off);
      d[0] = __uint_as_float(q.x << 16); d[1] = __uint_as_float(q.x & 0xffff0000u);
      d[2] = __uint_as_float(q.y << 16); d[3] = __uint_as_float(q.y & 0xffff0000u);
      d[4] = __uint_as_float(q.z << 16); d[5] = __uint_as_float(q.z & 0xffff0000u);
      d[6] = __uint_as_float(q.w << 16); d[7] = __uint_as_float(q.w & 0xffff0000u);
    } else {
      const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(dy) + off);
      const float4 a = p[0], b = p[1];
      d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
    }
    {
      const float4* p = reinterpret_cast<const float4*>(x + off);
      const float4 a = p[0], b = p[1];
      xv[0] = a.x; xv[1] = a.y; xv[2] = a.z; xv[3] = a.w; xv[4] = b.x; xv[5] = b.y; xv[6] = b.z; xv[7] = b.w;
    }
    const float mu = mean[live ? row : 0], rs = rstd[live ? row : 0];
    float xh[8], gg[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (!live) d[e] = 0.f;
      xh[e] = (xv[e] - mu) * rs;
      gg[e] = d[e] * gm[e];
      s1 += gg[e];
      s2 += gg[e] * xh[e];
      pg[e] += d[e] * xh[e];
      pb[e] += d[e];
    }
    s1 = ln_sum16(s1) * (1.f / 128.f);
    s2 = ln_sum16(s2) * (1.f / 128.f);
    if (live) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = rs * (gg[e] - s1 - xh[e] * s2);
      if (dres) {  // gradient of the residual branch that by-passes the normalisation: summed here, not by autograd
        const float4* rp = reinterpret_cast<const float4*>(dres + off);
        const float4 a = rp[0], b = rp[1];
        o[0] += a.x; o[1] += a.y; o[2] += a.z; o[3] += a.w; o[4] += b.x; o[5] += b.y; o[6] += b.z; o[7] += b.w;
      }
      // the producer of x zeroes these rows of its output, so it would zero them of its incoming gradient first thing in
      // its backward: done here on its behalf (dgamma / dbeta above are this node's own and are not affected)
      if (zero_rows && zero_rows[row]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
      }
      float4* op = reinterpret_cast<float4*>(dx + off);
      op[0] = make_float4(o[0], o[1], o[2], o[3]);
      op[1] = make_float4(o[4], o[5], o[6], o[7]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[0][grp][c0 + e] = pg[e];
    red[1][grp][c0 + e] = pb[e];
  }
  __syncthreads();
  const int which = threadIdx.x >> 7, c = threadIdx.x & 127;
  float a = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) a += red[which][w][c];
  if (a != 0.f) atomicAdd(which ? &dbeta[c] : &dgamma[c], a);
}

extern "C" int kantts_ln128_fwd(const float* x, const float* gamma, const float* beta, void* y, int y_bf16, float* mean,
                                float* rstd, int M, float eps, void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || M < 0) return KANTTS_E_BADARG;
  if (M == 0) return KANTTS_OK;
  if (y_bf16)
    hipLaunchKernelGGL(ln128_fwd_kernel<true>, dim3(kantts_cdiv(M, 16)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y,
                       mean, rstd, M, eps);
  else
    hipLaunchKernelGGL(ln128_fwd_kernel<false>, dim3(kantts_cdiv(M, 16)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta,
                       y, mean, rstd, M, eps);
  KANTTS_CHECK_LAUNCH();
}

extern "C" int kantts_ln128_bwd_rows(const void* dy, int dy_bf16, const float* x, const float* gamma, const float* mean,
                                     const float* rstd, const float* dres, float* dx, float* dgamma_accum,
                                     float* dbeta_accum, const unsigned char* zero_rows, int M, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma_accum || !dbeta_accum || M < 0) return KANTTS_E_BADARG;
  if (M == 0) return KANTTS_OK;
  // every block ends with 256 atomics onto the SAME 256 addresses: with one block per 16-row slab (408 at the decoder's
  // 6528 rows) the kernel took 13 us against 4.5 us for the forward pass -- same-address atomics serialise in L2
  // (profiles/r02_runD_sambert_kernel_stats_top.csv).  128 blocks walk ~3 slabs each instead.
  int blocks = kantts_cdiv(M, 16);
  static const char* cap_env = getenv("KANTTS_LN_BWD_BLOCKS");  // experiment switch
  const int cap = cap_env ? atoi(cap_env) : 128;
  if (blocks > cap) blocks = cap;
  if (dy_bf16)
    hipLaunchKernelGGL(ln128_bwd_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, x, gamma, mean, rstd,
                       dres, dx, dgamma_accum, dbeta_accum, zero_rows, M);
  else
    hipLaunchKernelGGL(ln128_bwd_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, x, gamma, mean, rstd,
                       dres, dx, dgamma_accum, dbeta_accum, zero_rows, M);
  KANTTS_CHECK_LAUNCH();
}

extern "C" int kantts_ln128_bwd(const void* dy, int dy_bf16, const float* x, const float* gamma, const float* mean,
                                const float* rstd, const float* dres, float* dx, float* dgamma_accum, float* dbeta_accum,
                                int M, void* stream) {
  return kantts_ln128_bwd_rows(dy, dy_bf16, x, gamma, mean, rstd, dres, dx, dgamma_accum, dbeta_accum, nullptr, M, stream);
}
