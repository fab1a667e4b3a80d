// LayerNorm forward / backward (eps inside the sqrt, biased variance -- torch.nn.LayerNorm).
// Replaces nn.LayerNorm(d, eps=1e-6) at kantts/models/sambert/__init__.py:63,130,198 and
// kantts/models/sambert/kantts_sambert.py:58,128.
//
// HBM-bound streaming kernels: one 64-lane wave per row, the row is held in registers
// (C <= 1024 -> <= 16 values per lane), two-pass mean / variance like ATen's CPU kernel.
// Backward: per-block partial dgamma/dbeta are reduced through LDS, one atomicAdd per column/block.
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
