// Convolutions with ONE output channel: the `conv_post` layers of HiFi-GAN (round 4).
//
//   generator            Conv1d(32 -> 1, k = 7) after F.leaky_relu(x, 0.01)   kantts/models/hifigan/hifigan.py:178-180
//   period discriminator Conv2d(1024 -> 1, (3, 1))                             kantts/models/hifigan/hifigan.py:238-262
//   scale discriminator  Conv1d(1024 -> 1, k = 3)                              kantts/models/hifigan/hifigan.py:373-402
//
// On the windowed-GEMM kernel (conv_win.hip) such a layer is ONE output tile, i.e. one workgroup for the whole tensor: 88 us
// per launch, 24 launches per GAN step, and the gradients went to the scalar fall-back kernels (conv_direct_kernel: 0.33 TB/s,
// conv_wgrad_direct_kernel: 1-4 workgroups per launch) -- 3.9 ms of kernel time per step on the branch streams of the
// discriminators (profiles/r04_runV: kernel trace of the captured GAN step).  A one-channel convolution is a dot product
// per output position, bound by reading x once:
//   forward          y[b,q,p]   = bias + sum_k sum_c w[k][c] act(x[b, q s + k d - pad, p, c])
//   input gradient   dx[b,t,p,c] = act'(x[b,t,p,c]) sum_k w[k][c] dy[b, (t + pad - k d) / s, p]       (exact divisions only)
//   weight gradient  dw[k][c]  += sum_{b,t,p} act(x[b,t,p,c]) dy[b, (t + pad - k d) / s, p];   db += sum dy
// x (B, Tsrc, inner, Cin) fp32 channels-last, Cin % 4 == 0, Cin <= 1024; y / dy (B, Tdst, inner).  The weight is addressed as
// w[k * w_ks + c * w_cs] (tap-major (K, 1, Cin): w_ks = Cin, w_cs = 1; parameter layout (1, Cin, K): w_ks = 1, w_cs = K).
#include "common.h"

#define N1_THREADS 256
#define N1_MAXW 16  // float4 weight registers per lane in the forward kernel: (Cin / 4 / lanes per token) * K

__device__ __forceinline__ float n1_act(float v, int on, float slope) { return (on && v < 0.f) ? v * slope : v; }

__device__ __forceinline__ float4 n1_w4(const kantts_conv_n1_args& g, int k, int c) {
  const float* p = g.w + (long long)k * g.w_ks + (long long)c * g.w_cs;
  if (g.w_cs == 1) return *reinterpret_cast<const float4*>(p);
  return make_float4(p[0], p[g.w_cs], p[2 * g.w_cs], p[3 * g.w_cs]);
}

// ---- forward: LC = min(64, Cin / 4) lanes share an output position, 64 / LC positions per wave, weights in registers
__global__ __launch_bounds__(N1_THREADS) void conv_n1_fwd_kernel(const kantts_conv_n1_args g) {
  const int C4 = g.Cin >> 2;
  const int LC = C4 < 64 ? C4 : 64;      // lanes per output position (a power of two: checked by the launcher)
  const int PER = C4 / LC;               // float4 columns per lane
  const int lane = threadIdx.x & 63, li = lane % LC, slot = lane / LC, TPW = 64 / LC;
  float4 wr[N1_MAXW];
#pragma unroll
  for (int i = 0; i < N1_MAXW; ++i) {
    const int k = i / PER, j = i % PER;
    wr[i] = (i < PER * g.K) ? n1_w4(g, k, 4 * (li + j * LC)) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float bias = g.bias ? g.bias[0] : 0.f;
  const long long total = (long long)g.B * g.Tdst * g.inner;
  const long long wave0 = ((long long)blockIdx.x * (N1_THREADS / 64) + (threadIdx.x >> 6)) * TPW;
  const long long step = (long long)gridDim.x * (N1_THREADS / 64) * TPW;
  for (long long m0 = wave0; m0 < total; m0 += step) {
    const long long m = m0 + slot;
    const bool live = m < total;
    const long long mm = live ? m : 0;
    const int p = (int)(mm % g.inner);
    const long long bq = mm / g.inner;
    const int q = (int)(bq % g.Tdst), b = (int)(bq / g.Tdst);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < N1_MAXW; ++i) {
      if (i < PER * g.K) {  // uniform
        const int k = i / PER, j = i % PER;
        const int t = q * g.stride + k * g.dil - g.pad;
        if (live && t >= 0 && t < g.Tsrc) {
          const float4 v = *reinterpret_cast<const float4*>(g.x + (((long long)b * g.Tsrc + t) * g.inner + p) * g.Cin +
                                                            4 * (li + j * LC));
          acc += n1_act(v.x, g.in_act, g.in_slope) * wr[i].x + n1_act(v.y, g.in_act, g.in_slope) * wr[i].y +
                 n1_act(v.z, g.in_act, g.in_slope) * wr[i].z + n1_act(v.w, g.in_act, g.in_slope) * wr[i].w;
        }
      }
    }
    for (int off = LC >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (live && li == 0) g.y[m] = acc + bias;
  }
}

// ---- input gradient: one thread per float4 of dx
__global__ __launch_bounds__(N1_THREADS) void conv_n1_dgrad_kernel(const kantts_conv_n1_args g) {
  const int C4 = g.Cin >> 2;
  const long long total = (long long)g.B * g.Tsrc * g.inner * C4;
  for (long long i = (long long)blockIdx.x * N1_THREADS + threadIdx.x; i < total; i += (long long)gridDim.x * N1_THREADS) {
    const int c4 = (int)(i % C4);
    const long long row = i / C4;
    const int p = (int)(row % g.inner);
    const long long bt = row / g.inner;
    const int t = (int)(bt % g.Tsrc), b = (int)(bt / g.Tsrc);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < g.K; ++k) {
      const int tq = t + g.pad - k * g.dil;
      if (tq < 0 || tq % g.stride) continue;
      const int q = tq / g.stride;
      if (q >= g.Tdst) continue;
      const float d = g.y[((long long)b * g.Tdst + q) * g.inner + p];
      const float4 w = n1_w4(g, k, 4 * c4);
      acc.x += d * w.x; acc.y += d * w.y; acc.z += d * w.z; acc.w += d * w.w;
    }
    if (g.in_act) {
      const float4 v = *reinterpret_cast<const float4*>(g.x + row * g.Cin + 4 * c4);
      acc.x *= v.x < 0.f ? g.in_slope : 1.f;
      acc.y *= v.y < 0.f ? g.in_slope : 1.f;
      acc.z *= v.z < 0.f ? g.in_slope : 1.f;
      acc.w *= v.w < 0.f ? g.in_slope : 1.f;
    }
    *reinterpret_cast<float4*>(g.dx + row * g.Cin + 4 * c4) = acc;
  }
}

// ---- weight gradient: thread (row slot, c4) walks input rows once; K accumulators per thread; LDS reduce over the row
//      slots; K * Cin atomics per workgroup (few workgroups: the layer reads x once, 4-130 MB)
#define N1_MAXK 8
__global__ __launch_bounds__(N1_THREADS) void conv_n1_wgrad_kernel(const kantts_conv_n1_args g) {
  extern __shared__ __attribute__((aligned(16))) float n1_lds[];  // [RS][K][Cin]
  __shared__ float red[4];
  const int C4 = g.Cin >> 2;
  const int RS = N1_THREADS / C4;  // row slots (launcher: C4 divides 256)
  const int c4 = threadIdx.x % C4, rs = threadIdx.x / C4;
  float4 acc[N1_MAXK];
#pragma unroll
  for (int k = 0; k < N1_MAXK; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  const long long rows = (long long)g.B * g.Tsrc * g.inner;
  for (long long row = (long long)blockIdx.x * RS + rs; row < rows; row += (long long)gridDim.x * RS) {
    const int p = (int)(row % g.inner);
    const long long bt = row / g.inner;
    const int t = (int)(bt % g.Tsrc), b = (int)(bt / g.Tsrc);
    float4 v = *reinterpret_cast<const float4*>(g.x + row * g.Cin + 4 * c4);
    v.x = n1_act(v.x, g.in_act, g.in_slope); v.y = n1_act(v.y, g.in_act, g.in_slope);
    v.z = n1_act(v.z, g.in_act, g.in_slope); v.w = n1_act(v.w, g.in_act, g.in_slope);
#pragma unroll
    for (int k = 0; k < N1_MAXK; ++k) {
      if (k < g.K) {
        const int tq = t + g.pad - k * g.dil;
        if (tq >= 0 && tq % g.stride == 0 && tq / g.stride < g.Tdst) {
          const float d = g.y[((long long)b * g.Tdst + tq / g.stride) * g.inner + p];
          acc[k].x += d * v.x; acc[k].y += d * v.y; acc[k].z += d * v.z; acc[k].w += d * v.w;
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < N1_MAXK; ++k)
    if (k < g.K) *reinterpret_cast<float4*>(&n1_lds[((size_t)rs * g.K + k) * g.Cin + 4 * c4]) = acc[k];
  __syncthreads();
  const int nflush = g.K * g.Cin;
  for (int i0 = threadIdx.x; i0 < nflush; i0 += N1_THREADS) {
    // every workgroup flushes the same K * Cin addresses: each starts at its own offset
    const int i = (int)((i0 + (long long)blockIdx.x * 68) % nflush);
    float s = 0.f;
    for (int r = 0; r < RS; ++r) s += n1_lds[(size_t)r * nflush + i];
    const int k = i / g.Cin, c = i % g.Cin;
    atomicAdd(&g.dw[(long long)k * g.w_ks + (long long)c * g.w_cs], s);
  }
  if (g.db) {
    const long long n = (long long)g.B * g.Tdst * g.inner;
    float part = 0.f;
    for (long long i = (long long)blockIdx.x * N1_THREADS + threadIdx.x; i < n; i += (long long)gridDim.x * N1_THREADS)
      part += g.y[i];
    part = kantts_block_sum(part, red);
    if (threadIdx.x == 0) atomicAdd(g.db, part);
  }
}

extern "C" int kantts_conv_n1_launch(const kantts_conv_n1_args* a, int mode, void* stream) {
  if (!a || mode < 0 || mode > 2) return KANTTS_E_BADARG;
  const kantts_conv_n1_args& g = *a;
  if (g.B < 0 || g.Tsrc < 0 || g.Tdst < 0 || g.Cin < 4 || g.K < 1 || g.stride < 1 || g.dil < 1 || g.inner < 1 || g.pad < 0)
    return KANTTS_E_BADARG;
  if (g.B == 0 || (g.Tdst == 0 && g.Tsrc == 0)) return KANTTS_OK;  // (empty tensors may carry NULL pointers)
  if (!g.x || !g.y || !g.w || (mode == 1 && !g.dx) || (mode == 2 && !g.dw)) return KANTTS_E_BADARG;
  const int C4 = g.Cin >> 2;
  if ((g.Cin & 3) || g.Cin > 1024 || (C4 & (C4 - 1)) || g.K > N1_MAXK || ((uintptr_t)g.x & 15) || (g.dx && ((uintptr_t)g.dx & 15)) ||
      (g.w_cs == 1 && (((uintptr_t)g.w & 15) || (g.w_ks & 3))))
    return KANTTS_E_UNSUPPORTED;
  const int LC = C4 < 64 ? C4 : 64;
  if ((C4 / LC) * g.K > N1_MAXW) return KANTTS_E_UNSUPPORTED;
  if (g.B == 0 || g.Tdst == 0 || g.Tsrc == 0) return KANTTS_OK;
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) {
    const long long total = (long long)g.B * g.Tdst * g.inner;
    const int per_wg = (N1_THREADS / 64) * (64 / LC);
    int grid = kantts_cdiv(total, per_wg);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(conv_n1_fwd_kernel, dim3(grid), dim3(N1_THREADS), 0, st, g);
  } else if (mode == 1) {
    const long long total = (long long)g.B * g.Tsrc * g.inner * C4;
    int grid = kantts_cdiv(total, N1_THREADS);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(conv_n1_dgrad_kernel, dim3(grid), dim3(N1_THREADS), 0, st, g);
  } else {
    const int RS = N1_THREADS / C4;
    const long long rows = (long long)g.B * g.Tsrc * g.inner;
    // workgroups: enough to stream x, few enough that the K * Cin atomics per workgroup stay under ~0.5 M per launch
    // (generator: 224 addresses -> up to 1024 workgroups over 33 MB; discriminators: 3072 addresses -> 170)
    int grid = kantts_cdiv(rows, (long long)RS * 16);
    int cap = (1 << 19) / (g.K * g.Cin);
    if (cap < 64) cap = 64;
    if (cap > 1024) cap = 1024;
    if (grid > cap) grid = cap;
    if (grid < 1) grid = 1;
    const size_t lds = (size_t)RS * g.K * g.Cin * sizeof(float);
    if (lds > 64 * 1024) return KANTTS_E_UNSUPPORTED;
    hipLaunchKernelGGL(conv_n1_wgrad_kernel, dim3(grid), dim3(N1_THREADS), lds, st, g);
  }
  KANTTS_CHECK_LAUNCH();
}
