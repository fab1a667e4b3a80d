// Single-input-channel convolutions (the first layer of every HiFi-GAN discriminator: Conv1d(1 -> 128, k=15)
// of the scale discriminators, Conv2d(1 -> 32, (5,1), stride (3,1)) of the period discriminators;
// kantts/models/hifigan/hifigan.py:217-267,332-407).
//
// With one input channel there is no reduction to speak of (K = 5 ... 15 multiply-adds per output), so these
// layers are pure streaming kernels bound by the (B, T, Cout) activation they write (forward) or read (both
// gradients): MFMA tiles would be 1/32 full.  All three kernels walk runs of QB consecutive output tokens of one
// (batch item, folded position) pair, keep the run's waveform window in LDS, and touch the big activation exactly
// once with 16-byte / fully coalesced accesses:
//   forward  y[b,q,p,n]  = LeakyReLU( bias[n] + sum_k x[b, q*s + k*d - pad, p] * w[n][k] )
//   dgrad    dx[b,t,p]  += sum_k sum_n gate(dy[b,q,p,n]) * w[n][k]      at t = q*s + k*d - pad    (LDS scatter, one global add per token)
//   wgrad    dw[n][k]   += sum_{b,q,p} gate(dy[b,q,p,n]) * x[b, q*s + k*d - pad, p];   db[n] += sum gate(dy)
// fp32 throughout (no operand rounding), so the result does not depend on the GEMM precision mode.
#include "common.h"

#define C1_THREADS 256
#define C1_QB 256      // output tokens per run
#define C1_MAXK 16

__device__ __forceinline__ float c1_gate(float d, float y, float slope) { return d * ((y > 0.f) ? 1.f : slope); }

// ------------------------------------------------------------------------------------------- forward
__global__ __launch_bounds__(C1_THREADS) void conv_c1_fwd_kernel(const kantts_conv_c1_args g) {
  extern __shared__ __attribute__((aligned(16))) float c1_lds[];
  const int runs = (g.Tdst + C1_QB - 1) / C1_QB;
  const int run = blockIdx.x % runs;
  const int bp = blockIdx.x / runs;
  const int b = bp / g.inner, p = bp % g.inner;
  const int q0 = run * C1_QB;
  const int nq = min(C1_QB, g.Tdst - q0);
  const int W = (nq - 1) * g.stride + (g.K - 1) * g.dil + 1;
  float* xs = c1_lds;               // [W]
  float* ws = c1_lds + ((C1_QB - 1) * g.stride + (g.K - 1) * g.dil + 1 + 3) / 4 * 4;  // [K][Cout]
  const int lo = q0 * g.stride - g.pad;
  for (int i = threadIdx.x; i < W; i += C1_THREADS) {
    const int t = lo + i;
    xs[i] = (t >= 0 && t < g.Tsrc) ? g.x[((long long)b * g.Tsrc + t) * g.inner + p] : 0.f;
  }
  for (int i = threadIdx.x; i < g.K * g.Cout; i += C1_THREADS) {
    const int k = i / g.Cout, n = i % g.Cout;
    ws[i] = g.w[n * g.K + k];
  }
  __syncthreads();
  const int NQ = g.Cout / 4;            // lanes per token
  const int TPB = C1_THREADS / NQ;      // tokens per pass
  const int n4 = (threadIdx.x % NQ) * 4, tl = threadIdx.x / NQ;
  if (tl >= TPB) return;
  float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g.bias) bias = *reinterpret_cast<const float4*>(g.bias + n4);
  for (int ql = tl; ql < nq; ql += TPB) {
    float4 acc = bias;
    for (int k = 0; k < g.K; ++k) {
      const float xv = xs[ql * g.stride + k * g.dil];
      const float4 wv = *reinterpret_cast<const float4*>(&ws[k * g.Cout + n4]);
      acc.x += xv * wv.x;
      acc.y += xv * wv.y;
      acc.z += xv * wv.z;
      acc.w += xv * wv.w;
    }
    if (g.out_act) {
      acc.x = acc.x > 0.f ? acc.x : acc.x * g.out_slope;
      acc.y = acc.y > 0.f ? acc.y : acc.y * g.out_slope;
      acc.z = acc.z > 0.f ? acc.z : acc.z * g.out_slope;
      acc.w = acc.w > 0.f ? acc.w : acc.w * g.out_slope;
    }
    *reinterpret_cast<float4*>(g.y + (((long long)b * g.Tdst + q0 + ql) * g.inner + p) * g.Cout + n4) = acc;
  }
}

// ------------------------------------------------------------------------------------------- input gradient
// One wave per dy token: lanes split the Cout channels, K wave reductions give the token's contribution to the K
// taps, lane k adds tap k into the run's dx window in LDS; the window goes to dx with one atomic per token.
__global__ __launch_bounds__(C1_THREADS) void conv_c1_dgrad_kernel(const kantts_conv_c1_args g) {
  extern __shared__ __attribute__((aligned(16))) float c1_lds[];
  const int runs = (g.Tdst + C1_QB - 1) / C1_QB;
  const int run = blockIdx.x % runs;
  const int bp = blockIdx.x / runs;
  const int b = bp / g.inner, p = bp % g.inner;
  const int q0 = run * C1_QB;
  const int nq = min(C1_QB, g.Tdst - q0);
  const int W = (nq - 1) * g.stride + (g.K - 1) * g.dil + 1;
  float* dxs = c1_lds;
  float* ws = c1_lds + ((C1_QB - 1) * g.stride + (g.K - 1) * g.dil + 1 + 3) / 4 * 4;  // [K][Cout]
  for (int i = threadIdx.x; i < W; i += C1_THREADS) dxs[i] = 0.f;
  for (int i = threadIdx.x; i < g.K * g.Cout; i += C1_THREADS) {
    const int k = i / g.Cout, n = i % g.Cout;
    ws[i] = g.w[n * g.K + k];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int ql = wave; ql < nq; ql += C1_THREADS / 64) {
    const long long row = (((long long)b * g.Tdst + q0 + ql) * g.inner + p) * g.Cout;
    float mine = 0.f;
    float part[C1_MAXK];
#pragma unroll
    for (int k = 0; k < C1_MAXK; ++k) part[k] = 0.f;
    for (int n = lane; n < g.Cout; n += 64) {
      float d = g.y[row + n];  // dy
      if (g.gate) d = c1_gate(d, g.gate[row + n], g.gate_slope);
#pragma unroll
      for (int k = 0; k < C1_MAXK; ++k)
        if (k < g.K) part[k] += d * ws[k * g.Cout + n];
    }
#pragma unroll
    for (int k = 0; k < C1_MAXK; ++k) {
      if (k < g.K) {
        const float s = kantts_wave_sum(part[k]);
        if (lane == k) mine = s;
      }
    }
    if (lane < g.K) atomicAdd(&dxs[ql * g.stride + lane * g.dil], mine);
  }
  __syncthreads();
  const int lo = q0 * g.stride - g.pad;
  for (int i = threadIdx.x; i < W; i += C1_THREADS) {
    const int t = lo + i;
    if (t >= 0 && t < g.Tsrc) atomicAdd(&g.dx[((long long)b * g.Tsrc + t) * g.inner + p], dxs[i]);
  }
}

// ------------------------------------------------------------------------------------------- weight / bias gradient
// Thread = (token lane, output channel); K accumulators in registers; the block's partial sums are reduced over
// the token lanes in LDS and added to dw / db with one atomic per (n, k) and block.
__global__ __launch_bounds__(C1_THREADS) void conv_c1_wgrad_kernel(const kantts_conv_c1_args g) {
  extern __shared__ __attribute__((aligned(16))) float c1_lds[];
  const int runs = (g.Tdst + C1_QB - 1) / C1_QB;
  const int run = blockIdx.x % runs;
  const int bp = blockIdx.x / runs;
  const int b = bp / g.inner, p = bp % g.inner;
  const int q0 = run * C1_QB;
  const int nq = min(C1_QB, g.Tdst - q0);
  const int W = (nq - 1) * g.stride + (g.K - 1) * g.dil + 1;
  float* xs = c1_lds;
  float* red = c1_lds + ((C1_QB - 1) * g.stride + (g.K - 1) * g.dil + 1 + 3) / 4 * 4;  // [(K+1)][Cout]
  const int lo = q0 * g.stride - g.pad;
  for (int i = threadIdx.x; i < W; i += C1_THREADS) {
    const int t = lo + i;
    xs[i] = (t >= 0 && t < g.Tsrc) ? g.x[((long long)b * g.Tsrc + t) * g.inner + p] : 0.f;
  }
  for (int i = threadIdx.x; i < (g.K + 1) * g.Cout; i += C1_THREADS) red[i] = 0.f;
  __syncthreads();
  const int TL = C1_THREADS / g.Cout;  // token lanes (Cout <= 256)
  const int n = threadIdx.x % g.Cout, tl = threadIdx.x / g.Cout;
  if (tl < TL) {
    float acc[C1_MAXK];
#pragma unroll
    for (int k = 0; k < C1_MAXK; ++k) acc[k] = 0.f;
    float bsum = 0.f;
    for (int ql = tl; ql < nq; ql += TL) {
      const long long o = (((long long)b * g.Tdst + q0 + ql) * g.inner + p) * g.Cout + n;
      float d = g.y[o];
      if (g.gate) d = c1_gate(d, g.gate[o], g.gate_slope);
      bsum += d;
#pragma unroll
      for (int k = 0; k < C1_MAXK; ++k)
        if (k < g.K) acc[k] += d * xs[ql * g.stride + k * g.dil];
    }
#pragma unroll
    for (int k = 0; k < C1_MAXK; ++k)
      if (k < g.K) atomicAdd(&red[k * g.Cout + n], acc[k]);
    atomicAdd(&red[g.K * g.Cout + n], bsum);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < g.K * g.Cout; i += C1_THREADS) {
    const int k = i / g.Cout, nn = i % g.Cout;
    atomicAdd(&g.dw[nn * g.K + k], red[i]);
  }
  if (g.db)
    for (int i = threadIdx.x; i < g.Cout; i += C1_THREADS) atomicAdd(&g.db[i], red[g.K * g.Cout + i]);
}

extern "C" int kantts_conv_c1_launch(const kantts_conv_c1_args* a, int mode, void* stream) {
  if (!a || !a->y || !a->w || (mode != 1 && !a->x) || (mode == 1 && !a->dx)) return KANTTS_E_BADARG;
  const kantts_conv_c1_args& g = *a;
  if (g.B < 0 || g.Tsrc < 0 || g.Tdst < 0 || g.K < 1 || g.Cout < 1 || g.stride < 1 || g.dil < 1 || g.inner < 1)
    return KANTTS_E_BADARG;
  if (mode < 0 || mode > 2 || (mode == 2 && !g.dw)) return KANTTS_E_BADARG;
  if (g.K > C1_MAXK || (g.Cout & 3) || g.Cout > 256 || (256 % (g.Cout / 4)) != 0 || (256 % g.Cout) != 0 ||
      ((uintptr_t)g.y & 15) || (g.bias && ((uintptr_t)g.bias & 15)))
    return KANTTS_E_UNSUPPORTED;
  if (g.B == 0 || g.Tdst == 0) return KANTTS_OK;
  const long long blocks = (long long)g.B * g.inner * kantts_cdiv(g.Tdst, C1_QB);
  if (blocks > 0x7fffffffLL) return KANTTS_E_UNSUPPORTED;
  const int wmax = (C1_QB - 1) * g.stride + (g.K - 1) * g.dil + 1;
  const size_t lds = ((size_t)(wmax + 3) / 4 * 4 + (size_t)(g.K + 1) * g.Cout) * sizeof(float);
  if (lds > 64 * 1024) return KANTTS_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0)
    hipLaunchKernelGGL(conv_c1_fwd_kernel, dim3((unsigned)blocks), dim3(C1_THREADS), lds, st, g);
  else if (mode == 1)
    hipLaunchKernelGGL(conv_c1_dgrad_kernel, dim3((unsigned)blocks), dim3(C1_THREADS), lds, st, g);
  else
    hipLaunchKernelGGL(conv_c1_wgrad_kernel, dim3((unsigned)blocks), dim3(C1_THREADS), lds, st, g);
  KANTTS_CHECK_LAUNCH();
}
