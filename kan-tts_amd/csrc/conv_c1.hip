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
#include <stdlib.h>

#include "common.h"
#include <atomic>
extern std::atomic<int> kantts_tune_c1_wgrad_wgs;  // csrc/gemm_bf16.hip: kantts_launch_tuning


#define C1_THREADS 256
#define C1_QB 256      // output tokens per run
#define C1_MAXK 16

__device__ __forceinline__ float c1_gate(float d, float y, float slope) { return d * ((y > 0.f) ? 1.f : slope); }

typedef __bf16 c1_bf16x4 __attribute__((ext_vector_type(4)));
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
    const long long o = (((long long)b * g.Tdst + q0 + ql) * g.inner + p) * g.Cout + n4;
    *reinterpret_cast<float4*>(g.y + o) = acc;
    if (g.y_bf16) {
      c1_bf16x4 v = {(__bf16)acc.x, (__bf16)acc.y, (__bf16)acc.z, (__bf16)acc.w};
      *reinterpret_cast<c1_bf16x4*>(reinterpret_cast<__bf16*>(g.y_bf16) + o) = v;
    }
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

// ------------------------------------------------------------------------------------------- MFMA forms of the gradients
// (round 3) The two kernels above spend their time in wave reductions (K shuffles trees per token) and per-thread tap
// loops, not on the 268 MB of dy + gate they read at batch 32 x 8192 x 128: 300 us where HBM needs 60.  Both sums are
// small contractions over the output channels / the tokens, so they go to v_mfma_f32_16x16x4_f32 -- fp32 operands, exact
// fp32 FMA chains (the result still does not depend on the precision mode), 1/16 of the bf16 rate and still 10x more than
// these layers need:
//   dgrad   P[token][tap] = sum_n gate(dy[token][n]) * w[n][tap]   16 tokens x 16 taps per accumulator; a lane loads
//           float4s of dy / gate (4 consecutive channels) and feeds element m to MFMA m, the weight operand uses the same
//           channel permutation; P goes to LDS and every window position GATHERS its <= K terms (no LDS atomics)
//   wgrad   dw[n][tap]    = sum_q gate(dy[q][n]) * x[q*s + tap*d - pad]   16 channels x 16 taps per accumulator over 4
//           tokens per MFMA; element m of a lane's float4 is channel 4*li + m of its 64-channel half.
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define C1_PLD 17  // pitch of the P tile

template <int G16>  // Cout = 16 * G16
__global__ __launch_bounds__(C1_THREADS) void conv_c1_dgrad_mfma_kernel(const kantts_conv_c1_args g) {
  extern __shared__ __attribute__((aligned(16))) float c1_lds[];
  constexpr int COUT = 16 * G16;
  const int runs = (g.Tdst + C1_QB - 1) / C1_QB;
  const int run = blockIdx.x % runs;
  const int bp = blockIdx.x / runs;
  const int b = bp / g.inner, p = bp % g.inner;
  const int q0 = run * C1_QB;
  const int nq = min(C1_QB, g.Tdst - q0);
  const int W = (nq - 1) * g.stride + (g.K - 1) * g.dil + 1;
  float* P = c1_lds;  // [C1_QB][C1_PLD]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, kg = lane >> 4;
  // weight operand of MFMA (group gg, element m): B[k = kg][j = li] = w[16*gg + 4*kg + m][tap li]
  float wb[G16][4];
#pragma unroll
  for (int gg = 0; gg < G16; ++gg)
#pragma unroll
    for (int m = 0; m < 4; ++m) wb[gg][m] = (li < g.K) ? g.w[(16 * gg + 4 * kg + m) * g.K + li] : 0.f;
  for (int tt = wave; tt * 16 < nq; tt += C1_THREADS / 64) {
    const int q = tt * 16 + li;
    const bool ok = q < nq;
    const long long row = (((long long)b * g.Tdst + q0 + (ok ? q : 0)) * g.inner + p) * COUT + 4 * kg;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int gg = 0; gg < G16; ++gg) {
      float4 d = *reinterpret_cast<const float4*>(g.y + row + 16 * gg);
      if (g.gate) {
        const float4 y = *reinterpret_cast<const float4*>(g.gate + row + 16 * gg);
        d.x = c1_gate(d.x, y.x, g.gate_slope);
        d.y = c1_gate(d.y, y.y, g.gate_slope);
        d.z = c1_gate(d.z, y.z, g.gate_slope);
        d.w = c1_gate(d.w, y.w, g.gate_slope);
      }
      if (!ok) d = make_float4(0.f, 0.f, 0.f, 0.f);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(d.x, wb[gg][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(d.y, wb[gg][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(d.z, wb[gg][2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(d.w, wb[gg][3], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) P[(tt * 16 + kg * 4 + r) * C1_PLD + li] = acc[r];  // token kg*4 + r, tap li
  }
  __syncthreads();
  const int lo = q0 * g.stride - g.pad;
  for (int i = threadIdx.x; i < W; i += C1_THREADS) {
    const int t = lo + i;
    if (t < 0 || t >= g.Tsrc) continue;
    float v = 0.f;
    for (int k = 0; k < g.K; ++k) {
      const int u = i - k * g.dil;
      if (u < 0) break;
      const int q = u / g.stride;
      if (q * g.stride == u && q < nq) v += P[q * C1_PLD + k];
    }
    atomicAdd(&g.dx[((long long)b * g.Tsrc + t) * g.inner + p], v);
  }
}

template <int H>  // Cout <= 64 * H (a multiple of 4)
__global__ __launch_bounds__(C1_THREADS) void conv_c1_wgrad_mfma_kernel(const kantts_conv_c1_args g, const int total_runs) {
  // [round 4] PERSISTENT over runs: a workgroup walks runs blockIdx.x, blockIdx.x + gridDim.x, ... with its accumulators in
  // registers and reduces (LDS atomics, then (K + 1) * Cout global atomics) ONCE.  One workgroup per run -- 2048 of them
  // for the first MSD layer at batch 64 x 8192 -- put 4.2 M fp32 atomics on 2048 addresses: 511 us per launch, atomics-bound
  // (profiles/r03_runFINAL_bench_kernel_stats_top.csv).
  extern __shared__ __attribute__((aligned(16))) float c1_lds[];
  const int runs = (g.Tdst + C1_QB - 1) / C1_QB;
  float* xs = c1_lds;
  float* red = c1_lds + ((C1_QB - 1) * g.stride + (g.K - 1) * g.dil + 1 + 3) / 4 * 4;  // [(K+1)][Cout]
  for (int i = threadIdx.x; i < (g.K + 1) * g.Cout; i += C1_THREADS) red[i] = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, kg = lane >> 4;
  f32x4 acc[H][4];
  float bs[H][4];
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      acc[h][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
      bs[h][m] = 0.f;
    }
  for (int blk = blockIdx.x; blk < total_runs; blk += gridDim.x) {
    const int run = blk % runs;
    const int bp = blk / runs;
    const int b = bp / g.inner, p = bp % g.inner;
    const int q0 = run * C1_QB;
    const int nq = min(C1_QB, g.Tdst - q0);
    const int W = (nq - 1) * g.stride + (g.K - 1) * g.dil + 1;
    const int lo = q0 * g.stride - g.pad;
    __syncthreads();  // the previous run's readers of xs are done (first pass: red is zeroed)
    for (int i = threadIdx.x; i < W; i += C1_THREADS) {
      const int t = lo + i;
      xs[i] = (t >= 0 && t < g.Tsrc) ? g.x[((long long)b * g.Tsrc + t) * g.inner + p] : 0.f;
    }
    __syncthreads();
    // two token groups per trip, every load of the trip issued before the first MFMA: the layer is a stream of dy (and
    // the gate) -- 268 MB for the first MSD layer at batch 32 x 8192 -- and ran at 1.06 TB/s with one group in flight per
    // wave (profiles/r04_runV: 253 us per launch, alone on the chip at the end of every sub-discriminator's backward)
    constexpr int WV = C1_THREADS / 64;
    for (int tg = wave; tg * 4 < nq; tg += 2 * WV) {
      float4 dv[2][H], gv[2][H];
      float bxv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q = (tg + u * WV) * 4 + kg;
        const bool ok = q < nq;
        bxv[u] = (ok && li < g.K) ? xs[q * g.stride + li * g.dil] : 0.f;  // B[k = token kg][j = tap li]
        const long long row = (((long long)b * g.Tdst + q0 + (ok ? q : 0)) * g.inner + p) * g.Cout;
#pragma unroll
        for (int h = 0; h < H; ++h) {
          const int n = 64 * h + 4 * li;
          const bool live = ok && n < g.Cout;
          dv[u][h] = live ? *reinterpret_cast<const float4*>(g.y + row + n) : make_float4(0.f, 0.f, 0.f, 0.f);
          gv[u][h] = (live && g.gate) ? *reinterpret_cast<const float4*>(g.gate + row + n) : make_float4(1.f, 1.f, 1.f, 1.f);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float bx = bxv[u];
#pragma unroll
        for (int h = 0; h < H; ++h) {
          float4 d = dv[u][h];
          if (g.gate) {
            const float4 y = gv[u][h];
            d.x = c1_gate(d.x, y.x, g.gate_slope);
            d.y = c1_gate(d.y, y.y, g.gate_slope);
            d.z = c1_gate(d.z, y.z, g.gate_slope);
            d.w = c1_gate(d.w, y.w, g.gate_slope);
          }
          bs[h][0] += d.x; bs[h][1] += d.y; bs[h][2] += d.z; bs[h][3] += d.w;
          acc[h][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(d.x, bx, acc[h][0], 0, 0, 0);
          acc[h][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(d.y, bx, acc[h][1], 0, 0, 0);
          acc[h][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(d.z, bx, acc[h][2], 0, 0, 0);
          acc[h][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(d.w, bx, acc[h][3], 0, 0, 0);
        }
      }
    }
  }
  // accumulator (h, m), register r, lane (li, kg): channel 64h + 4*(kg*4 + r) + m, tap li
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = 64 * h + 4 * (kg * 4 + r) + m;
        if (n < g.Cout && li < g.K) atomicAdd(&red[li * g.Cout + n], acc[h][m][r]);
      }
      const int nb = 64 * h + 4 * li + m;
      if (nb < g.Cout) atomicAdd(&red[g.K * g.Cout + nb], bs[h][m]);
    }
  __syncthreads();
  // every workgroup flushes the same (K + 1) * Cout addresses at about the same time: start each at its own offset, so
  // that the L2 atomic units see distinct addresses instead of 512 queued adds on one (the flush, not the stream, was
  // the launch: 1 M atomics ~ 250 us, profiles/r04_runW)
  const int nflush = g.K * g.Cout, rot = (int)(((long long)blockIdx.x * 61) % nflush);
  for (int i0 = threadIdx.x; i0 < nflush; i0 += C1_THREADS) {
    int i = i0 + rot;
    if (i >= nflush) i -= nflush;
    const int k = i / g.Cout, nn = i % g.Cout;
    atomicAdd(&g.dw[nn * g.K + k], red[i]);
  }
  if (g.db)
    for (int i0 = threadIdx.x; i0 < g.Cout; i0 += C1_THREADS) {
      const int i = (i0 + blockIdx.x * 5) % g.Cout;
      atomicAdd(&g.db[i], red[g.K * g.Cout + i]);
    }
}

extern "C" int kantts_conv_c1_launch(const kantts_conv_c1_args* a, int mode, void* stream) {
  if (!a || !a->y || !a->w || (mode != 1 && !a->x) || (mode == 1 && !a->dx)) return KANTTS_E_BADARG;
  const kantts_conv_c1_args& g = *a;
  if (g.B < 0 || g.Tsrc < 0 || g.Tdst < 0 || g.K < 1 || g.Cout < 1 || g.stride < 1 || g.dil < 1 || g.inner < 1)
    return KANTTS_E_BADARG;
  if (mode < 0 || mode > 2 || (mode == 2 && !g.dw)) return KANTTS_E_BADARG;
  if (g.K > C1_MAXK || (g.Cout & 3) || g.Cout > 256 || (256 % (g.Cout / 4)) != 0 || (256 % g.Cout) != 0 ||
      ((uintptr_t)g.y & 15) || (g.bias && ((uintptr_t)g.bias & 15)) || ((uintptr_t)g.y_bf16 & 7))
    return KANTTS_E_UNSUPPORTED;
  if (g.B == 0 || g.Tdst == 0) return KANTTS_OK;
  const long long blocks = (long long)g.B * g.inner * kantts_cdiv(g.Tdst, C1_QB);
  if (blocks > 0x7fffffffLL) return KANTTS_E_UNSUPPORTED;
  const int wmax = (C1_QB - 1) * g.stride + (g.K - 1) * g.dil + 1;
  const size_t lds = ((size_t)(wmax + 3) / 4 * 4 + (size_t)(g.K + 1) * g.Cout) * sizeof(float);
  if (lds > 64 * 1024) return KANTTS_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  static const char* no_mfma = getenv("KANTTS_C1_NO_MFMA");  // A/B switch (scripts/gpu_*.sh)
  const bool vec = !no_mfma && g.K <= 16 && (!g.gate || ((uintptr_t)g.gate & 15) == 0);
  if (mode == 0) {
    hipLaunchKernelGGL(conv_c1_fwd_kernel, dim3((unsigned)blocks), dim3(C1_THREADS), lds, st, g);
  } else if (mode == 1) {
    const size_t plds = (size_t)C1_QB * C1_PLD * sizeof(float);
    if (vec && g.Cout == 128)
      hipLaunchKernelGGL((conv_c1_dgrad_mfma_kernel<8>), dim3((unsigned)blocks), dim3(C1_THREADS), plds, st, g);
    else if (vec && g.Cout == 32)
      hipLaunchKernelGGL((conv_c1_dgrad_mfma_kernel<2>), dim3((unsigned)blocks), dim3(C1_THREADS), plds, st, g);
    else
      hipLaunchKernelGGL(conv_c1_dgrad_kernel, dim3((unsigned)blocks), dim3(C1_THREADS), lds, st, g);
  } else {
    // persistent grid: at most C1_WGRAD_WGS workgroups (one per CU), each reducing once: the launch is the sum of a stream
    // (268 MB of dy + gate for the first MSD layer) and of (K + 1) * Cout global atomics per workgroup
    const int wgs_set = kantts_tune_c1_wgrad_wgs.load(std::memory_order_relaxed);  // kantts_launch_tuning (sweeps / tests)
    const long long cap = wgs_set > 0 ? wgs_set : 256;  // 512: 255 us, 256: 198, 128: 241 (r04_runX)
    const unsigned pgrid = (unsigned)(blocks < cap ? blocks : cap);
    if (vec && g.Cout <= 64)
      hipLaunchKernelGGL((conv_c1_wgrad_mfma_kernel<1>), dim3(pgrid), dim3(C1_THREADS), lds, st, g, (int)blocks);
    else if (vec && g.Cout <= 128)
      hipLaunchKernelGGL((conv_c1_wgrad_mfma_kernel<2>), dim3(pgrid), dim3(C1_THREADS), lds, st, g, (int)blocks);
    else
      hipLaunchKernelGGL(conv_c1_wgrad_kernel, dim3((unsigned)blocks), dim3(C1_THREADS), lds, st, g);
  }
  KANTTS_CHECK_LAUNCH();
}
