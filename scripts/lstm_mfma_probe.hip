// MEASUREMENT PROTOTYPE, not part of the product (scripts/lstm_mfma_probe.py builds and times it).
//
// "a batch-16-per-workgroup LSTM recurrence on v_mfma_f32_16x16x32_bf16 ... if it loses on the
// device, keep the measurement".  This is that recurrence, forward, H = 128, as a real LSTM (its outputs are compared with
// the product's kantts_lstm_fwd): one workgroup of 8 waves owns SIXTEEN sequences; W_hh (512 x 128, rows permuted so that
// a 16-row MFMA tile holds the four gates of four cells) stays in registers as 16 A fragments per wave for the whole
// sequence; h_{t-1} of the sixteen sequences is the B operand (bf16, double-buffered in LDS); the accumulator lane
// (sequence li, k-group kg) receives the four gate pre-activations of ONE cell of ONE sequence, adds the input projection,
// applies the activations, updates c and h.  One LDS-only barrier per step.  Inputs / outputs use the layout the lanes
// touch (one 16-byte access per cell), which an integrated version would have the input-projection GEMM write directly.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define LP_H 128
#define LP_THREADS 512
#define LP_PITCH (LP_H + 8)  // bf16 elements per sequence row of h: 272 B = 16 mod 64 (conflict-free 16-byte reads)

__device__ __forceinline__ float lp_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float lp_tanh(float x) { return fmaf(2.f, lp_sigmoid(2.f * x), -1.f); }
__device__ __forceinline__ void lp_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// wfrag : W_hh permuted (row 4 c + gate) as a fragment-major bf16 image (512 x 128)
// gxp   : [T][groups][32 tiles][64 lanes] float4 = (i, f, g, o) input projections (+ biases) of cell 4 tile + kg, sequence li
// outp  : [T][groups][8 waves][64 lanes] float4 = h of the cells 4 (4 w + j) + kg, j = 0..3, sequence li
// gates : [T][groups][32 tiles][64 lanes] float4 (post-activation, saved for backward), cs : like outp (cell states)
__global__ __launch_bounds__(LP_THREADS) void lstm_mfma_fwd_kernel(const __bf16* __restrict__ wfrag,
                                                                   const f32x4* __restrict__ gxp, f32x4* __restrict__ outp,
                                                                   f32x4* __restrict__ gates, f32x4* __restrict__ cs, int T,
                                                                   int groups) {
  __shared__ __attribute__((aligned(16))) __bf16 hb[2][16][LP_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, kg = lane >> 4;
  const int grp = blockIdx.x;
  bf16x8 a[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      a[j][kk] = *reinterpret_cast<const bf16x8*>(wfrag + ((long long)((wave * 4 + j) * 4 + kk) * 64 + lane) * 8);
  for (int i = tid; i < 2 * 16 * LP_PITCH; i += LP_THREADS) (&hb[0][0][0])[i] = (__bf16)0.f;
  float c[4] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  f32x4 gq[4], gn[4];
  const long long tstride = (long long)groups * 32 * 64;
  const f32x4* gbase = gxp + ((long long)grp * 32 + wave * 4) * 64 + lane;
#pragma unroll
  for (int j = 0; j < 4; ++j) gq[j] = gbase[j * 64];
  int cur = 0;
  for (int t = 0; t < T; ++t) {
    const int tn = min(t + 1, T - 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) gn[j] = gbase[(long long)tn * tstride + j * 64];  // next step's input projections
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const bf16x8 bv = *reinterpret_cast<const bf16x8*>(&hb[cur][li][kk * 32 + kg * 8]);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j][kk], bv, acc[j], 0, 0, 0);
    }
    f32x4 hv, cv;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gi = lp_sigmoid(acc[j][0] + gq[j][0]), gf = lp_sigmoid(acc[j][1] + gq[j][1]);
      const float gg = lp_tanh(acc[j][2] + gq[j][2]), go = lp_sigmoid(acc[j][3] + gq[j][3]);
      c[j] = fmaf(gf, c[j], gi * gg);
      const float h = go * lp_tanh(c[j]);
      hv[j] = h;
      cv[j] = c[j];
      hb[cur ^ 1][li][4 * (4 * wave + j) + kg] = (__bf16)h;
      gates[((long long)t * groups + grp) * 32 * 64 + (wave * 4 + j) * 64 + lane] = (f32x4){gi, gf, gg, go};
    }
    const long long o = ((long long)t * groups + grp) * 8 * 64 + wave * 64 + lane;
    outp[o] = hv;
    cs[o] = cv;
#pragma unroll
    for (int j = 0; j < 4; ++j) gq[j] = gn[j];
    cur ^= 1;
    lp_barrier();
  }
}

extern "C" int lstm_mfma_probe_fwd(const void* wfrag, const float* gxp, float* outp, float* gates, float* cs, int T, int groups,
                                   void* stream) {
  hipLaunchKernelGGL(lstm_mfma_fwd_kernel, dim3(groups), dim3(LP_THREADS), 0, (hipStream_t)stream,
                     reinterpret_cast<const __bf16*>(wfrag), reinterpret_cast<const f32x4*>(gxp),
                     reinterpret_cast<f32x4*>(outp), reinterpret_cast<f32x4*>(gates), reinterpret_cast<f32x4*>(cs), T, groups);
  return (int)hipGetLastError();
}
