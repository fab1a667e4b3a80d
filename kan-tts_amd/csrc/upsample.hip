// HiFi-GAN transposed-convolution upsampling as an HBM-streaming kernel (north-star: ">= 60 % of the HBM roofline on
// HiFi-GAN upsampling"; round 1 ran this path at 10 %).
//
// CausalConvTranspose1d with kernel 2*s, stride s (kantts/models/hifigan/layers.py:125-165; the V1 generator's four
// layers, hifigan.py:67-80,160) in polyphase form:
//     y[b, t*s + r, co] = bias[co] + sum_{j=0,1} sum_ci lrelu(x[b, t - j, ci]) * w[ci, co, r + j*s]      (x[b,-1] = 0)
// i.e. per input token a (s*Cout) x (2*Cin) matrix times the token's [x_t | x_{t-1}] vector: y, as a row-major
// (B*T, s*Cout) matrix, is written exactly once and x is read exactly once.  For the narrow layers (Cin <= 128) the
// whole weight matrix fits in LDS (16 / 64 KB as bf16), so the kernel is a pure stream:
//   * activations are bf16 in HBM; every wave owns 16-token tiles and forms the MFMA B fragments (k = 8 consecutive
//     input channels of one token) straight from 16-byte global loads -- no LDS round trip, the loads of the next tile are
//     issued before the current tile is multiplied;
//   * the product is computed transposed (rows = output channels, columns = tokens): with the weight rows permuted on the
//     host so that a lane's two accumulator fragments hold 8 consecutive output channels of one token, every lane stores
//     16 bytes and a wave stores full 64-byte pieces of 16 output rows;
//   * LeakyReLU on the input is applied to the loaded fragment, bias (and an optional residual: the generator's
//     repeat-upsample branch) in the store.
// The wide early layers (Cin = 512 / 256, weights 4 / 1 MB) are ordinary contractions and go through kantts_bgemm_nt with
// two token-shifted segments (host layer).
#include <stdlib.h>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define UP_THREADS 512

__device__ __forceinline__ unsigned up_pack2(float a, float b) {
  typedef __bf16 v2 __attribute__((ext_vector_type(2)));
  v2 t = {(__bf16)a, (__bf16)b};
  return (unsigned&)t;
}
__device__ __forceinline__ float up_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float up_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned up_lrelu2(unsigned u, float slope) {
  float a = up_lo(u), b = up_hi(u);
  a = a > 0.f ? a : a * slope;
  b = b > 0.f ? b : b * slope;
  return up_pack2(a, b);
}

struct UpArgs {
  const __bf16* x;    // (B*T, CIN)
  const __bf16* wp;   // (S*COUT, 2*CIN) rows permuted (see kantts_upsample_stream)
  const float* bias;  // (COUT) or null
  const void* res;    // (B*T*S, COUT) like out, or null
  void* out;          // (B*T*S, COUT) bf16 or fp32
  int ntok, T;        // B*T tokens, tokens per sequence
  float slope;        // LeakyReLU slope on the input; 1 = identity
  int out_bf16;
  int early_fetch;    // launch-shape choice (no effect on results): see the kernel
};

template <int CIN, int COUT, int S>
__global__ __launch_bounds__(UP_THREADS) void upsample_stream_kernel(const UpArgs a) {
  constexpr int K = 2 * CIN;              // reduction: [x_t | x_{t-1}]
  constexpr int NR = S * COUT;            // weight rows = outputs per token
  constexpr int MT = NR / 16;             // MFMA row tiles
  constexpr int KS = K / 32;              // MFMA k-steps
  constexpr int CPR = K / 8;              // 16-byte chunks per weight row
  extern __shared__ __attribute__((aligned(16))) unsigned char up_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kg = lane >> 4;

  const int ntile = (a.ntok + 15) >> 4;
  const int stride = gridDim.x * (UP_THREADS / 64);
  int tile = blockIdx.x * (UP_THREADS / 64) + wave;

  u32x4 cur[KS], nxt[KS];
  auto fetch = [&](int tl, u32x4* f) {
    const int tok = tl * 16 + li;
    const bool ok0 = tok < a.ntok;
    const bool ok1 = ok0 && (tok % a.T) != 0;  // x[t-1] of the same sequence (causal: zero at t = 0)
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const int j = (kk * 32) / CIN;                 // tap of this k-step
      const int ci = (kk * 32) % CIN + kg * 8;
      const bool ok = j ? ok1 : ok0;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (ok) v = *reinterpret_cast<const u32x4*>(a.x + (long long)(tok - j) * CIN + ci);
      f[kk] = v;
    }
  };
  // ---- weights -> LDS once per workgroup, chunk index XORed with (row & 15): 16 rows of a fragment read 16 distinct slots
  // ([round 4] requesting the first tile's activations BEFORE this copy measured no better: 17.2 against 13.8-14.9 us on
  // other boxes for the 128 -> 64 stage, inside the box-to-box spread; the order of round 2 stays)
  if (a.early_fetch && tile < ntile) fetch(tile, cur);  // [round 6 A/B] the first tile's activations requested BEFORE the weights
  for (int id = tid; id < NR * CPR; id += UP_THREADS) {
    const int row = id / CPR, c = id % CPR;
    const u32x4 v = *reinterpret_cast<const u32x4*>(a.wp + (long long)row * K + c * 8);
    *reinterpret_cast<u32x4*>(up_lds + ((long long)row * CPR + (c ^ (row & 15))) * 16) = v;
  }
  __syncthreads();
  if (!a.early_fetch && tile < ntile) fetch(tile, cur);
  for (; tile < ntile; tile += stride) {
    const int tn = tile + stride;
    if (tn < ntile) fetch(tn, nxt);
    if (a.slope != 1.f) {
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        cur[kk].x = up_lrelu2(cur[kk].x, a.slope);
        cur[kk].y = up_lrelu2(cur[kk].y, a.slope);
        cur[kk].z = up_lrelu2(cur[kk].z, a.slope);
        cur[kk].w = up_lrelu2(cur[kk].w, a.slope);
      }
    }
    const int tok = tile * 16 + li;
    // one fragment PAIR (8 consecutive output channels of a phase) at a time: 2*KS weight fragments live, stored at once
#pragma unroll
    for (int p = 0; p < MT / 2; ++p) {
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      const int row0 = (2 * p) * 16 + li, row1 = row0 + 16;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        const bf16x8 bf = (bf16x8&)cur[kk];
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(up_lds + ((long long)row0 * CPR + ((kk * 4 + kg) ^ (row0 & 15))) * 16);
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(up_lds + ((long long)row1 * CPR + ((kk * 4 + kg) ^ (row1 & 15))) * 16);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bf, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bf, acc1, 0, 0, 0);
      }
      // fragment pair (2p, 2p+1) = channels co .. co+7 of output row tok*S + r, token li
      if (tok < a.ntok) {
        const int r = p / (COUT / 32), cb = p % (COUT / 32);
        const int co = cb * 32 + kg * 8;
        float o[8] = {acc0[0], acc0[1], acc0[2], acc0[3], acc1[0], acc1[1], acc1[2], acc1[3]};
        if (a.bias) {
          const float4 b0 = *reinterpret_cast<const float4*>(a.bias + co), b1 = *reinterpret_cast<const float4*>(a.bias + co + 4);
          o[0] += b0.x; o[1] += b0.y; o[2] += b0.z; o[3] += b0.w; o[4] += b1.x; o[5] += b1.y; o[6] += b1.z; o[7] += b1.w;
        }
        const long long off = ((long long)tok * S + r) * COUT + co;
        if (a.out_bf16) {
          if (a.res) {
            const u32x4 q = *reinterpret_cast<const u32x4*>(reinterpret_cast<const __bf16*>(a.res) + off);
            o[0] += up_lo(q.x); o[1] += up_hi(q.x); o[2] += up_lo(q.y); o[3] += up_hi(q.y);
            o[4] += up_lo(q.z); o[5] += up_hi(q.z); o[6] += up_lo(q.w); o[7] += up_hi(q.w);
          }
          u32x4 w = {up_pack2(o[0], o[1]), up_pack2(o[2], o[3]), up_pack2(o[4], o[5]), up_pack2(o[6], o[7])};
          *reinterpret_cast<u32x4*>(reinterpret_cast<__bf16*>(a.out) + off) = w;
        } else {
          if (a.res) {
            const float* rp = reinterpret_cast<const float*>(a.res) + off;
            const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
            o[0] += r0.x; o[1] += r0.y; o[2] += r0.z; o[3] += r0.w; o[4] += r1.x; o[5] += r1.y; o[6] += r1.z; o[7] += r1.w;
          }
          float* op = reinterpret_cast<float*>(a.out) + off;
          *reinterpret_cast<float4*>(op) = make_float4(o[0], o[1], o[2], o[3]);
          *reinterpret_cast<float4*>(op + 4) = make_float4(o[4], o[5], o[6], o[7]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the next pair's weight fragments from being hoisted above this store
    }
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) cur[kk] = nxt[kk];
  }
}

template <int CIN, int COUT, int S>
static int up_launch(const UpArgs& a, hipStream_t st) {
  constexpr size_t lds = (size_t)S * COUT * 2 * CIN * 2;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&upsample_stream_kernel<CIN, COUT, S>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int ntile = (a.ntok + 15) / 16;
  int blocks = kantts_cdiv(ntile, UP_THREADS / 64);
  int per_cu = lds > 40 * 1024 ? 2 : 4;  // workgroups a CU can hold (LDS bound)
  static const char* env_pc = getenv("KANTTS_UPSTREAM_WG_PER_CU");  // experiment switch
  if (env_pc && atoi(env_pc) > 0) per_cu = atoi(env_pc);
  if (blocks > 256 * per_cu) blocks = 256 * per_cu;
  hipLaunchKernelGGL((upsample_stream_kernel<CIN, COUT, S>), dim3(blocks), dim3(UP_THREADS), lds, st, a);
  KANTTS_CHECK_LAUNCH();
}


// [round 4] A variant with the weights in REGISTERS (a wave owning one pair of 16-row weight tiles for the whole launch,
// fragments loaded straight from L2, no LDS, no barrier) was written and measured: 25.8 / 14.4 us per stage against
// 14.9 / 12.5 us for the kernel above (profiles/r04_runE_upsampling_stream_forms.log) -- a lane's 16-byte piece of a
// 512-byte weight row coalesces badly, the LDS staging reads whole rows -- and was removed.

// x: (B*T, Cin) bf16 tokens; wp: (S*Cout, 2*Cin) bf16 with row (mt*16 + rho), mt = (r*(Cout/32) + cb)*2 + h, holding output
// channel co = cb*32 + (rho >> 2)*8 + h*4 + (rho & 3) of phase r, and column j*Cin + ci holding w[ci, co, r + j*S];
// out / res: (B*T*S, Cout) bf16 (out_bf16) or fp32.  Supported: (Cin, Cout, S) = (128, 64, 2), (64, 32, 2) -- the narrow
// layers of the V1 generator and of its half-width variants; KANTTS_E_UNSUPPORTED otherwise.
extern "C" int kantts_upsample_stream(const void* x_bf16, const void* wp_bf16, const float* bias, const void* res, void* out,
                                      int B, int T, int Cin, int Cout, int S, float in_slope, int out_bf16, void* stream) {
  if (!x_bf16 || !wp_bf16 || !out || B < 0 || T < 1) return KANTTS_E_BADARG;
  if (((uintptr_t)x_bf16 | (uintptr_t)wp_bf16 | (uintptr_t)out | (uintptr_t)res | (uintptr_t)bias) & 15) return KANTTS_E_UNSUPPORTED;
  if (B == 0) return KANTTS_OK;
  UpArgs a;
  a.x = reinterpret_cast<const __bf16*>(x_bf16);
  a.wp = reinterpret_cast<const __bf16*>(wp_bf16);
  a.bias = bias; a.res = res; a.out = out;
  a.ntok = B * T; a.T = T; a.slope = in_slope; a.out_bf16 = out_bf16;
  static const char* env_ef = getenv("KANTTS_UP_EARLY_FETCH");  // experiment switch, read once per process
  a.early_fetch = env_ef ? atoi(env_ef) : 0;
  hipStream_t st = (hipStream_t)stream;
  if (Cin == 128 && Cout == 64 && S == 2) return up_launch<128, 64, 2>(a, st);
  if (Cin == 64 && Cout == 32 && S == 2) return up_launch<64, 32, 2>(a, st);
  return KANTTS_E_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------ [round 5] calibration
// The achievable roof of a launch that must read `read_bytes` and write `write_bytes` once: every 16-byte chunk of the source
// is loaded, every 16-byte chunk of the destination stored (the value stored is the chunk loaded at the same index, folded
// with a running XOR so that no load is dead code).  bench.py runs it with the upsampling stages' algorithmic bytes inside
// the same 4-launch graph as the real kernels: what a 10-34 MB problem can reach on this chip including its launch ramp.
typedef unsigned int cr_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_roof_kernel(const cr_u32x4* __restrict__ src, long long nread,
                                                       cr_u32x4* __restrict__ dst, long long nwrite) {
  const long long n = nread > nwrite ? nread : nwrite;
  cr_u32x4 acc = {0u, 0u, 0u, 0u};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    if (i < nread) acc ^= src[i];
    if (i < nwrite) dst[i] = acc;
  }
}

extern "C" int kantts_copy_roof(const void* src, long long read_bytes, void* dst, long long write_bytes, void* stream) {
  if (!src || !dst || read_bytes < 0 || write_bytes < 0 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return KANTTS_E_BADARG;
  const long long nr = read_bytes >> 4, nw = write_bytes >> 4, n = nr > nw ? nr : nw;
  if (n == 0) return KANTTS_OK;
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(copy_roof_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const cr_u32x4*>(src), nr, reinterpret_cast<cr_u32x4*>(dst), nw);
  KANTTS_CHECK_LAUNCH();
}
