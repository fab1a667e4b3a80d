// The position-wise feed-forward pair of a SAM-BERT block as ONE launch (round 2).
//
// reference: kantts/models/sambert/__init__.py:134-149 (PositionwiseConvFeedForward.forward after the LayerNorm):
//   hid = mask(dropout(relu(Conv1d_k(x))));  out = mask(dropout(Conv1d_1(hid)) + residual)
// and its backward through both convolutions:
//   dz = gate_{hid > 0}(dropout(dy) . W2) / (1 - p_inner);  dx = dz . W1
//
// As two bgemm_nt launches (csrc/gemm_bf16.hip) the (M, 1024) hidden tensor is written by the first contraction and read
// back by the second: 13.4 MB each way at M = 6528, and each launch is a short chain of dependent latencies (13 + 16 us
// in the step for 1.7 GFLOP).  Here a workgroup owns 32 tokens and carries them through BOTH contractions:
//
//   phase 1   T^T[f][tok] = W1[f][:] . X[tok][:]       f in chunks of 256, reduction 128 per tap
//   phase 2   Y^T[n][tok] = W2[n][:] . T[tok][:]       n = 128 outputs, reduction F (<= 1024)
//
// The weights are the MFMA *A* operand (rows = output channels) and are never staged in LDS: with a 32-token tile every
// weight element is used by exactly one wave, so each lane loads its 16-byte fragment straight from L2 (the 0.5 MB of
// weights stay L2-resident).  The weight images are FRAGMENT-MAJOR (kantts_fragmajor_bf16: the 64 x 16 bytes one
// load instruction of a wave needs are 1 KB contiguous, the 8 loads of a step 8 KB contiguous): the first version read
// row-major weights (16 rows x 64 bytes per instruction; 2 KB row pitch in phase 2 = a quarter of the L2 channels) with
// one step of prefetch and took 27 us per launch, all of it load latency (profiles/r02_runK_ffn_pair_first_version.log).
// Now 8 waves keep FOUR steps of fragments in flight each (a 4-deep register ring, 256 KB per CU).  Tokens are the MFMA
// *B* operand: X (with its tap halo) and the 32 x F intermediate T live in LDS.  A lane's four accumulator rows are four
// CONSECUTIVE channels of one token, so T goes to LDS as one 8-byte write and the outputs go to HBM as 16-byte stores
// without a transposition stage.  T is also written to HBM (the weight gradients need it), by the wave that produced it
// and from its own LDS columns -- phase 1 needs no workgroup barrier at all; the only barrier separates the phases.
// Backward: the ReLU / dropout gate (the saved hidden activation) is staged, coalesced, into the very LDS cells that the
// gradient tile will overwrite.
//
// HBM bytes at M = 6528, F = 1024 (forward): X 1.7 MB + T 13.4 MB written + residual / output 6.7 MB + 0.5 MB weights
// = 22.3 MB against 35.6 MB for the two-launch form.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define FP_THREADS 512
#define FP_BM 32
#define FP_K1 128
#define FP_N 128
#define FP_F 1024
#define FP_XP (FP_K1 + 16)   // X tile pitch in elements: 288 B = 32 mod 64 (conflict-free 16-byte fragment reads)
#define FP_XROWS 40          // 32 tokens + halo of up to 4 rows either side
#define FP_TP (FP_F + 16)    // T tile pitch in elements

__device__ __forceinline__ unsigned fp_pack2(float a, float b) {
  bf16x4 t = {(__bf16)a, (__bf16)b, (__bf16)0.f, (__bf16)0.f};
  return ((u32x2&)t).x;
}
__device__ __forceinline__ float fp_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float fp_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// KT2 = taps of PHASE 2 (the backward pass of a k = 3 first convolution: dh[m] = sum_tap dz[m + pad - tap] . W1[tap]):
// the tile then covers 32 rows of the intermediate of which the inner 32 - (KT2 - 1) are output rows -- the halo rows are
// recomputed by the neighbouring workgroup (6 % more phase-1 work instead of a second launch that re-reads dz).
template <bool BWD, int KT2>
__global__ __launch_bounds__(FP_THREADS) void ffn_pair_kernel(const kantts_ffn_args g) {
#include "ffn_pair_body.inc"
}
static bool fp_aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

extern "C" int kantts_ffn_pair(const kantts_ffn_args* gp, void* stream) {
  if (!gp) return KANTTS_E_BADARG;
  const kantts_ffn_args& g = *gp;
  if (!g.x || !g.w1 || !g.w2 || !g.y || g.M < 0 || g.KT < 1) return KANTTS_E_BADARG;
  if (g.K1 != FP_K1 || g.N != FP_N || g.F != FP_F) return KANTTS_E_UNSUPPORTED;
  if (g.KT > 9 || g.pad < 0 || g.pad >= g.KT || (g.KT > 1 && g.T <= 0)) return KANTTS_E_UNSUPPORTED;
  if ((g.ldx & 7) || (g.ldy & 3) || (g.res && (g.ldr & 3))) return KANTTS_E_UNSUPPORTED;
  if (!fp_aligned16(g.x) || !fp_aligned16(g.w1) || !fp_aligned16(g.w2) || !fp_aligned16(g.y) ||
      (g.t_out && !fp_aligned16(g.t_out)) || (g.gate && !fp_aligned16(g.gate)) || (g.res && !fp_aligned16(g.res)) ||
      (g.bias1 && !fp_aligned16(g.bias1)) || (g.bias2 && !fp_aligned16(g.bias2)))
    return KANTTS_E_UNSUPPORTED;
  if (g.xdrop_p > 0.f && !g.x_f32) return KANTTS_E_UNSUPPORTED;
  if (g.gate && g.KT != 1) return KANTTS_E_UNSUPPORTED;
  const int kt2 = g.KT2 < 1 ? 1 : g.KT2;
  if (kt2 != 1 && kt2 != 3) return KANTTS_E_UNSUPPORTED;
  if (kt2 == 3 && (!g.gate || g.T <= 0 || (g.s2_step != 1 && g.s2_step != -1) || g.M % g.T)) return KANTTS_E_UNSUPPORTED;
  if (g.ln_out) {
    if (!g.ln_gamma || !g.ln_beta || !g.ln_mean || !g.ln_rstd) return KANTTS_E_BADARG;
    if (g.gate || kt2 != 1 || !fp_aligned16(g.ln_out) || !fp_aligned16(g.ln_gamma) || !fp_aligned16(g.ln_beta))
      return KANTTS_E_UNSUPPORTED;
  }
  if (g.M == 0) return KANTTS_OK;
  hipStream_t st = (hipStream_t)stream;
  if (kt2 == 3)
    hipLaunchKernelGGL((ffn_pair_kernel<true, 3>), dim3(kantts_cdiv(g.M, FP_BM - 2)), dim3(FP_THREADS), 0, st, g);
  else if (g.gate)
    hipLaunchKernelGGL((ffn_pair_kernel<true, 1>), dim3(kantts_cdiv(g.M, FP_BM)), dim3(FP_THREADS), 0, st, g);
  else
    hipLaunchKernelGGL((ffn_pair_kernel<false, 1>), dim3(kantts_cdiv(g.M, FP_BM)), dim3(FP_THREADS), 0, st, g);
  KANTTS_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------ fragment-major images
// A (R, K) matrix whose element (r, k) is src[src_off + r*sr + k*sk] (fp32 master weights, any orientation) becomes a
// bf16 image in which the 16 x 32 block (r / 16, k / 32) is 1 KB: lane = ((k % 32) / 8) * 16 + r % 16 holds the 8
// consecutive k it feeds to v_mfma_f32_16x16x32_bf16 as an A operand:
//   dst[dst_off + ((r/16)*(K/32) + k/32)*512 + lane*8 + k%8]
// One table entry per matrix, one launch for all of them (the parameter arena's refresh, also under graph capture).
// [round 5] A thread forms one lane's 16-byte fragment (eight consecutive k of one row): 32-bit index arithmetic once per
// eight elements, two 16-byte loads when the source is k-contiguous, one 16-byte store.  (Per element -- a 64-bit division,
// a scalar load and a 2-byte store each -- the refresh of a SAM-BERT step's images took 32 us of the serial tail behind the
// optimizer; profiles/r05_runPRE_sambert_steps_kernel_stats_top.csv.)
__global__ __launch_bounds__(256) void fragmajor_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst,
                                                            const kantts_fragmajor_desc* __restrict__ tab) {
  const kantts_fragmajor_desc d = tab[blockIdx.y];
  const long long total = (long long)d.R * d.K;
  const unsigned KB = (unsigned)d.K >> 5;
  const bool fast = total < (1ll << 31) && (d.dst_off & 7) == 0;
  if (fast) {
    const unsigned total8 = (unsigned)(total >> 3);
    const bool contig = d.sk == 1 && ((d.src_off | d.sr) & 3) == 0;
    for (unsigned o8 = blockIdx.x * blockDim.x + threadIdx.x; o8 < total8; o8 += gridDim.x * blockDim.x) {
      const unsigned lane = o8 & 63u, blk = o8 >> 6;
      const unsigned rb = blk / KB, kb = blk - rb * KB;
      const long long r = (long long)rb * 16 + (lane & 15u), k0 = (long long)kb * 32 + (lane >> 4) * 8;
      const float* sp = src + d.src_off + r * d.sr + k0 * d.sk;
      float v[8];
      if (contig) {
        const float4 a = reinterpret_cast<const float4*>(sp)[0], b = reinterpret_cast<const float4*>(sp)[1];
        v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = sp[(long long)e * d.sk];
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
      *reinterpret_cast<bf16x8*>(dst + d.dst_off + (long long)o8 * 8) = o;
    }
    return;
  }
  for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
    const int e = (int)(o & 7), lane = (int)((o >> 3) & 63);
    const long long blk = o >> 9;
    const int kb = (int)(blk % KB), rb = (int)(blk / KB);
    const long long r = (long long)rb * 16 + (lane & 15), k = (long long)kb * 32 + (lane >> 4) * 8 + e;
    dst[d.dst_off + o] = (__bf16)src[d.src_off + r * d.sr + k * d.sk];
  }
}

extern "C" int kantts_fragmajor_bf16(const float* src, void* dst, const kantts_fragmajor_desc* table_dev, int ndesc,
                                     int blocks_per_desc, void* stream) {
  if (!src || !dst || !table_dev || ndesc < 0 || blocks_per_desc < 1) return KANTTS_E_BADARG;
  if (ndesc == 0) return KANTTS_OK;
  hipLaunchKernelGGL(fragmajor_bf16_kernel, dim3(blocks_per_desc, ndesc), dim3(256), 0, (hipStream_t)stream, src,
                     reinterpret_cast<__bf16*>(dst), table_dev);
  KANTTS_CHECK_LAUNCH();
}
