// The attention sub-layer of an encoder FFT block as ONE launch (round 6).
//
// Reference: MultiHeadSelfAttention.forward, kantts/models/sambert/__init__.py:52-106 inside FFTBlock.forward :152-184
// (SelfAttentionEncoder, kantts_sambert.py:20-89): LayerNorm -> fused QKV projection -> 8-head scaled-dot-product attention
// with key padding (+ attention dropout) -> output projection + dropout + residual + zeroing of padded rows -> and, in the
// same epilogue, the LayerNorm of the feed-forward sub-layer that consumes the result.  Until round 5 that was three launches
// over M = B * T = 2048 rows (kantts_bgemm_nt 12 us, kantts_attn_fwd 9 us, kantts_bgemm_nt 8-12 us: each launch-latency
// sized, 2048 x 128 activations) per block, 8 blocks per forward pass.
//
// One workgroup per (sequence, block of 16 queries) (T <= 128 tokens: up to 8 workgroups per sequence, each recomputing the
// sequence's K and V -- 32 of its 36 projection MFMAs per wave -- so that the attention's VALU work, the softmax and the one
// dropout hash per four keys, spreads over 4 CUs), WAVE = HEAD (8 waves, 8 heads of 16 channels):
//   * Q^T, K^T of the wave's head come out of the bf16 MFMA (A = weight rows, B = normalised token rows from LDS) in the
//     accumulator layout "lane (kg, li): token li, channels 4 kg + r"; V comes out of the SAME fragments with the operand
//     roles swapped (A = token rows, B = weight rows): "lane (kg, li): channel li, tokens 4 kg + r".
//   * Scores S^T = K Q^T and contexts O^T = V^T P^T run on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains, like the VALU
//     attention of csrc/attn.hip) STRAIGHT FROM THOSE ACCUMULATORS: an fp32 MFMA step takes one float per lane for A[i][k] and
//     B[k][j] with (i | j) = lane % 16, k = lane / 16; a contraction index may be visited in any order, so step r takes the
//     index 4 kg + r -- which is accumulator element r of every lane.  K and Q need no transpose (contraction over channels),
//     the probabilities leave the score MFMA in exactly the layout the context MFMA wants as B (contraction over keys), and V
//     was produced transposed for it.  Nothing of the attention touches LDS.
//   * softmax statistics: a query is a column (li); its keys sit in the 4 accumulator elements, the 4 key blocks and the 4
//     lane groups kg: in-lane reductions + two shuffles.
//   * the contexts go to LDS as bf16 (the output projection's B operand), wave w then computes output channels 16 w .. +15
//     for all tokens; LayerNorm statistics of a token meet across the 8 waves through LDS (as in csrc/pnca_block.hip).
// What backward needs (the existing launches: kantts_attn_bwd, kantts_bgemm_nt/tn, LayerNorm backward) is written once:
// qkv (fp32), contexts, log-sum-exps, y1, its normalised rows and their statistics.  Same arithmetic as the three-launch
// chain (bf16 MFMA operands for the projections, fp32 attention, the chain's dropout streams); summation orders differ.
//
// The bound is latency, not a roof: 0.13 MB of weights per workgroup from L2, ~2 k MFMA cycles per wave.
// Measured (profiles/r06_run{J,K,L}_*): one workgroup per sequence with libm expf and a dropout hash per element 28.2 us;
// one hash per four keys 20.8 us; a workgroup per 16 queries: see DESIGN.md section 5.
#include <stdlib.h>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define EA_THREADS 512
#define EA_TMAX 128          // longest sequence (tokens): 8 token blocks of 16 (the 64-token instantiation holds 4)
#define EA_C 128
#define EA_H 8
#define EA_XP (EA_C + 16)   // bf16 row pitch of the token tiles: 288 B = 32 mod 64

__device__ __forceinline__ unsigned ea_pack2(float a, float b) {
  typedef __bf16 v2 __attribute__((ext_vector_type(2)));
  v2 t = {(__bf16)a, (__bf16)b};
  return (unsigned&)t;
}

__device__ __forceinline__ float ea_col_max(float v) {  // over the 4 lane groups kg that share a column li
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float ea_col_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

template <int EA_NB>  // token blocks of 16 a sequence may have (4: T <= 64, 8: T <= 128)
__global__ __launch_bounds__(EA_THREADS) void enc_attn_fwd_kernel(const kantts_enc_attn_args g) {
  __shared__ __attribute__((aligned(16))) __bf16 Xs[EA_NB * 16 * EA_XP];  // normalised rows; rows 0..15 later hold the contexts
  __shared__ float St[2 * EA_H * 16];                              // LayerNorm partial sums: [pass][wave][token]
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // = head
  const int b = blockIdx.x, qb = blockIdx.y, T = g.L;         // this workgroup: queries qb * 16 .. + 15 of sequence b
  const long long m0 = (long long)b * T;
  const int len = g.lens ? min(max(g.lens[b], 0), T) : T;
  const uint64_t seed_off = g.seed_dev ? *g.seed_dev : 0ull;
  const __bf16* __restrict__ wq = reinterpret_cast<const __bf16*>(g.wqkv);
  const __bf16* __restrict__ wf = reinterpret_cast<const __bf16*>(g.wfc);
  const float* dummy = reinterpret_cast<const float*>(g.wqkv);  // any mapped, 16-byte aligned address
  const float4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // ---- global loads in consumption order: token rows, the head's q / k / v weight fragments, biases
  constexpr int NXR = EA_NB / 2;  // 16 EA_NB rows x 16 chunks of 16 B: EA_NB / 2 per thread (source row clamped, zeroed below)
  u32x4 xr[NXR];
  {
    const __bf16* xn = reinterpret_cast<const __bf16*>(g.xn);
#pragma unroll
    for (int it = 0; it < NXR; ++it) {
      const int id = tid + EA_THREADS * it;
      const int row = min(id >> 4, T - 1);
      xr[it] = *reinterpret_cast<const u32x4*>(xn + (m0 + row) * EA_C + (id & 15) * 8);
    }
  }
  u32x4 wqf[3][4];  // row blocks wave (q), 8 + wave (k), 16 + wave (v) of the fragment-major (384 x 128) image
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      wqf[a][kk] = *reinterpret_cast<const u32x4*>(wq + ((long long)((a * EA_H + wave) * 4 + kk)) * 512 + lane * 8);
  float4 bq = *reinterpret_cast<const float4*>(g.bqkv ? g.bqkv + wave * 16 + kg * 4 : dummy);
  float4 bk = *reinterpret_cast<const float4*>(g.bqkv ? g.bqkv + EA_C + wave * 16 + kg * 4 : dummy);
  float bv = *(g.bqkv ? g.bqkv + 2 * EA_C + wave * 16 + li : dummy);
  if (!g.bqkv) {
    bq = bk = zero4;
    bv = 0.f;
  }
#pragma unroll
  for (int it = 0; it < NXR; ++it) {
    const int id = tid + EA_THREADS * it;
    const u32x4 v = ((id >> 4) < T) ? xr[it] : (u32x4){0u, 0u, 0u, 0u};
    *reinterpret_cast<u32x4*>(&Xs[(id >> 4) * EA_XP + (id & 15) * 8]) = v;
  }
  __syncthreads();

  // ---- the head's K^T (channels x tokens) and V (tokens x channels) of all tokens, Q^T of the workgroup's 16 queries
  f32x4 aq = {0.f, 0.f, 0.f, 0.f}, ak[EA_NB], av[EA_NB];
#pragma unroll
  for (int tb = 0; tb < EA_NB; ++tb) ak[tb] = av[tb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    {
      const u32x4 v = *reinterpret_cast<const u32x4*>(&Xs[(qb * 16 + li) * EA_XP + kk * 32 + kg * 8]);
      aq = __builtin_amdgcn_mfma_f32_16x16x32_bf16((const bf16x8&)wqf[0][kk], (const bf16x8&)v, aq, 0, 0, 0);
    }
#pragma unroll
    for (int tb = 0; tb < EA_NB; ++tb) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(&Xs[(tb * 16 + li) * EA_XP + kk * 32 + kg * 8]);
      const bf16x8 xf = (bf16x8&)v;
      ak[tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((const bf16x8&)wqf[1][kk], xf, ak[tb], 0, 0, 0);
      av[tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, (const bf16x8&)wqf[2][kk], av[tb], 0, 0, 0);
    }
  }
  // the output projection's weights (row block = wave) and this lane's epilogue vectors: requested now, used after the attention
  u32x4 wff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) wff[kk] = *reinterpret_cast<const u32x4*>(wf + ((long long)(wave * 4 + kk)) * 512 + lane * 8);
  const int n0 = wave * 16 + kg * 4;  // this lane's 4 output channels of the projection
  float4 bfc = *reinterpret_cast<const float4*>(g.bfc ? g.bfc + n0 : dummy);
  if (!g.bfc) bfc = zero4;
  const float4 l1g = *reinterpret_cast<const float4*>(g.ln1_gamma + n0), l1b = *reinterpret_cast<const float4*>(g.ln1_beta + n0);
  const int tq = qb * 16 + li;  // this lane's token (query / output row)
  const bool live = tq < T;
  const long long mq = m0 + min(tq, T - 1);
  const float4 xres = *reinterpret_cast<const float4*>(g.x + mq * EA_C + n0);
  const uint8_t rmq = *(g.rowmask ? g.rowmask + mq : reinterpret_cast<const uint8_t*>(dummy));
  const bool rz = g.rowmask && rmq != 0;
  // biases; q | k | v of the workgroup's OWN 16 tokens -> HBM for the backward pass (fp32, (M, 384))
  aq += (f32x4){bq.x, bq.y, bq.z, bq.w};
#pragma unroll
  for (int tb = 0; tb < EA_NB; ++tb) {
    ak[tb] += (f32x4){bk.x, bk.y, bk.z, bk.w};
    av[tb] += (f32x4){bv, bv, bv, bv};
    if (g.qkv && tb == qb) {  // (uniform)
      if (live) {
        float* row = g.qkv + (m0 + tq) * (3 * EA_C) + wave * 16 + kg * 4;
        *reinterpret_cast<f32x4*>(row) = aq;
        *reinterpret_cast<f32x4*>(row + EA_C) = ak[tb];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int tv = tb * 16 + kg * 4 + r;
        if (tv < T) g.qkv[(m0 + tv) * (3 * EA_C) + 2 * EA_C + wave * 16 + li] = av[tb][r];
      }
    }
  }

  // ---- attention of this head for the 16 queries (columns li): S^T[key][query] on the fp32 MFMA
  f32x4 ao;  // contexts, transposed: lane (kg, li) = query li, channels 4 kg + r
  {
    const uint64_t att_seed = g.att_seed + seed_off;
    f32x4 s[EA_NB];
#pragma unroll
    for (int kb = 0; kb < EA_NB; ++kb) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ak[kb][r], aq[r], acc, 0, 0, 0);
      s[kb] = acc;
    }
    // keys kb * 16 + 4 kg + r < len (mode 0 of csrc/attn.hip: every query, padded ones included, sees keys [0, len - 1])
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < EA_NB; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = kb * 16 + kg * 4 + r < len;
        s[kb][r] = ok ? s[kb][r] * 0.25f : -INFINITY;
        mx = fmaxf(mx, s[kb][r]);
      }
    mx = ea_col_max(mx);
    const int i = min(tq, T - 1);  // this lane's query
    const uint64_t rng_row = (((uint64_t)wave * g.B + b) * T + i) * (uint64_t)T;
    // One hash per FOUR keys: a lane's four keys are consecutive, kantts_dropout_scale4 serves them from one hash whenever
    // the row's first index is a multiple of 4 (one 64-bit hash per element was 13 of the first version's 28 us:
    // profiles/r06_runJ_*, where a whole sequence's 8 x 64 x 64 probabilities were evaluated on ONE CU).
    float l = 0.f;
    const bool quad_rng = (rng_row & 3ull) == 0ull;
#pragma unroll
    for (int kb = 0; kb < EA_NB; ++kb) {
      float e4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = kb * 16 + kg * 4 + r;
        e4[r] = (j < len) ? expf(s[kb][r] - mx) : 0.f;
        l += e4[r];
      }
      if (g.att_p > 0.f) {
        const uint64_t j0 = rng_row + (uint64_t)(kb * 16 + kg * 4);
        if (quad_rng) {
          kantts_dropout_scale4(g.att_p, att_seed, j0, e4);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) e4[r] *= kantts_dropout_scale(g.att_p, att_seed, j0 + r);
        }
      }
      s[kb] = (f32x4){e4[0], e4[1], e4[2], e4[3]};
    }
    l = ea_col_sum(l);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < EA_NB; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kb][r], s[kb][r], acc, 0, 0, 0);
    const float inv = (len > 0) ? 1.f / l : 0.f;
    ao = acc * inv;
    if (live) {
      if (g.o) *reinterpret_cast<f32x4*>(g.o + (m0 + tq) * EA_C + wave * 16 + kg * 4) = ao;
      if (kg == 0 && g.lse) g.lse[((long long)b * EA_H + wave) * T + tq] = (len > 0) ? (mx + logf(l)) : 0.f;
    }
  }
  __syncthreads();  // every wave has read the normalised rows: rows 0..15 of the tile become the context tile
  {
    const u32x2 pk = {ea_pack2(ao[0], ao[1]), ea_pack2(ao[2], ao[3])};
    *reinterpret_cast<u32x2*>(&Xs[li * EA_XP + wave * 16 + kg * 4]) = pk;
  }
  __syncthreads();

  // ---- y1 = rowmask(dropout(fc(context)) + x): wave w -> output channels 16 w .. 16 w + 15 of the 16 tokens
  float y1v[4];
  {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(&Xs[li * EA_XP + kk * 32 + kg * 8]);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16((const bf16x8&)wff[kk], (const bf16x8&)v, acc, 0, 0, 0);
    }
    const uint64_t sdf = g.fc_seed + seed_off;
    const long long m = m0 + tq;
    y1v[0] = acc[0] + bfc.x; y1v[1] = acc[1] + bfc.y; y1v[2] = acc[2] + bfc.z; y1v[3] = acc[3] + bfc.w;
    if (g.fc_p > 0.f) kantts_dropout_scale4(g.fc_p, sdf, (uint64_t)m * (uint64_t)EA_C + (uint64_t)n0, y1v);
    y1v[0] += xres.x; y1v[1] += xres.y; y1v[2] += xres.z; y1v[3] += xres.w;
    if (rz || !live) {
#pragma unroll
      for (int r = 0; r < 4; ++r) y1v[r] = 0.f;
    }
    if (live && g.y1) *reinterpret_cast<f32x4*>(g.y1 + m * EA_C + n0) = (f32x4){y1v[0], y1v[1], y1v[2], y1v[3]};
  }
  if (!g.xn1) return;  // (uniform) no consumer asked for the LayerNorm of y1

  // ---- LayerNorm(128) of the 16 tokens of y1: a token's channels sit in 4 lanes (kg) of each of the 8 waves
  float mu = 0.f, rs = 0.f;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float d = pass ? y1v[r] - mu : y1v[r];
      t += pass ? d * d : d;
    }
    t = ea_col_sum(t);
    if (kg == 0) St[(pass * EA_H + wave) * 16 + li] = t;
    __syncthreads();
    t = 0.f;
#pragma unroll
    for (int w = 0; w < EA_H; ++w) t += St[(pass * EA_H + w) * 16 + li];
    if (pass)
      rs = 1.0f / sqrtf(t * (1.f / 128.f) + g.ln1_eps);
    else
      mu = t * (1.f / 128.f);
  }
  if (live) {
    const long long m = m0 + tq;
    const float z0 = (y1v[0] - mu) * rs * l1g.x + l1b.x, z1 = (y1v[1] - mu) * rs * l1g.y + l1b.y;
    const float z2 = (y1v[2] - mu) * rs * l1g.z + l1b.z, z3 = (y1v[3] - mu) * rs * l1g.w + l1b.w;
    if (g.xn1_bf16) {
      const u32x2 pk = {ea_pack2(z0, z1), ea_pack2(z2, z3)};
      *reinterpret_cast<u32x2*>(reinterpret_cast<__bf16*>(g.xn1) + m * EA_C + n0) = pk;
    } else {
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.xn1) + m * EA_C + n0) = (f32x4){z0, z1, z2, z3};
    }
    if (wave == 0 && kg == 0 && g.mean1) {
      g.mean1[m] = mu;
      g.rstd1[m] = rs;
    }
  }
}

static bool ea_aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

extern "C" int kantts_enc_attn_fwd(const kantts_enc_attn_args* gp, void* stream) {
  if (!gp) return KANTTS_E_BADARG;
  const kantts_enc_attn_args& g = *gp;
  if (g.B < 0 || g.L < 0) return KANTTS_E_BADARG;
  if (g.B == 0 || g.L == 0) return KANTTS_OK;
  if (!g.x || !g.xn || !g.wqkv || !g.wfc || !g.ln1_gamma || !g.ln1_beta || !g.y1) return KANTTS_E_BADARG;
  if (g.L > EA_TMAX) return KANTTS_E_UNSUPPORTED;  // a workgroup holds K and V of a sequence of up to 128 tokens
  if (g.xn1 && (!g.mean1 || !g.rstd1)) return KANTTS_E_BADARG;
  const void* ps[] = {g.x, g.xn, g.wqkv, g.wfc, g.bqkv, g.bfc, g.ln1_gamma, g.ln1_beta, g.qkv, g.o, g.y1, g.xn1};
  for (const void* p : ps)
    if (!ea_aligned16(p)) return KANTTS_E_UNSUPPORTED;
  if (g.L <= 64)
    hipLaunchKernelGGL(enc_attn_fwd_kernel<4>, dim3(g.B, (g.L + 15) / 16), dim3(EA_THREADS), 0, (hipStream_t)stream, g);
  else
    hipLaunchKernelGGL(enc_attn_fwd_kernel<8>, dim3(g.B, (g.L + 15) / 16), dim3(EA_THREADS), 0, (hipStream_t)stream, g);
  KANTTS_CHECK_LAUNCH();
}
