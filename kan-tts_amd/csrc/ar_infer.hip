// Free-running inference loops as ONE launch each (round 5; SURVEY 8 row e1, bf16 mode).
//
//   pnca_decode_run_kernel : every step of the free-running mel decoder -- kantts/models/sambert/kantts_sambert.py:569-610
//                            (the loop), :208-253 (HybridAttentionDecoder.infer), kantts/models/sambert/__init__.py:217-306
//                            (PNCA attention under update_x_state / update_h_state), :109-149 (feed-forward).
//   dur_ar_run_kernel      : the autoregressive duration predictor -- kantts/models/sambert/adaptors.py:67-83.
//
// Until round 5 a decoder step was a replayed hipGraph of ~90 launches (0.36 ms per step at batch 1: 63 % of a batch-1
// utterance, profiles/r05_runT_inference_batch1_breakdown.log) and a duration token was eight launches issued from
// Python (20 %).  Both loops are chains of matrix-VECTOR products per sequence with no work shared between sequences, so
// here ONE workgroup owns one sequence and walks the whole loop: nothing returns to the host between steps.
//
// A step is bound by streaming the weights (8.6 MB of bf16 per decoder step) into one CU and by the latency of ~50
// dependent products, not by arithmetic.  The products run on v_mfma_f32_16x16x32_bf16 with the weight rows as the A operand
// (16 output channels x 32 inputs per instruction, 16-byte loads straight from the row-major bf16 matrix, no cross-lane
// reduction) and the sequence's vector broadcast into every column of B: the instruction does 16 x the useful work, at
// the same issue cost per weight as v_dot2 and without its 4-step lane reduction per output.  Up to 16 weight loads per
// lane are issued before the first is consumed.  Activations live in LDS; HBM sees the K / V cache and the output frames.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define AR_THREADS 512
#define AR_WAVES 8
#define AR_D 128        // model width
#define AR_H 8          // heads
#define AR_DH 16        // head width
#define AR_FF 1024      // feed-forward width
#define AR_PRE 256      // decoder prenet width
#define AR_KMAX 128     // keys per band (band width + 1)
#define AR_LK 16        // band rows kept in LDS per band (band widths up to 15); wider bands read the caches in place
#define AR_LP 260       // their row pitch in floats

__device__ __forceinline__ int ar_pad128(int k) { return (k + 127) / 128 * 128; }
__device__ __forceinline__ int ar_pad16(int n) { return (n + 15) / 16 * 16; }

// Loads whose address is "uniform base + 32-bit lane offset": written so that the base stays in SGPRs (the saddr form of
// global_load).  Folding the lane offset into the base pointer first makes every address a 64-bit VGPR pair, and the
// pipelined layer loop has ~100 of them live: 444 bytes of scratch per lane.
__device__ __forceinline__ bf16x8 ar_ldw(const __bf16* ubase, unsigned lane_elems) {
  return *reinterpret_cast<const bf16x8*>(ubase + lane_elems);
}
__device__ __forceinline__ f32x4 ar_ldf(const float* ubase, unsigned lane_elems) {
  return *reinterpret_cast<const f32x4*>(ubase + lane_elems);
}

// y[n] = sum_k W[n][k] x[k] for n < N (N % 16 == 0), K = 128 KU, x = xs[0 .. K) bf16 in LDS.  W is stored FRAGMENT-MAJOR:
// the 16 rows x 32 inputs one MFMA takes as its A operand are 1 KB contiguous, lane l's eight values at +16 l bytes
// (tile-major, then k-block; the layout of kantts_fragmajor_bf16), so one load instruction of a wave reads 1 KB of
// consecutive addresses.  (First version: row-major rows, lane (row li, chunk kg) -> sixteen different 128-byte lines per
// quarter-wave: 28 GB/s per CU, 9.5 us for the 262 KB of a feed-forward product -- profiles/r05_runW_decode_phases.log.)
// Wave w takes the 16-row tiles w, w + 8, ...; epi(n, y + bias[n]) runs once per output (lanes 0, 16, 32, 48 of the
// tile's wave, four consecutive outputs each).  TPI tiles x min(KU, 4) k-blocks of 128 = up to 16 weight loads in flight
// per lane; the bias (N floats in global memory, 16-byte aligned) is fetched with the first of them, not after the last.
template <int KU, class Epi>
__device__ __forceinline__ void ar_gemv(const __bf16* __restrict__ W, int N, const __bf16* xs,
                                        const float* __restrict__ bias, Epi epi) {
  constexpr int TPI = KU == 1 ? 4 : (KU == 2 ? 2 : 1);
  constexpr int UC = KU < 4 ? KU : 4;  // k-blocks per chunk
  // the wave index as a SCALAR: tile indices and matrix bases then live in SGPRs and a load is "scalar base + lane offset".
  // (As a vector value every one of the ~150 load addresses of the kernel became a hoisted 64-bit VGPR pair: 344 bytes of
  // scratch per lane.)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), li = lane & 15, kg = lane >> 4;
  const int ntile = N >> 4;
  const long long ldw = 128 * KU;
  // lane offsets the compiler cannot see through: otherwise "matrix base + lane offset" of every product of the kernel is
  // hoisted to the top as a 64-bit VGPR pair and most of them live in scratch
  unsigned lane8 = 8u * lane, kg4 = 4u * kg;
  KANTTS_OPAQUE_VGPR(lane8);
  KANTTS_OPAQUE_VGPR(kg4);
  for (int t0 = wave; t0 < ntile; t0 += AR_WAVES * TPI) {
    f32x4 acc[TPI], bia[TPI];
#pragma unroll
    for (int j = 0; j < TPI; ++j) {
      acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      bia[j] = ar_ldf(bias + min(t0 + j * AR_WAVES, ntile - 1) * 16, kg4);
    }
#pragma unroll
    for (int u0 = 0; u0 < KU; u0 += UC) {
      bf16x8 a[TPI][UC][4];
#pragma unroll
      for (int j = 0; j < TPI; ++j) {
        const int tile = min(t0 + j * AR_WAVES, ntile - 1);  // clamped: loads stay unconditional
        const __bf16* frag = W + (long long)tile * 16 * ldw;
#pragma unroll
        for (int u = 0; u < UC; ++u) {
          const int uu = (u0 + u < KU) ? u0 + u : KU - 1;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) a[j][u][kk] = ar_ldw(frag + (uu * 4 + kk) * 512, lane8);
        }
      }
      // every load above (and the bias) is issued before the first product: left alone, the scheduler sinks each load to
      // just in front of its MFMA (2-4 in flight instead of 16) and the bias into the epilogue's branch (a second round trip)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < UC; ++u) {
        if (u0 + u < KU) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const bf16x8 bv = *reinterpret_cast<const bf16x8*>(xs + (u0 + u) * 128 + kk * 32 + kg * 8);
#pragma unroll
            for (int j = 0; j < TPI; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j][u][kk], bv, acc[j], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < TPI; ++j) KANTTS_OPAQUE_VGPR(bia[j]);  // the bias exists HERE: its load cannot sink into the branch
    if (li == 0) {
#pragma unroll
      for (int j = 0; j < TPI; ++j) {
        const int tile = t0 + j * AR_WAVES;
        if (tile < ntile) {
#pragma unroll
          for (int r = 0; r < 4; ++r) epi(tile * 16 + 4 * kg + r, acc[j][r] + bia[j][r]);
        }
      }
    }
  }
}

// Sum over the 64 lanes of a wave, result in every lane: four DPP steps inside each row of 16 lanes, then two shuffles across
// the rows (six shuffles were 0.6 us per LayerNorm: a ds_bpermute round trip each, all dependent).
__device__ __forceinline__ float ar_row_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, false));  // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, false));  // row_mirror
  return v;
}
__device__ __forceinline__ float ar_row_max(float v) {
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, false)));
  return v;
}
__device__ __forceinline__ float ar_wave_sum(float v) {
  v = ar_row_sum(v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// LayerNorm of the 128-wide row xs (fp32, LDS) into vout (bf16, LDS): wave 0 only, two elements per lane; the caller
// synchronises the workgroup afterwards.  Two-pass variance like csrc/norm.hip.  The affine parameters arrive in registers
// (ArLnParams, loaded by lanes of wave 0 BEFORE the product that precedes the LayerNorm: their latency is off the chain).
struct ArLnParams {
  float g0, g1, b0, b1;
};
__device__ __forceinline__ ArLnParams ar_ln_load(const float* __restrict__ gamma_beta) {
  ArLnParams p = {0.f, 0.f, 0.f, 0.f};
  if (threadIdx.x < 64) {
    const int l = threadIdx.x;
    p.g0 = gamma_beta[l], p.g1 = gamma_beta[l + 64], p.b0 = gamma_beta[AR_D + l], p.b1 = gamma_beta[AR_D + l + 64];
  }
  return p;
}
__device__ __forceinline__ void ar_layernorm(const float* xs, const ArLnParams p, float eps, __bf16* vout) {
  if (threadIdx.x < 64) {
    const int l = threadIdx.x;
    const float a = xs[l], b = xs[l + 64];
    const float mean = ar_wave_sum(a + b) * (1.f / AR_D);
    const float da = a - mean, db = b - mean;
    const float var = ar_wave_sum(da * da + db * db) * (1.f / AR_D);
    const float rstd = rsqrtf(var + eps);
    vout[l] = (__bf16)fmaf(da * rstd, p.g0, p.b0);
    vout[l + 64] = (__bf16)fmaf(db * rstd, p.g1, p.b1);
  }
}

// Phase timing of the decoder kernel (-DAR_PROFILE variant build only, scripts/build_arprof.sh): thread 0 of workgroup 0
// adds the 100 MHz wall-clock ticks between two marks to ar_prof[phase]; kantts_ar_profile_read copies them out.
#ifdef AR_PROFILE
__device__ unsigned long long ar_prof[16];
#define AR_MARK(i)                                        \
  do {                                                    \
    if (threadIdx.x == 0 && blockIdx.x == 0) {            \
      const unsigned long long now_ = wall_clock64();     \
      ar_prof[i] += now_ - ar_last;                       \
      ar_last = now_;                                     \
    }                                                     \
  } while (0)
extern "C" int kantts_ar_profile_read(unsigned long long* out16, int reset) {
  if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(ar_prof), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(ar_prof), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#else
#define AR_MARK(i)
#endif

// ------------------------------------------------------------------------------------------------ decoder
struct ArDecLayout {
  long long w_p1, w_p2, w_p3, w_in, w_layer0, w_layer, w_out, w_total;
  long long f_p1, f_p2, f_p3, f_in, f_layer0, f_layer, f_lnf, f_out, f_total;
  int k_p1, k_in, n_out;
};
// offsets inside a layer's slice
#define AR_WL_QKV 0
#define AR_WL_FC (384 * 128)
#define AR_WL_W1 (AR_WL_FC + 128 * 256)
#define AR_WL_W2 (AR_WL_W1 + AR_FF * 128)
#define AR_WL_SIZE (AR_WL_W2 + 128 * AR_FF)
#define AR_FL_LN0 0
#define AR_FL_BQKV 256
#define AR_FL_BFC (256 + 384)
#define AR_FL_LN1 (AR_FL_BFC + 128)
#define AR_FL_BW1 (AR_FL_LN1 + 256)
#define AR_FL_BW2 (AR_FL_BW1 + AR_FF)
#define AR_FL_SIZE (AR_FL_BW2 + 128)

__host__ __device__ inline ArDecLayout ar_dec_layout(int d_mel, int d_mem, int d_out, int n_layer) {
  ArDecLayout y;
  y.k_p1 = (d_mel + 127) / 128 * 128;
  y.k_in = (d_mem + AR_D + 127) / 128 * 128;
  y.n_out = (d_out + 15) / 16 * 16;
  y.w_p1 = 0;
  y.w_p2 = y.w_p1 + (long long)AR_PRE * y.k_p1;
  y.w_p3 = y.w_p2 + AR_PRE * AR_PRE;
  y.w_in = y.w_p3 + AR_D * AR_PRE;
  y.w_layer0 = y.w_in + (long long)AR_D * y.k_in;
  y.w_layer = AR_WL_SIZE;
  y.w_out = y.w_layer0 + y.w_layer * n_layer;
  y.w_total = y.w_out + (long long)y.n_out * AR_D;
  y.f_p1 = 0;
  y.f_p2 = AR_PRE;
  y.f_p3 = 2 * AR_PRE;
  y.f_in = 2 * AR_PRE + AR_D;
  y.f_layer0 = 2 * AR_PRE + 2 * AR_D;
  y.f_layer = AR_FL_SIZE;
  y.f_lnf = y.f_layer0 + y.f_layer * n_layer;
  y.f_out = y.f_lnf + 2 * AR_D;
  y.f_total = y.f_out + y.n_out;
  return y;
}

__global__ __launch_bounds__(AR_THREADS) void pnca_decode_run_kernel(const kantts_decode_args g) {
  __shared__ __attribute__((aligned(16))) __bf16 vA[AR_FF];
  __shared__ __attribute__((aligned(16))) __bf16 vB[AR_FF];
  __shared__ __attribute__((aligned(16))) float xs[AR_D];
  __shared__ __attribute__((aligned(16))) float qkv[3 * AR_D];
  __shared__ float sc[2][AR_H][AR_KMAX];
  __shared__ __attribute__((aligned(16))) float kvs[2][AR_LK][AR_LP];  // K | V rows of the two bands (bw < AR_LK)
  __shared__ float linv[2][AR_H];
  __shared__ float frame[AR_D];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int L = g.L, d_mem = g.d_mem, d_mel = g.d_mel, d_out = g.d_out, NL = g.n_layer;
  const ArDecLayout lay = ar_dec_layout(d_mel, d_mem, d_out, NL);
  const __bf16* W = reinterpret_cast<const __bf16*>(g.w);
  const float* F = g.f;
  const int len = min(g.lens ? g.lens[b] : L, L);
  const int bw = g.bw_seq ? g.bw_seq[b] : g.bw;
  float* outb = g.out + (long long)b * L * d_out;
  if (bw + 1 > AR_KMAX || bw < 0) {  // only reachable with a device-side band width: poison instead of a wrong answer
    for (long long i = tid; i < (long long)L * d_out; i += AR_THREADS) outb[i] = __builtin_nanf("");
    return;
  }
  const long long hkv_ld = (long long)NL * 256;
  const float* memb = g.memory + (long long)b * L * d_mem;
  const float* hkvb = g.hkv + (long long)b * L * hkv_ld;
  if (tid < AR_D) frame[tid] = 0.f;
  __syncthreads();
  const ArLnParams lnf = ar_ln_load(F + lay.f_lnf);
#ifdef AR_PROFILE
  unsigned long long ar_last = wall_clock64();
#endif
  for (int step = 0; step < L; ++step) {
    const bool live = step < len;
    if (live) {
      // this step's memory row: in flight while the prenet runs
      float memv = 0.f;
      if (tid < d_mem) memv = memb[(long long)step * d_mem + tid];
      // ---- prenet: d_mel -> 256 -> 256 -> 128 (ReLU, ReLU, none); dropout is off outside training
      for (int k = tid; k < lay.k_p1; k += AR_THREADS) vA[k] = (__bf16)(k < d_mel ? frame[k] : 0.f);
      __syncthreads();
      ar_gemv<1>(W + lay.w_p1, AR_PRE, vA, F + lay.f_p1, [&](int n, float v) { vB[n] = (__bf16)fmaxf(v, 0.f); });
      __syncthreads();
      ar_gemv<2>(W + lay.w_p2, AR_PRE, vB, F + lay.f_p2, [&](int n, float v) { vA[n] = (__bf16)fmaxf(v, 0.f); });
      __syncthreads();
      // input of the entry projection: [memory[b, step, :] | prenet] (zero padded to the pitch)
      for (int k = tid; k < lay.k_in; k += AR_THREADS)
        if (k < d_mem || k >= d_mem + AR_D) vB[k] = (__bf16)(k < d_mem ? memv : 0.f);
      ar_gemv<2>(W + lay.w_p3, AR_D, vA, F + lay.f_p3, [&](int n, float v) { vB[d_mem + n] = (__bf16)v; });
      __syncthreads();
      ArLnParams ln = ar_ln_load(F + lay.f_layer0 + AR_FL_LN0);
      auto in_epi = [&](int n, float v) { xs[n] = v * g.in_scale; };
      if (lay.k_in == 256)
        ar_gemv<2>(W + lay.w_in, AR_D, vB, F + lay.f_in, in_epi);
      else if (lay.k_in == 384)
        ar_gemv<3>(W + lay.w_in, AR_D, vB, F + lay.f_in, in_epi);
      else
        ar_gemv<4>(W + lay.w_in, AR_D, vB, F + lay.f_in, in_epi);
      __syncthreads();
      AR_MARK(0);
      // (A software-pipelined form of this loop -- each product's first 16 weight loads issued during the previous phase --
      // was built and measured: 182 us per step against 158.  A product is bound by what one CU can pull from L2
      // (~85 GB/s here), not by the latency in front of it, and the pipeline's registers spilled.
      // profiles/r05_runZ_decode_phases_pipelined.log)
      const bool kv_lds = bw < AR_LK;
      const int lo_x = max(0, step - bw);
      for (int layer = 0; layer < NL; ++layer) {
        const __bf16* Wl = W + lay.w_layer0 + (long long)layer * lay.w_layer;
        const float* Fl = F + lay.f_layer0 + (long long)layer * lay.f_layer;
        float* xkvb = g.xkv + ((long long)layer * g.B + b) * L * 256;
        // the K | V rows both attentions of this layer will read -- earlier steps of the own cache, the look-ahead rows of
        // the memory projection -- do not depend on this step's arithmetic: their loads are in flight during the LayerNorm
        // and land in LDS (row r of band 0 = cache row lo_x + r, of band 1 = memory row step + r)
        f32x4 kvp[4];
        if (kv_lds) {
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const int idx = tid + AR_THREADS * q4, band = idx >> 10, row = (idx >> 6) & 15, c4 = idx & 63;
            const int j = band == 0 ? min(lo_x + row, max(step - 1, 0)) : min(step + row, L - 1);
            const float* src = band == 0 ? xkvb + (long long)j * 256 : hkvb + (long long)j * hkv_ld + layer * 256;
            kvp[q4] = *reinterpret_cast<const f32x4*>(src + 4 * c4);
          }
        }
        ar_layernorm(xs, ln, g.eps, vA);
        if (kv_lds) {
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const int idx = tid + AR_THREADS * q4, band = idx >> 10, row = (idx >> 6) & 15, c4 = idx & 63;
            *reinterpret_cast<f32x4*>(&kvs[band][row][4 * c4]) = kvp[q4];
          }
        }
        __syncthreads();
        AR_MARK(1);
        // q | k | v of this step; k, v are appended to the sequence's cache
        ar_gemv<1>(Wl + AR_WL_QKV, 3 * AR_D, vA, Fl + AR_FL_BQKV, [&](int n, float v) {
          qkv[n] = v;
          if (n >= AR_D) xkvb[(long long)step * 256 + (n - AR_D)] = v;
        });
        __syncthreads();
        AR_MARK(2);
        // ---- both attentions: causal band over the own cache, look-ahead band over the memory K / V.
        // 32 lanes per (band, head): a key each (more than 32 keys: strided), softmax statistics by DPP + one shuffle.
        {
          const int band = tid >> 8, head = (tid >> 5) & 7, jl = tid & 31;
          const int lo = band == 0 ? lo_x : step;
          const int hi = band == 0 ? step : min(min(step + bw, L - 1), len - 1);
          const int nk = hi - lo + 1;  // >= 1 on a live step
          const float* kbase = (band == 0 ? xkvb : hkvb + layer * 256) + head * AR_DH;
          const long long kst = band == 0 ? 256 : hkv_ld;
          float q[AR_DH];
#pragma unroll
          for (int d = 0; d < AR_DH; ++d) q[d] = qkv[head * AR_DH + d];
          float sv[AR_KMAX / 32];
          float m = -INFINITY;
#pragma unroll
          for (int c = 0; c < AR_KMAX / 32; ++c) {
            const int j = lo + jl + 32 * c;
            float s = -INFINITY;
            if (c * 32 < nk) {
              const bool ok = j <= hi;
              const bool own = (band == 0) && (j == step);  // this step's key never left the CU: read it from LDS
              const float4* kp = kv_lds ? reinterpret_cast<const float4*>(&kvs[band][ok && !own ? j - lo : 0][head * AR_DH])
                                        : reinterpret_cast<const float4*>(kbase + (long long)(ok ? j : lo) * kst);
              float acc = 0.f;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float4 kv = kp[e];
                if (own) kv = *reinterpret_cast<const float4*>(&qkv[AR_D + head * AR_DH + 4 * e]);
                acc = fmaf(q[4 * e], kv.x, acc);
                acc = fmaf(q[4 * e + 1], kv.y, acc);
                acc = fmaf(q[4 * e + 2], kv.z, acc);
                acc = fmaf(q[4 * e + 3], kv.w, acc);
              }
              if (ok) s = acc * 0.25f;
            }
            sv[c] = s;
            m = fmaxf(m, s);
          }
          m = ar_row_max(m);
          m = fmaxf(m, __shfl_xor(m, 16));
          float l = 0.f;
#pragma unroll
          for (int c = 0; c < AR_KMAX / 32; ++c) {
            if (c * 32 < nk) {
              const float e = (lo + jl + 32 * c <= hi) ? expf(sv[c] - m) : 0.f;
              sc[band][head][jl + 32 * c] = e;
              l += e;
            }
          }
          l = ar_row_sum(l);
          l += __shfl_xor(l, 16);
          if (jl == 0) linv[band][head] = 1.f / l;
        }
        __syncthreads();
        AR_MARK(3);
        if (tid < 2 * AR_D) {
          const int band = tid >> 7, head = (tid >> 4) & 7, d = tid & 15;
          const int lo = band == 0 ? lo_x : step;
          const int hi = band == 0 ? step : min(min(step + bw, L - 1), len - 1);
          const int nk = hi - lo + 1;
          const float* vbase = (band == 0 ? xkvb : hkvb + layer * 256) + AR_D + head * AR_DH + d;
          const long long kst = band == 0 ? 256 : hkv_ld;
          float o = 0.f;
          for (int j0 = 0; j0 < nk; j0 += 4) {
            float vv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int j = lo + min(j0 + u, nk - 1);
              const bool own = (band == 0) && (j == step);
              vv[u] = kv_lds ? kvs[band][own ? 0 : j - lo][AR_D + head * AR_DH + d] : vbase[(long long)j * kst];
              if (own) vv[u] = qkv[2 * AR_D + head * AR_DH + d];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (j0 + u < nk) o = fmaf(sc[band][head][j0 + u], vv[u], o);
          }
          vB[band * AR_D + head * AR_DH + d] = (__bf16)(o * linv[band][head]);
        }
        __syncthreads();
        AR_MARK(4);
        // fc_x(ox) + fc_h(oh) + residual
        ln = ar_ln_load(Fl + AR_FL_LN1);
        ar_gemv<2>(Wl + AR_WL_FC, AR_D, vB, Fl + AR_FL_BFC, [&](int n, float v) { xs[n] = v + xs[n]; });
        __syncthreads();
        AR_MARK(5);
        ar_layernorm(xs, ln, g.eps, vA);
        __syncthreads();
        AR_MARK(6);
        ar_gemv<1>(Wl + AR_WL_W1, AR_FF, vA, Fl + AR_FL_BW1, [&](int n, float v) { vB[n] = (__bf16)fmaxf(v, 0.f); });
        __syncthreads();
        AR_MARK(7);
        if (layer + 1 < NL) ln = ar_ln_load(Fl + lay.f_layer + AR_FL_LN0);
        ar_gemv<8>(Wl + AR_WL_W2, AR_D, vB, Fl + AR_FL_BW2, [&](int n, float v) { xs[n] = v + xs[n]; });
        __syncthreads();
        AR_MARK(8);
      }
    } else {
      // a finished sequence: the reference zeroes the row after every sub-layer, so x = 0 enters the final LayerNorm
      if (tid < AR_D) xs[tid] = 0.f;
      __syncthreads();
    }
    ar_layernorm(xs, lnf, g.eps, vA);
    __syncthreads();
    ar_gemv<1>(W + lay.w_out, lay.n_out, vA, F + lay.f_out, [&](int n, float v) {
      if (n < d_out) {
        outb[(long long)step * d_out + n] = v;
        if (n >= d_out - d_mel) frame[n - (d_out - d_mel)] = v;
      }
    });
    __syncthreads();
    AR_MARK(9);
  }
}

extern "C" int kantts_pnca_decode_blob_sizes(int d_mel, int d_mem, int d_out, int n_layer, long long* w_elems,
                                             long long* f_elems) {
  if (d_mel < 1 || d_mel > AR_D || d_mem < 1 || d_mem + AR_D > 512 || d_out < d_mel || n_layer < 0) return KANTTS_E_UNSUPPORTED;
  const ArDecLayout lay = ar_dec_layout(d_mel, d_mem, d_out, n_layer);
  if (w_elems) *w_elems = lay.w_total;
  if (f_elems) *f_elems = lay.f_total;
  return KANTTS_OK;
}

extern "C" int kantts_pnca_decode_run(const kantts_decode_args* a, void* stream) {
  if (!a || !a->w || !a->f || !a->memory || !a->hkv || !a->xkv || !a->out || a->B < 0 || a->L < 0) return KANTTS_E_BADARG;
  if (a->d_mel < 1 || a->d_mel > AR_D || a->d_mem < 1 || a->d_mem + AR_D > 512 || a->d_out < a->d_mel || a->n_layer < 0)
    return KANTTS_E_UNSUPPORTED;
  if (!a->bw_seq && (a->bw < 0 || a->bw + 1 > AR_KMAX)) return KANTTS_E_UNSUPPORTED;
  if (a->B == 0 || a->L == 0) return KANTTS_OK;
  hipLaunchKernelGGL(pnca_decode_run_kernel, dim3(a->B), dim3(AR_THREADS), 0, (hipStream_t)stream, *a);
  KANTTS_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------ duration predictor
#define DA_H 128
#define DA_W_P2 0
#define DA_W_G0 (DA_H * DA_H)
#define DA_W_G1 (DA_W_G0 + 4 * DA_H * 2 * DA_H)
#define DA_F_WP1 0
#define DA_F_BP1 128
#define DA_F_BP2 256
#define DA_F_BG1 384
#define DA_F_WFC (384 + 512)
#define DA_F_BFC (DA_F_WFC + 128)

__device__ __forceinline__ float da_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(AR_THREADS) void dur_ar_run_kernel(const kantts_durar_args g) {
  __shared__ __attribute__((aligned(16))) __bf16 vA[2 * DA_H];
  __shared__ __attribute__((aligned(16))) __bf16 vB[2 * DA_H];
  __shared__ __attribute__((aligned(16))) float gate[4 * DA_H];
  __shared__ float part[2];
  __shared__ float xprev;
  const int tid = threadIdx.x, b = blockIdx.x, T = g.T;
  const __bf16* W = reinterpret_cast<const __bf16*>(g.w);
  const float* F = g.f;
  const int len = min(g.lens ? g.lens[b] : T, T);
  const float* gcb = g.gc + (long long)b * T * 4 * DA_H;
  float* outb = g.out + (long long)b * T;
  float c0 = 0.f, c1 = 0.f;  // cell states of unit tid (threads < 128)
  if (tid == 0) xprev = 0.f;
  if (tid < DA_H) {
    vB[DA_H + tid] = (__bf16)0.f;  // h0 of "token -1"
    vA[DA_H + tid] = (__bf16)0.f;  // h1
  }
  __syncthreads();
  for (int i = 0; i < len; ++i) {
    // prenet layer 1 (one input) and the one-row output layer stay fp32: the bf16 contractions of this mode need extents
    // that are multiples of 8, the per-launch path ran these two in fp32 as well
    if (tid < DA_H) vA[tid] = (__bf16)fmaxf(fmaf(F[DA_F_WP1 + tid], xprev, F[DA_F_BP1 + tid]), 0.f);
    __syncthreads();
    // vA[128..255] still holds h1 of the previous token, but this product only reads the first 128 inputs
    ar_gemv<1>(W + DA_W_P2, DA_H, vA, F + DA_F_BP2, [&](int n, float v) { vB[n] = (__bf16)fmaxf(v, 0.f); });
    __syncthreads();
    // cell 0: gates = gc[i] + [W_ih0[:, :128] | W_hh0] . [prenet | h0]
    ar_gemv<2>(W + DA_W_G0, 4 * DA_H, vB, gcb + (long long)i * 4 * DA_H, [&](int n, float v) { gate[n] = v; });
    __syncthreads();
    if (tid < DA_H) {
      const float gi = da_sigmoid(gate[tid]), gf = da_sigmoid(gate[DA_H + tid]);
      const float gg = tanhf(gate[2 * DA_H + tid]), go = da_sigmoid(gate[3 * DA_H + tid]);
      c0 = gf * c0 + gi * gg;
      const __bf16 h = (__bf16)(go * tanhf(c0));
      vA[tid] = h;           // input of cell 1: [h0 | h1]
      vB[DA_H + tid] = h;    // recurrent input of cell 0 at the next token
    }
    __syncthreads();
    ar_gemv<2>(W + DA_W_G1, 4 * DA_H, vA, F + DA_F_BG1, [&](int n, float v) { gate[n] = v; });
    __syncthreads();
    if (tid < DA_H) {
      const float gi = da_sigmoid(gate[tid]), gf = da_sigmoid(gate[DA_H + tid]);
      const float gg = tanhf(gate[2 * DA_H + tid]), go = da_sigmoid(gate[3 * DA_H + tid]);
      c1 = gf * c1 + gi * gg;
      const float h = go * tanhf(c1);
      vA[DA_H + tid] = (__bf16)h;
      // output layer: one row
      const float p = ar_wave_sum(F[DA_F_WFC + tid] * h);
      if ((tid & 63) == 0) part[tid >> 6] = p;
    }
    __syncthreads();
    if (tid == 0) {
      const float y = fmaxf(part[0] + part[1] + F[DA_F_BFC], 0.f);
      xprev = y;
      outb[i] = y;
    }
    __syncthreads();
  }
  for (int i = len + tid; i < T; i += AR_THREADS) outb[i] = 0.f;
}

// ---- fp32 twin (round 6).  In bf16 mode the token-level front of inference (text encoder, variance adaptor, this loop)
// stays fp32 so that the INDEX tensors of inference -- durations = int(exp(log_dur) - 1 + 0.5), the regulated lengths, the
// band widths (kantts_sambert.py:455-460, 989-993) -- are the reference's bit for bit; the bf16 loop above flipped ~0.2 % of
// the durations.  Same structure, a workgroup per sequence; the products are plain fp32 FMAs in k order.
//   w : fp32 blob, each matrix (N, K) stored K-CHUNK-MAJOR: element (n, k) at ((k / 4) * N + n) * 4 + k % 4, so that
//       thread n reads 16 bytes per chunk and a wave 1 KB of consecutive addresses:  P2 128 x 128 | G0 512 x 256 | G1 512 x 256
//   f : as above.
// The bound is the 1.1 MB weight stream per token from L2 into one CU (the bf16 loop streams half of it).
template <int NCH>
__device__ __forceinline__ float da_dot_f32(const float* __restrict__ Wn, int pitch, const float* xs) {
  // Wn = chunk 0 of this thread's output; pitch = floats between two chunks (4 N); xs = the chunked inputs in LDS
  float acc = 0.f;
  constexpr int G = NCH < 16 ? NCH : 16;
#pragma unroll
  for (int c0 = 0; c0 < NCH; c0 += G) {
    f32x4 w[G];
#pragma unroll
    for (int j = 0; j < G; ++j) w[j] = *reinterpret_cast<const f32x4*>(Wn + (long long)(c0 + j) * pitch);
    __builtin_amdgcn_sched_barrier(0);  // all G loads in flight before the first FMA (see ar_gemv)
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(xs + 4 * (c0 + j));
      acc = fmaf(w[j][0], x[0], acc);
      acc = fmaf(w[j][1], x[1], acc);
      acc = fmaf(w[j][2], x[2], acc);
      acc = fmaf(w[j][3], x[3], acc);
    }
  }
  return acc;
}

__global__ __launch_bounds__(AR_THREADS) void dur_ar_run_f32_kernel(const kantts_durar_args g) {
  static_assert(AR_THREADS == 4 * DA_H, "a thread per gate pre-activation");
  __shared__ __attribute__((aligned(16))) float vA[2 * DA_H];
  __shared__ __attribute__((aligned(16))) float vB[2 * DA_H];
  __shared__ __attribute__((aligned(16))) float gate[4 * DA_H];
  __shared__ float part[2];
  __shared__ float xprev;
  const int tid = threadIdx.x, b = blockIdx.x, T = g.T;
  const float* W = reinterpret_cast<const float*>(g.w);
  const float* F = g.f;
  const int len = min(g.lens ? g.lens[b] : T, T);
  const float* gcb = g.gc + (long long)b * T * 4 * DA_H;
  float* outb = g.out + (long long)b * T;
  float c0 = 0.f, c1 = 0.f;
  if (tid == 0) xprev = 0.f;
  if (tid < DA_H) {
    vB[DA_H + tid] = 0.f;
    vA[DA_H + tid] = 0.f;
  }
  __syncthreads();
  const int pn = tid & (DA_H - 1), pq = tid >> 7;  // prenet layer 2: output pn, k quarter pq
  for (int i = 0; i < len; ++i) {
    const float gci = gcb[(long long)i * 4 * DA_H + tid];  // requested before the products, used after them
    if (tid < DA_H) vA[tid] = fmaxf(fmaf(F[DA_F_WP1 + tid], xprev, F[DA_F_BP1 + tid]), 0.f);
    __syncthreads();
    // prenet layer 2 (128 x 128): four threads per output, a quarter of k each; combined in a fixed order
    gate[pq * DA_H + pn] = da_dot_f32<8>(W + DA_W_P2 + (long long)(pq * 8) * 4 * DA_H + 4 * pn, 4 * DA_H, vA + 32 * pq);
    __syncthreads();
    if (tid < DA_H)
      vB[tid] = fmaxf(((gate[tid] + gate[DA_H + tid]) + (gate[2 * DA_H + tid] + gate[3 * DA_H + tid])) + F[DA_F_BP2 + tid], 0.f);
    __syncthreads();
    // cell 0: gates = gc[i] + [W_ih0[:, :128] | W_hh0] . [prenet | h0]; thread = gate row
    gate[tid] = da_dot_f32<64>(W + DA_W_G0 + 4 * tid, 4 * 4 * DA_H, vB) + gci;
    __syncthreads();
    if (tid < DA_H) {
      const float gi = da_sigmoid(gate[tid]), gf = da_sigmoid(gate[DA_H + tid]);
      const float gg = tanhf(gate[2 * DA_H + tid]), go = da_sigmoid(gate[3 * DA_H + tid]);
      c0 = gf * c0 + gi * gg;
      const float h = go * tanhf(c0);
      vA[tid] = h;
      vB[DA_H + tid] = h;
    }
    __syncthreads();
    gate[tid] = da_dot_f32<64>(W + DA_W_G1 + 4 * tid, 4 * 4 * DA_H, vA) + F[DA_F_BG1 + tid];
    __syncthreads();
    if (tid < DA_H) {
      const float gi = da_sigmoid(gate[tid]), gf = da_sigmoid(gate[DA_H + tid]);
      const float gg = tanhf(gate[2 * DA_H + tid]), go = da_sigmoid(gate[3 * DA_H + tid]);
      c1 = gf * c1 + gi * gg;
      const float h = go * tanhf(c1);
      vA[DA_H + tid] = h;
      const float p = ar_wave_sum(F[DA_F_WFC + tid] * h);
      if ((tid & 63) == 0) part[tid >> 6] = p;
    }
    __syncthreads();
    if (tid == 0) {
      const float y = fmaxf(part[0] + part[1] + F[DA_F_BFC], 0.f);
      xprev = y;
      outb[i] = y;
    }
    __syncthreads();
  }
  for (int i = len + tid; i < T; i += AR_THREADS) outb[i] = 0.f;
}

extern "C" int kantts_dur_ar_run(const kantts_durar_args* a, void* stream) {
  if (!a || !a->w || !a->f || !a->gc || !a->out || a->B < 0 || a->T < 0) return KANTTS_E_BADARG;
  if (a->B == 0 || a->T == 0) return KANTTS_OK;
  hipLaunchKernelGGL(dur_ar_run_kernel, dim3(a->B), dim3(AR_THREADS), 0, (hipStream_t)stream, *a);
  KANTTS_CHECK_LAUNCH();
}

extern "C" int kantts_dur_ar_run_f32(const kantts_durar_args* a, void* stream) {
  if (!a || !a->w || !a->f || !a->gc || !a->out || a->B < 0 || a->T < 0) return KANTTS_E_BADARG;
  if (a->B == 0 || a->T == 0) return KANTTS_OK;
  hipLaunchKernelGGL(dur_ar_run_f32_kernel, dim3(a->B), dim3(AR_THREADS), 0, (hipStream_t)stream, *a);
  KANTTS_CHECK_LAUNCH();
}
