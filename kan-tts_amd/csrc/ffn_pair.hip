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
  __shared__ __attribute__((aligned(16))) __bf16 Xs[FP_XROWS * FP_XP];
  __shared__ __attribute__((aligned(16))) __bf16 Ts[FP_BM * FP_TP];
  __shared__ __attribute__((aligned(16))) float B1s[FP_F];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kg = lane >> 4;
  // phase-2 tap t reads tile row (output slot + HL + s2_first + t*s2_step)
  constexpr int OUT = FP_BM - (KT2 - 1);
  const int s2_lo = KT2 == 1 ? 0 : min(g.s2_first, g.s2_first + (KT2 - 1) * g.s2_step);
  const int HL = KT2 == 1 ? 0 : -s2_lo;                 // halo rows in front of the first output row
  const int mo0 = blockIdx.x * OUT;                     // first output row of this workgroup
  const int m0 = mo0 - HL;                              // global row of tile row 0 (may be negative)
  const int M = g.M, KT = g.KT, pad = g.pad, T = g.T;
  constexpr int F = FP_F, TP = FP_TP;
  const uint64_t seed_off = g.seed_dev ? *g.seed_dev : 0ull;
  const __bf16* __restrict__ w1 = reinterpret_cast<const __bf16*>(g.w1);
  const __bf16* __restrict__ w2 = reinterpret_cast<const __bf16*>(g.w2);

  // ---- weight stream: units of 8 fragment loads (8 KB per wave, contiguous in the fragment-major images)
  //   phase 1, step s = (chunk c, tap): rows tap*F + c*256 + wave*32 + {0, 16}, 4 k-blocks of 32 each
  //   phase 2, unit u: rows (wave & 3)*32 + {0, 16}, k-blocks (wave >> 2)*16 + u*4 + {0..3}
  const int steps = (F >> 8) * KT;  // multiple of 4
  u32x4 ring[4][8];
  auto load1 = [&](u32x4* w, int s) {
    const int c = s / KT, tap = s - c * KT;
    const __bf16* p = w1 + ((long long)(tap * (F >> 4) + c * 16 + wave * 2) * 4) * 512 + lane * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = *reinterpret_cast<const u32x4*>(p + j * 512);
  };
  auto load2 = [&](u32x4* w, int u) {
    const int nq = wave & 3, kh = wave >> 2;
    const int tap = u >> 2, uu = u & 3;  // four units per tap and reduction half
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const __bf16* p = w2 + ((long long)(tap * (FP_N >> 4) + nq * 2 + a) * (F >> 5) + kh * 16 + uu * 4) * 512 + lane * 8;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) w[a * 4 + kk] = *reinterpret_cast<const u32x4*>(p + kk * 512);
    }
  };
#pragma unroll
  for (int j = 0; j < 4; ++j) load1(ring[j], j);

  // ---- X tile: LDS row j <-> global row m0 - pad + j
  {
    const int rows = FP_BM + KT - 1;
    for (int id = tid; id < rows * 16; id += FP_THREADS) {
      const int j = id >> 4, ch = id & 15;
      const long long src = (long long)m0 - pad + j;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (src >= 0 && src < M && !(g.xrowmask && g.xrowmask[src])) {
        if (g.x_f32) {
          const float* p = reinterpret_cast<const float*>(g.x) + src * g.ldx + ch * 8;
          float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
          if (g.xdrop_p > 0.f) {
            const uint64_t base = (uint64_t)src * (uint64_t)FP_K1 + (uint64_t)(ch * 8);
            const uint64_t sd = g.xdrop_seed + seed_off;
            float lo4[4] = {a.x, a.y, a.z, a.w}, hi4[4] = {b.x, b.y, b.z, b.w};
            kantts_dropout_scale4(g.xdrop_p, sd, base, lo4);
            kantts_dropout_scale4(g.xdrop_p, sd, base + 4, hi4);
            a = make_float4(lo4[0], lo4[1], lo4[2], lo4[3]);
            b = make_float4(hi4[0], hi4[1], hi4[2], hi4[3]);
          }
          v.x = fp_pack2(a.x, a.y);
          v.y = fp_pack2(a.z, a.w);
          v.z = fp_pack2(b.x, b.y);
          v.w = fp_pack2(b.z, b.w);
        } else {
          v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const __bf16*>(g.x) + src * g.ldx + ch * 8);
        }
      }
      *reinterpret_cast<u32x4*>(&Xs[j * FP_XP + ch * 8]) = v;
    }
  }

  // Everything the chunk epilogues need from global memory is fetched NOW: the vector-memory counter retires in order,
  // so a load issued inside the weight stream could only be waited for by draining the whole 4-deep ring.
  if (!BWD && tid < FP_F / 4) {
    float4 bv = {0.f, 0.f, 0.f, 0.f};
    if (g.bias1) bv = *reinterpret_cast<const float4*>(g.bias1 + tid * 4);
    *reinterpret_cast<float4*>(&B1s[tid * 4]) = bv;
  }
  // position of this lane's two tokens inside their sequences (tap validity at sequence boundaries), their row masks
  int tpos[2];
  bool rz1[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const long long m = (long long)m0 + b * 16 + li;
    tpos[b] = ((KT > 1 || KT2 > 1) && T > 0) ? (int)(((m % T) + T) % T) : 0;
    rz1[b] = !BWD && g.rowmask1 && m >= 0 && m < M && g.rowmask1[m] != 0;
  }

  // wave-local coordinates of the T tile copy-out: the wave owns 32 columns (64 bytes) of a chunk: 4 lanes x 16 bytes
  // per row, 16 rows per pass
  const int crow = lane >> 2, ccol = (lane & 3) * 8;
  // backward: the gate (saved hidden activation, 32 x F bf16) goes into the cells of the T tile that the gradient will
  // overwrite -- all of it now, next to the X tile (one exposed load latency for both; see the note on the counter above)
  if (BWD) {
    const __bf16* gp = reinterpret_cast<const __bf16*>(g.gate);
    u32x4 gq[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int id = tid + FP_THREADS * it;
      const long long row = max(0ll, min((long long)m0 + (id >> 7), (long long)M - 1));
      gq[it] = *reinterpret_cast<const u32x4*>(gp + row * F + (id & 127) * 8);
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int id = tid + FP_THREADS * it;
      *reinterpret_cast<u32x4*>(&Ts[(id >> 7) * TP + (id & 127) * 8]) = gq[it];
    }
  }
  __syncthreads();  // X tile complete

  f32x4 acc[2][2];
  auto mfma1 = [&](const u32x4* w, int s) {
    const int c = s / KT, tap = s - c * KT;
    if (tap == 0) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    bool ok[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int q = tpos[b] + tap - pad;
      ok[b] = (KT == 1) || (q >= 0 && q < T);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 bf[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        u32x4 v = *reinterpret_cast<const u32x4*>(&Xs[(b * 16 + li + tap) * FP_XP + kk * 32 + kg * 8]);
        if (!ok[b]) v = (u32x4){0u, 0u, 0u, 0u};
        bf[b] = (bf16x8&)v;
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((const bf16x8&)w[a * 4 + kk], bf[b], acc[a][b], 0, 0, 0);
    }
    if (tap != KT - 1) return;
    // ---- epilogue of chunk c: four consecutive channels of one token per accumulator
    const uint64_t sd1 = g.drop1_seed + seed_off;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int f0 = c * 256 + wave * 32 + a * 16 + kg * 4;
      float4 bs = {0.f, 0.f, 0.f, 0.f};
      if (!BWD) bs = *reinterpret_cast<const float4*>(&B1s[f0]);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int tok = b * 16 + li;
        const long long m = (long long)m0 + tok;
        float o[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
        if (BWD) {
          const u32x2 q = *reinterpret_cast<const u32x2*>(&Ts[tok * TP + f0]);
          const float gv[4] = {fp_lo(q.x), fp_hi(q.x), fp_lo(q.y), fp_hi(q.y)};
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (gv[r] > 0.f) ? o[r] * g.alpha1 : 0.f;
        } else {
          o[0] += bs.x; o[1] += bs.y; o[2] += bs.z; o[3] += bs.w;
          if (g.relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = fmaxf(o[r], 0.f);
          }
          if (g.drop1_p > 0.f) {
            const uint64_t base = (uint64_t)m * (uint64_t)F + (uint64_t)f0;  // f0 % 4 == 0: one hash
            kantts_dropout_scale4(g.drop1_p, sd1, base, o);
          }
          if (rz1[b]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = 0.f;
          }
        }
        const u32x2 pk = {fp_pack2(o[0], o[1]), fp_pack2(o[2], o[3])};
        *reinterpret_cast<u32x2*>(&Ts[tok * TP + f0]) = pk;
      }
    }
    // the wave's own 32 columns of T -> HBM (LDS operations of a wave execute in order: no barrier)
    KANTTS_WAVE_ORDERED();
    if (g.t_out) {
      __bf16* tp = reinterpret_cast<__bf16*>(g.t_out);
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int i = crow + 16 * it;
        const u32x4 v = *reinterpret_cast<const u32x4*>(&Ts[i * TP + c * 256 + wave * 32 + ccol]);
        if (i >= HL && i < HL + OUT && m0 + i < M)  // rows this workgroup owns
          *reinterpret_cast<u32x4*>(tp + ((long long)m0 + i) * F + c * 256 + wave * 32 + ccol) = v;
      }
    }
  };

  // set j of the ring holds unit s = j (mod 4); it is refilled with unit s + 4 -- the tail of phase 1 pulls in the four
  // units of phase 2, which are therefore in flight across the barrier
  for (int s = 0; s < steps; s += 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mfma1(ring[j], s + j);
      if (s + 4 < steps)
        load1(ring[j], s + 4 + j);
      else
        load2(ring[j], j);
    }
  }
  __syncthreads();  // T tile complete

  // ---- phase 2: 32 output channels x 32 tokens x half of the reduction per wave
  const int nq = wave & 3, kh = wave >> 2;
  // what the output epilogue needs from memory is requested before the contraction (it lands while the ring drains)
  const int n0 = nq * 32 + kh * 16 + kg * 4;
  // (unconditional loads -- a dummy address when an operand is absent: a load inside a branch makes hipcc drain the
  // counter where the branch re-joins)
  const float4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const float* dummy = reinterpret_cast<const float*>(g.w2);
  float4 bs2 = *reinterpret_cast<const float4*>(g.bias2 ? g.bias2 + n0 : dummy), rv[2];
  if (!g.bias2) bs2 = zero4;
  const bool ln_epi = !BWD && KT2 == 1 && g.ln_out != nullptr;
  const float4 lg4 = *reinterpret_cast<const float4*>(ln_epi ? g.ln_gamma + n0 : dummy);
  const float4 lb4 = *reinterpret_cast<const float4*>(ln_epi ? g.ln_beta + n0 : dummy);
  bool rz2[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const long long m = min((long long)mo0 + b * 16 + li, (long long)M - 1);
    rv[b] = *reinterpret_cast<const float4*>(g.res ? g.res + m * g.ldr + n0 : dummy);
    if (!g.res) rv[b] = zero4;
    const uint8_t q = *(g.rowmask2 ? g.rowmask2 + m : reinterpret_cast<const uint8_t*>(dummy));
    rz2[b] = g.rowmask2 && q != 0;
  }
  f32x4 acc2[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc2[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int tpo[2];  // position of the two OUTPUT tokens inside their sequences
#pragma unroll
  for (int b = 0; b < 2; ++b) tpo[b] = (KT2 > 1 && T > 0) ? (mo0 + b * 16 + li) % T : 0;
  auto mfma2 = [&](const u32x4* w, int u) {
    const int tap = u >> 2, uu = u & 3;
    const int sh = KT2 == 1 ? 0 : g.s2_first + tap * g.s2_step;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 bf[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int row = min(b * 16 + li + HL + sh, FP_BM - 1);  // slots beyond OUT read a clamped row (never stored)
        u32x4 v = *reinterpret_cast<const u32x4*>(&Ts[row * TP + (kh * 16 + uu * 4 + kk) * 32 + kg * 8]);
        if (KT2 > 1) {
          const int q = tpo[b] + sh;
          if (q < 0 || q >= T) v = (u32x4){0u, 0u, 0u, 0u};
        }
        bf[b] = (bf16x8&)v;
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc2[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((const bf16x8&)w[a * 4 + kk], bf[b], acc2[a][b], 0, 0, 0);
    }
  };
#pragma unroll
  for (int u0 = 0; u0 < 4 * KT2; u0 += 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mfma2(ring[j], u0 + j);
      if (u0 + 4 + j < 4 * KT2) load2(ring[j], u0 + 4 + j);
    }
  }
  // the two halves of the reduction meet through LDS (the T tile is dead): wave (nq, kh) finishes row block a = kh and
  // hands its partial sums of the other block to its partner
  __syncthreads();
  {
    float* Ex = reinterpret_cast<float*>(Ts);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const f32x4 snd = kh ? acc2[0][b] : acc2[1][b];
      *reinterpret_cast<f32x4*>(&Ex[(((nq * 2 + (kh ^ 1)) * 2 + b) * 64 + lane) * 4]) = snd;
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(&Ex[(((nq * 2 + kh) * 2 + b) * 64 + lane) * 4]);
      acc2[0][b] = (kh ? acc2[1][b] : acc2[0][b]) + v;
    }
  }

  // ---- output: four consecutive channels of one token per accumulator -> 16-byte (fp32) / 8-byte (bf16) stores
  const uint64_t sd2 = g.drop2_seed + seed_off;
  float ov[2][4];
  bool live[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const long long m = (long long)mo0 + b * 16 + li;
    live[b] = m < M && b * 16 + li < OUT;
    float* o = ov[b];
    o[0] = acc2[0][b][0] + bs2.x; o[1] = acc2[0][b][1] + bs2.y; o[2] = acc2[0][b][2] + bs2.z; o[3] = acc2[0][b][3] + bs2.w;
    if (g.drop2_p > 0.f) {
      const uint64_t base = (uint64_t)m * (uint64_t)FP_N + (uint64_t)n0;
      kantts_dropout_scale4(g.drop2_p, sd2, base, o);
    }
    o[0] += rv[b].x; o[1] += rv[b].y; o[2] += rv[b].z; o[3] += rv[b].w;
    if (rz2[b]) {
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = 0.f;
    }
    if (!live[b]) continue;
    if (g.y_bf16) {
      const u32x2 pk = {fp_pack2(o[0], o[1]), fp_pack2(o[2], o[3])};
      *reinterpret_cast<u32x2*>(reinterpret_cast<__bf16*>(g.y) + m * g.ldy + n0) = pk;
    } else {
      const f32x4 v = {o[0], o[1], o[2], o[3]};
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.y) + m * g.ldy + n0) = v;
    }
  }
  if (ln_epi) {
    // LayerNorm(128) of the rows just formed (the pre-LN sub-layer that consumes y): a token's 128 channels sit in 4 lanes
    // (kg) of each of the 8 waves -> two-pass statistics through 2 x 256 floats of LDS (the X tile is dead)
    float* St = reinterpret_cast<float*>(Xs);
    float mu[2], rs[2];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      float ps[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = pass ? ov[b][r] - mu[b] : ov[b][r];
          t += pass ? d * d : d;
        }
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        ps[b] = t;
      }
      if (kg == 0) {
        St[pass * 256 + (wave * 2 + 0) * 16 + li] = ps[0];
        St[pass * 256 + (wave * 2 + 1) * 16 + li] = ps[1];
      }
      __syncthreads();
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += St[pass * 256 + (w * 2 + b) * 16 + li];
        if (pass)
          rs[b] = 1.0f / sqrtf(t * (1.f / 128.f) + g.ln_eps);
        else
          mu[b] = t * (1.f / 128.f);
      }
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const long long m = (long long)mo0 + b * 16 + li;
      if (!live[b]) continue;
      const float* o = ov[b];
      const float y0 = (o[0] - mu[b]) * rs[b] * lg4.x + lb4.x, y1 = (o[1] - mu[b]) * rs[b] * lg4.y + lb4.y;
      const float y2 = (o[2] - mu[b]) * rs[b] * lg4.z + lb4.z, y3 = (o[3] - mu[b]) * rs[b] * lg4.w + lb4.w;
      if (g.ln_out_bf16) {
        const u32x2 pk = {fp_pack2(y0, y1), fp_pack2(y2, y3)};
        *reinterpret_cast<u32x2*>(reinterpret_cast<__bf16*>(g.ln_out) + m * FP_N + n0) = pk;
      } else {
        const f32x4 v = {y0, y1, y2, y3};
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.ln_out) + m * FP_N + n0) = v;
      }
      if (wave == 0 && kg == 0) {
        g.ln_mean[m] = mu[b];
        g.ln_rstd[m] = rs[b];
      }
    }
  }
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
__global__ __launch_bounds__(256) void fragmajor_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst,
                                                            const kantts_fragmajor_desc* __restrict__ tab) {
  const kantts_fragmajor_desc d = tab[blockIdx.y];
  const long long total = (long long)d.R * d.K;
  const int KB = d.K >> 5;
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
