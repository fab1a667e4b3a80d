// Weight (and bias) gradient of the channels-last 1-D convolutions, with LDS-resident token windows.
//
//   dw[k][n][c] += sum_{b,p,q} gate(dy[b,q,p,n]) * act(x[b, q*stride + k*dil - pad, p, g*CR + c])      (g = n / NG)
//   db[n]       += sum_{b,p,q} gate(dy[b,q,p,n])
//
// The reduction runs over tokens, so on MFMA the token axis is the k dimension of BOTH operands, while memory
// (channels-last) has channels contiguous.  gfx950's LDS transpose read (ds_read_b64_tr_b16) forms the
// k-contiguous fragments straight from the natural [token][channel] LDS image, and its per-lane row address
// absorbs the tap shift: one 32-token dy tile and ONE window of x tokens serve every tap of the block's tap
// group -- the per-tap segmented GEMM (gemm_fast.hip + z_taps) re-read dy and x from L2/HBM once per tap
// (41 times for the scale discriminators) and transposed both operands through registers.
//
// Block: 64 output channels x CT input channels x TG taps, accumulated over a slab of 32-token steps and
// added to dw with fp32 atomics (slabs of different blocks overlap in dw only, never in a token).
// CT = 64: waves 2(n) x 2(c), 2x2 fragments per tap, TG <= 8;  CT = 32: waves 4(n) x 1(c), 1x2 fragments, TG <= 16.
// Strided windows are stored de-interleaved by (token mod stride) so that consecutive q are consecutive LDS
// rows (conflict-free transpose reads at row pitch 80 / 48 elements).
//
// Reference: the autograd of Conv1d / Conv2d((k,1)) in kantts/models/hifigan/layers.py:15-91 and
// hifigan.py:217-267,332-407 (ATen computes these with MIOpen/im2col in the reference).
#include <stdlib.h>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;

#define WG_THREADS 256
#define WG_BQ 32  // tokens per step = one bf16 MFMA k-step

__device__ __forceinline__ bf16x4 wg_read_tr4(const __bf16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(p));
}

template <bool BF16>
__device__ __forceinline__ void wg_store4(void* base, int idx, float v0, float v1, float v2, float v3) {
  if (BF16) {
    bf16x4 p = {(__bf16)v0, (__bf16)v1, (__bf16)v2, (__bf16)v3};
    *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(base) + idx) = p;
  } else {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + idx) = make_float4(v0, v1, v2, v3);
  }
}

template <bool BF16, int CT>
__global__ __launch_bounds__(WG_THREADS, BF16 ? 2 : 1) void conv_wgrad_kernel(const kantts_convw_args g, int ntaps_per_block,
                                                                int steps_per_block) {
  constexpr int TG = (CT == 64) ? 8 : 16;
  constexpr int WN = (CT == 64) ? 2 : 4;  // waves along n
  constexpr int WC = 4 / WN;
  constexpr int NF = 64 / (16 * WN);      // n fragments per wave
  constexpr int CF = CT / (16 * WC);      // c fragments per wave (2)
  constexpr int PN = 80;                  // dy tile pitch (elements)
  constexpr int PC = (CT == 64) ? 80 : 48;
  constexpr int ESZ = BF16 ? 2 : 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char wg_lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wc = wave / WN;
  const int ntn = (g.NG + 63) / 64, ntc = (g.CR + CT - 1) / CT;
  // XCD-aware order: workgroup L runs on XCD L % 8; all channel tiles / tap groups of one token slab get consecutive
  // virtual ids so that the slab's dy and x rows are fetched into one L2 only
  int bx = blockIdx.x, bytap = blockIdx.y, bz = blockIdx.z;
  {
    const int gx = gridDim.x, gy = gridDim.y;
    const long long total = (long long)gx * gy * gridDim.z;
    if (total >= 64 && total < 0x7fffffffLL && gx * gy > 1) {
      const int L = (bz * gy + bytap) * gx + bx, k = L & 7, j = L >> 3;
      const int q = (int)(total >> 3), r = (int)(total & 7);
      const int vid = k * q + (k < r ? k : r) + j;
      bz = vid / (gx * gy);
      const int rem = vid - bz * (gx * gy);
      bytap = rem / gx;
      bx = rem - bytap * gx;
    }
  }
  const int ct = bx % ntc;
  bx /= ntc;
  const int nt = bx % ntn;
  const int grp = bx / ntn;
  const int n0 = grp * g.NG + nt * 64, n_end = (grp + 1) * g.NG;
  const int c0 = ct * CT;  // inside the group
  const int k0 = bytap * ntaps_per_block;
  const int nk = min(ntaps_per_block, g.K - k0);
  const int s = g.stride;

  const int steps_per_seq = (g.Tdst + WG_BQ - 1) / WG_BQ;
  const long long total_steps = (long long)g.B * g.inner * steps_per_seq;
  const long long step_lo = (long long)bz * steps_per_block;
  const long long step_hi = min(total_steps, step_lo + steps_per_block);
  if (step_lo >= step_hi || nk <= 0) return;

  // up > 1: x is read through a nearest-neighbour upsampling (virtual token u = source token u / up): the window
  // holds source tokens and every lane computes its own (repeating) row for the transpose reads
  const int up = g.up > 1 ? g.up : 1;
  const int dm = (up > 1) ? 1 : s;
  const int Wmax = (up > 1) ? ((WG_BQ - 1) * s + (nk - 1) * g.dil) / up + 3 : (WG_BQ - 1) * s + (nk - 1) * g.dil + 1;
  const int Wp = (Wmax + dm - 1) / dm;
  void* dyt = wg_lds;
  void* xw = wg_lds + WG_BQ * PN * ESZ;

  // window row of every tap of the block, computed once (two scalar divisions per tap and step otherwise: the PMC run
  // showed ~12 SALU instructions per MFMA)
  int trow[TG];
#pragma unroll
  for (int t = 0; t < TG; ++t) {
    const int a_ = t * g.dil;
    trow[t] = (a_ % dm) * Wp + a_ / dm;
  }
  f32x4 acc[TG][NF][CF];
#pragma unroll
  for (int t = 0; t < TG; ++t)
#pragma unroll
    for (int a = 0; a < NF; ++a)
#pragma unroll
      for (int c = 0; c < CF; ++c) acc[t][a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;  // threads 0..63: column sums of dy (bias gradient)
  const bool do_bias = (g.db != nullptr) && ct == 0 && bytap == 0;

  const long long x_pitch = (long long)g.inner * g.Cin_tot, dy_pitch = (long long)g.inner * g.Ntot;
  const int li = lane & 15, kg = lane >> 4;

  for (long long st = step_lo; st < step_hi; ++st) {
    const int q0 = (int)(st % steps_per_seq) * WG_BQ;
    const long long bp = st / steps_per_seq;
    const int b = (int)(bp / g.inner), pi = (int)(bp % g.inner);
    const float* dy_b = g.dy + ((long long)b * g.Tdst * g.inner + pi) * g.Ntot + n0;
    const float* gt_b = g.dy_gate ? g.dy_gate + ((long long)b * g.Tdst * g.inner + pi) * g.Ntot + n0 : nullptr;
    const float* x_b = g.x + ((long long)b * g.Tsrc * g.inner + pi) * g.Cin_tot + (long long)grp * g.CR + c0;
    __syncthreads();  // previous step's fragments are consumed
    // ---- dy tile: 32 tokens x 64 channels (2 float4 per thread)
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int row = (tid >> 4) + 16 * v, col = (tid & 15) * 4;
      const int q = q0 + row;
      const bool ok = q < g.Tdst && (n0 + col) < n_end;
      const long long o = ok ? ((long long)q * dy_pitch + col) : 0;
      float4 d = *reinterpret_cast<const float4*>(dy_b + o);
      if (!ok) d = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gt_b && ok) {
        const float4 y = *reinterpret_cast<const float4*>(gt_b + o);
        d.x *= (y.x > 0.f) ? 1.f : g.dy_gate_slope;
        d.y *= (y.y > 0.f) ? 1.f : g.dy_gate_slope;
        d.z *= (y.z > 0.f) ? 1.f : g.dy_gate_slope;
        d.w *= (y.w > 0.f) ? 1.f : g.dy_gate_slope;
      }
      wg_store4<BF16>(dyt, row * PN + col, d.x, d.y, d.z, d.w);
    }
    // ---- x window: W tokens x CT channels, de-interleaved by (token mod stride)
    {
      constexpr int LPR = CT / 4, RPP = WG_THREADS / LPR;
      const int c4 = (tid % LPR) * 4;
      const bool cok = (c0 + c4) < g.CR;
      const int lo_u = q0 * s - g.pad + k0 * g.dil;
      const int lo = (up > 1) ? ((lo_u >= 0) ? lo_u / up : -((-lo_u + up - 1) / up)) : lo_u;
      const int W = (up > 1) ? (lo_u + (WG_BQ - 1) * s + (nk - 1) * g.dil >= 0
                                    ? (lo_u + (WG_BQ - 1) * s + (nk - 1) * g.dil) / up - lo + 1 : 1)
                             : Wmax;
      for (int r0 = tid / LPR; r0 < W; r0 += RPP * 4) {
        float4 xv[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int rel = r0 + RPP * u;
          const int t = lo + rel;
          ok[u] = cok && rel < W && t >= 0 && t < g.Tsrc;
          xv[u] = *reinterpret_cast<const float4*>(x_b + (ok[u] ? ((long long)t * x_pitch + c4) : 0));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int rel = r0 + RPP * u;
          if (rel >= W) continue;
          float v0 = xv[u].x, v1 = xv[u].y, v2 = xv[u].z, v3 = xv[u].w;
          if (!ok[u]) v0 = v1 = v2 = v3 = 0.f;
          if (g.x_act) {
            v0 = v0 > 0.f ? v0 : v0 * g.x_slope;
            v1 = v1 > 0.f ? v1 : v1 * g.x_slope;
            v2 = v2 > 0.f ? v2 : v2 * g.x_slope;
            v3 = v3 > 0.f ? v3 : v3 * g.x_slope;
          }
          wg_store4<BF16>(xw, ((rel % dm) * Wp + rel / dm) * PC + c4, v0, v1, v2, v3);
        }
      }
    }
    __syncthreads();
    if (do_bias && tid < 64) {
      float q = 0.f;
      if (BF16) {
        const __bf16* col = reinterpret_cast<const __bf16*>(dyt) + tid;
        for (int r = 0; r < WG_BQ; ++r) q += (float)col[r * PN];
      } else {
        const float* col = reinterpret_cast<const float*>(dyt) + tid;
        for (int r = 0; r < WG_BQ; ++r) q += col[r * PN];
      }
      bsum += q;
    }
    if (BF16) {
      const __bf16* Dh = reinterpret_cast<const __bf16*>(dyt);
      const __bf16* Xh = reinterpret_cast<const __bf16*>(xw);
      bf16x8 af[NF];
#pragma unroll
      for (int a = 0; a < NF; ++a) {
        const __bf16* p = &Dh[(kg * 4 + (li >> 2)) * PN + (wn * NF + a) * 16 + (li & 3) * 4];
        const bf16x4 lo4 = wg_read_tr4(p), hi4 = wg_read_tr4(p + 16 * PN);
        af[a] = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int t = 0; t < TG; ++t) {
        if (t < nk) {
          const int a_ = t * g.dil;
          int rb = trow[t] + kg * 4 + (li >> 2), rb2 = rb + 16;
          if (up > 1) {
            const int lo_u = q0 * s - g.pad + k0 * g.dil;
            const int lo = (lo_u >= 0) ? lo_u / up : -((-lo_u + up - 1) / up);
            const int u1 = lo_u + (kg * 4 + (li >> 2)) * s + a_, u2 = u1 + 16 * s;
            // negative virtual tokens only occur inside the zero padding: any zero row will do (row 0 is token lo <= -1)
            rb = (u1 >= 0) ? u1 / up - lo : 0;
            rb2 = (u2 >= 0) ? u2 / up - lo : 0;
          }
          bf16x8 bfr[CF];
#pragma unroll
          for (int c = 0; c < CF; ++c) {
            const __bf16* p = &Xh[rb * PC + (wc * CF + c) * 16 + (li & 3) * 4];
            const __bf16* p2 = &Xh[rb2 * PC + (wc * CF + c) * 16 + (li & 3) * 4];
            const bf16x4 lo4 = wg_read_tr4(p), hi4 = wg_read_tr4(p2);
            bfr[c] = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
          }
#pragma unroll
          for (int a = 0; a < NF; ++a)
#pragma unroll
            for (int c = 0; c < CF; ++c)
              acc[t][a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[c], acc[t][a][c], 0, 0, 0);
        }
      }
    } else {
      const float* Df = reinterpret_cast<const float*>(dyt);
      const float* Xf = reinterpret_cast<const float*>(xw);
#pragma unroll
      for (int ks = 0; ks < WG_BQ / 4; ++ks) {
        const int qq = ks * 4 + kg;
        float af[NF];
#pragma unroll
        for (int a = 0; a < NF; ++a) af[a] = Df[qq * PN + (wn * NF + a) * 16 + li];
#pragma unroll
        for (int t = 0; t < TG; ++t) {
          if (t < nk) {
            const int a_ = t * g.dil;
            int rb = trow[t] + qq;
            if (up > 1) {
              const int lo_u = q0 * s - g.pad + k0 * g.dil;
              const int lo = (lo_u >= 0) ? lo_u / up : -((-lo_u + up - 1) / up);
              const int u1 = lo_u + qq * s + a_;
              rb = (u1 >= 0) ? u1 / up - lo : 0;
            }
            float bfr[CF];
#pragma unroll
            for (int c = 0; c < CF; ++c) bfr[c] = Xf[rb * PC + (wc * CF + c) * 16 + li];
#pragma unroll
            for (int a = 0; a < NF; ++a)
#pragma unroll
              for (int c = 0; c < CF; ++c)
                acc[t][a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a], bfr[c], acc[t][a][c], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- write-out: fp32 atomics into the tap-major gradient (K, Ntot, CR)
#pragma unroll
  for (int t = 0; t < TG; ++t) {
    if (t < nk) {
#pragma unroll
      for (int a = 0; a < NF; ++a)
#pragma unroll
        for (int c = 0; c < CF; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = n0 + (wn * NF + a) * 16 + kg * 4 + r;
            const int cc = c0 + (wc * CF + c) * 16 + li;
            if (n < n_end && cc < g.CR) atomicAdd(&g.dw[((long long)(k0 + t) * g.Ntot + n) * g.CR + cc], acc[t][a][c][r]);
          }
    }
  }
  if (do_bias && tid < 64 && (n0 + tid) < n_end) atomicAdd(&g.db[n0 + tid], bsum);
}

// ---------------------------------------------------------------------------------------------------------
// Direct form for channel counts the MFMA tiles cannot use (CR or NG not a multiple of 4: the 1/2/4-channel
// wavelet / projection convolutions and the Cout = 1 output convolutions).  A thread owns one (n, c) pair and
// keeps its <= 16 taps in registers; it walks x tokens (x[t][c] is read once, coalesced over c, and meets the
// K output tokens that tap it -- dy[q][n] is a broadcast).  With few pairs the block's threads also split the
// tokens (PP pairs x 256/PP token lanes) and meet in LDS; every block adds its sums to dw with one atomic each.
#define WD_MAXK 16
__global__ __launch_bounds__(256) void conv_wgrad_direct_kernel(const kantts_convw_args g, int PP, int slab, int k0) {
  extern __shared__ float wd_red[];  // [PP][WD_MAXK + 1] when token lanes > 1
  const int npairs = g.Ntot * g.CR;
  const int TL = 256 / PP;
  const int pl = threadIdx.x % PP, tl = threadIdx.x / PP;
  const int pair = blockIdx.x * PP + pl;
  const bool own = pair < npairs;
  const int n = own ? pair / g.CR : 0;
  const int c = (own ? pair % g.CR : 0) + (n / g.NG) * g.CR;  // input channel inside the group of n
  const int nk = min(WD_MAXK, g.K - k0);
  const long long seqs = (long long)g.B * g.inner;            // (b, p) sequences
  const int slabs_per_seq = (g.Tsrc + slab - 1) / slab;
  const long long sq = blockIdx.y / slabs_per_seq;
  const int t_lo = (int)(blockIdx.y % slabs_per_seq) * slab, t_hi = min(g.Tsrc, t_lo + slab);
  const int b = (int)(sq / g.inner), pi = (int)(sq % g.inner);
  (void)seqs;
  float acc[WD_MAXK];
#pragma unroll
  for (int k = 0; k < WD_MAXK; ++k) acc[k] = 0.f;
  float bsum = 0.f;
  const bool do_bias = g.db != nullptr && k0 == 0 && own && (pair % g.CR) == 0;
  const float* xb = g.x + ((long long)b * g.Tsrc * g.inner + pi) * g.Cin_tot + c;
  const long long x_pitch = (long long)g.inner * g.Cin_tot, dy_pitch = (long long)g.inner * g.Ntot;
  const float* dyb = g.dy + ((long long)b * g.Tdst * g.inner + pi) * g.Ntot + n;
  const float* gtb = g.dy_gate ? g.dy_gate + ((long long)b * g.Tdst * g.inner + pi) * g.Ntot + n : nullptr;
  if (own) {
    for (int t = t_lo + tl; t < t_hi; t += TL) {
      float v = xb[(long long)t * x_pitch];
      if (g.x_act) v = v > 0.f ? v : v * g.x_slope;
#pragma unroll
      for (int k = 0; k < WD_MAXK; ++k) {
        if (k < nk) {
          const int u = t + g.pad - (k0 + k) * g.dil;  // = q * stride
          const int q = u / g.stride;
          if (u >= 0 && q * g.stride == u && q < g.Tdst) {
            float d = dyb[(long long)q * dy_pitch];
            if (gtb) d *= (gtb[(long long)q * dy_pitch] > 0.f) ? 1.f : g.dy_gate_slope;
            acc[k] += d * v;
          }
        }
      }
    }
    if (do_bias) {  // bias: the slab of OUTPUT tokens with the same index range (Tdst <= Tsrc-ish; cover the rest in the last slab)
      const int q_hi = (t_hi == g.Tsrc) ? g.Tdst : min(g.Tdst, t_hi);
      for (int q = t_lo + tl; q < q_hi; q += TL) {
        float d = dyb[(long long)q * dy_pitch];
        if (gtb) d *= (gtb[(long long)q * dy_pitch] > 0.f) ? 1.f : g.dy_gate_slope;
        bsum += d;
      }
    }
  }
  if (TL > 1) {
    for (int i = threadIdx.x; i < PP * (WD_MAXK + 1); i += 256) wd_red[i] = 0.f;
    __syncthreads();
    if (own) {
#pragma unroll
      for (int k = 0; k < WD_MAXK; ++k)
        if (k < nk) atomicAdd(&wd_red[pl * (WD_MAXK + 1) + k], acc[k]);
      if (do_bias) atomicAdd(&wd_red[pl * (WD_MAXK + 1) + WD_MAXK], bsum);
    }
    __syncthreads();
    if (tl != 0 || !own) return;
#pragma unroll
    for (int k = 0; k < WD_MAXK; ++k) acc[k] = wd_red[pl * (WD_MAXK + 1) + k];
    bsum = wd_red[pl * (WD_MAXK + 1) + WD_MAXK];
  }
  if (!own) return;
#pragma unroll
  for (int k = 0; k < WD_MAXK; ++k)
    if (k < nk) atomicAdd(&g.dw[((long long)(k0 + k) * g.Ntot + n) * g.CR + (pair % g.CR)], acc[k]);
  if (do_bias) atomicAdd(&g.db[n], bsum);
}

template <bool BF16, int CT>
static int wg_launch(const kantts_convw_args& g, hipStream_t st) {
  constexpr int TG = (CT == 64) ? 8 : 16;
  constexpr int PC = (CT == 64) ? 80 : 48;
  constexpr int ESZ = BF16 ? 2 : 4;
  const int ntg = kantts_cdiv(g.K, TG);
  const int tpb = kantts_cdiv(g.K, ntg);  // taps per block, balanced over the tap groups
  const int ntn = kantts_cdiv(g.NG, 64), ntc = kantts_cdiv(g.CR, CT);
  const int up = g.up > 1 ? g.up : 1;
  const int dm = (up > 1) ? 1 : g.stride;
  const int W = (up > 1) ? ((WG_BQ - 1) * g.stride + (tpb - 1) * g.dil) / up + 3 : (WG_BQ - 1) * g.stride + (tpb - 1) * g.dil + 1;
  const int Wp = kantts_cdiv(W, dm);
  const size_t lds = (size_t)WG_BQ * 80 * ESZ + (size_t)Wp * dm * PC * ESZ;
  if (lds > 150 * 1024) return KANTTS_E_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<BF16, CT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const long long total_steps = (long long)g.B * g.inner * kantts_cdiv(g.Tdst, WG_BQ);
  const long long xy = (long long)g.groups * ntn * ntc * ntg;
  // enough token slabs to fill the chip (~2048 blocks), but at least 4 steps per block to amortise the atomics ...
  long long slabs = (2048 + xy - 1) / xy;
  if (slabs > (total_steps + 3) / 4) slabs = (total_steps + 3) / 4;
  // ... and within an atomics budget: every slab adds the whole dw once, and fp32 atomics retire at ~200 G/s (measured on
  // the SAM-BERT weight gradients, round 2); A/B on the V1 GAN step: no budget 69.1 ms of conv launches, 3 M 74.5 ms (too few
  // blocks), 12 M 63.9 ms (profiles/r02_runG_conv_shapes_cap*.log).  The generator's 32 / 64-channel residual convolutions have tiny weight
  // tensors (11 K elements at 32 ch, k = 11) and many tokens: 2048 slabs meant 23 M atomics = 137 us for a 6 GFLOP
  // contraction (profiles/r01_hifigan_conv_shapes_packed.log).  Never below ~256 blocks.
  static const char* cap_env = getenv("KANTTS_WGRAD_ATOMICS");  // A/B switch: "0" disables the budget, else millions
  const long long budget = cap_env ? (long long)(atof(cap_env) * 1048576.0) : (12ll << 20);
  if (budget > 0) {
    const long long dw_elems = (long long)g.K * g.Ntot * g.CR;
    long long cap = budget / (dw_elems > 0 ? dw_elems : 1);
    const long long floor_slabs = (256 + xy - 1) / xy;
    if (cap < floor_slabs) cap = floor_slabs;
    if (slabs > cap) slabs = cap;
  }
  if (slabs < 1) slabs = 1;
  if (slabs > 65535) slabs = 65535;
  const int spb = (int)((total_steps + slabs - 1) / slabs);
  const int nz = (int)((total_steps + spb - 1) / spb);
  dim3 grid((unsigned)(g.groups * ntn * ntc), (unsigned)ntg, (unsigned)nz);
  hipLaunchKernelGGL((conv_wgrad_kernel<BF16, CT>), grid, dim3(WG_THREADS), lds, st, g, tpb, spb);
  KANTTS_CHECK_LAUNCH();
}

extern "C" int kantts_conv_wgrad_launch(const kantts_convw_args* a, void* stream) {
  if (!a || !a->x || !a->dy || !a->dw) return KANTTS_E_BADARG;
  const kantts_convw_args& g = *a;
  if (g.B < 0 || g.Tsrc < 0 || g.Tdst < 0 || g.K < 1 || g.groups < 1 || g.NG < 1 || g.CR < 1 || g.stride < 1 ||
      g.dil < 1 || g.inner < 1 || g.up < 0)
    return KANTTS_E_BADARG;
  if (g.Ntot != g.groups * g.NG || g.Cin_tot != g.groups * g.CR) return KANTTS_E_BADARG;
  if (g.B == 0 || g.Tdst == 0) return KANTTS_OK;
  if ((g.CR & 3) || (g.NG & 3)) {
    if (g.up > 1) return KANTTS_E_UNSUPPORTED;
    const int npairs = g.Ntot * g.CR;
    int PP = 256;
    while (PP > 1 && PP / 2 >= npairs) PP /= 2;
    const long long gx = (npairs + PP - 1) / PP;
    const long long seqs = (long long)g.B * g.inner;
    // enough token slabs for ~2048 blocks, at least 8 x tokens per thread
    long long want = (2048 + gx - 1) / gx;
    long long slab = ((long long)g.Tsrc * seqs + want - 1) / want;
    const int TL = 256 / PP;
    if (slab < 8LL * TL) slab = 8LL * TL;
    if (slab > g.Tsrc) slab = g.Tsrc > 0 ? g.Tsrc : 1;
    const long long gy = seqs * ((g.Tsrc + slab - 1) / slab);
    if (gy > 65535 || gx > 0x7fffffffLL) return KANTTS_E_UNSUPPORTED;
    const size_t lds = (TL > 1) ? (size_t)PP * (WD_MAXK + 1) * sizeof(float) : 0;
    for (int k0 = 0; k0 < g.K; k0 += WD_MAXK)
      hipLaunchKernelGGL(conv_wgrad_direct_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), lds, (hipStream_t)stream, g,
                         PP, (int)slab, k0);
    KANTTS_CHECK_LAUNCH();
  }
  if ((g.CR & 3) || (g.NG & 3) || ((uintptr_t)g.x & 15) || ((uintptr_t)g.dy & 15) ||
      (g.dy_gate && ((uintptr_t)g.dy_gate & 15)))
    return KANTTS_E_UNSUPPORTED;
  if ((long long)g.groups * kantts_cdiv(g.NG, 64) * kantts_cdiv(g.CR, 32) > 65535LL * 16) return KANTTS_E_UNSUPPORTED;
  if (g.B == 0 || g.Tdst == 0) return KANTTS_OK;
  hipStream_t st = (hipStream_t)stream;
  const bool wide = g.CR > 32;
  if (g.precision == 1) return wide ? wg_launch<true, 64>(g, st) : wg_launch<true, 32>(g, st);
  if (g.precision == 0) return wide ? wg_launch<false, 64>(g, st) : wg_launch<false, 32>(g, st);
  return KANTTS_E_BADARG;
}
