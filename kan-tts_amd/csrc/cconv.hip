// bf16 convolution contractions for the HiFi-GAN layers (round 3).
//
// conv_win.hip / conv_wgrad.hip read fp32 activations, round them to bf16 while staging through VGPRs and run 32-deep
// chunks with two barriers per tap: 100-370 TFLOP/s, 73 % of a GAN training step (profiles/r02_runT_*).  The yardstick
// (profiles/r02_runU_blas_reference.log) is 840 TFLOP/s for the same contractions as plain bf16 GEMMs.  What costs the
// old kernels is the staging skeleton -- fp32 loads, conversion, ds_write_b128 at ~80 B/clk/CU (MI355X_MICROARCH.md
// §LDS) -- not the MFMAs.  Here
//   * both operands ARE bf16 in memory: activations are written pre-activated by their producer (or by the one-pass
//     kantts_act_cast_bf16), weights are a bf16 tap-major image;
//   * tiles are filled by `global_load_lds` (16 bytes per lane straight into LDS, no VGPRs, no ds_write): a lane's
//     source address is its row's shifted token -- or a 16-byte block of zeros for padding, sequence / phase / group
//     edges and ragged channel counts -- so the same copy loop serves every stride, dilation, fold and upsampling rule;
//   * the LDS image is the row-major [row][64] bf16 tile with the 16-byte slot index XORed by (row >> 1) & 7
//     (conflict-free ds_read_b128 fragment reads, same rule as gemm_bf16.hip); since the DMA writes lane-linear, the
//     swizzle is applied to the SOURCE chunk a lane fetches (cdna_hip_programming.md §5.4 rule 21);
//   * the reduction runs over 64-deep steps made of two 32-channel halves, each half its own (tap, channel block):
//     a 32-channel layer pairs two taps per step, an 80-channel layer pads the third half with zeros;
//   * an NSTAGE-deep ring of LDS stages with counted `s_waitcnt vmcnt` and ONE raw s_barrier per step keeps the loads of
//     the next NSTAGE-1 steps in flight across the barrier;
//   * rows of a tile run across batch items (the period discriminators' deep layers have 10-110 rows per item).
//
//   cconv_kernel        forward / input gradient      out[M, N]   = sum_(tap, c) in[row(m, tap), c] . w[tap][n][c]
//   cconv_wgrad_kernel  weight / bias gradient        dw[tap][n][c] = sum_m dy[m][n] . x[row(m, tap)][c]
//       tokens are the reduction axis of both operands: [token][channel] LDS images, both MFMA fragments by
//       ds_read_tr16_b64; one workgroup owns a (tap, n tile, c tile) and walks a contiguous slice of the tokens.
//
// Reference: Conv1d / CausalConv1d / ConvTranspose1d of kantts/models/hifigan/layers.py:15-165, the residual blocks
// (layers.py:168-226) and the discriminator stacks (hifigan.py:200-267, 305-407).
#include <stdlib.h>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

typedef __attribute__((address_space(3))) void cc_lds_void;
typedef __attribute__((address_space(1))) const void cc_gl_void;
typedef __attribute__((address_space(3))) bf16x4 cc_lds_bf16x4;

#define CC_THREADS 256
#define CC_MAXPH 8

// the block every masked lane copies from
__device__ __attribute__((aligned(16))) unsigned int cc_zero16[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ void cc_glds16(const void* src, unsigned char* lds_dst) {
  __builtin_amdgcn_global_load_lds((cc_gl_void*)src, (cc_lds_void*)lds_dst, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void cc_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void cc_barrier() {
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ unsigned cc_pack2(float a, float b) {
  bf16x4 t = {(__bf16)a, (__bf16)b, (__bf16)0.f, (__bf16)0.f};
  return ((u32x2&)t).x;
}
__device__ __forceinline__ float cc_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float cc_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ bf16x4 cc_tr4(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((cc_lds_bf16x4*)(p));
}

struct CcPhase {
  int nv, kfirst, off0;  // valid taps of the phase: k = kfirst + j*kper, source offset off0 + j*doff, j < nv
};
struct CcArgs {
  kantts_cconv_args a;
  int kper, doff;
  int hpt;             // 32-channel halves per tap = ceil(CR / 32)
  unsigned up_magic;   // floor(2^32 / up) + 1: n / up == mulhi(n, up_magic) for 0 <= n < 2^31 / up
  CcPhase ph[CC_MAXPH];
};

// ================================================================================================ forward / dgrad
template <int BM, int BN, int WM, int WN, int NSTAGE>
__global__ __launch_bounds__(CC_THREADS) void cconv_kernel(const CcArgs P) {
  static_assert(WM * WN == 4, "four waves");
  constexpr int MREP = BM / WM / 16, NREP = BN / WN / 16;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int NA = BM / 32, NB = BN / 32;  // 1 KB copies per wave and step
  constexpr int LPS = NA + NB;
  constexpr int CLD = BN + 4;
  constexpr int EPI_BYTES = BM * CLD * 4;
  constexpr int MAIN_BYTES = (NSTAGE * STAGE > EPI_BYTES) ? NSTAGE * STAGE : EPI_BYTES;
  // ONE LDS object (a second __shared__ array makes hipcc drain vmcnt before every fragment read, guide §5 item 4a)
  extern __shared__ __attribute__((aligned(16))) unsigned char cc_lds[];
  long long* t_in = reinterpret_cast<long long*>(cc_lds + MAIN_BYTES);  // element offset of (b, token 0, p', group)
  long long* t_out = t_in + BM;                                          // element offset of the output row, -1 = no row
  int* t_mm = reinterpret_cast<int*>(t_out + BM);                        // m * in_mul

  const kantts_cconv_args& g = P.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
  const int li = lane & 15, kg = lane >> 4;

  int bx = blockIdx.x, by = blockIdx.y;
  {  // consecutive tiles of one XCD share their A rows (the n tiles of a row tile sit behind one L2)
    const int gx = gridDim.x, total = gx * gridDim.y;
    // (with a single n tile the band order -- neighbouring row tiles behind one L2 for their shared tap rows -- was measured
    // and is slower: GAN step 27.03 against 26.81 ms, profiles/r06_runGX_cconv_band_single_n_tile_ab.log)
    if (total >= 64 && gx > 1) {
      const int L = by * gx + bx, k = L & 7, j = L >> 3;
      const int q = total >> 3, r = total & 7;
      const int vid = k * q + (k < r ? k : r) + j;
      by = vid / gx;
      bx = vid - by * gx;
    }
  }
  const int phase = blockIdx.z;
  const int ntpg = (g.NG + BN - 1) / BN;
  const int grp = bx / ntpg;
  const int n0 = grp * g.NG + (bx % ntpg) * BN;
  const int n_end = (grp + 1) * g.NG;
  const int inner = g.inner;
  const int mrows = (g.Tdst - phase + g.phases - 1) / g.phases;
  const int R = mrows * inner;              // folded rows of one batch item in this phase
  const long long Mtot = (long long)g.B * R;
  const long long m0 = (long long)by * BM;
  if (m0 >= Mtot) return;

  // ---- row table: one decomposition (two integer divisions) per tile row, shared by the loaders and the epilogue
  if (tid < BM) {
    const long long G = m0 + tid;
    long long ib = 0, ob = -1;
    int mm = -(1 << 28);  // every shifted token of a missing row falls outside the sequence
    if (G < Mtot) {
      const int b = (int)(G / R);
      const int r = (int)(G - (long long)b * R);
      const int m = (inner > 1) ? r / inner : r;
      const int p = r - m * inner;
      ib = ((long long)b * g.Tsrc * inner + p) * g.Cin_tot + (long long)grp * g.CR;
      ob = (((long long)b * g.Tdst + (m * g.phases + phase)) * inner + p) * g.Ntot;
      mm = m * g.in_mul;
    }
    t_in[tid] = ib;
    t_out[tid] = ob;
    t_mm[tid] = mm;
  }
  __syncthreads();

  // ---- per-thread copy coordinates.  Wave w, copy v fills rows (v*4 + w)*8 .. +7 of a tile; lane l sits at row
  // (l >> 3), slot (l & 7) and fetches source chunk slot ^ ((row >> 1) & 7) -- the same chunk for every v
  const int sw_ld = (((wave & 1) << 2) | (lane >> 4)) & 7;
  const int chunk = (lane & 7) ^ sw_ld;  // 8-channel chunk of the 64-deep step
  const int half = chunk >> 2;           // which 32-channel half of the step
  const int ce8 = (chunk & 3) * 8;       // channel offset inside the half
  const unsigned char* pa[NA];
  int amm[NA];
#pragma unroll
  for (int v = 0; v < NA; ++v) {
    const int r = (v * 4 + wave) * 8 + (lane >> 3);
    pa[v] = reinterpret_cast<const unsigned char*>(g.in) + (t_in[r] + ce8) * 2;
    amm[v] = t_mm[r];
  }
  const unsigned char* pb[NB];
  bool bok[NB];
#pragma unroll
  for (int v = 0; v < NB; ++v) {
    const int n = n0 + (v * 4 + wave) * 8 + (lane >> 3);
    bok[v] = n < n_end;
    pb[v] = reinterpret_cast<const unsigned char*>(g.w) + ((long long)(bok[v] ? n : n0) * g.CR + ce8) * 2;
  }
  const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(cc_zero16);

  const int nv = P.ph[phase].nv, kfirst = P.ph[phase].kfirst, off0 = P.ph[phase].off0;
  const int hpt = P.hpt;
  const int nsteps = (nv * hpt + 1) >> 1;
  const int up = g.up > 1 ? g.up : 1;
  const unsigned lim = (unsigned)(g.Tsrc * up);
  const int tokstride = inner * g.Cin_tot;  // elements between consecutive tokens of one (b, p')
  const int wtap = g.Ntot * g.CR;           // elements between taps of the weight image

  f32x4 acc[MREP][NREP];
#pragma unroll
  for (int m = 0; m < MREP; ++m)
#pragma unroll
    for (int n = 0; n < NREP; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // next half-step to issue (wave-uniform): tap ordinal hj, half index hc inside the tap
  int hj = 0, hc = 0;
  auto issue = [&](int buf) {
    // the two halves of this step
    const int j0 = hj, c0 = hc;
    int j1 = j0, c1 = c0 + 1;
    if (c1 == hpt) { c1 = 0; ++j1; }
    hj = j1; hc = c1 + 1;
    if (hc == hpt) { hc = 0; ++hj; }
    const int jh = half ? j1 : j0;
    const int ce = (half ? c1 : c0) * 32 + ce8;  // channel of this lane's chunk inside the group
    const bool cok = jh < nv && ce < g.CR;
    const int off = off0 + jh * P.doff;
    const int ktap = kfirst + jh * P.kper;
    unsigned char* Ab = cc_lds + buf * STAGE;
    unsigned char* Bb = Ab + A_BYTES;
#pragma unroll
    for (int v = 0; v < NA; ++v) {
      const int tu = amm[v] + off;
      const bool ok = cok && (unsigned)tu < lim;
      const int tok = (up > 1) ? (int)__umulhi((unsigned)tu, P.up_magic) : tu;
      const long long eo = ((long long)tok * tokstride + (half ? c1 : c0) * 32) * 2;
      const unsigned char* src = ok ? pa[v] + eo : zsrc;
      cc_glds16(src, Ab + (v * 4 + wave) * 1024);
    }
    const long long wo = ((long long)ktap * wtap + (half ? c1 : c0) * 32) * 2;
#pragma unroll
    for (int v = 0; v < NB; ++v) {
      const unsigned char* src = (cok && bok[v]) ? pb[v] + wo : zsrc;
      cc_glds16(src, Bb + (v * 4 + wave) * 1024);
    }
  };
  auto compute = [&](int buf) {
    const unsigned char* Ab = cc_lds + buf * STAGE;
    const unsigned char* Bb = Ab + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[MREP], bf[NREP];
#pragma unroll
      for (int m = 0; m < MREP; ++m) {
        const int r = wr * (MREP * 16) + m * 16 + li;
        af[m] = *reinterpret_cast<const bf16x8*>(Ab + r * 128 + (((kk * 4 + kg) ^ ((r >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int n = 0; n < NREP; ++n) {
        const int r = wc * (NREP * 16) + n * 16 + li;
        bf[n] = *reinterpret_cast<const bf16x8*>(Bb + r * 128 + (((kk * 4 + kg) ^ ((r >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int n = 0; n < NREP; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bf[n], acc[m][n], 0, 0, 0);
    }
  };

  // ---- ring: step i lives in stage i % NSTAGE.  Every issue() is LPS copies per wave, also past the last step (those
  // lanes copy zeros into a stage nobody reads), so the vmcnt distance is a compile-time constant.
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s) issue(s);
  for (int i = 0; i < nsteps; ++i) {
    cc_wait_vm<(NSTAGE - 2) * LPS>();  // this wave's copies of step i have landed ...
    cc_barrier();                      // ... and everybody's; stage (i - 1) % NSTAGE is free
    int nb = i + NSTAGE - 1;
    nb -= (nb / NSTAGE) * NSTAGE;
    issue(nb);
    int cb = i - (i / NSTAGE) * NSTAGE;
    compute(cb);
  }
  cc_wait_vm<0>();
  cc_barrier();

  // ---- epilogue: accumulators through LDS, BN / 8 threads write one contiguous row piece
  float* Cs = reinterpret_cast<float*>(cc_lds);
#pragma unroll
  for (int m = 0; m < MREP; ++m)
#pragma unroll
    for (int n = 0; n < NREP; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Cs[(wr * (MREP * 16) + m * 16 + kg * 4 + r) * CLD + wc * (NREP * 16) + n * 16 + li] = acc[m][n][r];
  __syncthreads();

  constexpr int TPR = BN / 8;            // threads per row
  constexpr int RPP = CC_THREADS / TPR;  // rows per pass
  const int jc = (tid % TPR) * 8;
  const int j = n0 + jc;
  if (j >= n_end) return;
  float bs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bs[e] = 0.f;
  if (g.bias) {
    const float4 b0 = *reinterpret_cast<const float4*>(g.bias + j), b1 = *reinterpret_cast<const float4*>(g.bias + j + 4);
    bs[0] = b0.x; bs[1] = b0.y; bs[2] = b0.z; bs[3] = b0.w; bs[4] = b1.x; bs[5] = b1.y; bs[6] = b1.z; bs[7] = b1.w;
  }
#pragma unroll 2
  for (int pass = 0; pass < BM / RPP; ++pass) {
    const int rl = tid / TPR + pass * RPP;
    const long long ob = t_out[rl];
    if (ob < 0) continue;
    const long long o = ob + j;
    const float4 c0 = *reinterpret_cast<const float4*>(&Cs[rl * CLD + jc]);
    const float4 c1 = *reinterpret_cast<const float4*>(&Cs[rl * CLD + jc + 4]);
    float v[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = v[e] + bs[e];
      if (g.out_act) x = x > 0.f ? x : x * g.out_slope;
      v[e] = x;
    }
    float rv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (g.res) {
      const float4 r0 = *reinterpret_cast<const float4*>(g.res + o), r1 = *reinterpret_cast<const float4*>(g.res + o + 4);
      rv[0] = r0.x; rv[1] = r0.y; rv[2] = r0.z; rv[3] = r0.w; rv[4] = r1.x; rv[5] = r1.y; rv[6] = r1.z; rv[7] = r1.w;
      if (!g.res_after_gate) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rv[e];
      }
    }
    if (g.out_gate) {
      float gv[8];
      if (g.out_gate_bf16) {
        const u32x4 q = *reinterpret_cast<const u32x4*>(reinterpret_cast<const __bf16*>(g.out_gate) + o);
        gv[0] = cc_lo(q.x); gv[1] = cc_hi(q.x); gv[2] = cc_lo(q.y); gv[3] = cc_hi(q.y);
        gv[4] = cc_lo(q.z); gv[5] = cc_hi(q.z); gv[6] = cc_lo(q.w); gv[7] = cc_hi(q.w);
      } else {
        const float* gp = reinterpret_cast<const float*>(g.out_gate) + o;
        const float4 q0 = *reinterpret_cast<const float4*>(gp), q1 = *reinterpret_cast<const float4*>(gp + 4);
        gv[0] = q0.x; gv[1] = q0.y; gv[2] = q0.z; gv[3] = q0.w; gv[4] = q1.x; gv[5] = q1.y; gv[6] = q1.z; gv[7] = q1.w;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= (gv[e] > 0.f) ? 1.f : g.out_gate_slope;
    }
    if (g.res && g.res_after_gate) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += rv[e];
    }
    if (g.out) {
      f32x4 w0 = {v[0], v[1], v[2], v[3]}, w1 = {v[4], v[5], v[6], v[7]};
      *reinterpret_cast<f32x4*>(g.out + o) = w0;
      *reinterpret_cast<f32x4*>(g.out + o + 4) = w1;
    }
    if (g.out_bf) {
      if (g.bf_act) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * g.bf_slope;
      }
      u32x4 w = {cc_pack2(v[0], v[1]), cc_pack2(v[2], v[3]), cc_pack2(v[4], v[5]), cc_pack2(v[6], v[7])};
      *reinterpret_cast<u32x4*>(reinterpret_cast<__bf16*>(g.out_bf) + o) = w;
    }
  }
}

template <int BM, int BN, int WM, int WN, int NSTAGE>
static int cc_launch(const CcArgs& P, hipStream_t st) {
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int EPI_BYTES = BM * (BN + 4) * 4;
  constexpr int MAIN_BYTES = (NSTAGE * STAGE > EPI_BYTES) ? NSTAGE * STAGE : EPI_BYTES;
  constexpr size_t LDS = MAIN_BYTES + BM * 20;
  static_assert(LDS <= 160 * 1024, "LDS");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cconv_kernel<BM, BN, WM, WN, NSTAGE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const kantts_cconv_args& g = P.a;
  const long long mrows = (long long)((g.Tdst + g.phases - 1) / g.phases) * g.inner;
  const int ntpg = (g.NG + BN - 1) / BN;
  const long long ytiles = kantts_cdiv((long long)g.B * mrows, BM);
  if (ytiles > 65535) return KANTTS_E_UNSUPPORTED;
  dim3 grid(g.groups * ntpg, (unsigned)ytiles, g.phases);
  hipLaunchKernelGGL((cconv_kernel<BM, BN, WM, WN, NSTAGE>), grid, dim3(CC_THREADS), LDS, st, P);
  KANTTS_CHECK_LAUNCH();
}

// ================================================================================================ narrow forward / dgrad
// Channel groups of at most 64 input channels (the generator's 64- / 32-channel residual stacks, the scale
// discriminators' grouped k = 41 layers and their input gradients): in the gather form above every tap re-copies its
// 64-deep A tile from L2 -- K copies of every activation row -- while the MFMAs of a 32-channel layer need a quarter of
// a tile.  Here a workgroup copies the WINDOW of input tokens its BM outputs can touch
//   (BM - 1) * in_mul + (off_max - off_min) + 1 rows of one batch item, all CR channels
// ONCE, keeps it in LDS for every tap (tap j of output m reads window row (m - m0) * in_mul + off_j - off_min), and
// streams the weights in groups of TG taps (double-buffered copies).  One (batch item, phase) per blockIdx.z.
struct CnArgs {
  kantts_cconv_args a;
  int kper, doff;
  int tg;        // taps per weight group
  int wcopies;   // window copies per wave
  int wr;        // window rows (worst case over the phases)
  CcPhase ph[CC_MAXPH];
};

__device__ __forceinline__ int cn_swz(int row, int rowbytes) {
  // 64-byte rows: chunk ^= (-(row >> 2)) & 3;  128-byte rows: chunk ^= (row >> 1) & 7  (conflict-free b128 fragment reads)
  return rowbytes == 64 ? ((-(row >> 2)) & 3) : ((row >> 1) & 7);
}

template <int BM, int BN, int CP>  // CP = padded input channels per group (32 or 64)
__global__ __launch_bounds__(CC_THREADS) void cconv_narrow_kernel(const CnArgs P) {
  constexpr int MREP = BM / 64, NREP = BN / 16, KK = CP / 32;
  constexpr int ROWB = CP * 2;
  constexpr int RPC = 1024 / ROWB;  // rows per 1 KB copy
  constexpr int CPRW = ROWB / 16;   // chunks per row
  constexpr int CLD = BN + 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char cc_lds[];
  const kantts_cconv_args& g = P.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kg = lane >> 4;
  // ([round 6] walking the tiles in XCD-band order -- neighbouring row tiles share the (K - 1) * dilation rows of their
  // windows -- was measured on the GAN step and the generator forward and gained nothing: 27.27 / 26.19 / 27.09 ms in id order,
  // 27.28 / 27.13 / 27.21 ms banded; profiles/r06_runNB_cconv_narrow_xcd_band_ab.log.  The id order stays.)
  const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  const int phase = bz % g.phases, b = bz / g.phases;
  const int ntpg = (g.NG + BN - 1) / BN;
  const int grp = bx / ntpg;
  const int n0 = grp * g.NG + (bx % ntpg) * BN;
  const int n_end = (grp + 1) * g.NG;
  const int mrows = (g.Tdst - phase + g.phases - 1) / g.phases;
  const int m0 = by * BM;
  if (m0 >= mrows) return;
  const int nv = P.ph[phase].nv, kfirst = P.ph[phase].kfirst, off0 = P.ph[phase].off0;
  const int off_last = off0 + (nv - 1) * P.doff;
  const int off_min = nv > 0 ? min(off0, off_last) : 0;
  const int win_bytes = P.wcopies * 4 * 1024;
  const int wg_bytes = P.tg * BN * ROWB;  // one weight group
  unsigned char* Win = cc_lds;
  unsigned char* Wt = cc_lds + win_bytes;
  const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(cc_zero16);

  // ---- window: row r <-> token m0 * in_mul + off_min + r of batch item b
  const int t_base = m0 * g.in_mul + off_min;
  const int rsub = lane / CPRW, slot = lane % CPRW;
  const unsigned char* xb = reinterpret_cast<const unsigned char*>(g.in) + ((long long)b * g.Tsrc * g.Cin_tot + (long long)grp * g.CR) * 2;
  for (int v = 0; v < P.wcopies; ++v) {
    const int row = (v * 4 + wave) * RPC + rsub;
    const int chunk = slot ^ cn_swz(row, ROWB);
    const int t = t_base + row;
    const bool ok = row < P.wr && (unsigned)t < (unsigned)g.Tsrc && chunk * 8 < g.CR;
    cc_glds16(ok ? xb + ((long long)t * g.Cin_tot + chunk * 8) * 2 : zsrc, Win + (v * 4 + wave) * 1024);
  }
  // ---- weight groups: tap slot s of group q holds tap ordinal q * tg + s; rows n0 .. n0 + BN - 1
  const int wcop = (P.tg * BN + 4 * RPC - 1) / (4 * RPC);  // copies per wave and group
  auto issue_w = [&](int q, int buf) {
    unsigned char* dst = Wt + buf * wg_bytes;
    for (int v = 0; v < wcop; ++v) {
      const int row = (v * 4 + wave) * RPC + rsub;  // row inside the group image: s * BN + n
      const int sidx = row / BN, nl = row - sidx * BN;
      const int chunk = slot ^ cn_swz(row, ROWB);
      const int j = q * P.tg + sidx;
      const bool ok = sidx < P.tg && j < nv && (n0 + nl) < n_end && chunk * 8 < g.CR;
      const long long wo = (((long long)(kfirst + j * P.kper) * g.Ntot + n0 + nl) * g.CR + chunk * 8) * 2;
      if ((v * 4 + wave) * 1024 < wg_bytes)
        cc_glds16(ok ? reinterpret_cast<const unsigned char*>(g.w) + wo : zsrc, dst + (v * 4 + wave) * 1024);
    }
  };
  const int ngroups = (nv + P.tg - 1) / P.tg;
  if (ngroups > 0) issue_w(0, 0);

  f32x4 acc[MREP][NREP];
#pragma unroll
  for (int m = 0; m < MREP; ++m)
#pragma unroll
    for (int n = 0; n < NREP; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int q = 0; q < ngroups; ++q) {
    cc_wait_vm<0>();
    cc_barrier();  // group q (and, first time round, the window) has landed; group q - 1 is no longer read
    if (q + 1 < ngroups) issue_w(q + 1, (q + 1) & 1);
    const unsigned char* Wq = Wt + (q & 1) * wg_bytes;
    const int jend = min(P.tg, nv - q * P.tg);
    for (int s_ = 0; s_ < jend; ++s_) {
      const int offj = off0 + (q * P.tg + s_) * P.doff - off_min;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        bf16x8 af[MREP], bf[NREP];
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
          const int r = (wave * (BM / 4) + m * 16 + li) * g.in_mul + offj;
          af[m] = *reinterpret_cast<const bf16x8*>(Win + r * ROWB + (((kk * 4 + kg) ^ cn_swz(r, ROWB)) << 4));
        }
#pragma unroll
        for (int n = 0; n < NREP; ++n) {
          const int r = s_ * BN + n * 16 + li;
          bf[n] = *reinterpret_cast<const bf16x8*>(Wq + r * ROWB + (((kk * 4 + kg) ^ cn_swz(r, ROWB)) << 4));
        }
#pragma unroll
        for (int m = 0; m < MREP; ++m)
#pragma unroll
          for (int n = 0; n < NREP; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bf[n], acc[m][n], 0, 0, 0);
      }
    }
  }
  cc_wait_vm<0>();
  cc_barrier();

  // ---- epilogue through LDS (the window / weight images are dead)
  float* Cs = reinterpret_cast<float*>(cc_lds);
#pragma unroll
  for (int m = 0; m < MREP; ++m)
#pragma unroll
    for (int n = 0; n < NREP; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Cs[(wave * (BM / 4) + m * 16 + kg * 4 + r) * CLD + n * 16 + li] = acc[m][n][r];
  __syncthreads();
  constexpr int TPR = BN / 8;
  constexpr int RPP = CC_THREADS / TPR;
  const int jc = (tid % TPR) * 8;
  const int j = n0 + jc;
  if (j >= n_end) return;
  float bs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bs[e] = 0.f;
  if (g.bias) {
    const float4 b0 = *reinterpret_cast<const float4*>(g.bias + j), b1 = *reinterpret_cast<const float4*>(g.bias + j + 4);
    bs[0] = b0.x; bs[1] = b0.y; bs[2] = b0.z; bs[3] = b0.w; bs[4] = b1.x; bs[5] = b1.y; bs[6] = b1.z; bs[7] = b1.w;
  }
#pragma unroll 2
  for (int pass = 0; pass < BM / RPP; ++pass) {
    const int rl = tid / TPR + pass * RPP;
    const int m = m0 + rl;
    if (m >= mrows) continue;
    const long long o = ((long long)b * g.Tdst + (long long)m * g.phases + phase) * g.Ntot + j;
    const float4 c0 = *reinterpret_cast<const float4*>(&Cs[rl * CLD + jc]);
    const float4 c1 = *reinterpret_cast<const float4*>(&Cs[rl * CLD + jc + 4]);
    float v[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = v[e] + bs[e];
      if (g.out_act) x = x > 0.f ? x : x * g.out_slope;
      v[e] = x;
    }
    float rv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (g.res) {
      const float4 r0 = *reinterpret_cast<const float4*>(g.res + o), r1 = *reinterpret_cast<const float4*>(g.res + o + 4);
      rv[0] = r0.x; rv[1] = r0.y; rv[2] = r0.z; rv[3] = r0.w; rv[4] = r1.x; rv[5] = r1.y; rv[6] = r1.z; rv[7] = r1.w;
      if (!g.res_after_gate) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rv[e];
      }
    }
    if (g.out_gate) {
      float gv[8];
      if (g.out_gate_bf16) {
        const u32x4 q = *reinterpret_cast<const u32x4*>(reinterpret_cast<const __bf16*>(g.out_gate) + o);
        gv[0] = cc_lo(q.x); gv[1] = cc_hi(q.x); gv[2] = cc_lo(q.y); gv[3] = cc_hi(q.y);
        gv[4] = cc_lo(q.z); gv[5] = cc_hi(q.z); gv[6] = cc_lo(q.w); gv[7] = cc_hi(q.w);
      } else {
        const float* gp = reinterpret_cast<const float*>(g.out_gate) + o;
        const float4 q0 = *reinterpret_cast<const float4*>(gp), q1 = *reinterpret_cast<const float4*>(gp + 4);
        gv[0] = q0.x; gv[1] = q0.y; gv[2] = q0.z; gv[3] = q0.w; gv[4] = q1.x; gv[5] = q1.y; gv[6] = q1.z; gv[7] = q1.w;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= (gv[e] > 0.f) ? 1.f : g.out_gate_slope;
    }
    if (g.res && g.res_after_gate) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += rv[e];
    }
    if (g.out) {
      f32x4 w0 = {v[0], v[1], v[2], v[3]}, w1 = {v[4], v[5], v[6], v[7]};
      *reinterpret_cast<f32x4*>(g.out + o) = w0;
      *reinterpret_cast<f32x4*>(g.out + o + 4) = w1;
    }
    if (g.out_bf) {
      if (g.bf_act) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * g.bf_slope;
      }
      u32x4 w = {cc_pack2(v[0], v[1]), cc_pack2(v[2], v[3]), cc_pack2(v[4], v[5]), cc_pack2(v[6], v[7])};
      *reinterpret_cast<u32x4*>(reinterpret_cast<__bf16*>(g.out_bf) + o) = w;
    }
  }
}

template <int BM, int BN, int CP>
static int cn_launch(CnArgs& P, int span, hipStream_t st) {
  const kantts_cconv_args& g = P.a;
  constexpr int ROWB = CP * 2, RPC = 1024 / ROWB;
  P.wr = (BM - 1) * g.in_mul + span + 1;
  P.wcopies = kantts_cdiv(kantts_cdiv(P.wr, RPC), 4);
  int tg = (16 * 1024) / (BN * ROWB);  // ~16 KB of weights per group
  if (tg < 1) tg = 1;
  if (tg > g.K) tg = g.K;
  P.tg = tg;
  const size_t wg_alloc = (size_t)kantts_cdiv(kantts_cdiv(tg * BN, RPC), 4) * 4 * 1024;  // whole copies
  size_t lds = (size_t)P.wcopies * 4096 + 2 * wg_alloc;
  const size_t epi = (size_t)BM * (BN + 4) * 4;
  if (lds < epi) lds = epi;
  if (lds > 72 * 1024) return KANTTS_E_UNSUPPORTED;  // (two workgroups per CU)
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cconv_narrow_kernel<BM, BN, CP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int mrows = (g.Tdst + g.phases - 1) / g.phases;
  const int ntpg = (g.NG + BN - 1) / BN;
  if ((long long)g.B * g.phases > 65535) return KANTTS_E_UNSUPPORTED;
  dim3 grid(g.groups * ntpg, kantts_cdiv(mrows, BM), g.B * g.phases);
  hipLaunchKernelGGL((cconv_narrow_kernel<BM, BN, CP>), grid, dim3(CC_THREADS), lds, st, P);
  KANTTS_CHECK_LAUNCH();
}

static bool cc_aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
static int cc_floordiv(int a, int b) {
  int q = a / b;
  if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
  return q;
}
static int cc_gcd(int a, int b) {
  a = abs(a);
  b = abs(b);
  while (b) {
    int t = a % b;
    a = b;
    b = t;
  }
  return a;
}

extern "C" int kantts_cconv_launch(const kantts_cconv_args* ap, void* stream) {
  if (!ap || !ap->in || !ap->w || (!ap->out && !ap->out_bf)) return KANTTS_E_BADARG;
  const kantts_cconv_args& g = *ap;
  if (g.B < 0 || g.Tsrc < 0 || g.Tdst < 0 || g.K < 1 || g.groups < 1 || g.NG < 1 || g.CR < 1 || g.in_mul < 1 ||
      g.in_div < 1 || g.phases < 1 || g.inner < 1 || g.up < 0)
    return KANTTS_E_BADARG;
  if (g.Ntot != g.groups * g.NG || g.Cin_tot != g.groups * g.CR) return KANTTS_E_BADARG;
  if (g.B == 0 || g.Tdst == 0) return KANTTS_OK;
  if ((g.CR & 7) || (g.NG & 7) || g.K > 64 || g.phases > CC_MAXPH) return KANTTS_E_UNSUPPORTED;
  if (!cc_aligned16(g.in) || !cc_aligned16(g.w) || (g.bias && !cc_aligned16(g.bias)) || (g.res && !cc_aligned16(g.res)) ||
      (g.out_gate && !cc_aligned16(g.out_gate)) || (g.out && !cc_aligned16(g.out)) || (g.out_bf && !cc_aligned16(g.out_bf)))
    return KANTTS_E_UNSUPPORTED;
  const int up = g.up > 1 ? g.up : 1;
  // 32-bit element offsets inside one batch item / one weight image; mulhi division needs tokens < 2^31 / up
  if ((long long)g.Tsrc * g.inner * g.Cin_tot >= (1ll << 30) || (long long)g.K * g.Ntot * g.CR >= (1ll << 30) ||
      (long long)g.Tsrc * up * up >= (1ll << 31) || up > 64)
    return KANTTS_E_UNSUPPORTED;

  CcArgs P = {};
  P.a = g;
  const int gd = cc_gcd(g.in_kstep, g.in_div);
  P.kper = (g.in_kstep == 0) ? 1 : g.in_div / gd;
  P.doff = (g.in_kstep == 0) ? 0 : g.in_kstep * P.kper / g.in_div;
  P.hpt = (g.CR + 31) / 32;
  P.up_magic = (unsigned)((1ull << 32) / (unsigned)up) + 1u;
  for (int ph = 0; ph < g.phases; ++ph) {
    int nv = 0, kf = 0, o0 = 0;
    for (int k = 0; k < g.K; ++k) {
      const int u = g.in_add + ph + k * g.in_kstep;
      const int q = cc_floordiv(u, g.in_div);
      if (q * g.in_div != u) continue;
      if (nv == 0) {
        kf = k;
        o0 = q;
      } else if (k != kf + nv * P.kper || q != o0 + nv * P.doff) {
        return KANTTS_E_UNSUPPORTED;  // (cannot happen: the valid taps of a phase are an arithmetic progression)
      }
      ++nv;
    }
    P.ph[ph].nv = nv;
    P.ph[ph].kfirst = kf;
    P.ph[ph].off0 = o0;
  }
  hipStream_t st = (hipStream_t)stream;
  const long long rows = (long long)g.B * ((g.Tdst + g.phases - 1) / g.phases) * g.inner;
  static const char* no_narrow = getenv("KANTTS_NO_CCONV_NARROW");
  if (!no_narrow && (g.tile == 0 || g.tile == 1) && g.inner == 1 && up == 1 && g.CR <= 64 &&
      (g.Tdst + g.phases - 1) / g.phases >= 64) {
    // window form: every activation row is copied once per workgroup instead of once per tap
    CnArgs N = {};
    N.a = g;
    N.kper = P.kper;
    N.doff = P.doff;
    int span = 0;
    for (int ph = 0; ph < g.phases; ++ph) {
      N.ph[ph] = P.ph[ph];
      if (P.ph[ph].nv > 0) {
        const int sp = abs((P.ph[ph].nv - 1) * P.doff);
        if (sp > span) span = sp;
      }
    }
    const int mrows = (g.Tdst + g.phases - 1) / g.phases;
    const bool big = mrows >= 192;
    int rc;
    if (g.CR <= 32) {
      if (g.NG > 32)
        rc = big ? cn_launch<256, 64, 32>(N, span, st) : cn_launch<128, 64, 32>(N, span, st);
      else
        rc = big ? cn_launch<256, 32, 32>(N, span, st) : cn_launch<128, 32, 32>(N, span, st);
    } else {
      if (g.NG > 32)
        rc = big ? cn_launch<256, 64, 64>(N, span, st) : cn_launch<128, 64, 64>(N, span, st);
      else
        rc = big ? cn_launch<256, 32, 64>(N, span, st) : cn_launch<128, 32, 64>(N, span, st);
    }
    if (rc != KANTTS_E_UNSUPPORTED) return rc;
    if (g.tile == 1) return rc;
  }
  int tile = g.tile == 1 ? 0 : g.tile;
  int nst_arg = 0;  // args->tile = stages * 1000000 + tile code: experiments / the upsampling stages pick both
  if (tile >= 1000000) {
    nst_arg = tile / 1000000;
    tile %= 1000000;
  }
  if (tile == 0) {
    // (measured, scripts/bench_native/cconv_test: 128 x 64 beats 256 x 64 at every 64-channel shape of the model)
    if (g.NG > 64) {
      const long long t128 = kantts_cdiv(rows, 128) * g.groups * kantts_cdiv(g.NG, 128) * g.phases;
      tile = (t128 >= 200) ? 128128 : 128064;
      // [round 4] few rows, a shallow reduction (the first transposed convolution of the generator: 1024 rows x 2048
      // columns x 1024 deep): 128 x 64 tiles give 256 workgroups whose 16 reduction steps each wait for one L2 round
      // trip; 64 x 64 tiles with a four-deep ring put two workgroups on every CU with three steps in flight
      // (profiles/r04_runD_upsampling_tile_sweep.log: 17.0 -> 13.2 us; deeper reductions and larger grids lose)
      if (t128 < 200 && (long long)g.K * g.CR <= 2048 && nst_arg == 0) {
        tile = 64064;
        nst_arg = 4;
      }
    } else if (g.NG > 32) {
      tile = 128064;
    } else {
      tile = 256032;
    }
  }
  static const char* env_stage = getenv("KANTTS_CCONV_STAGES");
  int nst = nst_arg ? nst_arg : (env_stage ? atoi(env_stage) : 0);
  if (nst == 0 && tile == 128128) {
    // a grid that cannot give every CU two tiles anyway gains nothing from two co-resident workgroups: spend the LDS
    // on a 4-deep ring instead (three steps of loads in flight hide the L2 / HBM round trip of a short reduction)
    const long long t128 = kantts_cdiv(rows, 128) * g.groups * kantts_cdiv(g.NG, 128) * g.phases;
    if (t128 <= 320) nst = 4;
  }
  switch (tile) {
    case 128128:
      if (nst == 3) return cc_launch<128, 128, 2, 2, 3>(P, st);
      if (nst == 4) return cc_launch<128, 128, 2, 2, 4>(P, st);
      return cc_launch<128, 128, 2, 2, 2>(P, st);
    case 256064:
      if (nst == 3) return cc_launch<256, 64, 4, 1, 3>(P, st);
      return cc_launch<256, 64, 4, 1, 2>(P, st);
    case 128064:
      if (nst == 3) return cc_launch<128, 64, 2, 2, 3>(P, st);
      if (nst == 4) return cc_launch<128, 64, 2, 2, 4>(P, st);
      return cc_launch<128, 64, 2, 2, 2>(P, st);
    case 64128:
      if (nst == 3) return cc_launch<64, 128, 2, 2, 3>(P, st);
      if (nst == 4) return cc_launch<64, 128, 2, 2, 4>(P, st);
      return cc_launch<64, 128, 2, 2, 2>(P, st);
    case 64064:
      if (nst == 3) return cc_launch<64, 64, 2, 2, 3>(P, st);
      if (nst == 4) return cc_launch<64, 64, 2, 2, 4>(P, st);
      return cc_launch<64, 64, 2, 2, 2>(P, st);
    case 256032:
      if (nst == 3) return cc_launch<256, 32, 4, 1, 3>(P, st);
      return cc_launch<256, 32, 4, 1, 2>(P, st);
    default:
      return KANTTS_E_BADARG;
  }
}

// ================================================================================================ weight gradient
// dw[tap][n][c] (+)= sum_G dy[G][n] * x[src(G, tap)][c];  db[n] += sum_G dy[G][n]  (c tile 0, tap 0, through one extra
// MFMA against a fragment of ones: no LDS column walk).  TW x TW output tile, 64 tokens per step.
struct CwArgs {
  kantts_cconvw_args a;
  int slices;
  int to_ws;  // slices > 1 with a workspace: every slice stores its partial tile, cconv_wgrad_reduce_kernel sums them
  unsigned up_magic;
  int xcd_map;  // [round 6] 1-D grid, the tiles of one token slice on one XCD (see the kernel); 0 = the 3-D grid
};

template <int TW, int NSTAGE>
__global__ __launch_bounds__(CC_THREADS) void cconv_wgrad_kernel(const CwArgs P) {
  constexpr int REP = TW / 32;            // 16-wide fragments per wave and axis (waves 2 x 2)
  constexpr int ROWB = TW * 2;            // bytes of one image row
  constexpr int IMG = 64 * ROWB;          // one [64 tokens][TW channels] image
  constexpr int STAGE = 2 * IMG;
  constexpr int NI = IMG / (CC_THREADS * 16);  // 16-byte copies per thread and image (4 / 2)
  constexpr int LPS = 2 * NI;
  constexpr int CPR = TW / 8;             // chunks per row (16 / 8)
  constexpr int CLD = TW + 4;
  constexpr int EPI_BYTES = TW * CLD * 4;
  constexpr int MAIN_BYTES = (NSTAGE * STAGE > EPI_BYTES) ? NSTAGE * STAGE : EPI_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char cc_lds[];
  const kantts_cconvw_args& g = P.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int li = lane & 15, kg = lane >> 4;

  const int ctiles = (g.CR + TW - 1) / TW;
  const int ntpg = (g.NG + TW - 1) / TW;
  // Workgroup -> (channel tile, output-channel tile, tap, token slice).  Every tile and every tap of one token slice reads
  // the SAME rows of dy and (shifted by the tap) of x.  On the 3-D grid (x = channel tile, y = output tile, z = tap * slices
  // + slice) the hardware deals consecutive ids to the 8 XCDs in turn, so the K taps of a slice ran behind up to 8
  // different L2s and its rows crossed the fabric once per XCD.  [round 6] With at least 8 slices the launch is a 1-D grid in
  // which a slice's ctiles * ntiles * K workgroups are consecutive workgroups of ONE XCD (id % 8 = XCD, id / 8 = position
  // inside it; the last round of slices is padded with idle workgroups).
  int bx = blockIdx.x, by = blockIdx.y, tap_ = blockIdx.z / P.slices, slice_ = blockIdx.z % P.slices;
  if (P.xcd_map) {
    const int ny = g.groups * ntpg, per = ctiles * ny * g.K;
    const int j = blockIdx.x >> 3;
    slice_ = (blockIdx.x & 7) + 8 * (j / per);
    if (slice_ >= P.slices) return;
    int m = j % per;
    bx = m % ctiles;
    m /= ctiles;
    by = m % ny;
    tap_ = m / ny;
  }
  const int ct = bx % ctiles;
  const int grp = by / ntpg;
  const int n0 = grp * g.NG + (by % ntpg) * TW;
  const int n_end = (grp + 1) * g.NG;
  const int c0 = ct * TW;  // channel offset inside the group
  const int tap = tap_, slice = slice_;
  const int inner = g.inner;
  const int up = g.up > 1 ? g.up : 1;
  const unsigned lim = (unsigned)(g.Tsrc * up);
  const long long Mtot = (long long)g.B * g.Tdst * inner;
  const int NT = (int)((Mtot + 63) / 64);
  const int t_lo = (int)((long long)NT * slice / P.slices), t_hi = (int)((long long)NT * (slice + 1) / P.slices);
  if (t_lo >= t_hi) return;
  const int shift = tap * g.dil - g.pad;

  // ---- copy coordinates: copy v of wave w fills bytes (v*4 + w)*1024 .. of an image = rows of 64 / CPR lanes each
  // row = id / CPR, slot = id % CPR with id = (v*4 + w)*64 + lane; the fetched chunk is slot ^ swz(row), constant in v
  constexpr int RPC = 64 / CPR;  // rows per copy (4 / 8)
  const int rsub = lane / CPR;   // row inside the copy
  const int slot = lane % CPR;
  // TW = 128: 256-byte rows, chunk ^= (row & 7) << 1;  TW = 64: 128-byte rows, chunk ^= ((row >> 1) & 3) << 1
  const int row_lo = (wave * RPC + rsub);  // row of copy v = 0 (v adds multiples of 16 / 32 rows)
  const int swz = (TW == 128) ? ((row_lo & 7) << 1) : (((row_lo >> 1) & 3) << 1);
  const int chunk = slot ^ swz;
  const bool a_cok = (n0 + chunk * 8) < n_end;
  const bool b_cok = (c0 + chunk * 8) < g.CR;
  const unsigned char* dy_base = reinterpret_cast<const unsigned char*>(g.dy) + (long long)(n0 + chunk * 8) * 2;
  const unsigned char* x_base = reinterpret_cast<const unsigned char*>(g.x) + ((long long)grp * g.CR + c0 + chunk * 8) * 2;
  const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(cc_zero16);

  // token state of this thread's NI rows: (b, q, p') of dy row G, advanced by 64 rows per step without divisions
  int sb[NI], sq[NI], sp[NI];
  const int dq64 = 64 / inner, dp64 = 64 - dq64 * inner;
#pragma unroll
  for (int v = 0; v < NI; ++v) {
    const long long G = (long long)t_lo * 64 + (v * 4 + wave) * RPC + rsub;
    const long long tq = G / inner;
    sp[v] = (int)(G - tq * inner);
    sb[v] = (int)(tq / g.Tdst);
    sq[v] = (int)(tq - (long long)sb[v] * g.Tdst);
  }

  f32x4 acc[REP][REP];
#pragma unroll
  for (int m = 0; m < REP; ++m)
#pragma unroll
    for (int n = 0; n < REP; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 accb[REP];
#pragma unroll
  for (int m = 0; m < REP; ++m) accb[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool do_bias = g.db != nullptr && ct == 0 && tap == 0 && wc == 0;
  const __bf16 one = (__bf16)1.0f;
  const bf16x8 ones = {one, one, one, one, one, one, one, one};

  auto issue = [&](int buf) {
    unsigned char* Ab = cc_lds + buf * STAGE;
    unsigned char* Bb = Ab + IMG;
#pragma unroll
    for (int v = 0; v < NI; ++v) {
      const bool rok = sb[v] < g.B;
      const long long rowy = ((long long)sb[v] * g.Tdst + sq[v]) * inner + sp[v];
      const unsigned char* sa = (rok && a_cok) ? dy_base + rowy * g.Ntot * 2 : zsrc;
      cc_glds16(sa, Ab + (v * 4 + wave) * 1024);
      const int tu = sq[v] * g.stride + shift;
      const bool tok_ok = (unsigned)tu < lim;
      const int tok = (up > 1) ? (int)__umulhi((unsigned)tu, P.up_magic) : tu;
      const long long rowx = ((long long)sb[v] * g.Tsrc + tok) * inner + sp[v];
      const unsigned char* sx = (rok && b_cok && tok_ok) ? x_base + rowx * g.Cin_tot * 2 : zsrc;
      cc_glds16(sx, Bb + (v * 4 + wave) * 1024);
      // next step: 64 rows further
      sp[v] += dp64;
      sq[v] += dq64;
      if (sp[v] >= inner) {
        sp[v] -= inner;
        ++sq[v];
      }
      while (sq[v] >= g.Tdst) {
        sq[v] -= g.Tdst;
        ++sb[v];
      }
    }
  };
  auto compute = [&](int buf) {
    const unsigned char* Ab = cc_lds + buf * STAGE;
    const unsigned char* Bb = Ab + IMG;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[REP], bf[REP];
      // transpose reads: lane (li, kg) reads 4 consecutive channels of token kk*32 + kg*4 + (li >> 2) (and + 16)
      const int row = kk * 32 + kg * 4 + (li >> 2);
      const int rs = (TW == 128) ? ((row & 7) << 1) : (((row >> 1) & 3) << 1);  // same for row + 16
      const int sub = ((li & 3) * 4 & 7) * 2;                                    // byte offset inside the 16-byte chunk
#pragma unroll
      for (int m = 0; m < REP; ++m) {
        const int col = wr * (REP * 16) + m * 16 + (li & 3) * 4;
        const unsigned char* p = Ab + row * ROWB + (((col >> 3) ^ rs) << 4) + sub;
        const bf16x4 lo = cc_tr4(p), hi = cc_tr4(p + 16 * ROWB);
        af[m] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int n = 0; n < REP; ++n) {
        const int col = wc * (REP * 16) + n * 16 + (li & 3) * 4;
        const unsigned char* p = Bb + row * ROWB + (((col >> 3) ^ rs) << 4) + sub;
        const bf16x4 lo = cc_tr4(p), hi = cc_tr4(p + 16 * ROWB);
        bf[n] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int m = 0; m < REP; ++m)
#pragma unroll
        for (int n = 0; n < REP; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bf[n], acc[m][n], 0, 0, 0);
      if (do_bias) {
#pragma unroll
        for (int m = 0; m < REP; ++m) accb[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], ones, accb[m], 0, 0, 0);
      }
    }
  };

  const int nsteps = t_hi - t_lo;
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s) issue(s);
  for (int i = 0; i < nsteps; ++i) {
    cc_wait_vm<(NSTAGE - 2) * LPS>();
    cc_barrier();
    int nb = i + NSTAGE - 1;
    nb -= (nb / NSTAGE) * NSTAGE;
    issue(nb);  // (past the slice's last step these rows belong to the next slice: copied, never multiplied)
    int cb = i - (i / NSTAGE) * NSTAGE;
    compute(cb);
  }
  cc_wait_vm<0>();
  cc_barrier();

  // ---- epilogue through LDS: TW / 4 threads per output row, float4 read-modify-write (one slice) or atomics
  float* Cs = reinterpret_cast<float*>(cc_lds);
#pragma unroll
  for (int m = 0; m < REP; ++m)
#pragma unroll
    for (int n = 0; n < REP; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Cs[(wr * (REP * 16) + m * 16 + kg * 4 + r) * CLD + wc * (REP * 16) + n * 16 + li] = acc[m][n][r];
  if (do_bias && li == 0) {
#pragma unroll
    for (int m = 0; m < REP; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wr * (REP * 16) + m * 16 + kg * 4 + r;
        const float v = accb[m][r];
        if (n < n_end && v != 0.f) atomicAdd(&g.db[n], v);
      }
  }
  __syncthreads();
  constexpr int TPR = TW / 4;
  constexpr int RPP = CC_THREADS / TPR;
  const int jc = (tid % TPR) * 4;
  if (c0 + jc >= g.CR) return;  // CR % 8 == 0: whole float4 groups are live or dead
  float* dwt = (P.to_ws ? g.workspace + (long long)slice * g.K * g.Ntot * g.CR : g.dw) + (long long)tap * g.Ntot * g.CR;
#pragma unroll 2
  for (int pass = 0; pass < TW / RPP; ++pass) {
    const int rl = tid / TPR + pass * RPP;
    const int n = n0 + rl;
    if (n >= n_end) continue;
    float* dst = dwt + (long long)n * g.CR + c0 + jc;
    const float4 c = *reinterpret_cast<const float4*>(&Cs[rl * CLD + jc]);
    if (P.to_ws) {
      *reinterpret_cast<float4*>(dst) = c;
    } else if (P.slices == 1) {
      float4 o = *reinterpret_cast<float4*>(dst);
      o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w;
      *reinterpret_cast<float4*>(dst) = o;
    } else {
      if (c.x != 0.f) atomicAdd(dst + 0, c.x);
      if (c.y != 0.f) atomicAdd(dst + 1, c.y);
      if (c.z != 0.f) atomicAdd(dst + 2, c.z);
      if (c.w != 0.f) atomicAdd(dst + 3, c.w);
    }
  }
}

// dw[e] += sum_s ws[s][e]: the token slices of a weight gradient, summed in a fixed order (no atomics: a launch is
// bit-reproducible, and 50-120 slices adding into the same 50 k addresses at the same moment ran at ~60 G atomics/s).
// A block owns 16 float4 columns; its 16 thread rows each sum every 16th slice, then meet in LDS -- a narrow gradient
// (a 3-tap 32 x 32 layer is 768 float4) with 512 slices would otherwise be 768 threads doing 512 dependent-latency loads.
__global__ __launch_bounds__(256) void cconv_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                                long long n4, int slices) {
  __shared__ float4 part[16][17];
  const int col = threadIdx.x & 15, sg = threadIdx.x >> 4;
  const long long i = (long long)blockIdx.x * 16 + col;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    for (int s = sg; s < slices; s += 16) {
      const float4 v = reinterpret_cast<const float4*>(ws)[(long long)s * n4 + i];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  part[sg][col] = a;
  __syncthreads();
  if (sg == 0 && i < n4) {
    float4 t = reinterpret_cast<const float4*>(dw)[i];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float4 v = part[q][col];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    reinterpret_cast<float4*>(dw)[i] = t;
  }
}

template <int TW, int NSTAGE>
static int cw_launch2(const CwArgs& P, hipStream_t st) {
  constexpr int STAGE = 2 * 64 * TW * 2;
  constexpr int EPI_BYTES = TW * (TW + 4) * 4;
  constexpr size_t LDS = (NSTAGE * STAGE > EPI_BYTES) ? NSTAGE * STAGE : EPI_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cconv_wgrad_kernel<TW, NSTAGE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const kantts_cconvw_args& g = P.a;
  dim3 grid((g.CR + TW - 1) / TW, g.groups * ((g.NG + TW - 1) / TW), g.K * P.slices);
  if (P.xcd_map) grid = dim3(8u * grid.x * grid.y * (unsigned)g.K * (unsigned)kantts_cdiv(P.slices, 8), 1, 1);
  hipLaunchKernelGGL((cconv_wgrad_kernel<TW, NSTAGE>), grid, dim3(CC_THREADS), LDS, st, P);
  if (P.to_ws) {
    const long long n4 = (long long)g.K * g.Ntot * g.CR / 4;
    hipLaunchKernelGGL(cconv_wgrad_reduce_kernel, dim3((unsigned)kantts_cdiv(n4, 16)), dim3(256), 0, st, g.workspace, g.dw, n4,
                       P.slices);
  }
  KANTTS_CHECK_LAUNCH();
}

// ================================================================================================ narrow weight gradient
// Channel groups of at most 64 x 64 (the generator's 64- and 32-channel residual stacks, the scale discriminators' grouped
// k = 41 layers): a (tap, 128 x 128) tile wastes 3/4 .. 15/16 of its MFMAs and re-reads dy and x once per tap.  Here a
// workgroup keeps the accumulators of ALL taps of its channel group in registers (taps dealt to the four waves), and
// per step of 64 output tokens loads the 64 dy rows and the x WINDOW those tokens can touch
//   (64 - 1) * stride + (K - 1) * dil + 1 rows
// once: tap k of output token j multiplies window row j * stride + k * dil.  dy and x cross HBM -> LDS once per launch
// instead of once per tap.  Both operands are [token][channel] images read through ds_read_tr16_b64 as above.
// Steps never straddle two batch items (the window of a step is one contiguous token range of one item).
struct CtArgs {
  kantts_cconvw_args a;
  int slices, to_ws;
  int rs;    // 64-token sub-steps per step (1, 2 or 4): one barrier and one load round trip per 64 * rs tokens
  int spi;   // steps per batch item = ceil(Tdst / (64 * rs))
  int wr;    // window rows
  int ncp;   // window copies per wave and step
};

template <int NF, int CF, int TPW>
__global__ __launch_bounds__(CC_THREADS) void cconv_wgrad_taps_kernel(const CtArgs P) {
  constexpr int NT = NF * 16, CT = CF * 16;       // channel tiles of dy / x (32 or 64)
  constexpr int DYB = NT * 2, XB = CT * 2;        // row bytes
  constexpr int DY_CP = 64 * DYB / 4096;          // dy copies per wave and 64 tokens (1 or 2)
  constexpr int X_RPC = 1024 / XB;                // window rows per 1 KB copy (16 or 8)
  extern __shared__ __attribute__((aligned(16))) unsigned char cc_lds[];
  const kantts_cconvw_args& g = P.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kg = lane >> 4;
  const int RS = P.rs;
  const int DY_IMG = RS * 64 * DYB;
  const int X_IMG = P.ncp * 4 * 1024;
  const int STAGE = DY_IMG + X_IMG;

  const int tgroup = blockIdx.x;                 // taps tgroup*4*TPW .. + 4*TPW
  const int grp = blockIdx.y;
  const int slice = blockIdx.z;
  const int NS = g.B * P.spi;
  const int s_lo = (int)((long long)NS * slice / P.slices), s_hi = (int)((long long)NS * (slice + 1) / P.slices);
  const int n0 = grp * g.NG, c0g = grp * g.CR;
  const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(cc_zero16);

  // ---- copy coordinates.  dy image: copy v of wave w = rows (v*4 + w) * (1024 / DYB) ..; 64-byte rows swizzle the
  // 16-byte chunk by ((row >> 2) & 1) << 1, 128-byte rows by ((row >> 1) & 3) << 1 (conflict-free transpose reads)
  constexpr int DY_RPC = 1024 / DYB;
  const int dy_rsub = lane / (DYB / 16), dy_slot = lane % (DYB / 16);
  const int x_rsub = lane / (XB / 16), x_slot = lane % (XB / 16);
  auto swz = [](int row, int rowbytes) { return rowbytes == 64 ? (((row >> 2) & 1) << 1) : (((row >> 1) & 3) << 1); };

  f32x4 acc[TPW][NF][CF];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
      for (int c = 0; c < CF; ++c) acc[t][n][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 accb[NF];
#pragma unroll
  for (int n = 0; n < NF; ++n) accb[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool do_bias = g.db != nullptr && tgroup == 0 && wave == 0;
  const __bf16 one = (__bf16)1.0f;
  const bf16x8 ones = {one, one, one, one, one, one, one, one};

  auto issue = [&](int step, int buf) {
    unsigned char* Ab = cc_lds + buf * STAGE;
    unsigned char* Xb = Ab + DY_IMG;
    const bool live = step < s_hi;
    const int b = step / P.spi;
    const int q0 = (step - b * P.spi) * 64 * RS;
    for (int v = 0; v < DY_CP * RS; ++v) {
      const int row = (v * 4 + wave) * DY_RPC + dy_rsub;
      const int chunk = dy_slot ^ swz(row, DYB);
      const int q = q0 + row;
      const bool ok = live && q < g.Tdst && chunk * 8 < g.NG;
      const unsigned char* src = ok ? reinterpret_cast<const unsigned char*>(g.dy) +
                                          (((long long)b * g.Tdst + q) * g.Ntot + n0 + chunk * 8) * 2
                                    : zsrc;
      cc_glds16(src, Ab + (v * 4 + wave) * 1024);
    }
    const int t0 = q0 * g.stride - g.pad;
    for (int v = 0; v < P.ncp; ++v) {
      const int row = (v * 4 + wave) * X_RPC + x_rsub;
      const int chunk = x_slot ^ swz(row, XB);
      const int t = t0 + row;
      const bool ok = live && row < P.wr && (unsigned)t < (unsigned)g.Tsrc && chunk * 8 < g.CR;
      const unsigned char* src = ok ? reinterpret_cast<const unsigned char*>(g.x) +
                                          (((long long)b * g.Tsrc + t) * g.Cin_tot + c0g + chunk * 8) * 2
                                    : zsrc;
      cc_glds16(src, Xb + (v * 4 + wave) * 1024);
    }
  };
  auto compute = [&](int buf) {
    const unsigned char* Ab = cc_lds + buf * STAGE;
    const unsigned char* Xb = Ab + DY_IMG;
    for (int kk = 0; kk < 2 * RS; ++kk) {
      const int j = kk * 32 + kg * 4 + (li >> 2);  // output token of this lane's transpose read (and j + 16)
      const int sub = ((li & 3) * 4 & 7) * 2;
      bf16x8 af[NF];
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        const int col = n * 16 + (li & 3) * 4;
        const unsigned char* p = Ab + j * DYB + (((col >> 3) ^ swz(j, DYB)) << 4) + sub;  // swz(j) == swz(j + 16)
        const bf16x4 lo = cc_tr4(p), hi = cc_tr4(p + 16 * DYB);
        af[n] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
      if (do_bias) {
#pragma unroll
        for (int n = 0; n < NF; ++n) accb[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[n], ones, accb[n], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int k = (tgroup * TPW + t) * 4 + wave;  // taps interleaved over the waves
        if (k < g.K) {
          const int r_lo = j * g.stride + k * g.dil, r_hi = r_lo + 16 * g.stride;
          bf16x8 bf[CF];
#pragma unroll
          for (int c = 0; c < CF; ++c) {
            const int col = c * 16 + (li & 3) * 4;
            const bf16x4 lo = cc_tr4(Xb + r_lo * XB + (((col >> 3) ^ swz(r_lo, XB)) << 4) + sub);
            const bf16x4 hi = cc_tr4(Xb + r_hi * XB + (((col >> 3) ^ swz(r_hi, XB)) << 4) + sub);
            bf[c] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          }
#pragma unroll
          for (int n = 0; n < NF; ++n)
#pragma unroll
            for (int c = 0; c < CF; ++c)
              acc[t][n][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[n], bf[c], acc[t][n][c], 0, 0, 0);
        }
      }
    }
  };

  issue(s_lo, 0);
  for (int s = s_lo; s < s_hi; ++s) {
    cc_wait_vm<0>();
    cc_barrier();
    issue(s + 1, (s - s_lo + 1) & 1);
    compute((s - s_lo) & 1);
  }
  cc_wait_vm<0>();

  // ---- epilogue straight from the accumulators: lane (li, kg) holds dy channels kg*4 .. +3 x x channel li of a fragment
  float* base = P.to_ws ? g.workspace + (long long)slice * g.K * g.Ntot * g.CR : g.dw;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int k = (tgroup * TPW + t) * 4 + wave;
    if (k >= g.K) continue;
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
      for (int c = 0; c < CF; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int nn = n * 16 + kg * 4 + r, cc = c * 16 + li;
          if (nn < g.NG && cc < g.CR) {
            float* dst = base + ((long long)k * g.Ntot + n0 + nn) * g.CR + cc;
            const float v = acc[t][n][c][r];
            if (P.to_ws)
              *dst = v;
            else if (P.slices == 1)
              *dst += v;
            else if (v != 0.f)
              atomicAdd(dst, v);
          }
        }
  }
  if (do_bias && li == 0) {
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nn = n * 16 + kg * 4 + r;
        const float v = accb[n][r];
        if (nn < g.NG && v != 0.f) atomicAdd(&g.db[n0 + nn], v);
      }
  }
}

template <int NF, int CF, int TPW>
static int ct_launch(const CtArgs& P, hipStream_t st) {
  const kantts_cconvw_args& g = P.a;
  const size_t LDS = 2 * (size_t)(P.rs * 64 * NF * 32 + P.ncp * 4096);
  if (LDS > 150 * 1024) return KANTTS_E_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cconv_wgrad_taps_kernel<NF, CF, TPW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  dim3 grid(kantts_cdiv(g.K, 4 * TPW), g.groups, P.slices);
  hipLaunchKernelGGL((cconv_wgrad_taps_kernel<NF, CF, TPW>), grid, dim3(CC_THREADS), LDS, st, P);
  if (P.to_ws) {
    const long long n4 = (long long)g.K * g.Ntot * g.CR / 4;
    hipLaunchKernelGGL(cconv_wgrad_reduce_kernel, dim3((unsigned)kantts_cdiv(n4, 16)), dim3(256), 0, st, g.workspace, g.dw, n4,
                       P.slices);
  }
  KANTTS_CHECK_LAUNCH();
}

// 64 x 64 groups need two taps per wave: beyond one tap group (8 taps) the per-tap tiles above are as good
// (scripts/bench_native/cconv_test: 16 groups of 64 x 64, k = 41: 69 vs 127 us)
static bool ct_applies(const kantts_cconvw_args& g) {
  static const char* off = getenv("KANTTS_NO_WGRAD_TAPS");
  if (off || g.inner != 1 || g.up > 1 || g.CR > 64 || g.NG > 64 || g.K > 64) return false;
  return !(g.CR > 32 && g.NG > 32 && g.K > 8);
}
// 64-token sub-steps per step: as many (4, 2, 1) as the sequence length and ~36 KB of LDS per stage allow
static int ct_rs(const kantts_cconvw_args& g) {
  static const char* e = getenv("KANTTS_WGRAD_TAPS_RS");
  const int xb = (g.CR > 32) ? 128 : 64, dyb = (g.NG > 32) ? 128 : 64;
  for (int rs = e ? atoi(e) : 4; rs > 1; rs >>= 1) {
    const long long wr = (long long)(64 * rs - 1) * g.stride + (long long)(g.K - 1) * g.dil + 1;
    if (g.Tdst >= 64 * rs && wr * xb + 64ll * rs * dyb <= 36 * 1024) return rs;
  }
  return 1;
}
// taps per wave: "deep" keeps up to 44 / 20 / 12 taps per workgroup in ~400 registers (one wave per SIMD), the default
// 24 / 12 / 8 taps in < 256 (two waves per SIMD, more workgroups re-reading the step's rows)
static bool ct_deep() {
  static const char* e = getenv("KANTTS_WGRAD_TAPS_DEEP");
  return e && atoi(e) != 0;
}
static int ct_tpw(const kantts_cconvw_args& g) {
  const int nf = g.NG > 32 ? 4 : 2, cf = g.CR > 32 ? 4 : 2;
  if (ct_deep()) return nf * cf == 4 ? 11 : (nf * cf == 8 ? 5 : 3);
  return nf * cf == 4 ? 6 : (nf * cf == 8 ? 3 : 2);
}
static int ct_slices(const kantts_cconvw_args& g, bool have_ws) {
  const long long NS = (long long)g.B * kantts_cdiv(g.Tdst, 64 * ct_rs(g));
  long long slices = g.slices;
  if (slices == 0) {
    const long long wgs = (long long)kantts_cdiv(g.K, 4 * ct_tpw(g)) * g.groups;
    slices = (512 + wgs - 1) / wgs;
    if (!have_ws) {
      const long long cap = (6ll << 20) / ((long long)g.K * g.Ntot * g.CR) + 1;
      if (slices > cap) slices = cap;
    }
    if (slices > NS / 4) slices = NS / 4;
    if (slices < 1) slices = 1;
  }
  if (slices > NS) slices = NS;
  if (slices > 65535) slices = 65535;
  return (int)slices;
}

// token slices of a launch: explicit, or enough workgroups for two per CU; with a workspace the partial tiles are
// stored and summed by a second kernel, without one they meet in fp32 atomics (capped: contended atomics are slow)
static int cw_slices(const kantts_cconvw_args& g, bool have_ws) {
  const int TW = (g.CR > 64 || g.NG > 64) ? 128 : 64;
  const long long tiles = (long long)kantts_cdiv(g.CR, TW) * g.groups * kantts_cdiv(g.NG, TW) * g.K;
  const long long NT = kantts_cdiv((long long)g.B * g.Tdst * g.inner, 64);
  long long slices = g.slices;
  if (slices == 0) {
    slices = 1;
    if (tiles < 200) {
      // [round 6] the launch is dealt slice by slice to the 8 XCDs (cconv_wgrad_kernel), each with 64 places (32 CUs x 2
      // workgroups): as many slices per XCD as fit in ONE round of its places.  (The rule of rounds 4-5, ceil(512 / tiles)
      // slices, put 6 x 11 = 66 workgroups on an XCD for the 128 -> 128, k = 11 layers -- a second round for two of them:
      // 91 us against 65 us at 36 slices; 128 -> 128 k = 7: 61 -> 44 us, 256 -> 256 k = 11: 54 -> 45 us;
      // profiles/r06_runSL_cconv_wgrad_slices.log.)  More than 64 tiles per slice: fewer than 8 slices, the 3-D grid.
      slices = tiles <= 64 ? 8 * (64 / tiles) : 512 / tiles;
      if (!have_ws) {
        const long long cap = (6ll << 20) / ((long long)g.K * g.Ntot * g.CR) + 1;  // <= ~6 M atomics per launch
        if (slices > cap) slices = cap;
      }
      if (slices > NT / 8) slices = NT / 8;  // at least 8 steps of 64 tokens per workgroup
      if (slices < 1) slices = 1;
    }
  }
  if (slices > NT) slices = NT;
  return (int)slices;
}

extern "C" long long kantts_cconv_wgrad_ws_floats(const kantts_cconvw_args* ap) {
  if (!ap || ap->K < 1 || ap->groups < 1 || ap->NG < 1 || ap->CR < 1 || ap->inner < 1 || ap->B < 1 || ap->Tdst < 1) return 0;
  const int slices = ct_applies(*ap) ? ct_slices(*ap, true) : cw_slices(*ap, true);
  return slices > 1 ? (long long)slices * ap->K * ap->Ntot * ap->CR : 0;
}

extern "C" int kantts_cconv_wgrad_launch(const kantts_cconvw_args* ap, void* stream) {
  if (!ap || !ap->x || !ap->dy || !ap->dw) return KANTTS_E_BADARG;
  const kantts_cconvw_args& g = *ap;
  if (g.B < 0 || g.Tsrc < 0 || g.Tdst < 0 || g.K < 1 || g.groups < 1 || g.NG < 1 || g.CR < 1 || g.stride < 1 ||
      g.inner < 1 || g.up < 0 || g.slices < 0)
    return KANTTS_E_BADARG;
  if (g.Ntot != g.groups * g.NG || g.Cin_tot != g.groups * g.CR) return KANTTS_E_BADARG;
  if (g.B == 0 || g.Tdst == 0) return KANTTS_OK;
  if ((g.CR & 7) || (g.NG & 7) || !cc_aligned16(g.x) || !cc_aligned16(g.dy) || !cc_aligned16(g.dw)) return KANTTS_E_UNSUPPORTED;
  const int up = g.up > 1 ? g.up : 1;
  if (up > 64 || (long long)g.Tsrc * up * up >= (1ll << 31) || (long long)g.B * g.Tsrc * g.inner >= (1ll << 30) ||
      (long long)g.B * g.Tdst * g.inner >= (1ll << 30) || g.inner > 64)
    return KANTTS_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (ct_applies(g)) {
    CtArgs T = {};
    T.a = g;
    int slices = ct_slices(g, g.workspace != nullptr);
    bool ws = false;
    if (slices > 1 && g.workspace) {
      if (g.ws_floats >= (long long)slices * g.K * g.Ntot * g.CR && cc_aligned16(g.workspace))
        ws = true;
      else
        slices = ct_slices(g, false);
    }
    T.slices = slices;
    T.to_ws = ws ? 1 : 0;
    T.rs = ct_rs(g);
    T.spi = kantts_cdiv(g.Tdst, 64 * T.rs);
    T.wr = (64 * T.rs - 1) * g.stride + (g.K - 1) * g.dil + 1;
    const int xrpc = (g.CR > 32) ? 8 : 16;
    T.ncp = kantts_cdiv(kantts_cdiv(T.wr, xrpc), 4);
    const int nf = g.NG > 32 ? 4 : 2, cf = g.CR > 32 ? 4 : 2;
    int rc;
    if (ct_deep()) {
      if (nf == 2 && cf == 2)
        rc = ct_launch<2, 2, 11>(T, st);
      else if (nf == 2 && cf == 4)
        rc = ct_launch<2, 4, 5>(T, st);
      else if (nf == 4 && cf == 2)
        rc = ct_launch<4, 2, 5>(T, st);
      else
        rc = ct_launch<4, 4, 3>(T, st);
    } else {
      if (nf == 2 && cf == 2)
        rc = ct_launch<2, 2, 6>(T, st);
      else if (nf == 2 && cf == 4)
        rc = ct_launch<2, 4, 3>(T, st);
      else if (nf == 4 && cf == 2)
        rc = ct_launch<4, 2, 3>(T, st);
      else
        rc = ct_launch<4, 4, 2>(T, st);
    }
    if (rc != KANTTS_E_UNSUPPORTED) return rc;
  }
  CwArgs P = {};
  P.a = g;
  P.up_magic = (unsigned)((1ull << 32) / (unsigned)up) + 1u;
  const int TW = (g.CR > 64 || g.NG > 64) ? 128 : 64;
  int slices = cw_slices(g, g.workspace != nullptr);
  bool to_ws = false;
  if (slices > 1 && g.workspace) {
    if (g.ws_floats >= (long long)slices * g.K * g.Ntot * g.CR && cc_aligned16(g.workspace))
      to_ws = true;
    else
      slices = cw_slices(g, false);
  }
  if ((long long)g.K * slices > 65535) return KANTTS_E_UNSUPPORTED;
  P.to_ws = to_ws ? 1 : 0;
  P.slices = slices;
  static const bool no_xcd = getenv("KANTTS_CCONV_WGRAD_NO_XCD_MAP") != nullptr;  // A/B switch: the 3-D grid of rounds 4-5
  // a slice per XCD and round: fewer slices than XCDs, or a last round that leaves more than a quarter of them idle -> 3-D grid
  P.xcd_map = (slices >= 8 && slices * 4 >= 3 * 8 * kantts_cdiv(slices, 8) && !no_xcd) ? 1 : 0;
  static const char* env_stage = getenv("KANTTS_CCONV_STAGES");
  const int nst = env_stage ? atoi(env_stage) : 0;
  if (TW == 128) {
    if (nst == 3) return cw_launch2<128, 3>(P, st);
    return cw_launch2<128, 2>(P, st);
  }
  if (nst == 3) return cw_launch2<64, 3>(P, st);
  return cw_launch2<64, 2>(P, st);
}

// ================================================================================================ bf16 operand images
template <bool GATE_BF16>
__global__ __launch_bounds__(256) void act_cast_kernel(const float* __restrict__ src, const void* __restrict__ gate,
                                                      __bf16* __restrict__ dst, int act, float slope, long long n8) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const float4 a = reinterpret_cast<const float4*>(src)[2 * i], b = reinterpret_cast<const float4*>(src)[2 * i + 1];
  float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  if (gate) {
    float q[8];
    if (GATE_BF16) {
      const u32x4 t = reinterpret_cast<const u32x4*>(gate)[i];
      q[0] = cc_lo(t.x); q[1] = cc_hi(t.x); q[2] = cc_lo(t.y); q[3] = cc_hi(t.y);
      q[4] = cc_lo(t.z); q[5] = cc_hi(t.z); q[6] = cc_lo(t.w); q[7] = cc_hi(t.w);
    } else {
      const float4 t0 = reinterpret_cast<const float4*>(gate)[2 * i], t1 = reinterpret_cast<const float4*>(gate)[2 * i + 1];
      q[0] = t0.x; q[1] = t0.y; q[2] = t0.z; q[3] = t0.w; q[4] = t1.x; q[5] = t1.y; q[6] = t1.z; q[7] = t1.w;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= (q[e] > 0.f) ? 1.f : slope;
  } else if (act) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
  }
  u32x4 w = {cc_pack2(v[0], v[1]), cc_pack2(v[2], v[3]), cc_pack2(v[4], v[5]), cc_pack2(v[6], v[7])};
  reinterpret_cast<u32x4*>(dst)[i] = w;
}

extern "C" int kantts_act_cast_bf16(const float* src, const void* gate, int gate_bf16, void* dst, int act, float slope,
                                    long long n, void* stream) {
  if (!src || !dst || n < 0 || (n & 7) || !cc_aligned16(src) || !cc_aligned16(dst) || (gate && !cc_aligned16(gate)))
    return KANTTS_E_BADARG;
  if (n == 0) return KANTTS_OK;
  const long long n8 = n / 8;
  dim3 grid((unsigned)kantts_cdiv(n8, 256)), block(256);
  if (gate_bf16)
    hipLaunchKernelGGL((act_cast_kernel<true>), grid, block, 0, (hipStream_t)stream, src, gate, reinterpret_cast<__bf16*>(dst), act,
                       slope, n8);
  else
    hipLaunchKernelGGL((act_cast_kernel<false>), grid, block, 0, (hipStream_t)stream, src, gate, reinterpret_cast<__bf16*>(dst),
                       act, slope, n8);
  KANTTS_CHECK_LAUNCH();
}
