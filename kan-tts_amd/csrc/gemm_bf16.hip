// Contractions of the SAM-BERT step with bf16 operands in HBM (round 2).
//
// gemm_fast.hip keeps activations and weights fp32 in HBM and rounds them to bf16 while staging: its PMC profile
// (profiles/r01_gemm_ablation.log) puts 11-15 us of a 19-23 us launch into that conversion / predication skeleton.
// Here the operands a contraction reads ARE bf16 in memory (weights: the arena's bf16 shadow, refreshed once per
// step; activations: written bf16 by the producing kernel), so staging is a 16-byte copy; fp32 operands (the residual
// stream, incoming gradients of fp32 tensors) are still accepted and rounded once per tile.
//
//   bgemm_nt_kernel : C[M,N] = epi( sum_seg A_s[M,K_s] . B_s^T )      forward and input gradients
//       B_s is (N, K_s) k-contiguous (forward: the weight itself) or, with b_kn, (K_s, N) n-contiguous (input gradient
//       with the SAME weight buffer: the MFMA B fragments are formed by LDS transpose reads, no transposed copy).
//       Conv taps are segments whose A rows are shifted tokens.  Epilogue: bias, alpha, ReLU, dropout, fp32 residual,
//       gate by a saved activation (ReLU backward), row zeroing; bf16 or fp32 output, 16-byte stores.
//   bgemm_tn_kernel : dW[N,K] += A[M,N]^T . B[M,K]   (+ bias gradient)  weight gradients, tokens are the reduction
//       axis of both operands: natural [token][channel] LDS images, both fragments by transpose reads.
//
// Tiles: 256 threads = 4 waves.  NT: BM x 128 outputs (BM = 64: waves 2x2, 32x64 each; BM = 32: waves 1x4, 32x32
// each), reduction tile 64, register-prefetched double-buffered LDS, one barrier per tile.  k-contiguous LDS images are
// [row][64] bf16 with the 16-byte chunk index XORed by (row >> 1) & 7 (conflict-free 16- and 8-byte fragment reads);
// n-contiguous images are [k][128 + 16] (8 consecutive k rows start 8 banks apart, as in gemm_fast.hip).
// TN: 128 x 128 outputs, waves 2x2 with 64x64 each, token tile 32, atomics into the fp32 gradient.
#include <stdlib.h>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define BG_THREADS 256
#define BG_BN 128
#define BG_BK 64
#define BG_LDKN (BG_BN + 16)  // pitch (elements) of an n-contiguous image

typedef __attribute__((address_space(3))) bf16x4 bg_lds_bf16x4;
__device__ __forceinline__ bf16x4 bg_tr4(const __bf16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bg_lds_bf16x4*)(p));
}

__device__ __forceinline__ unsigned bg_pack2(float a, float b) {
  bf16x4 t = {(__bf16)a, (__bf16)b, (__bf16)0.f, (__bf16)0.f};
  return ((u32x2&)t).x;
}
__device__ __forceinline__ float bg_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bg_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// 8 consecutive elements of a row as bf16: either a 16-byte copy or two float4 loads rounded once.  `drop` regenerates
// an epilogue dropout of the forward pass on the incoming gradient (element index = logical offset in the tensor).
template <bool F32>
__device__ __forceinline__ u32x4 bg_load8(const void* base, long long elem, bool ok, float drop_p, uint64_t seed,
                                          uint64_t logical) {
  // [round 5] The load is UNCONDITIONAL (a predicated-off chunk reads element 0 of the tensor and is zeroed afterwards): with
  // `if (!ok) return` in front of it every chunk was a branch around its loads and `s_waitcnt vmcnt(0)` where the branch
  // re-joins -- the staging of a tile was a chain of 4-6 memory round trips instead of one (the weight-gradient kernel
  // issued its 48 loads two at a time; found by scanning the ISA of every kernel for load / wait bursts).
  u32x4 r = {0u, 0u, 0u, 0u};
  const long long e_ = ok ? elem : 0;
  if (F32) {
    const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + e_);
    float4 a = p[0], b = p[1];
    if (!ok) a = b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (drop_p > 0.f) {  // logical % 8 == 0 (8-element granularity of every extent)
      float lo4[4] = {a.x, a.y, a.z, a.w}, hi4[4] = {b.x, b.y, b.z, b.w};
      kantts_dropout_scale4(drop_p, seed, logical, lo4);
      kantts_dropout_scale4(drop_p, seed, logical + 4, hi4);
      a = make_float4(lo4[0], lo4[1], lo4[2], lo4[3]);
      b = make_float4(hi4[0], hi4[1], hi4[2], hi4[3]);
    }
    r.x = bg_pack2(a.x, a.y);
    r.y = bg_pack2(a.z, a.w);
    r.z = bg_pack2(b.x, b.y);
    r.w = bg_pack2(b.z, b.w);
  } else {
    r = *reinterpret_cast<const u32x4*>(reinterpret_cast<const __bf16*>(base) + e_);
    if (!ok) r = (u32x4){0u, 0u, 0u, 0u};
  }
  return r;
}

// Staging in two halves (round 5): bg_fetch8 only ISSUES the loads of a chunk (unconditional, clamped address) and keeps
// the raw bits; bg_finish8 -- at commit time, one tile later -- zeroes a predicated-off chunk, applies the A-operand dropout
// and rounds fp32 to bf16.  With conversion right behind each load (bg_load8) the compiler produced `load pair; branch;
// s_waitcnt vmcnt(0)` per chunk: the staging of a tile was a chain of 4-6 memory round trips that the compute of the
// previous tile could not hide.
template <bool F32>
struct BgRaw {
  u32x4 v[F32 ? 2 : 1];
};
template <bool F32>
__device__ __forceinline__ BgRaw<F32> bg_fetch8(const void* base, long long elem, bool ok) {
  BgRaw<F32> r;
  const long long e_ = ok ? elem : 0;
  if (F32) {
    const u32x4* p = reinterpret_cast<const u32x4*>(reinterpret_cast<const float*>(base) + e_);
    r.v[0] = p[0];
    r.v[F32 ? 1 : 0] = p[1];
  } else {
    r.v[0] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const __bf16*>(base) + e_);
  }
  return r;
}
template <bool F32>
__device__ __forceinline__ u32x4 bg_finish8(const BgRaw<F32>& raw, bool ok, float drop_p, uint64_t seed, uint64_t logical) {
  u32x4 r = {0u, 0u, 0u, 0u};
  if (!ok) return r;
  if (F32) {
    const u32x4 a = raw.v[0], b = raw.v[F32 ? 1 : 0];
    float lo4[4] = {__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w)};
    float hi4[4] = {__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w)};
    if (drop_p > 0.f) {  // logical % 8 == 0 (8-element granularity of every extent)
      kantts_dropout_scale4(drop_p, seed, logical, lo4);
      kantts_dropout_scale4(drop_p, seed, logical + 4, hi4);
    }
    r.x = bg_pack2(lo4[0], lo4[1]);
    r.y = bg_pack2(lo4[2], lo4[3]);
    r.z = bg_pack2(hi4[0], hi4[1]);
    r.w = bg_pack2(hi4[2], hi4[3]);
  } else {
    r = raw.v[0];
  }
  return r;
}

__device__ __forceinline__ float bg_sum16(float v) {  // over the 16 lanes that share (tid >> 4)
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}

__device__ __forceinline__ void bg_xcd_remap(int& bx, int& by) {
  // workgroups of one XCD get consecutive tiles in row-major order (same rule as gemm_fast.hip): the column tiles
  // that share an A row tile stay behind one L2
  const int gx = gridDim.x, total = gx * gridDim.y;
  if (total >= 64) {
    const int L = by * gx + bx, k = L & 7, j = L >> 3;
    const int q = total >> 3, r = total & 7;
    const int vid = k * q + (k < r ? k : r) + j;
    by = vid / gx;
    bx = vid - by * gx;
  }
}

// ================================================================================================ NT
// LNB: the epilogue is the backward of a LayerNorm(128) whose output gradient is this contraction's result
// (kantts_bgemm_nt_lnbwd): dx / dgamma / dbeta leave instead of (or beside) the result itself.
template <int BM, bool A_F32, bool B_KN>
__global__ __launch_bounds__(BG_THREADS) void bgemm_nt_kernel(const kantts_bgemm_args g) {
  constexpr bool LNB = false;
  const kantts_lnbwd_args* const lnb = nullptr;
#include "gemm_bf16_nt_body.inc"
}
template <int BM, bool A_F32>
__global__ __launch_bounds__(BG_THREADS) void bgemm_nt_lnb_kernel(const kantts_bgemm_args g, const kantts_lnbwd_args lnb_args) {
  constexpr bool LNB = true, B_KN = true;
  const kantts_lnbwd_args* const lnb = &lnb_args;
#include "gemm_bf16_nt_body.inc"
}

template <int BM>
static void bg_launch_nt(const kantts_bgemm_args& g, dim3 grid, hipStream_t st) {
  const bool af = g.a_f32 != 0, kn = g.b_kn != 0;
  if (af) {
    if (kn)
      hipLaunchKernelGGL((bgemm_nt_kernel<BM, true, true>), grid, dim3(BG_THREADS), 0, st, g);
    else
      hipLaunchKernelGGL((bgemm_nt_kernel<BM, true, false>), grid, dim3(BG_THREADS), 0, st, g);
  } else {
    if (kn)
      hipLaunchKernelGGL((bgemm_nt_kernel<BM, false, true>), grid, dim3(BG_THREADS), 0, st, g);
    else
      hipLaunchKernelGGL((bgemm_nt_kernel<BM, false, false>), grid, dim3(BG_THREADS), 0, st, g);
  }
}

static bool bg_aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

extern "C" int kantts_bgemm_nt(const kantts_bgemm_args* gp, void* stream) {
  if (!gp) return KANTTS_E_BADARG;
  const kantts_bgemm_args& g = *gp;
  if (g.nseg < 1 || g.nseg > KANTTS_BGEMM_MAX_SEG || g.M < 0 || g.N < 0 || !g.c) return KANTTS_E_BADARG;
  if (g.M == 0 || g.N == 0) return KANTTS_OK;
  // 16-byte vector access everywhere: 8-element granularity of every extent / leading dimension
  if ((g.N & 7) || (g.ldc & 7) || !bg_aligned16(g.c)) return KANTTS_E_UNSUPPORTED;
  if (g.res && ((g.ldr & 3) || !bg_aligned16(g.res))) return KANTTS_E_UNSUPPORTED;
  if (g.gate && ((g.ldg & 7) || !bg_aligned16(g.gate))) return KANTTS_E_UNSUPPORTED;
  if ((g.bias && !bg_aligned16(g.bias)) || (g.bias2 && !bg_aligned16(g.bias2))) return KANTTS_E_UNSUPPORTED;
  if (g.ln_out) {
    if (!g.ln_gamma || !g.ln_beta || !g.ln_mean || !g.ln_rstd) return KANTTS_E_BADARG;
    if (g.N != 128 || g.c_bf16 || !bg_aligned16(g.ln_out) || !bg_aligned16(g.ln_gamma) || !bg_aligned16(g.ln_beta))
      return KANTTS_E_UNSUPPORTED;
  }
  for (int s = 0; s < g.nseg; ++s) {
    const kantts_bgemm_seg& sg = g.seg[s];
    if (!sg.a || !sg.b || sg.klen < 8 || (sg.klen & 7) || (sg.lda & 7) || (sg.ldb & 7)) return KANTTS_E_UNSUPPORTED;
    if (!bg_aligned16(sg.a) || !bg_aligned16(sg.b)) return KANTTS_E_UNSUPPORTED;
    if (sg.a_shift != 0 && g.T <= 0) return KANTTS_E_BADARG;
  }
  // 64-row tiles once they give every CU a workgroup, else 32-row tiles.  128-row tiles (one round of 408 workgroups
  // instead of 816 in two) were measured and lost: 2 waves per SIMD and a 64-register epilogue made the 6528 x 128 -> 1024
  // projection 17.0 us instead of 12.8 (profiles/r02_runH_bgemm_bm128.log); KANTTS_BGEMM_BM=128 still selects them.
  const long long nt = kantts_cdiv(g.N, BG_BN);
  const long long wg64 = kantts_cdiv(g.M, 64) * nt;
  static const char* force_bm = getenv("KANTTS_BGEMM_BM");
  // fp32 A operands (two float4 loads + a rounding pass per 8 elements) do better with 32-row tiles at every size measured:
  // M = 19584, 512 -> 256: 28.1 us against 34.9 (profiles/r03_runU_postnet_gemm_tiles.log)
  int bm = (wg64 >= 256 && !g.a_f32) ? 64 : 32;
  if (force_bm) bm = atoi(force_bm);
  hipStream_t st = (hipStream_t)stream;
  if (bm == 128)
    bg_launch_nt<128>(g, dim3(nt, kantts_cdiv(g.M, 128)), st);
  else if (bm == 64)
    bg_launch_nt<64>(g, dim3(nt, kantts_cdiv(g.M, 64)), st);
  else
    bg_launch_nt<32>(g, dim3(nt, kantts_cdiv(g.M, 32)), st);
  KANTTS_CHECK_LAUNCH();
}

extern "C" int kantts_bgemm_nt_lnbwd(const kantts_bgemm_args* gp, const kantts_lnbwd_args* lp, void* stream) {
  if (!gp || !lp) return KANTTS_E_BADARG;
  const kantts_bgemm_args& g = *gp;
  const kantts_lnbwd_args& l = *lp;
  if (g.nseg < 1 || g.nseg > KANTTS_BGEMM_MAX_SEG || g.M < 0) return KANTTS_E_BADARG;
  if (!l.x || !l.gamma || !l.mean || !l.rstd || !l.dx || (!l.part_rows && (!l.dgamma_accum || !l.dbeta_accum)))
    return KANTTS_E_BADARG;
  // what the model has: the input gradient of a projection (weights stored (out, in) = [k][n]: b_kn) of LayerNorm-ed rows
  if (g.N != 128 || !g.b_kn || g.ln_out || g.gate || g.relu || g.drop_p > 0.f) return KANTTS_E_UNSUPPORTED;
  if (g.M == 0) return KANTTS_OK;
  if (g.c && ((g.ldc & 7) || !bg_aligned16(g.c))) return KANTTS_E_UNSUPPORTED;
  if (g.res && ((g.ldr & 3) || !bg_aligned16(g.res))) return KANTTS_E_UNSUPPORTED;
  if ((g.bias && !bg_aligned16(g.bias)) || (g.bias2 && !bg_aligned16(g.bias2))) return KANTTS_E_UNSUPPORTED;
  if (!bg_aligned16(l.x) || !bg_aligned16(l.gamma) || !bg_aligned16(l.dx) || (l.dres && !bg_aligned16(l.dres)))
    return KANTTS_E_UNSUPPORTED;
  for (int s = 0; s < g.nseg; ++s) {
    const kantts_bgemm_seg& sg = g.seg[s];
    if (!sg.a || !sg.b || sg.klen < 8 || (sg.klen & 7) || (sg.lda & 7) || (sg.ldb & 7)) return KANTTS_E_UNSUPPORTED;
    if (!bg_aligned16(sg.a) || !bg_aligned16(sg.b)) return KANTTS_E_UNSUPPORTED;
    if (sg.a_shift != 0 && g.T <= 0) return KANTTS_E_BADARG;
  }
  // 32-row tiles: 2 x 256 same-address atomics per workgroup, and one round of workgroups at the decoder's 6528 rows
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(1, kantts_cdiv(g.M, 32));
  if (g.a_f32)
    hipLaunchKernelGGL((bgemm_nt_lnb_kernel<32, true>), grid, dim3(BG_THREADS), 0, st, g, l);
  else
    hipLaunchKernelGGL((bgemm_nt_lnb_kernel<32, false>), grid, dim3(BG_THREADS), 0, st, g, l);
  KANTTS_CHECK_LAUNCH();
}

// ================================================================================================ TN (weight gradients)
// dW[n][k] (+)= alpha * sum_m A[m][n] * B[m + shift][k];  db[n] += alpha * sum_m A[m][n]  (k-tile 0, tap 0 only).
// grid = (k tiles of 128, n tiles of 64, taps * slices) -- since round 6 as a 1-D grid in XCD-aware order (see the kernel);
// slice z walks token tiles z, z + slices, ...
// What bounds this kernel is the fp32 atomics of the split over tokens (~200 G atomics/s measured: the first version,
// 128x128 tiles x 64 slices = 8.4 M atomics, took 48 us where the round-1 kernel took 24) and the chain of dependent
// load latencies of a short token tile.  Hence: 64 x 128 output tiles (more tiles, fewer slices for the same number of
// workgroups), token tiles of 64 with the loads of two tiles in flight, and a slice count capped by the atomics budget.
#include <atomic>
// Launch-shape knobs for sweeps and tests (kantts_launch_tuning): explicit state instead of an environment read per launch.
std::atomic<int> kantts_tune_tn_tile{0}, kantts_tune_tn_slices{0}, kantts_tune_c1_wgrad_wgs{0};
extern "C" int kantts_launch_tuning(int tn_tile, int tn_slices, int c1_wgrad_wgs) {
  if (tn_slices < 0 || c1_wgrad_wgs < 0) return KANTTS_E_BADARG;
  kantts_tune_tn_tile.store(tn_tile, std::memory_order_relaxed);
  kantts_tune_tn_slices.store(tn_slices, std::memory_order_relaxed);
  kantts_tune_c1_wgrad_wgs.store(c1_wgrad_wgs, std::memory_order_relaxed);
  return KANTTS_OK;
}
#define TN_BT 64                 // tokens per tile (two MFMA k-steps)
// The grouped form (kantts_bgemm_tn_grouped) runs up to KANTTS_TN_MAX_GROUP problems of one shape in a single launch:
// weight gradients are leaves of the backward graph, so the host defers them and issues every layer's gradient of one
// shape together -- each problem then needs only a 1/n_problems share of the token split (n_problems times fewer
// atomics) and ~150 small launches per training step become ~15.
struct TnGroupArgs {
  kantts_bgemm_tn_args g;  // shape, dtypes, alpha, dropout probability of the whole group; a/b/c/db/a_drop_seed of problem 0
  int nprob;
  const void* a[KANTTS_TN_MAX_GROUP];
  const void* b[KANTTS_TN_MAX_GROUP];
  float* c[KANTTS_TN_MAX_GROUP];
  float* db[KANTTS_TN_MAX_GROUP];
  uint64_t a_drop_seed[KANTTS_TN_MAX_GROUP];
  // [round 6] XCD-aware launch (1-D grid): ktiles * ntiles output tiles of one (problem, tap, slice) = one GROUP; 0 = the
  // 3-D grid of rounds 2-5
  int ktiles, ntiles, ngroups;
};

// BN x BK = output tile (channels of A x channels of B).  64 x 128 is the round-2 tile; [round 4] 128 x 256 for the large
// problems: with 64 x 128 tiles every element of A is re-read K / 128 times and every element of B N / 64 times from L2, and
// the grouped launches of the decoder / postnet weight gradients ran AT the L2 bandwidth that implies (e.g. four
// 19 584 x 256 x 512 problems: 640 MB of tile traffic in 145 us; three 512 x 256 ones: 600 MB in 214 us -- kernel trace
// profiles/r04_runH_*).  The four-times larger tile halves both re-read factors.
template <bool A_F32, bool B_F32, int BN, int BK>
__global__ __launch_bounds__(BG_THREADS) void bgemm_tn_kernel(const TnGroupArgs ga) {
  constexpr int LDA = BN + 16, LDB = BK + 16;  // pitches of the [token][channel] images
  constexpr int IMG_A = TN_BT * LDA * 2, IMG_B = TN_BT * LDB * 2, STAGE = IMG_A + IMG_B;
  constexpr int NA = TN_BT * (BN / 8) / BG_THREADS, NB = TN_BT * (BK / 8) / BG_THREADS;  // 16-byte chunks per thread
  constexpr int CA = BN / 8, CB = BK / 8;                                                // chunks per token row
  constexpr int MR = BN / 32, NR = BK / 32;  // 16-wide fragments per wave and axis (waves 2 x 2)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  kantts_bgemm_tn_args g = ga.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;  // wave tile: BN / 2 (n) x BK / 2 (k)
  const int li = lane & 15, kg = lane >> 4;
  // Workgroup -> (k tile, n tile, group).  The output tiles of one group read the SAME token rows of A and B (ntiles x and
  // ktiles x over); the hardware deals consecutive workgroup ids round-robin over the 8 XCDs, each with its own L2, so on
  // the 3-D grid the tiles of a group landed on 8 different L2s and every re-read went back to HBM / the memory-side cache
  // (PMC: 2.3-2.8 x the algorithmic traffic, profiles/r05_runPMC_*).  Here a group's tiles are the consecutive workgroups of
  // ONE XCD (id % 8 = XCD, id / 8 = position inside it): they start together, walk the tokens in step, and every token
  // tile is fetched from HBM once and re-read from that XCD's L2.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (ga.ngroups > 0) {
    const int P = ga.ktiles * ga.ntiles, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    bz = xcd + 8 * (j / P);
    if (bz >= ga.ngroups) return;  // padding of the last round of groups (uniform over the workgroup)
    const int tile = j % P;
    bx = tile % ga.ktiles;
    by = tile / ga.ktiles;
  }
  const int n0 = by * BN, c0 = bx * BK;
  const int per_prob = g.ntaps * g.slices;
  const int prob = bz / per_prob, zr = bz % per_prob;
  const int tap = zr / g.slices, slice = zr % g.slices;
  g.a = ga.a[prob];
  g.b = ga.b[prob];
  g.c = ga.c[prob];
  g.db = ga.db[prob];
  g.a_drop_seed = ga.a_drop_seed[prob];
  const int shift = g.shift0 + tap * g.shift_step;
  const uint64_t seed_off = g.seed_dev ? *g.seed_dev : 0ull;

  f32x4 acc[MR][NR];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int n = 0; n < NR; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float colsum = 0.f;
  const bool do_bias = g.db && bx == 0 && tap == 0;

  BgRaw<A_F32> ra0[NA], ra1[NA];
  BgRaw<B_F32> rb0[NB], rb1[NB];
  const int ntile = (g.M + TN_BT - 1) / TN_BT;

  // fetch: loads only (see bg_fetch8); the tile index is clamped by the caller so that the loads are never inside a branch
  auto fetch = [&](int t, BgRaw<A_F32>* ra, BgRaw<B_F32>* rb) {
    const int m0 = t * TN_BT;
#pragma unroll
    for (int v = 0; v < NA; ++v) {
      const int id = tid + BG_THREADS * v;
      const int m = m0 + id / CA, nc = n0 + (id % CA) * 8;
      const bool ok = m < g.M && nc < g.N;
      ra[v] = bg_fetch8<A_F32>(g.a, (long long)m * g.lda + nc, ok);
    }
#pragma unroll
    for (int v = 0; v < NB; ++v) {
      const int id = tid + BG_THREADS * v;
      const int m = m0 + id / CB, kc = c0 + (id % CB) * 8;
      bool ok = m < g.M && kc < g.K;
      long long src = m;
      if (shift != 0) {
        const int tt = m % g.T + shift;
        ok = ok && tt >= 0 && tt < g.T;
        src = (long long)m + shift;
      }
      rb[v] = bg_fetch8<B_F32>(g.b, src * g.ldb + kc, ok);
    }
  };
  // commit: predicate / dropout / rounding of the chunks fetched for tile t, then the LDS images
  auto commit = [&](int buf, int t, const BgRaw<A_F32>* ra, const BgRaw<B_F32>* rb) {
    unsigned char* Ab = lds + buf * STAGE;
    unsigned char* Bb = Ab + IMG_A;
    const int m0 = t * TN_BT;
#pragma unroll
    for (int v = 0; v < NA; ++v) {
      const int id = tid + BG_THREADS * v;
      const int m = m0 + id / CA, nc = n0 + (id % CA) * 8;
      const bool ok = m < g.M && nc < g.N;
      *reinterpret_cast<u32x4*>(Ab + ((id / CA) * LDA + (id % CA) * 8) * 2) =
          bg_finish8<A_F32>(ra[v], ok, A_F32 ? g.a_drop_p : 0.f, g.a_drop_seed + seed_off,
                            (uint64_t)m * (uint64_t)g.N + (uint64_t)nc);
    }
#pragma unroll
    for (int v = 0; v < NB; ++v) {
      const int id = tid + BG_THREADS * v;
      const int m = m0 + id / CB, kc = c0 + (id % CB) * 8;
      bool ok = m < g.M && kc < g.K;
      if (shift != 0) {
        const int tt = m % g.T + shift;
        ok = ok && tt >= 0 && tt < g.T;
      }
      *reinterpret_cast<u32x4*>(Bb + ((id / CB) * LDB + (id % CB) * 8) * 2) = bg_finish8<B_F32>(rb[v], ok, 0.f, 0ull, 0ull);
    }
  };
  auto compute = [&](int buf) {
    const __bf16* Ah = reinterpret_cast<const __bf16*>(lds + buf * STAGE);
    const __bf16* Bh = reinterpret_cast<const __bf16*>(lds + buf * STAGE + IMG_A);
    if (do_bias && tid < BN) {
      float q = 0.f;
#pragma unroll 8
      for (int m = 0; m < TN_BT; ++m) q += (float)Ah[m * LDA + tid];
      colsum += q;
    }
#pragma unroll
    for (int kk = 0; kk < TN_BT / 32; ++kk) {
      bf16x8 af[MR], bf[NR];
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        const __bf16* p = Ah + (kk * 32 + kg * 4 + (li >> 2)) * LDA + wr * (BN / 2) + m * 16 + (li & 3) * 4;
        const bf16x4 lo = bg_tr4(p), hi = bg_tr4(p + 16 * LDA);
        af[m] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int n = 0; n < NR; ++n) {
        const __bf16* p = Bh + (kk * 32 + kg * 4 + (li >> 2)) * LDB + wc * (BK / 2) + n * 16 + (li & 3) * 4;
        const bf16x4 lo = bg_tr4(p), hi = bg_tr4(p + 16 * LDB);
        bf[n] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int n = 0; n < NR; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bf[n], acc[m][n], 0, 0, 0);
    }
  };

  // register set / LDS buffer i & 1 for this slice's i-th tile; same one-barrier-per-tile argument as the NT kernel
  const int st = g.slices;
  int t = slice;
  const int tlast = ntile - 1;  // fetches past the slice's last tile re-read that tile: loads stay unconditional
  fetch(min(t, tlast), ra0, rb0);
  fetch(min(t + st, tlast), ra1, rb1);
  for (; t < ntile; t += 2 * st) {
    commit(0, t, ra0, rb0);
    __syncthreads();
    fetch(min(t + 2 * st, tlast), ra0, rb0);
    compute(0);
    if (t + st < ntile) {
      commit(1, t + st, ra1, rb1);
      __syncthreads();
      fetch(min(t + 3 * st, tlast), ra1, rb1);
      compute(1);
    }
  }

  if (do_bias && tid < BN && (n0 + tid) < g.N && colsum != 0.f) atomicAdd(&g.db[n0 + tid], colsum * g.alpha);
  float* cbase = g.c + (long long)tap * g.c_ts;
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int n = 0; n < NR; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = n0 + wr * (BN / 2) + m * 16 + kg * 4 + r;  // output row  = channel of A
        const int j = c0 + wc * (BK / 2) + n * 16 + li;          // output col  = channel of B
        if (i < g.N && j < g.K) {
          const float v = acc[m][n][r] * g.alpha;
#ifdef TN_NO_ATOMICS  // timing experiment only (scripts/build_tnprobe.sh): what the split-token atomics cost; results are wrong
          if (v != 0.f) cbase[(long long)i * g.c_ns + (long long)j * g.c_ks] = v;
#else
          if (v != 0.f) atomicAdd(cbase + (long long)i * g.c_ns + (long long)j * g.c_ks, v);
#endif
        }
      }
}

template <bool A_F32, bool B_F32, int BN, int BK>
static int bg_tn_go(TnGroupArgs& ga, dim3 grid, hipStream_t st) {
  constexpr size_t LDS = 2 * (size_t)TN_BT * ((BN + 16) + (BK + 16)) * 2;
  static bool attr_set = false;
  if (!attr_set && LDS > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bgemm_tn_kernel<A_F32, B_F32, BN, BK>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL((bgemm_tn_kernel<A_F32, B_F32, BN, BK>), grid, dim3(BG_THREADS), LDS, st, ga);
  KANTTS_CHECK_LAUNCH();
}

template <int BN, int BK>
static int bg_tn_dispatch(TnGroupArgs& ga, dim3 grid, hipStream_t st) {
  const kantts_bgemm_tn_args& g = ga.g;
  if (g.a_f32) {
    if (g.b_f32) return bg_tn_go<true, true, BN, BK>(ga, grid, st);
    return bg_tn_go<true, false, BN, BK>(ga, grid, st);
  }
  if (g.b_f32) return bg_tn_go<false, true, BN, BK>(ga, grid, st);
  return bg_tn_go<false, false, BN, BK>(ga, grid, st);
}

static int bg_tn_tile_rule(const kantts_bgemm_tn_args& g, int nprob) {
  // [round 4] 128 x 128 / 64 x 256 / 128 x 256 tiles were measured over every grouped shape of the training step
  // (profiles/r04_runL_tn_tile_sweep.log): the 64 x 128 tile with the right slice count is within a few per cent of the
  // best everywhere except two shapes where 128 x 128 gains 10-15 %; the larger tiles stay available to the sweep only
  (void)g;
  (void)nprob;
  return 64128;
}

static int bg_tn_launch(TnGroupArgs& ga, hipStream_t st) {
  kantts_bgemm_tn_args& g = ga.g;
  if (g.M < 0 || g.N < 1 || g.K < 1 || g.ntaps < 1 || ga.nprob < 1 || ga.nprob > KANTTS_TN_MAX_GROUP) return KANTTS_E_BADARG;
  if (g.M == 0) return KANTTS_OK;
  if ((g.N & 7) || (g.K & 7) || (g.lda & 7) || (g.ldb & 7)) return KANTTS_E_UNSUPPORTED;
  for (int p = 0; p < ga.nprob; ++p) {
    if (!ga.a[p] || !ga.b[p] || !ga.c[p]) return KANTTS_E_BADARG;
    if (!bg_aligned16(ga.a[p]) || !bg_aligned16(ga.b[p])) return KANTTS_E_UNSUPPORTED;
  }
  if ((g.shift0 != 0 || g.shift_step != 0) && g.T <= 0) return KANTTS_E_BADARG;
  // tile (BN x BK) and token slices.  kantts_launch_tuning(tn_tile = BN * 1000 + BK (64128 / 128128 / 64256 / 128256),
  // tn_slices, ...) forces them (scripts/tn_sweep.py, tests; the host layer maps KANTTS_TN_TILE / KANTTS_TN_SLICES onto it:
  // the library itself reads no environment here); the rules below are what that sweep measured
  // (profiles/r04_runL_tn_tile_sweep.log).
  int code = kantts_tune_tn_tile.load(std::memory_order_relaxed);  // kantts_launch_tuning (sweeps / tests); 0 = the rule
  const int forced_slices = kantts_tune_tn_slices.load(std::memory_order_relaxed);
  // (code + 1, e.g. 64129: that tile on the 3-D grid of rounds 2-5, without the XCD-aware mapping -- for A/B runs)
  bool xcd_map = true;
  if (code == 64129 || code == 128129 || code == 64257 || code == 128257) code -= 1, xcd_map = false;
  if (code != 64128 && code != 128128 && code != 64256 && code != 128256) code = bg_tn_tile_rule(g, ga.nprob);
  const int BN = code / 1000, BK = code % 1000;
  const int tiles = kantts_cdiv(g.N, BN) * kantts_cdiv(g.K, BK) * g.ntaps * ga.nprob;
  const int ntile = kantts_cdiv(g.M, TN_BT);
  int slices = g.slices;
  if (forced_slices > 0) slices = forced_slices;
  if (slices <= 0) {
    // [round 4, scripts/tn_sweep.py] as many token slices as still give ONE resident round of workgroups (two per CU: 512;
    // 576 workgroups = a second, mostly empty round cost 123 us where 384 took 95), at most ~4 M fp32 atomics per launch
    // (round 2's budget of 1.5 M left the 19 584-row postnet problems at 4 slices: 126 us where 8 slices take 80)
    static const bool r2_rule = getenv("KANTTS_TN_SLICE_RULE_R2") != nullptr;  // A/B switch: rounds 2-3's rule
    slices = r2_rule ? kantts_cdiv(320, tiles) : 512 / tiles;
    if (xcd_map && !r2_rule) {
      // [round 6] groups (= problem x tap x slice) are dealt to the 8 XCDs whole, each XCD has 64 places (32 CUs x 2
      // workgroups): as many groups per XCD as fit in ONE round of its places (7 problems of 6 tiles: 12 slices were 11
      // groups = 66 workgroups on an XCD, a second round for two of them)
      const int P = kantts_cdiv(g.N, BN) * kantts_cdiv(g.K, BK);
      const int per_xcd = P <= 64 ? 64 / P : 1;
      slices = 8 * per_xcd / (g.ntaps * ga.nprob);
    }
    if (slices < 1) slices = 1;
    if (slices > 16 && !r2_rule) slices = 16;
    const long long cap = ((r2_rule ? 3ll << 19 : 1ll << 22)) / ((long long)g.N * g.K * g.ntaps * ga.nprob) + 1;
    if (slices > cap) slices = (int)cap;
  }
  if (slices > ntile) slices = ntile;
  if (slices < 1) slices = 1;
  g.slices = slices;
  dim3 grid(kantts_cdiv(g.K, BK), kantts_cdiv(g.N, BN), ga.nprob * g.ntaps * slices);
  ga.ktiles = ga.ntiles = ga.ngroups = 0;
  // a group per XCD and round: with fewer than 8 groups, or a last round that leaves more than a quarter of the XCDs idle,
  // the 3-D grid (every XCD busy, every re-read through the fabric) is the better of two evils
  if (xcd_map && (int)grid.z * 4 < 3 * 8 * kantts_cdiv((int)grid.z, 8)) xcd_map = false;
  if (xcd_map) {
    ga.ktiles = (int)grid.x, ga.ntiles = (int)grid.y, ga.ngroups = (int)grid.z;
    grid = dim3(8u * grid.x * grid.y * (unsigned)kantts_cdiv(ga.ngroups, 8), 1, 1);
  }
  switch (code) {
    case 128256: return bg_tn_dispatch<128, 256>(ga, grid, st);
    case 128128: return bg_tn_dispatch<128, 128>(ga, grid, st);
    case 64256: return bg_tn_dispatch<64, 256>(ga, grid, st);
    default: return bg_tn_dispatch<64, 128>(ga, grid, st);
  }
}

extern "C" int kantts_bgemm_tn(const kantts_bgemm_tn_args* gp, void* stream) {
  if (!gp) return KANTTS_E_BADARG;
  TnGroupArgs ga = {};
  ga.g = *gp;
  ga.nprob = 1;
  ga.a[0] = gp->a; ga.b[0] = gp->b; ga.c[0] = gp->c; ga.db[0] = gp->db; ga.a_drop_seed[0] = gp->a_drop_seed;
  return bg_tn_launch(ga, (hipStream_t)stream);
}

extern "C" int kantts_bgemm_tn_grouped(const kantts_bgemm_tn_args* shape, int nprob, const void* const* a,
                                       const void* const* b, float* const* c, float* const* db,
                                       const uint64_t* a_drop_seed, void* stream) {
  if (!shape || !a || !b || !c || nprob < 1 || nprob > KANTTS_TN_MAX_GROUP) return KANTTS_E_BADARG;
  TnGroupArgs ga = {};
  ga.g = *shape;
  ga.nprob = nprob;
  for (int p = 0; p < nprob; ++p) {
    ga.a[p] = a[p]; ga.b[p] = b[p]; ga.c[p] = c[p];
    ga.db[p] = db ? db[p] : nullptr;
    ga.a_drop_seed[p] = a_drop_seed ? a_drop_seed[p] : shape->a_drop_seed;
  }
  return bg_tn_launch(ga, (hipStream_t)stream);
}

// ================================================================================================ casts
// fp32 -> bf16 over a flat buffer (the parameter arena's shadow; activations that feed several contractions)
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long long n8) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const float4 a = reinterpret_cast<const float4*>(src)[2 * i], b = reinterpret_cast<const float4*>(src)[2 * i + 1];
  u32x4 w = {bg_pack2(a.x, a.y), bg_pack2(a.z, a.w), bg_pack2(b.x, b.y), bg_pack2(b.z, b.w)};
  reinterpret_cast<u32x4*>(dst)[i] = w;
}

extern "C" int kantts_cast_f32_bf16(const float* src, void* dst, long long n, void* stream) {
  if (!src || !dst || n < 0 || (n & 7) || !bg_aligned16(src) || !bg_aligned16(dst)) return KANTTS_E_BADARG;
  if (n == 0) return KANTTS_OK;
  const long long n8 = n / 8;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(kantts_cdiv(n8, 256)), dim3(256), 0, (hipStream_t)stream, src,
                     reinterpret_cast<__bf16*>(dst), n8);
  KANTTS_CHECK_LAUNCH();
}

// Conv1d weights (N, Cin, KT) fp32 -> tap-major (KT, N, Cin) bf16, a table of them in one launch.
__global__ __launch_bounds__(256) void tapmajor_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst,
                                                           const kantts_tapmajor_desc* __restrict__ tab) {
  const kantts_tapmajor_desc d = tab[blockIdx.y];
  const long long total = (long long)d.N * d.Cin * d.KT;
  for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(o % d.Cin);
    const long long q = o / d.Cin;
    const int n = (int)(q % d.N), tap = (int)(q / d.N);
    dst[d.dst_off + o] = (__bf16)src[d.src_off + ((long long)n * d.Cin + c) * d.KT + tap];
  }
}

extern "C" int kantts_tapmajor_bf16(const float* src, void* dst, const kantts_tapmajor_desc* table_dev, int ndesc,
                                    int blocks_per_desc, void* stream) {
  if (!src || !dst || !table_dev || ndesc < 0 || blocks_per_desc < 1) return KANTTS_E_BADARG;
  if (ndesc == 0) return KANTTS_OK;
  hipLaunchKernelGGL(tapmajor_bf16_kernel, dim3(blocks_per_desc, ndesc), dim3(256), 0, (hipStream_t)stream, src,
                     reinterpret_cast<__bf16*>(dst), table_dev);
  KANTTS_CHECK_LAUNCH();
}

// ReLU(+dropout) backward on a saved activation: dz = (y > 0) ? dy * scale : 0, all operands bf16 or fp32, output bf16.
// Used by the generic fused linear in bf16 mode (the fused FFN applies the same gate in its contraction epilogue).
template <bool DY_BF16, bool Y_BF16>
__global__ __launch_bounds__(256) void relu_gate_kernel(const void* __restrict__ dy, const void* __restrict__ y,
                                                       __bf16* __restrict__ dz, float scale, long long n8) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  float d[8], a[8];
  if (DY_BF16) {
    const u32x4 q = reinterpret_cast<const u32x4*>(dy)[i];
    d[0] = bg_lo(q.x); d[1] = bg_hi(q.x); d[2] = bg_lo(q.y); d[3] = bg_hi(q.y);
    d[4] = bg_lo(q.z); d[5] = bg_hi(q.z); d[6] = bg_lo(q.w); d[7] = bg_hi(q.w);
  } else {
    const float4 p = reinterpret_cast<const float4*>(dy)[2 * i], q = reinterpret_cast<const float4*>(dy)[2 * i + 1];
    d[0] = p.x; d[1] = p.y; d[2] = p.z; d[3] = p.w; d[4] = q.x; d[5] = q.y; d[6] = q.z; d[7] = q.w;
  }
  if (Y_BF16) {
    const u32x4 q = reinterpret_cast<const u32x4*>(y)[i];
    a[0] = bg_lo(q.x); a[1] = bg_hi(q.x); a[2] = bg_lo(q.y); a[3] = bg_hi(q.y);
    a[4] = bg_lo(q.z); a[5] = bg_hi(q.z); a[6] = bg_lo(q.w); a[7] = bg_hi(q.w);
  } else {
    const float4 p = reinterpret_cast<const float4*>(y)[2 * i], q = reinterpret_cast<const float4*>(y)[2 * i + 1];
    a[0] = p.x; a[1] = p.y; a[2] = p.z; a[3] = p.w; a[4] = q.x; a[5] = q.y; a[6] = q.z; a[7] = q.w;
  }
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = a[e] > 0.f ? d[e] * scale : 0.f;
  u32x4 w = {bg_pack2(o[0], o[1]), bg_pack2(o[2], o[3]), bg_pack2(o[4], o[5]), bg_pack2(o[6], o[7])};
  reinterpret_cast<u32x4*>(dz)[i] = w;
}

extern "C" int kantts_relu_gate_bf16(const void* dy, int dy_bf16, const void* y, int y_bf16, void* dz_bf16, float scale,
                                     long long n, void* stream) {
  if (!dy || !y || !dz_bf16 || n < 0 || (n & 7) || !bg_aligned16(dy) || !bg_aligned16(y) || !bg_aligned16(dz_bf16))
    return KANTTS_E_BADARG;
  if (n == 0) return KANTTS_OK;
  const long long n8 = n / 8;
  dim3 grid(kantts_cdiv(n8, 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  __bf16* o = reinterpret_cast<__bf16*>(dz_bf16);
  if (dy_bf16) {
    if (y_bf16)
      hipLaunchKernelGGL((relu_gate_kernel<true, true>), grid, block, 0, st, dy, y, o, scale, n8);
    else
      hipLaunchKernelGGL((relu_gate_kernel<true, false>), grid, block, 0, st, dy, y, o, scale, n8);
  } else {
    if (y_bf16)
      hipLaunchKernelGGL((relu_gate_kernel<false, true>), grid, block, 0, st, dy, y, o, scale, n8);
    else
      hipLaunchKernelGGL((relu_gate_kernel<false, false>), grid, block, 0, st, dy, y, o, scale, n8);
  }
  KANTTS_CHECK_LAUNCH();
}
