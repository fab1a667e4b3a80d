// Lean specialisations of the segmented GEMM for the shapes that carry the SAM-BERT step.
//
// gemm.hip's kernel interprets every feature of the segment descriptor at run time (token maps,
// groups, scalar layouts ...); its instruction footprint (~18k instructions) does not fit the
// instruction cache and it pays integer divisions per staged element.  The kernels here are compiled
// per staging layout (float4 along k / 4x4 register-transposed blocks along rows, for A and for B),
// support only the plain descriptor (optional token shift, gate, dropout regeneration, kmask, the full
// epilogue, split-K, bias row sums) and hoist the index arithmetic out of the reduction loop.
//
// Tile: 256 threads (4 waves, 2x2), BM x 64 outputs, reduction tile BK = 128 (bf16) / 64 (fp32):
// the SAM-BERT contractions have K = 128 ... 1024, so a tile covers 1/1 ... 1/8 of K and the
// load-latency / barrier cost is paid 4x less often than with BK = 32 (PMC of the BK = 32 version:
// 50 % of wave cycles in s_waitcnt/barrier, profiles/r01_gemm_fwd128x1024_pmc.txt).  Operands are
// rounded to bf16 while they are staged (8-byte packed LDS stores); LDS rows are padded by 16 B so the
// 16-byte MFMA fragment reads of 16 consecutive rows hit 16 distinct 16-byte slots.  Global loads of
// tile t+1 are issued before the MFMAs of tile t (register double buffer).
#include <stdlib.h>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define F_BN 64
#define F_THREADS 256

__device__ __forceinline__ float f_epilogue(const kantts_gemm_args& g, float acc, int i, int j, bool first_slice,
                                            uint64_t seed_off, int grp) {
  float v = acc;
  if (first_slice && g.bias) v += g.bias[j + grp * g.bias_gs];
  if (first_slice && g.bias2) v += g.bias2[j + grp * g.bias_gs];
  v *= g.alpha;
  if (g.relu) v = fmaxf(v, 0.f);
  if (g.out_act) v = v > 0.f ? v : v * g.out_slope;
  if (g.drop_p > 0.f)
    v *= kantts_dropout_scale(g.drop_p, g.drop_seed + seed_off, (uint64_t)i * (uint64_t)g.N + (uint64_t)j);
  if (first_slice && g.res) v += g.res[(long long)i * g.r_is + (long long)j * g.r_js + grp * g.r_gs];
  if (g.gate) {
    const float gv = g.gate[(long long)i * g.c_is + (long long)j * g.c_js + grp * g.c_gs];
    v *= (gv > 0.f) ? 1.f : g.gate_slope;
  }
  if (g.rowmask && g.rowmask[i]) v = 0.f;
  return v;
}

// Token shifts (conv taps) of the plain kind only: a row / reduction index is a token of a (B, T) grid and
// the tap reads token t + shift of the same sequence (zero outside [0, T)).  Strided / folded / upsampled token
// maps go to conv_win.hip / conv_wgrad.hip (or the generic kernel of gemm.hip).
__device__ __forceinline__ bool f_shift_token(int tok, int shift, int T, long long& row) {
  row = (long long)tok + shift;
  if (shift == 0) return true;
  const int t = tok % T + shift;
  return t >= 0 && t < T;
}

// 4 consecutive elements of the LDS image at element index idx (idx % 4 == 0)
template <bool BF16>
__device__ __forceinline__ void lds_store4(void* base, int idx, float v0, float v1, float v2, float v3) {
  if (BF16) {
    bf16x4 p = {(__bf16)v0, (__bf16)v1, (__bf16)v2, (__bf16)v3};
    *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(base) + idx) = p;
  } else {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + idx) = make_float4(v0, v1, v2, v3);
  }
}

// gfx950 LDS transpose read: the 16 lanes of a group each point at 4 contiguous bf16 of a [4 k][16 rows] block
// (lane i: k-row i>>2, rows 4*(i&3)..+3, any row pitch) and receive column i: the 4 k values of row i.
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
__device__ __forceinline__ bf16x4 lds_read_tr4(const __bf16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p));
}

// One operand (A when IS_A, else B) of a tile: fetch() issues the global loads, commit() writes LDS.
template <bool BF16, int ROWS, int BK, bool ROWVEC, bool IS_A, int LD, bool GATE>
struct Stager {
  // k-vector layout: LPR lanes cover one row, NV vectors per thread
  static constexpr int LPR = BK / 4;
  static constexpr int RPP = F_THREADS / LPR;
  static constexpr int NVK = ROWS / RPP;
  // row-vector layout: 4x4 blocks (4 rows x 4 k), NB blocks per thread
  static constexpr int RG = ROWS / 4;
  static constexpr int NBLK = RG * (BK / 4);
  static constexpr int NB = (NBLK + F_THREADS - 1) / F_THREADS;
  static constexpr int NREG = ROWVEC ? NB * 4 : NVK;

  float4 r[NREG], gt[GATE ? NREG : 1];
  unsigned ok;  // bit per register vector
  int cur_k0;

  // operands are addressed as (uniform 64-bit base) + (32-bit byte offset): one VGPR per vector instead of two,
  // and the loads use the SGPR-base addressing mode.  kantts_gemm_try_fast() rejects operands >= 4 GiB.
  const char* p;
  const char* gp;
  uint32_t ks4;  // k stride in bytes (row-vector layout)
  uint32_t base[ROWVEC ? NB : NVK];
  bool rok[ROWVEC ? NB : NVK];
  int klen, shift, tok_axis, T, act;
  float slope;
  const uint8_t* kmask;

  __device__ __forceinline__ void setup(const kantts_gemm_seg& s, const kantts_gemm_args& g, int row0, int nrows,
                                        int tap, int grp) {
    const int tid = threadIdx.x;
    const long long goff = (long long)grp * (IS_A ? g.a_gs : g.b_gs) + (IS_A ? 0 : (long long)tap * s.b_tap);
    p = reinterpret_cast<const char*>((IS_A ? s.a : s.b) + goff);
    gp = (GATE && IS_A && s.a_gate) ? reinterpret_cast<const char*>(s.a_gate + goff) : nullptr;
    klen = s.klen;
    T = (IS_A ? s.a_Tq : s.b_Tq) > 0 ? (IS_A ? s.a_Tq : s.b_Tq) : g.T;
    act = IS_A ? s.a_act : s.b_act;
    slope = IS_A ? s.a_slope : s.b_slope;
    tok_axis = IS_A ? s.a_tok_axis : s.b_tok_axis;
    shift = tok_axis ? (IS_A ? s.a_shift0 + tap * s.a_shift_step : s.b_shift0 + tap * s.b_shift_step) : 0;
    kmask = IS_A ? g.kmask : nullptr;
    if (ROWVEC) {
      ks4 = (uint32_t)(IS_A ? s.a_ks : s.b_ks) * 4u;
#pragma unroll
      for (int v = 0; v < NB; ++v) {
        const int id = tid + F_THREADS * v;
        const int row = row0 + (id % RG) * 4;
        rok[v] = (id < NBLK) && (row < nrows);
        base[v] = (uint32_t)row * 4u;
      }
    } else {
      const long long rs = IS_A ? s.a_is : s.b_js;
      ks4 = 4u;
#pragma unroll
      for (int v = 0; v < NVK; ++v) {
        const int row = row0 + tid / LPR + RPP * v;
        bool okr = row < nrows;
        long long rr = row;
        if (IS_A && tok_axis == 1) okr = f_shift_token(row, shift, T, rr) && okr;
        rok[v] = okr;
        base[v] = (uint32_t)((rr * rs + (tid % LPR) * 4) * 4);
      }
    }
  }

  __device__ __forceinline__ void fetch(int k0) {
    const int tid = threadIdx.x;
    ok = 0;
    cur_k0 = k0;
    if (ROWVEC) {
#pragma unroll
      for (int v = 0; v < NB; ++v) {
        const int id = tid + F_THREADS * v;
        const int kb = k0 + (id / RG) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int kk = kb + e;
          bool o = rok[v] && kk < klen;
          long long kq = kk;
          if (tok_axis == 2) o = f_shift_token(kk, shift, T, kq) && o;
          if (kmask && o) o = kmask[kk] == 0;
          const uint32_t f = o ? ((uint32_t)kq * ks4 + base[v]) : 0u;
          r[v * 4 + e] = *reinterpret_cast<const float4*>(p + f);
          if (GATE && gp) gt[v * 4 + e] = *reinterpret_cast<const float4*>(gp + f);
          if (o) ok |= 1u << (v * 4 + e);
        }
      }
    } else {
#pragma unroll
      for (int v = 0; v < NVK; ++v) {
        const bool o = rok[v] && (k0 + (tid % LPR) * 4) < klen;
        const uint32_t f = o ? (base[v] + (uint32_t)k0 * 4u) : 0u;
        r[v] = *reinterpret_cast<const float4*>(p + f);
        if (GATE && gp) gt[v] = *reinterpret_cast<const float4*>(gp + f);
        if (o) ok |= 1u << v;
      }
    }
  }

  __device__ __forceinline__ float fix(float x, float gate, int reg, int lane_e, const kantts_gemm_seg& s,
                                       uint64_t seed_off) const {
    float v = ((ok >> reg) & 1u) ? x : 0.f;
    if (act) v = v > 0.f ? v : v * slope;
    if (IS_A) {
      if (GATE && gp && !(gate > 0.f)) v *= s.a_gate_slope;
      if (s.a_drop_p > 0.f) {
        // element offset of A (the dropout counter), recomputed instead of kept in registers
        long long o;
        if (ROWVEC) {
          const int id = threadIdx.x + F_THREADS * (reg >> 2);
          long long kq = cur_k0 + (id / RG) * 4 + (reg & 3);
          if (tok_axis == 2) kq += shift;
          o = (long long)(((uint32_t)kq * ks4 + base[reg >> 2]) >> 2) + lane_e;
        } else {
          o = (long long)(base[reg] >> 2) + cur_k0 + lane_e;
        }
        v *= kantts_dropout_scale(s.a_drop_p, s.a_drop_seed + seed_off, (uint64_t)o);
      }
    }
    return v;
  }

  __device__ __forceinline__ void commit(void* lds, const kantts_gemm_seg& s, uint64_t seed_off) const {
    const int tid = threadIdx.x;
    if (ROWVEC) {
      // K-major image [k][rows] (pitch LD): a float4 of 4 consecutive rows at one k is stored as it was loaded;
      // the MFMA fragments are formed by transpose reads (bf16) / scalar reads (fp32), never in registers
#pragma unroll
      for (int v = 0; v < NB; ++v) {
        const int id = tid + F_THREADS * v;
        if (NBLK % F_THREADS != 0 && id >= NBLK) continue;
        const int rl = (id % RG) * 4, kl = (id / RG) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float4 x = r[v * 4 + e];
          const float4 q = (GATE && gp) ? gt[v * 4 + e] : make_float4(1.f, 1.f, 1.f, 1.f);
          lds_store4<BF16>(lds, (kl + e) * LD + rl, fix(x.x, q.x, v * 4 + e, 0, s, seed_off),
                           fix(x.y, q.y, v * 4 + e, 1, s, seed_off), fix(x.z, q.z, v * 4 + e, 2, s, seed_off),
                           fix(x.w, q.w, v * 4 + e, 3, s, seed_off));
        }
      }
    } else {
#pragma unroll
      for (int v = 0; v < NVK; ++v) {
        const float4 x = r[v];
        const float4 q = (GATE && gp) ? gt[v] : make_float4(1.f, 1.f, 1.f, 1.f);
        lds_store4<BF16>(lds, (tid / LPR + RPP * v) * LD + (tid % LPR) * 4, fix(x.x, q.x, v, 0, s, seed_off),
                         fix(x.y, q.y, v, 1, s, seed_off), fix(x.z, q.z, v, 2, s, seed_off),
                         fix(x.w, q.w, v, 3, s, seed_off));
      }
    }
  }
};

template <bool BF16, int BM, bool BIGK, bool A_ROW, bool B_ROW, bool GATE>
__global__ __launch_bounds__(F_THREADS) void gemm_fast_kernel(const kantts_gemm_args g) {
  // deep reduction tiles only when there are many of them to amortise (K >= 512, split-K weight gradients);
  // short-K launches (the 128 -> 1024 projections) want occupancy instead: BK = 32 keeps them at ~70 VGPRs
#ifdef F_VARIANT_BK64
  constexpr int BK = BIGK ? (BF16 ? 128 : 64) : (BF16 ? 64 : 32);
#else
  constexpr int BK = BIGK ? (BF16 ? 128 : 64) : 32;
#endif
  constexpr int ESZ = BF16 ? 2 : 4;
  // LDS images.  k-vector operands: [rows][BK] with a row pitch chosen for the fragment read: 16-byte reads
  // (both operands k-vector) are conflict-free at pitch = 32 B (mod 64 B), the 8-byte reads of the permuted-k
  // convention (below) at pitch = 16 B (mod 32 B).  Row-vector operands: K-major [BK][rows], pitch 80 / 48
  // elements (64 / 32 rows): 8 consecutive k-rows start 8 banks apart, which makes the transpose reads (bf16)
  // and the scalar reads (fp32) of a 32-lane group touch every bank once.
  constexpr bool MIXED = A_ROW || B_ROW;
  constexpr int LDK = BF16 ? (MIXED ? BK + 8 : BK + 16) : (BK + 4);
  constexpr int PA = (BM == 64) ? 80 : 48, PB = 80;
  constexpr int LDA = A_ROW ? PA : LDK, LDB = B_ROW ? PB : LDK;
  constexpr int A_BYTES = (A_ROW ? BK * PA : BM * LDK) * ESZ, B_BYTES = (B_ROW ? BK * PB : F_BN * LDK) * ESZ;
  constexpr int OPB = A_BYTES + B_BYTES, CSB = BM * (F_BN + 4) * 4;  // operand tiles / epilogue staging
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[OPB > CSB ? OPB : CSB];
  void* Al = lds_raw;
  void* Bl = lds_raw + A_BYTES;
  constexpr int MREP = BM / 32;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
#ifdef F_DEBUG  // ablation mask (experiment builds only, scripts/build_gdbg.sh): 1 no operand loads, 2 no output stores, 8 no MFMA
  const int dbg = g.precision >> 8;
#else
  constexpr int dbg = 0;
#endif
  // XCD-aware tile order.  The dispatcher deals workgroup L = y*gx + x to XCD L % 8, so the gx column tiles that
  // share one A row tile would land on 8 different L2s and the A operand would cross the fabric 8 times
  // (measured: 27 MB fetched for 3.3 MB of A at 6528x1024x128).  Workgroups of one XCD are given consecutive
  // tiles in row-major order instead: XCD k owns a contiguous band of rows.
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const int gx = gridDim.x, total = gx * gridDim.y;
    if (total >= 64) {
      const int L = by * gx + bx, k = L & 7, j = L >> 3;
      const int q = total >> 3, r = total & 7;
      const int vid = k * q + (k < r ? k : r) + j;
      by = vid / gx;
      bx = vid - by * gx;
    }
  }
  const int i0 = by * BM;
  const int j0 = bx * F_BN;
  int ztap = -1, grp = 0, zslice = 0;
  if (gridDim.z > 1) {  // z = (tap slab, group, split-K slice); the common single-slice launch skips the divisions
    const int zper = g.groups * g.splitk;
    ztap = g.z_taps > 0 ? (int)(blockIdx.z / zper) : -1;  // one tap per z-slab (conv weight gradients)
    const int zrem = blockIdx.z % zper;
    grp = zrem / g.splitk;
    zslice = zrem % g.splitk;
  } else if (g.z_taps > 0) {
    ztap = 0;
  }
  const uint64_t seed_off = g.seed_dev ? *g.seed_dev : 0ull;

  f32x4 acc[MREP][2];
#pragma unroll
  for (int m = 0; m < MREP; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float rowsum = 0.f;
  const bool do_rowsum = (g.a_rowsum != nullptr) && (bx == 0) && (ztap <= 0);

  Stager<BF16, BM, BK, A_ROW, true, LDA, GATE> sa;
  Stager<BF16, F_BN, BK, B_ROW, false, LDB, false> sb;
  int tile_counter = 0;

  for (int sidx = 0; sidx < g.nseg; ++sidx) {
    const kantts_gemm_seg& s = g.seg[sidx];
    const int ntile = (s.klen + BK - 1) / BK;
    for (int tap = 0; tap < s.ntaps; ++tap) {
      if (ztap >= 0 && tap != ztap) continue;
      sa.setup(s, g, i0, g.M, tap, grp);
      sb.setup(s, g, j0, g.N, tap, grp);
      // rows are block-relative inside the stagers' LDS writes: shift the bases instead of the row ids
      auto next_owned = [&](int from) {
        int q = from;
        while (g.splitk > 1 && q < ntile && ((tile_counter + q) % g.splitk) != zslice) ++q;
        return q;
      };
      int t = next_owned(0);
      if (t < ntile && !(dbg & 1)) {
        sa.fetch(t * BK);
        sb.fetch(t * BK);
      }
      while (t < ntile) {
        sa.commit(Al, s, seed_off);
        sb.commit(Bl, s, seed_off);
        __syncthreads();
        const int tn = next_owned(t + 1);
        if (tn < ntile && !(dbg & 1)) {
          sa.fetch(tn * BK);
          sb.fetch(tn * BK);
        }
        if (do_rowsum && sidx == 0 && tid < BM) {
          float q = 0.f;
          const int rstep = A_ROW ? LDA : 1, rbase = A_ROW ? tid : tid * LDA;
          if (BF16) {
            const __bf16* row = reinterpret_cast<const __bf16*>(Al) + rbase;
            for (int k = 0; k < BK; ++k) q += (float)row[k * rstep];
          } else {
            const float* row = reinterpret_cast<const float*>(Al) + rbase;
            for (int k = 0; k < BK; ++k) q += row[k * rstep];
          }
          rowsum += q;
        }
        if (BF16) {
          const __bf16* Ah = reinterpret_cast<const __bf16*>(Al);
          const __bf16* Bh = reinterpret_cast<const __bf16*>(Bl);
          const int li = lane & 15, kg = lane >> 4;
#pragma unroll
          for (int kk = 0; kk < BK / 32; ++kk) {
            bf16x8 af[MREP], bfr[2];
            // MIXED: lane group kg holds k = kg*4..+3 and 16+kg*4..+3 of the 32-k step (what two transpose reads
            // deliver); the k-vector operand follows with two 8-byte reads.  Otherwise 8 consecutive k, one 16-byte read.
#pragma unroll
            for (int m = 0; m < MREP; ++m) {
              const int r0 = wr * (BM / 2) + m * 16;
              if (A_ROW) {
                const __bf16* p = &Ah[(kk * 32 + kg * 4 + (li >> 2)) * LDA + r0 + (li & 3) * 4];
                const bf16x4 lo = lds_read_tr4(p), hi = lds_read_tr4(p + 16 * LDA);
                af[m] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
              } else if (MIXED) {
                const __bf16* p = &Ah[(r0 + li) * LDA + kk * 32 + kg * 4];
                const bf16x4 lo = *reinterpret_cast<const bf16x4*>(p), hi = *reinterpret_cast<const bf16x4*>(p + 16);
                af[m] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
              } else {
                af[m] = *reinterpret_cast<const bf16x8*>(&Ah[(r0 + li) * LDA + kk * 32 + kg * 8]);
              }
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
              const int c0 = wc * 32 + n * 16;
              if (B_ROW) {
                const __bf16* p = &Bh[(kk * 32 + kg * 4 + (li >> 2)) * LDB + c0 + (li & 3) * 4];
                const bf16x4 lo = lds_read_tr4(p), hi = lds_read_tr4(p + 16 * LDB);
                bfr[n] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
              } else if (MIXED) {
                const __bf16* p = &Bh[(c0 + li) * LDB + kk * 32 + kg * 4];
                const bf16x4 lo = *reinterpret_cast<const bf16x4*>(p), hi = *reinterpret_cast<const bf16x4*>(p + 16);
                bfr[n] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
              } else {
                bfr[n] = *reinterpret_cast<const bf16x8*>(&Bh[(c0 + li) * LDB + kk * 32 + kg * 8]);
              }
            }
            if (dbg & 8) continue;
#pragma unroll
            for (int m = 0; m < MREP; ++m)
#pragma unroll
              for (int n = 0; n < 2; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bfr[n], acc[m][n], 0, 0, 0);
          }
        } else {
          const float* Af = reinterpret_cast<const float*>(Al);
          const float* Bf = reinterpret_cast<const float*>(Bl);
#pragma unroll
          for (int ks = 0; ks < BK / 4; ++ks) {
            float af[MREP], bfr[2];
            const int kq = ks * 4 + (lane >> 4);
#pragma unroll
            for (int m = 0; m < MREP; ++m) {
              const int r = wr * (BM / 2) + m * 16 + (lane & 15);
              af[m] = A_ROW ? Af[kq * LDA + r] : Af[r * LDA + kq];
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
              const int c = wc * 32 + n * 16 + (lane & 15);
              bfr[n] = B_ROW ? Bf[kq * LDB + c] : Bf[c * LDB + kq];
            }
#pragma unroll
            for (int m = 0; m < MREP; ++m)
#pragma unroll
              for (int n = 0; n < 2; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bfr[n], acc[m][n], 0, 0, 0);
          }
        }
        __syncthreads();
        t = tn;
      }
      tile_counter += ntile;
    }
  }

  if (do_rowsum && tid < BM && (i0 + tid) < g.M) atomicAdd(&g.a_rowsum[i0 + tid + grp * g.bias_gs], rowsum);

  const bool first_slice = (zslice == 0);
  // ---- coalesced epilogue: accumulators go through LDS so that 16 lanes write one 256-byte output row
  // (the MFMA C layout gives each lane 4 rows x 1 column: 64-byte pieces, which made the 26 MB output of the
  // 128->1024 projections the slowest part of the kernel)
  const bool vec_out = !g.accumulate && g.c_js == 1 && (g.c_is & 3) == 0 && ((uintptr_t)g.c & 15) == 0 &&
                       (!g.res || (g.r_js == 1 && (g.r_is & 3) == 0 && ((uintptr_t)g.res & 15) == 0)) &&
                       (g.N & 3) == 0 && g.groups <= 1 && !g.gate && ztap < 0;
#ifdef F_VARIANT_DIRECT_EPI
  if (false) {
#else
  if (vec_out) {
#endif
    constexpr int CLD = F_BN + 4;
    float* Cs = reinterpret_cast<float*>(lds_raw);  // BM x 68 floats <= the operand tiles
#pragma unroll
    for (int m = 0; m < MREP; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Cs[(wr * (BM / 2) + m * 16 + (lane >> 4) * 4 + r) * CLD + wc * 32 + n * 16 + (lane & 15)] = acc[m][n][r];
    __syncthreads();
#pragma unroll
    for (int v = 0; v < BM / 16; ++v) {
      const int rl = (tid >> 4) + 16 * v;
      const int i = i0 + rl, j = j0 + (tid & 15) * 4;
      if (i < g.M && j < g.N && !(dbg & 2)) {
        const float4 a4 = *reinterpret_cast<const float4*>(&Cs[rl * CLD + (tid & 15) * 4]);
        float o[4] = {a4.x, a4.y, a4.z, a4.w};
        float rr[4] = {0.f, 0.f, 0.f, 0.f};
        if (g.res) {
          const float4 r4 = *reinterpret_cast<const float4*>(g.res + (long long)i * g.r_is + j);
          rr[0] = r4.x; rr[1] = r4.y; rr[2] = r4.z; rr[3] = r4.w;
        }
        const bool masked = g.rowmask && g.rowmask[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float val = o[e];
          if (g.bias) val += g.bias[j + e];
          if (g.bias2) val += g.bias2[j + e];
          val *= g.alpha;
          if (g.relu) val = fmaxf(val, 0.f);
          if (g.out_act) val = val > 0.f ? val : val * g.out_slope;
          if (g.drop_p > 0.f)
            val *= kantts_dropout_scale(g.drop_p, g.drop_seed + seed_off, (uint64_t)i * (uint64_t)g.N + (uint64_t)(j + e));
          val += rr[e];
          o[e] = masked ? 0.f : val;
        }
        {
          // streaming (nontemporal) store: the tile is not re-read by this kernel; measured -7 % / -9 % on the two
          // FFN contractions with 26 MB outputs, neutral elsewhere (profiles/r01_gemm_variants.log)
          f32x4 nt = {o[0], o[1], o[2], o[3]};
          __builtin_nontemporal_store(nt, reinterpret_cast<f32x4*>(g.c + (long long)i * g.c_is + j));
        }
      }
    }
    return;
  }
#pragma unroll
  for (int m = 0; m < MREP; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + wr * (BM / 2) + m * 16 + (lane >> 4) * 4 + r;
        const int j = j0 + wc * 32 + n * 16 + (lane & 15);
        if (i < g.M && j < g.N && !(dbg & 2)) {
          const float v = f_epilogue(g, acc[m][n][r], i, j, first_slice, seed_off, grp);
          float* dst = &g.c[(long long)i * g.c_is + (long long)j * g.c_js + (long long)grp * g.c_gs +
                            (ztap > 0 ? (long long)ztap * g.c_tap : 0)];
          if (g.accumulate)
            atomicAdd(dst, v);
          else
            *dst = v;
        }
      }
}

template <bool BF16, int BM, bool BIGK, bool GATE>
static void launch_fast3(const kantts_gemm_args& g, bool a_row, bool b_row, dim3 grid, hipStream_t st) {
  if (a_row) {
    if (b_row)
      hipLaunchKernelGGL((gemm_fast_kernel<BF16, BM, BIGK, true, true, GATE>), grid, dim3(F_THREADS), 0, st, g);
    else
      hipLaunchKernelGGL((gemm_fast_kernel<BF16, BM, BIGK, true, false, GATE>), grid, dim3(F_THREADS), 0, st, g);
  } else {
    if (b_row)
      hipLaunchKernelGGL((gemm_fast_kernel<BF16, BM, BIGK, false, true, GATE>), grid, dim3(F_THREADS), 0, st, g);
    else
      hipLaunchKernelGGL((gemm_fast_kernel<BF16, BM, BIGK, false, false, GATE>), grid, dim3(F_THREADS), 0, st, g);
  }
}

template <bool BF16, int BM, bool BIGK>
static void launch_fast2(const kantts_gemm_args& g, bool a_row, bool b_row, dim3 grid, hipStream_t st) {
  bool gate = false;
  for (int s = 0; s < g.nseg; ++s) gate = gate || (g.seg[s].a_gate != nullptr);
  if (gate)
    launch_fast3<BF16, BM, BIGK, true>(g, a_row, b_row, grid, st);
  else
    launch_fast3<BF16, BM, BIGK, false>(g, a_row, b_row, grid, st);
}

template <bool BF16, int BM>
static void launch_fast(const kantts_gemm_args& g, bool a_row, bool b_row, dim3 grid, hipStream_t st) {
  long long ktot = 0;
  int kmin = 1 << 30;
  for (int s = 0; s < g.nseg; ++s) {
    ktot += (long long)g.seg[s].klen * g.seg[s].ntaps;
    kmin = g.seg[s].klen < kmin ? g.seg[s].klen : kmin;
  }
  static const char* force_bk = getenv("KANTTS_GEMM_BIGK");
  // a reduction tile never spans two taps / segments: deep tiles only pay when every run is at least one tile
  bool big = (ktot >= 512) && (kmin >= (BF16 ? 128 : 64));
  if (force_bk) big = (force_bk[0] == '1');
  if (big)
    launch_fast2<BF16, BM, true>(g, a_row, b_row, grid, st);
  else
    launch_fast2<BF16, BM, false>(g, a_row, b_row, grid, st);
}

// Returns 1 when the launch was taken by a fast kernel, 0 when the descriptor does not qualify.
int kantts_gemm_try_fast(const kantts_gemm_args& g_in, hipStream_t st) {
  if (g_in.precision > 1) return 0;
#ifdef F_DEBUG
  kantts_gemm_args g = g_in;
  static const char* dbg_env = getenv("KANTTS_GEMM_DBG");
  const int dbg_mask = dbg_env ? atoi(dbg_env) : 0;
#else
  const kantts_gemm_args& g = g_in;
#endif
  const int am = g.seg[0].a_mode, bm = g.seg[0].b_mode;
  if (am < 2 || bm < 2) return 0;
  for (int s = 0; s < g.nseg; ++s) {
    const kantts_gemm_seg& sg = g.seg[s];
    if (sg.a_mode != am || sg.b_mode != bm) return 0;
    if (sg.a_tok_axis == 1 && am != 2) return 0;
    if (sg.a_tok_axis == 2 && am != 3) return 0;
    if (sg.b_tok_axis == 2 && bm != 3) return 0;
    if (sg.b_tok_axis == 1) return 0;
    if (sg.a_drop_p > 0.f && g.groups > 1) return 0;  // dropout counters are group-relative here
    // plain token maps only (see f_shift_token)
    if (sg.a_inner > 1 || sg.a_mul > 1 || sg.a_div > 1 || sg.a_up > 1 || (sg.a_Tsrc > 0 && sg.a_Tsrc != (sg.a_Tq > 0 ? sg.a_Tq : g.T)))
      return 0;
    if (sg.b_inner > 1 || sg.b_mul > 1 || sg.b_div > 1 || sg.b_up > 1 || (sg.b_Tsrc > 0 && sg.b_Tsrc != (sg.b_Tq > 0 ? sg.b_Tq : g.T)))
      return 0;
    // 32-bit byte offsets inside an operand: bound the farthest element either side can touch
    const long long amul = sg.a_mul > 0 ? sg.a_mul : 1, bmul = sg.b_mul > 0 ? sg.b_mul : 1;
    const long long a_rows = (long long)g.M * (sg.a_tok_axis == 1 ? amul : 1) + 8;
    const long long a_ks_n = (long long)sg.klen * (sg.a_tok_axis == 2 ? amul : 1) + 8;
    const long long b_ks_n = (long long)sg.klen * (sg.b_tok_axis == 2 ? bmul : 1) + 8;
    const long long ext_a = a_rows * llabs(sg.a_is) + a_ks_n * llabs(sg.a_ks);
    const long long ext_b = ((long long)g.N + 8) * llabs(sg.b_js) + b_ks_n * llabs(sg.b_ks);
    if (ext_a >= (1ll << 30) || ext_b >= (1ll << 30)) return 0;
    if (sg.a_is < 0 || sg.a_ks < 0 || sg.b_js < 0 || sg.b_ks < 0) return 0;
  }
  const int splitk = g.splitk < 1 ? 1 : g.splitk;
  const int groups = g.groups < 1 ? 1 : g.groups;
  const int ztaps = g.z_taps > 0 ? g.z_taps : 1;
  const long long blocks64 = (long long)kantts_cdiv(g.N, F_BN) * kantts_cdiv(g.M, 64) * splitk * groups * ztaps;
  static const char* force_bm = getenv("KANTTS_GEMM_BM");
  bool small = blocks64 < 512;
  if (force_bm) small = (force_bm[0] == '3');
  const int bmr = small ? 32 : 64;
  dim3 grid(kantts_cdiv(g.N, F_BN), kantts_cdiv(g.M, bmr), splitk * groups * ztaps);
  const bool a_row = (am == 3), b_row = (bm == 3);
  const int prec = g.precision;
#ifdef F_DEBUG
  g.precision |= dbg_mask << 8;
#endif
  if (prec == 1) {
    if (small)
      launch_fast<true, 32>(g, a_row, b_row, grid, st);
    else
      launch_fast<true, 64>(g, a_row, b_row, grid, st);
  } else {
    if (small)
      launch_fast<false, 32>(g, a_row, b_row, grid, st);
    else
      launch_fast<false, 64>(g, a_row, b_row, grid, st);
  }
  return 1;
}
