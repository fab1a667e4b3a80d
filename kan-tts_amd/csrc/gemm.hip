// Segmented GEMM on CDNA4 matrix cores -- see include/kantts_hip.h for the contract.
//
// Block = 256 threads (4 waves, 2x2), output tile 64x64, reduction tile BK = 32.
// Each wave owns a 32x32 sub-tile = 2x2 MFMA 16x16 fragments (f32x4 accumulators).
//   precision 0: v_mfma_f32_16x16x4_f32   (8 k-steps per tile; exact fp32 FMA chain)
//   precision 1: v_mfma_f32_16x16x32_bf16 (1 k-step per tile; fp32 operands are rounded to bf16
//                when they are staged into LDS, accumulation stays fp32)
// LDS images are [row][k] with k contiguous: fp32 rows are padded to 34 words (conflict-free
// ds_read_b32 for the 16x16x4 fragment: bank = 2*row + k), bf16 rows to 40 halfwords (80 B, keeps
// the 16-byte fragment reads aligned).
// The operand loaders are generic (strides, conv taps as token shifts, strided / upsampled /
// period-folded token maps, gating, masks, fused LeakyReLU) so that one kernel serves forward, dgrad
// and wgrad of Linear / Conv1d / ConvTranspose1d / (k,1)-Conv2d in channels-last layout.  Lanes walk
// whichever operand dimension has unit stride so global loads coalesce.
#include <stdlib.h>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define G_BN 64
#define G_BK 32
#define G_THREADS 256
#define G_LDF 34 /* fp32 LDS row stride (words)     */
#define G_LDH 40 /* bf16 LDS row stride (halfwords) */

struct TokMap {
  int inner, Tq, Tsrc, mul, div, up;
};

__device__ __forceinline__ TokMap make_map(int inner, int Tq, int Tsrc, int mul, int div, int up, int T) {
  TokMap m;
  m.inner = inner > 0 ? inner : 1;
  m.Tq = Tq > 0 ? Tq : T;
  m.Tsrc = Tsrc > 0 ? Tsrc : T;
  m.mul = mul > 0 ? mul : 1;
  m.div = div > 0 ? div : 1;
  m.up = up > 0 ? up : 1;
  return m;
}

// token of the (B, Tq, inner) domain -> source row; false when the tap falls outside the sequence
__device__ __forceinline__ bool map_token(int tok, int shift, const TokMap& m, long long& row) {
  const int pi = tok % m.inner;
  const int bq = tok / m.inner;
  const int q = bq % m.Tq;
  const int b = bq / m.Tq;
  int t = q * m.mul + shift;
  if (t < 0) return false;
  if (m.div > 1) {
    if (t % m.div) return false;
    t /= m.div;
  }
  if (t >= m.Tsrc * m.up) return false;
  t /= m.up;
  row = ((long long)b * m.Tsrc + t) * m.inner + pi;
  return true;
}

__device__ __forceinline__ float g_load_a(const kantts_gemm_seg& s, const kantts_gemm_args& g, int i, int kk,
                                          int shift, const TokMap& m, long long goff) {
  if (i >= g.M || kk >= s.klen) return 0.f;
  long long ii = i, kq = kk;
  if (s.a_tok_axis == 1) {
    if (!map_token(i, shift, m, ii)) return 0.f;
  } else if (s.a_tok_axis == 2) {
    if (!map_token(kk, shift, m, kq)) return 0.f;
  }
  if (g.kmask && g.kmask[kk]) return 0.f;
  long long off = ii * s.a_is + kq * s.a_ks + goff;
  float v = s.a[off];
  if (s.a_act) v = v > 0.f ? v : v * s.a_slope;
  if (s.a_gate && !(s.a_gate[off] > 0.f)) v *= s.a_gate_slope;
  if (s.a_drop_p > 0.f)
    v *= kantts_dropout_scale(s.a_drop_p, s.a_drop_seed + (g.seed_dev ? *g.seed_dev : 0ull), (uint64_t)off);
  return v;
}

__device__ __forceinline__ float g_load_b(const kantts_gemm_seg& s, const kantts_gemm_args& g, int j, int kk,
                                          int shift, int tap, const TokMap& m, long long goff) {
  if (j >= g.N || kk >= s.klen) return 0.f;
  long long kq = kk;
  if (s.b_tok_axis == 2) {
    if (!map_token(kk, shift, m, kq)) return 0.f;
  }
  float v = s.b[(long long)j * s.b_js + kq * s.b_ks + (long long)tap * s.b_tap + goff];
  if (s.b_act) v = v > 0.f ? v : v * s.b_slope;
  return v;
}

__device__ __forceinline__ float g_epilogue(const kantts_gemm_args& g, float acc, int i, int j, bool first_slice,
                                            int grp) {
  float v = acc;
  if (first_slice && g.bias) v += g.bias[j + grp * g.bias_gs];
  if (first_slice && g.bias2) v += g.bias2[j + grp * g.bias_gs];
  v *= g.alpha;
  if (g.relu) v = fmaxf(v, 0.f);
  if (g.out_act) v = v > 0.f ? v : v * g.out_slope;
  if (g.drop_p > 0.f)
    v *= kantts_dropout_scale(g.drop_p, g.drop_seed + (g.seed_dev ? *g.seed_dev : 0ull),
                              (uint64_t)i * (uint64_t)g.N + (uint64_t)j);
  if (first_slice && g.res) v += g.res[(long long)i * g.r_is + (long long)j * g.r_js + grp * g.r_gs];
  if (g.gate) {
    const float gv = g.gate[(long long)i * g.c_is + (long long)j * g.c_js + grp * g.c_gs];
    v *= (gv > 0.f) ? 1.f : g.gate_slope;
  }
  if (g.rowmask && g.rowmask[i]) v = 0.f;
  return v;
}

// ---------------------------------------------------------------------------------------------
// Operand staging.  Every thread owns 8 elements of the A tile and 8 of the B tile per reduction
// tile, in one of four layouts chosen per segment by the host (seg.a_mode / seg.b_mode):
//   0  scalar, lanes along k            (generic fallback, e.g. conv weights with tap stride)
//   1  scalar, lanes along rows         (unit row stride, rows not a multiple of 4)
//   2  float4 along k                   (unit k stride: activations / weights / dY of forward & dgrad)
//   3  float4 along rows                (unit row stride: both operands of the weight gradient)
// Addresses and validity are pure ALU; loads are issued unconditionally at a clamped address so
// that all 16 loads of a tile are in flight together (a branch per load would serialise them on
// s_waitcnt), and they are issued for tile t+1 before the MFMAs of tile t (register double buffer).
struct TileCur {
  int sidx, tap, k0;
};

__device__ __forceinline__ bool cur_step(TileCur& c, const kantts_gemm_args& g, int ztap) {
  for (;;) {
    if (c.sidx >= g.nseg) return false;
    const kantts_gemm_seg& s = g.seg[c.sidx];
    if (c.tap < s.ntaps && (ztap < 0 || c.tap == ztap)) {
      c.k0 += G_BK;
      if (c.k0 < s.klen) return true;
    }
    c.k0 = -G_BK;
    c.tap++;
    if (c.tap >= s.ntaps) {
      c.tap = 0;
      c.sidx++;
    }
  }
}

__device__ __forceinline__ bool addr_a(const kantts_gemm_seg& s, const kantts_gemm_args& g, int i, int kk, int shift,
                                       const TokMap& m, long long goff, long long& off) {
  long long ii = i, kq = kk;
  bool ok = (i < g.M) && (kk < s.klen);
  if (s.a_tok_axis == 1) {
    ok = ok && map_token(i, shift, m, ii);
  } else if (s.a_tok_axis == 2) {
    ok = ok && map_token(kk, shift, m, kq);
  }
  if (g.kmask && ok) ok = (g.kmask[kk] == 0);
  off = ok ? (ii * s.a_is + kq * s.a_ks + goff) : 0;
  return ok;
}

__device__ __forceinline__ bool addr_b(const kantts_gemm_seg& s, const kantts_gemm_args& g, int j, int kk, int shift,
                                       int tap, const TokMap& m, long long goff, long long& off) {
  long long kq = kk;
  bool ok = (j < g.N) && (kk < s.klen);
  if (s.b_tok_axis == 2) ok = ok && map_token(kk, shift, m, kq);
  off = ok ? ((long long)j * s.b_js + kq * s.b_ks + (long long)tap * s.b_tap + goff) : 0;
  return ok;
}

template <int BM>
__device__ __forceinline__ void elem_rk(int mode, int tid, int e, int rows, int& r, int& k) {
  // rows = BM for A, 64 for B;  8 elements per thread when rows == 64, 4 when rows == 32
  if (mode == 0) {            // lanes along k
    k = tid & 31;
    r = (tid >> 5) + 8 * e;
  } else if (mode == 1) {     // lanes along rows
    r = tid % rows;
    k = tid / rows + (G_THREADS / rows) * e;
  } else if (mode == 2) {     // float4 along k: 8 lanes per row
    r = (tid >> 3) + 32 * (e >> 2);
    k = ((tid & 7) << 2) + (e & 3);
  } else {                    // float4 along rows
    const int rg = rows >> 2;
    r = ((tid % rg) << 2) + (e & 3);
    k = tid / rg + (G_THREADS / rg) * (e >> 2);
  }
}

template <bool BF16, int BM>
__global__ __launch_bounds__(G_THREADS) void gemm_seg_mfma_kernel(const kantts_gemm_args g) {
  __shared__ __attribute__((aligned(16))) float lds_raw[(BM + G_BN) * G_LDF];
  float* Af = lds_raw;
  float* Bf = lds_raw + BM * G_LDF;
  __bf16* Ah = reinterpret_cast<__bf16*>(lds_raw);
  __bf16* Bh = Ah + BM * G_LDH;
  constexpr int EA = BM / 8;   // staged A elements per thread (8 or 4)
  constexpr int MREP = BM / 32;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int i0 = blockIdx.y * BM;
  const int j0 = blockIdx.x * G_BN;
  const int zper = g.groups * g.splitk;
  const int ztap = g.z_taps > 0 ? (int)(blockIdx.z / zper) : -1;
  const int zrem = blockIdx.z % zper;
  const int grp = zrem / g.splitk;
  const int zslice = zrem % g.splitk;
  const uint64_t seed_off = g.seed_dev ? *g.seed_dev : 0ull;

  f32x4 acc[MREP][2];
#pragma unroll
  for (int m = 0; m < MREP; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float rowsum = 0.f;
  const bool do_rowsum = (g.a_rowsum != nullptr) && (blockIdx.x == 0) && (ztap <= 0);

  float ra[EA], rga[EA], rb[8];
  unsigned oka = 0, okb = 0;
  int tile_counter = 0;

  // fetch(): issue the global loads of tile `c` into registers (no dependent use here)
  auto fetch = [&](const TileCur& c) {
    const kantts_gemm_seg& s = g.seg[c.sidx];
    const TokMap am = make_map(s.a_inner, s.a_Tq, s.a_Tsrc, s.a_mul, s.a_div, s.a_up, g.T);
    const TokMap bm = make_map(s.b_inner, s.b_Tq, s.b_Tsrc, s.b_mul, s.b_div, s.b_up, g.T);
    const long long a_goff = (long long)grp * g.a_gs, b_goff = (long long)grp * g.b_gs;
    const int a_shift = s.a_tok_axis ? s.a_shift0 + c.tap * s.a_shift_step : 0;
    const int b_shift = s.b_tok_axis ? s.b_shift0 + c.tap * s.b_shift_step : 0;
    oka = 0;
    okb = 0;
    if (s.a_mode >= 2) {
#pragma unroll
      for (int v = 0; v < EA / 4; ++v) {
        int r, k;
        elem_rk<BM>(s.a_mode, tid, 4 * v, BM, r, k);
        long long off;
        const bool ok = addr_a(s, g, i0 + r, c.k0 + k, a_shift, am, a_goff, off);
        const float4 t = *reinterpret_cast<const float4*>(s.a + off);
        ra[4 * v] = t.x; ra[4 * v + 1] = t.y; ra[4 * v + 2] = t.z; ra[4 * v + 3] = t.w;
        if (s.a_gate) {
          const float4 u = *reinterpret_cast<const float4*>(s.a_gate + off);
          rga[4 * v] = u.x; rga[4 * v + 1] = u.y; rga[4 * v + 2] = u.z; rga[4 * v + 3] = u.w;
        }
        if (ok) oka |= 0xFu << (4 * v);
      }
    } else {
#pragma unroll
      for (int e = 0; e < EA; ++e) {
        int r, k;
        elem_rk<BM>(s.a_mode, tid, e, BM, r, k);
        long long off;
        const bool ok = addr_a(s, g, i0 + r, c.k0 + k, a_shift, am, a_goff, off);
        ra[e] = s.a[off];
        if (s.a_gate) rga[e] = s.a_gate[off];
        if (ok) oka |= 1u << e;
      }
    }
    if (s.b_mode >= 2) {
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        int r, k;
        elem_rk<BM>(s.b_mode, tid, 4 * v, G_BN, r, k);
        long long off;
        const bool ok = addr_b(s, g, j0 + r, c.k0 + k, b_shift, c.tap, bm, b_goff, off);
        const float4 t = *reinterpret_cast<const float4*>(s.b + off);
        rb[4 * v] = t.x; rb[4 * v + 1] = t.y; rb[4 * v + 2] = t.z; rb[4 * v + 3] = t.w;
        if (ok) okb |= 0xFu << (4 * v);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int r, k;
        elem_rk<BM>(s.b_mode, tid, e, G_BN, r, k);
        long long off;
        const bool ok = addr_b(s, g, j0 + r, c.k0 + k, b_shift, c.tap, bm, b_goff, off);
        rb[e] = s.b[off];
        if (ok) okb |= 1u << e;
      }
    }
  };

  // commit(): masks / activation / gate / dropout, convert and write the staged tile to LDS
  auto commit = [&](const TileCur& c) {
    const kantts_gemm_seg& s = g.seg[c.sidx];
#pragma unroll
    for (int e = 0; e < EA; ++e) {
      int r, k;
      elem_rk<BM>(s.a_mode, tid, e, BM, r, k);
      float v = ((oka >> e) & 1u) ? ra[e] : 0.f;
      if (s.a_act) v = v > 0.f ? v : v * s.a_slope;
      if (s.a_gate && !(rga[e] > 0.f)) v *= s.a_gate_slope;
      if (s.a_drop_p > 0.f) {
        const TokMap am = make_map(s.a_inner, s.a_Tq, s.a_Tsrc, s.a_mul, s.a_div, s.a_up, g.T);
        const int a_shift = s.a_tok_axis ? s.a_shift0 + c.tap * s.a_shift_step : 0;
        long long off;
        int rr = r, kk2 = k;
        if (s.a_mode == 2) { rr = r; kk2 = k; }
        addr_a(s, g, i0 + rr, c.k0 + kk2, a_shift, am, (long long)grp * g.a_gs, off);
        v *= kantts_dropout_scale(s.a_drop_p, s.a_drop_seed + seed_off, (uint64_t)off);
      }
      if (BF16)
        Ah[r * G_LDH + k] = (__bf16)v;
      else
        Af[r * G_LDF + k] = v;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int r, k;
      elem_rk<BM>(s.b_mode, tid, e, G_BN, r, k);
      float v = ((okb >> e) & 1u) ? rb[e] : 0.f;
      if (s.b_act) v = v > 0.f ? v : v * s.b_slope;
      if (BF16)
        Bh[r * G_LDH + k] = (__bf16)v;
      else
        Bf[r * G_LDF + k] = v;
    }
  };

  auto next_owned = [&](TileCur& c) -> bool {
    for (;;) {
      if (!cur_step(c, g, ztap)) return false;
      const bool mine = (tile_counter % g.splitk) == zslice;
      ++tile_counter;
      if (mine) return true;
    }
  };

  TileCur cur = {0, 0, -G_BK};
  bool has = next_owned(cur);
  if (has) fetch(cur);
  while (has) {
    commit(cur);
    __syncthreads();
    const bool count_rowsum = do_rowsum && cur.sidx == 0;
    TileCur nxt = cur;
    const bool has_next = next_owned(nxt);
    if (has_next) fetch(nxt);  // in flight during the MFMAs below
    if (count_rowsum && tid < BM) {
      float t = 0.f;
      if (BF16) {
        for (int k = 0; k < G_BK; ++k) t += (float)Ah[tid * G_LDH + k];
      } else {
        for (int k = 0; k < G_BK; ++k) t += Af[tid * G_LDF + k];
      }
      rowsum += t;
    }
    if (BF16) {
      bf16x8 af[MREP], bfr[2];
#pragma unroll
      for (int m = 0; m < MREP; ++m)
        af[m] = *reinterpret_cast<const bf16x8*>(
            &Ah[(wr * (BM / 2) + m * 16 + (lane & 15)) * G_LDH + (lane >> 4) * 8]);
#pragma unroll
      for (int n = 0; n < 2; ++n)
        bfr[n] = *reinterpret_cast<const bf16x8*>(&Bh[(wc * 32 + n * 16 + (lane & 15)) * G_LDH + (lane >> 4) * 8]);
#pragma unroll
      for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bfr[n], acc[m][n], 0, 0, 0);
    } else {
#pragma unroll
      for (int ks = 0; ks < G_BK / 4; ++ks) {
        float af[MREP], bfr[2];
#pragma unroll
        for (int m = 0; m < MREP; ++m)
          af[m] = Af[(wr * (BM / 2) + m * 16 + (lane & 15)) * G_LDF + ks * 4 + (lane >> 4)];
#pragma unroll
        for (int n = 0; n < 2; ++n) bfr[n] = Bf[(wc * 32 + n * 16 + (lane & 15)) * G_LDF + ks * 4 + (lane >> 4)];
#pragma unroll
        for (int m = 0; m < MREP; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bfr[n], acc[m][n], 0, 0, 0);
      }
    }
    __syncthreads();
    cur = nxt;
    has = has_next;
  }

  if (do_rowsum && tid < BM && (i0 + tid) < g.M) atomicAdd(&g.a_rowsum[i0 + tid + grp * g.bias_gs], rowsum);

  // ---- epilogue: C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
  const bool first_slice = (zslice == 0);
#pragma unroll
  for (int m = 0; m < MREP; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int i = i0 + wr * (BM / 2) + m * 16 + (lane >> 4) * 4 + r;
        int j = j0 + wc * 32 + n * 16 + (lane & 15);
        if (i < g.M && j < g.N) {
          float v = g_epilogue(g, acc[m][n][r], i, j, first_slice, grp);
          float* dst = &g.c[(long long)i * g.c_is + (long long)j * g.c_js + (long long)grp * g.c_gs +
                            (ztap > 0 ? (long long)ztap * g.c_tap : 0)];
          if (g.accumulate)
            atomicAdd(dst, v);
          else
            *dst = v;
        }
      }
}

// Scalar fp32 reference of the same contract (debug / cross-check of the MFMA fragment maps).
__global__ void gemm_seg_ref_kernel(const kantts_gemm_args g) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int grp = blockIdx.y % g.groups;
  const int ztap = g.z_taps > 0 ? (int)(blockIdx.y / g.groups) : -1;
  if (idx >= (long long)g.M * g.N) return;
  int i = (int)(idx / g.N), j = (int)(idx % g.N);
  float acc = 0.f, rs = 0.f;
  for (int sidx = 0; sidx < g.nseg; ++sidx) {
    const kantts_gemm_seg& s = g.seg[sidx];
    const TokMap am = make_map(s.a_inner, s.a_Tq, s.a_Tsrc, s.a_mul, s.a_div, s.a_up, g.T);
    const TokMap bm = make_map(s.b_inner, s.b_Tq, s.b_Tsrc, s.b_mul, s.b_div, s.b_up, g.T);
    const long long a_goff = (long long)grp * g.a_gs, b_goff = (long long)grp * g.b_gs;
    for (int tap = 0; tap < s.ntaps; ++tap) {
      if (ztap >= 0 && tap != ztap) continue;
      const int a_shift = s.a_tok_axis ? s.a_shift0 + tap * s.a_shift_step : 0;
      const int b_shift = s.b_tok_axis ? s.b_shift0 + tap * s.b_shift_step : 0;
      for (int kk = 0; kk < s.klen; ++kk) {
        float a = g_load_a(s, g, i, kk, a_shift, am, a_goff);
        acc = fmaf(a, g_load_b(s, g, j, kk, b_shift, tap, bm, b_goff), acc);
        if (sidx == 0) rs += a;
      }
    }
  }
  if (g.a_rowsum && j == 0 && ztap <= 0) atomicAdd(&g.a_rowsum[i + grp * g.bias_gs], rs);
  float v = g_epilogue(g, acc, i, j, true, grp);
  float* dst = &g.c[(long long)i * g.c_is + (long long)j * g.c_js + (long long)grp * g.c_gs +
                    (ztap > 0 ? (long long)ztap * g.c_tap : 0)];
  if (g.accumulate)
    atomicAdd(dst, v);
  else
    *dst = v;
}

int kantts_gemm_try_fast(const kantts_gemm_args& g, hipStream_t st);  // gemm_fast.hip

extern "C" int kantts_gemm_seg_launch(const kantts_gemm_args* a, void* stream) {
  if (!a || a->nseg < 1 || a->nseg > KANTTS_GEMM_MAX_SEG || a->M < 0 || a->N < 0 || !a->c) return KANTTS_E_BADARG;
  if (a->M == 0 || a->N == 0) return KANTTS_OK;
  int splitk = a->splitk < 1 ? 1 : a->splitk;
  int groups = a->groups < 1 ? 1 : a->groups;
  if (splitk > 1 && (!a->accumulate || a->relu || a->out_act || a->gate || a->drop_p > 0.f)) return KANTTS_E_BADARG;
  for (int s = 0; s < a->nseg; ++s) {
    const kantts_gemm_seg& sg = a->seg[s];
    if (!sg.a || !sg.b || sg.klen < 0 || sg.ntaps < 1) return KANTTS_E_BADARG;
    if ((sg.a_tok_axis && sg.a_Tq <= 0 && a->T <= 0) || (sg.b_tok_axis && sg.b_Tq <= 0 && a->T <= 0))
      return KANTTS_E_BADARG;
  }
  int ztaps = a->z_taps > 0 ? a->z_taps : 1;
  if (a->z_taps > 0 && (a->nseg != 1 || a->z_taps != a->seg[0].ntaps)) return KANTTS_E_BADARG;
  if ((long long)groups * splitk * ztaps > 65535) return KANTTS_E_BADARG;
  kantts_gemm_args g = *a;
  g.splitk = splitk;
  g.groups = groups;
  hipStream_t st = (hipStream_t)stream;
  static const bool no_fast = getenv("KANTTS_GEMM_NOFAST") != nullptr;
  if (!no_fast && g.precision != 2 && kantts_gemm_try_fast(g, st)) {
    KANTTS_CHECK_LAUNCH();
  }
  if (g.precision == 2) {
    g.splitk = 1;
    long long total = (long long)g.M * g.N;
    hipLaunchKernelGGL(gemm_seg_ref_kernel, dim3(kantts_cdiv(total, 256), groups * ztaps), dim3(256), 0, st, g);
  } else {
    // 32-row tiles when 64-row tiles would leave most of the 256 CUs idle
    const long long blocks64 = (long long)kantts_cdiv(g.N, G_BN) * kantts_cdiv(g.M, 64) * groups * splitk * ztaps;
    const bool small = blocks64 < 512;
    const int bm = small ? 32 : 64;
    dim3 grid(kantts_cdiv(g.N, G_BN), kantts_cdiv(g.M, bm), groups * splitk * ztaps);
    if (g.precision == 1) {
      if (small)
        hipLaunchKernelGGL((gemm_seg_mfma_kernel<true, 32>), grid, dim3(G_THREADS), 0, st, g);
      else
        hipLaunchKernelGGL((gemm_seg_mfma_kernel<true, 64>), grid, dim3(G_THREADS), 0, st, g);
    } else if (g.precision == 0) {
      if (small)
        hipLaunchKernelGGL((gemm_seg_mfma_kernel<false, 32>), grid, dim3(G_THREADS), 0, st, g);
      else
        hipLaunchKernelGGL((gemm_seg_mfma_kernel<false, 64>), grid, dim3(G_THREADS), 0, st, g);
    } else {
      return KANTTS_E_BADARG;
    }
  }
  KANTTS_CHECK_LAUNCH();
}

extern "C" int kantts_abi_version(void) { return 1; }
extern "C" const char* kantts_target_arch(void) { return "gfx950"; }
