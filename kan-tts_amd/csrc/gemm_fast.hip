// Lean specialisations of the segmented GEMM for the shapes that carry the SAM-BERT step.
//
// gemm.hip's kernel interprets every feature of the segment descriptor at run time (token maps,
// groups, scalar layouts ...); its instruction footprint (~18k instructions) does not fit the
// instruction cache and it pays integer divisions per staged element.  The kernels here are compiled
// per staging layout (float4 along k / float4 along rows for A and for B), support only the plain
// descriptor (optional token shift, gate, dropout regeneration, kmask, the full epilogue, split-K,
// bias row sums) and hoist all index arithmetic out of the reduction loop: per tile a thread issues
// 2+2 float4 loads at precomputed bases.  The host (kantts/_hip/__init__.py) routes a launch here when
// every segment qualifies; anything else goes to the generic kernel.  Same tile / MFMA structure:
// 256 threads, BMx64x32 tile, 16x16 MFMA fragments, register double buffer.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define F_BN 64
#define F_BK 32
#define F_THREADS 256
#define F_LDF 34
#define F_LDH 40

__device__ __forceinline__ float f_epilogue(const kantts_gemm_args& g, float acc, int i, int j, bool first_slice,
                                            uint64_t seed_off) {
  float v = acc;
  if (first_slice && g.bias) v += g.bias[j];
  if (first_slice && g.bias2) v += g.bias2[j];
  v *= g.alpha;
  if (g.relu) v = fmaxf(v, 0.f);
  if (g.drop_p > 0.f)
    v *= kantts_dropout_scale(g.drop_p, g.drop_seed + seed_off, (uint64_t)i * (uint64_t)g.N + (uint64_t)j);
  if (first_slice && g.res) v += g.res[(long long)i * g.r_is + (long long)j * g.r_js];
  if (g.rowmask && g.rowmask[i]) v = 0.f;
  return v;
}

template <bool BF16, int BM, bool A_ROW, bool B_ROW>
__global__ __launch_bounds__(F_THREADS) void gemm_fast_kernel(const kantts_gemm_args g) {
  __shared__ __attribute__((aligned(16))) float lds_raw[(BM + F_BN) * F_LDF];
  float* Af = lds_raw;
  float* Bf = lds_raw + BM * F_LDF;
  __bf16* Ah = reinterpret_cast<__bf16*>(lds_raw);
  __bf16* Bh = Ah + BM * F_LDH;
  constexpr int NVA = BM / 32;  // float4 vectors of A per thread and tile
  constexpr int NVB = 2;
  constexpr int MREP = BM / 32;
  constexpr int ARG = BM / 4;          // row groups of A (row-vector layout)
  constexpr int AKS = F_THREADS / ARG;  // k step between the vectors of a thread

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int i0 = blockIdx.y * BM;
  const int j0 = blockIdx.x * F_BN;
  const int zslice = blockIdx.z;
  const uint64_t seed_off = g.seed_dev ? *g.seed_dev : 0ull;

  f32x4 acc[MREP][2];
#pragma unroll
  for (int m = 0; m < MREP; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float rowsum = 0.f;
  const bool do_rowsum = (g.a_rowsum != nullptr) && (blockIdx.x == 0);

  // static element ownership
  int a_r[NVA], a_k[NVA], b_r[NVB], b_k[NVB];
#pragma unroll
  for (int v = 0; v < NVA; ++v) {
    if (A_ROW) {
      a_r[v] = (tid % ARG) * 4;
      a_k[v] = tid / ARG + AKS * v;
    } else {
      a_r[v] = (tid >> 3) + 32 * v;
      a_k[v] = (tid & 7) * 4;
    }
  }
#pragma unroll
  for (int v = 0; v < NVB; ++v) {
    if (B_ROW) {
      b_r[v] = (tid & 15) * 4;
      b_k[v] = (tid >> 4) + 16 * v;
    } else {
      b_r[v] = (tid >> 3) + 32 * v;
      b_k[v] = (tid & 7) * 4;
    }
  }

  float4 ra[NVA], rg[NVA], rb[NVB];
  long long ao[NVA];
  bool oka[NVA], okb[NVB];
  int tile_counter = 0;

  for (int sidx = 0; sidx < g.nseg; ++sidx) {
    const kantts_gemm_seg& s = g.seg[sidx];
    const float* __restrict__ ap = s.a;
    const float* __restrict__ gp = s.a_gate;
    const float* __restrict__ bp = s.b;
    const int klen = s.klen;
    for (int tap = 0; tap < s.ntaps; ++tap) {
      const int a_shift = s.a_tok_axis ? s.a_shift0 + tap * s.a_shift_step : 0;
      const int b_shift = s.b_tok_axis ? s.b_shift0 + tap * s.b_shift_step : 0;
      // ---- per (segment, tap) bases (everything that does not depend on k0)
      long long a_base[NVA], b_base[NVB];
      bool a_rok[NVA], b_rok[NVB];
#pragma unroll
      for (int v = 0; v < NVA; ++v) {
        const int i = i0 + a_r[v];
        bool ok = i < g.M;
        long long ii = i;
        if (!A_ROW && s.a_tok_axis == 1 && a_shift != 0) {
          const int t = i % g.T + a_shift;
          ok = ok && t >= 0 && t < g.T;
          ii = i + a_shift;
        }
        a_rok[v] = ok;
        a_base[v] = A_ROW ? (long long)i : (ii * s.a_is + a_k[v]);
      }
#pragma unroll
      for (int v = 0; v < NVB; ++v) {
        const int j = j0 + b_r[v];
        b_rok[v] = j < g.N;
        b_base[v] = (B_ROW ? (long long)j : ((long long)j * s.b_js + b_k[v])) + (long long)tap * s.b_tap;
      }

      auto fetch = [&](int k0) {
#pragma unroll
        for (int v = 0; v < NVA; ++v) {
          bool ok = a_rok[v];
          long long off;
          if (A_ROW) {
            const int kk = k0 + a_k[v];
            ok = ok && kk < klen;
            long long kq = kk;
            if (s.a_tok_axis == 2 && a_shift != 0) {
              const int t = kk % g.T + a_shift;
              ok = ok && t >= 0 && t < g.T;
              kq = kk + a_shift;
            }
            if (g.kmask && ok) ok = g.kmask[kk] == 0;
            off = kq * s.a_ks + a_base[v];
          } else {
            ok = ok && (k0 + a_k[v]) < klen;
            off = a_base[v] + k0;
          }
          off = ok ? off : 0;
          oka[v] = ok;
          ao[v] = off;
          ra[v] = *reinterpret_cast<const float4*>(ap + off);
          if (gp) rg[v] = *reinterpret_cast<const float4*>(gp + off);
        }
#pragma unroll
        for (int v = 0; v < NVB; ++v) {
          bool ok = b_rok[v];
          long long off;
          if (B_ROW) {
            const int kk = k0 + b_k[v];
            ok = ok && kk < klen;
            long long kq = kk;
            if (s.b_tok_axis == 2 && b_shift != 0) {
              const int t = kk % g.T + b_shift;
              ok = ok && t >= 0 && t < g.T;
              kq = kk + b_shift;
            }
            off = kq * s.b_ks + b_base[v];
          } else {
            ok = ok && (k0 + b_k[v]) < klen;
            off = b_base[v] + k0;
          }
          okb[v] = ok;
          rb[v] = *reinterpret_cast<const float4*>(bp + (ok ? off : 0));
        }
      };

      auto commit = [&]() {
#pragma unroll
        for (int v = 0; v < NVA; ++v) {
          float x[4] = {ra[v].x, ra[v].y, ra[v].z, ra[v].w};
          const float gt[4] = {rg[v].x, rg[v].y, rg[v].z, rg[v].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float val = oka[v] ? x[e] : 0.f;
            if (gp && !(gt[e] > 0.f)) val *= s.a_gate_slope;
            if (s.a_drop_p > 0.f) val *= kantts_dropout_scale(s.a_drop_p, s.a_drop_seed + seed_off, (uint64_t)(ao[v] + e));
            const int r = A_ROW ? a_r[v] + e : a_r[v];
            const int k = A_ROW ? a_k[v] : a_k[v] + e;
            if (BF16)
              Ah[r * F_LDH + k] = (__bf16)val;
            else
              Af[r * F_LDF + k] = val;
          }
        }
#pragma unroll
        for (int v = 0; v < NVB; ++v) {
          const float x[4] = {rb[v].x, rb[v].y, rb[v].z, rb[v].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float val = okb[v] ? x[e] : 0.f;
            const int r = B_ROW ? b_r[v] + e : b_r[v];
            const int k = B_ROW ? b_k[v] : b_k[v] + e;
            if (BF16)
              Bh[r * F_LDH + k] = (__bf16)val;
            else
              Bf[r * F_LDF + k] = val;
          }
        }
      };

      // ---- software-pipelined reduction over this (segment, tap): tiles owned by this split-K slice
      const int ntile = (klen + F_BK - 1) / F_BK;
      int t = 0;
      auto next_owned = [&](int from) {
        int q = from;
        while (q < ntile && ((tile_counter + q) % g.splitk) != zslice) ++q;
        return q;
      };
      t = next_owned(0);
      if (t < ntile) fetch(t * F_BK);
      while (t < ntile) {
        commit();
        __syncthreads();
        const int tn = next_owned(t + 1);
        if (tn < ntile) fetch(tn * F_BK);
        if (do_rowsum && sidx == 0 && tid < BM) {
          float q = 0.f;
          if (BF16) {
            for (int k = 0; k < F_BK; ++k) q += (float)Ah[tid * F_LDH + k];
          } else {
            for (int k = 0; k < F_BK; ++k) q += Af[tid * F_LDF + k];
          }
          rowsum += q;
        }
        if (BF16) {
          bf16x8 af[MREP], bfr[2];
#pragma unroll
          for (int m = 0; m < MREP; ++m)
            af[m] = *reinterpret_cast<const bf16x8*>(
                &Ah[(wr * (BM / 2) + m * 16 + (lane & 15)) * F_LDH + (lane >> 4) * 8]);
#pragma unroll
          for (int n = 0; n < 2; ++n)
            bfr[n] = *reinterpret_cast<const bf16x8*>(&Bh[(wc * 32 + n * 16 + (lane & 15)) * F_LDH + (lane >> 4) * 8]);
#pragma unroll
          for (int m = 0; m < MREP; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bfr[n], acc[m][n], 0, 0, 0);
        } else {
#pragma unroll
          for (int ks = 0; ks < F_BK / 4; ++ks) {
            float af[MREP], bfr[2];
#pragma unroll
            for (int m = 0; m < MREP; ++m)
              af[m] = Af[(wr * (BM / 2) + m * 16 + (lane & 15)) * F_LDF + ks * 4 + (lane >> 4)];
#pragma unroll
            for (int n = 0; n < 2; ++n) bfr[n] = Bf[(wc * 32 + n * 16 + (lane & 15)) * F_LDF + ks * 4 + (lane >> 4)];
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

  if (do_rowsum && tid < BM && (i0 + tid) < g.M) atomicAdd(&g.a_rowsum[i0 + tid], rowsum);

  const bool first_slice = (zslice == 0);
#pragma unroll
  for (int m = 0; m < MREP; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + wr * (BM / 2) + m * 16 + (lane >> 4) * 4 + r;
        const int j = j0 + wc * 32 + n * 16 + (lane & 15);
        if (i < g.M && j < g.N) {
          const float v = f_epilogue(g, acc[m][n][r], i, j, first_slice, seed_off);
          float* dst = &g.c[(long long)i * g.c_is + (long long)j * g.c_js];
          if (g.accumulate)
            atomicAdd(dst, v);
          else
            *dst = v;
        }
      }
}

template <bool BF16, int BM>
static void launch_fast(const kantts_gemm_args& g, bool a_row, bool b_row, dim3 grid, hipStream_t st) {
  if (a_row) {
    if (b_row)
      hipLaunchKernelGGL((gemm_fast_kernel<BF16, BM, true, true>), grid, dim3(F_THREADS), 0, st, g);
    else
      hipLaunchKernelGGL((gemm_fast_kernel<BF16, BM, true, false>), grid, dim3(F_THREADS), 0, st, g);
  } else {
    if (b_row)
      hipLaunchKernelGGL((gemm_fast_kernel<BF16, BM, false, true>), grid, dim3(F_THREADS), 0, st, g);
    else
      hipLaunchKernelGGL((gemm_fast_kernel<BF16, BM, false, false>), grid, dim3(F_THREADS), 0, st, g);
  }
}

// Returns 1 when the launch was taken by a fast kernel, 0 when the descriptor does not qualify.
int kantts_gemm_try_fast(const kantts_gemm_args& g, hipStream_t st) {
  if (g.precision > 1 || g.groups > 1 || g.z_taps > 0 || g.out_act || g.gate) return 0;
  const int am = g.seg[0].a_mode, bm = g.seg[0].b_mode;
  if (am < 2 || bm < 2) return 0;
  for (int s = 0; s < g.nseg; ++s) {
    const kantts_gemm_seg& sg = g.seg[s];
    if (sg.a_mode != am || sg.b_mode != bm) return 0;
    if (sg.a_inner || sg.a_Tq || sg.a_Tsrc || sg.a_mul || sg.a_div || sg.a_up) return 0;
    if (sg.b_inner || sg.b_Tq || sg.b_Tsrc || sg.b_mul || sg.b_div || sg.b_up) return 0;
    if (sg.a_act || sg.b_act) return 0;
    if (sg.a_tok_axis == 1 && am != 2) return 0;
    if (sg.a_tok_axis == 2 && am != 3) return 0;
    if (sg.b_tok_axis == 2 && bm != 3) return 0;
    if (sg.b_tok_axis == 1) return 0;
  }
  const int splitk = g.splitk < 1 ? 1 : g.splitk;
  const long long blocks64 = (long long)kantts_cdiv(g.N, F_BN) * kantts_cdiv(g.M, 64) * splitk;
  const bool small = blocks64 < 512;
  const int bmr = small ? 32 : 64;
  dim3 grid(kantts_cdiv(g.N, F_BN), kantts_cdiv(g.M, bmr), splitk);
  const bool a_row = (am == 3), b_row = (bm == 3);
  if (g.precision == 1) {
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
