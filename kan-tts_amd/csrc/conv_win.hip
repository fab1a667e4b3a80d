// Channels-last 1-D convolution with an LDS-resident input window (im2col-free).
//
// Replaces the per-tap token-shifted GEMM of gemm_fast.hip for the HiFi-GAN convolutions: there every
// tap of a K-tap kernel re-reads its 64 x BK activation tile from L2/HBM (K = 3 ... 41 reads of every
// input element).  Here a workgroup loads the window of input tokens its BQ outputs can touch
//        W = (BQ-1)*in_mul + (off_max - off_min) + 1   tokens x 32 channels
// ONCE per 32-channel chunk into LDS (LeakyReLU / LeakyReLU' gate applied and rounded to bf16 on the way
// in) and walks the K taps over it: tap k multiplies the window rows  m*in_mul + off_k  with the tap's
// 32 x BN weight tile on MFMA.  Only the weight tile (BN x 32, L2 resident) is fetched per tap, register
// double-buffered against the MFMAs of the previous tap.
//
// One kernel covers
//   forward   y[b,q,n]  = post( bias[n] + sum_k sum_c pre(x[b, q*stride + k*dil - pad, c]) w[k][n][c] )
//   dgrad     dx[b,t,c] = post( sum_k sum_n pre(dy[b, (t + pad - k*dil)/stride, n]) w'[k][c][n] )
// through the token rule   src = m*in_mul + (in_add + phase + k*in_kstep) / in_div   (exact division only),
// dst = m*phases + phase: a strided convolution's input gradient is computed per output phase, so its
// rows are dense (no zero-stuffed rows reach the MFMAs).  Strided windows are stored de-interleaved by
// (token mod in_mul) so the 16 rows of an MFMA fragment are consecutive LDS rows for every stride.
//
// Reference: Conv1d / CausalConv1d of kantts/models/hifigan/layers.py:15-91 as used by the generator's
// residual blocks (layers.py:168-288), conv_pre/conv_post (hifigan.py:40-66,118-160) and the scale
// discriminators (hifigan.py:332-407).
#include <stdlib.h>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define CW_THREADS 256
#define CW_CK 32        // channels per chunk = one bf16 MFMA k-step
#define CW_MAXTAPS 64
#define CW_HDR (3 * CW_MAXTAPS * 4 + 16)  // tap table ahead of the tiles (multiple of 16 bytes)

__device__ __forceinline__ int cw_floordiv(int a, int b) {
  int q = a / b;
  if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
  return q;
}

template <bool BF16>
__device__ __forceinline__ void cw_store4(void* base, int idx, float v0, float v1, float v2, float v3) {
  if (BF16) {
    bf16x4 p = {(__bf16)v0, (__bf16)v1, (__bf16)v2, (__bf16)v3};
    *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(base) + idx) = p;
  } else {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + idx) = make_float4(v0, v1, v2, v3);
  }
}

// WM waves along the output tokens (4 or 2), the other 4/WM along the output channels;
// every wave owns MREP x 4 accumulator fragments (16*MREP tokens x 64 channels).
template <bool BF16, int WM, int MREP>
__global__ __launch_bounds__(CW_THREADS) void conv_win_kernel(kantts_conv_args g) {
  constexpr int WN = 4 / WM;
  constexpr int BQ = WM * MREP * 16;
  constexpr int BN = WN * 64;
  // bf16 rows are 96 B apart (32 mod 64: conflict-free ds_read_b128 over 16 consecutive rows); fp32 rows 144 B
  constexpr int LDW = BF16 ? 48 : 36;
  constexpr int ESZ = BF16 ? 2 : 4;
  constexpr int NBV = BN * 8 / CW_THREADS;  // float4 per thread of one weight tile
  // ALL LDS is dynamic: a static __shared__ array ahead of the dynamic region would shift its base off
  // 16 bytes, and a misaligned ds_read_b128 is replayed at 64 cycles per wave-instruction
  extern __shared__ __attribute__((aligned(16))) unsigned char cw_lds_raw[];
  int* s_off = reinterpret_cast<int*>(cw_lds_raw);
  int* s_tap = s_off + CW_MAXTAPS;
  int* s_row = s_tap + CW_MAXTAPS;  // window row (before the fold multiplier) the tap starts at
  int* s_nv_p = s_row + CW_MAXTAPS;
  unsigned char* cw_lds = cw_lds_raw + CW_HDR;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
#ifdef CW_DEBUG  // ablation mask (experiment builds only): 1 no window loads, 2 no output stores, 4 no weight loads, 8 no MFMA
  const int dbg = g.up >> 16;
  g.up &= 0xffff;
#else
  constexpr int dbg = 0;
#endif
  const int ntpg = (g.NG + BN - 1) / BN;
  // XCD-aware order inside one (batch, phase) slab: workgroup L = y*gx + x runs on XCD L % 8; the gx channel tiles
  // of one token window are dealt to ONE XCD (consecutive virtual ids) so the window crosses the fabric once
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const int gx = gridDim.x, total = gx * gridDim.y;
    if (total >= 64 && gx > 1) {
      const int L = by * gx + bx, k = L & 7, j = L >> 3;
      const int q = total >> 3, r = total & 7;
      const int vid = k * q + (k < r ? k : r) + j;
      by = vid / gx;
      bx = vid - by * gx;
    }
  }
  const int grp = bx / ntpg;
  const int n0 = grp * g.NG + (bx % ntpg) * BN;
  const int n_end = (grp + 1) * g.NG;
  const int phase = blockIdx.z % g.phases;
  const int b = blockIdx.z / g.phases;
  // The folded axis (MPD period p = inner) is part of the ROW axis of a tile: row m' = m*inner + p'.  A period
  // discriminator's deep layers have 10-80 tokens per (batch, p') sequence; one tile per sequence would leave the
  // MFMA rows 16-44 % full, folded tiles are dense.  Global rows (token, p') are contiguous, so the window of a tile
  // is still one contiguous row range.
  const int inner = g.inner;
  const int m0 = by * BQ;  // first row (m' units) of the tile
  const int mrows = (g.Tdst - phase + g.phases - 1) / g.phases;
  const int R = mrows * inner;
  if (m0 >= R) return;
  const int m_lo = m0 / inner, m_hi = min(R - 1, m0 + BQ - 1) / inner;

  if (tid == 0) {
    int nv = 0;
    for (int k = 0; k < g.K; ++k) {
      const int u = g.in_add + phase + k * g.in_kstep;
      const int q = cw_floordiv(u, g.in_div);
      if (q * g.in_div != u) continue;
      s_off[nv] = q;
      s_tap[nv] = k;
      ++nv;
    }
    *s_nv_p = nv;
  }
  __syncthreads();
  const int nv = *s_nv_p;
  int offmin = 0, offmax = 0;
  if (nv > 0) {
    offmin = s_off[0];
    offmax = s_off[0];
    for (int i = 1; i < nv; ++i) {
      offmin = min(offmin, s_off[i]);
      offmax = max(offmax, s_off[i]);
    }
  }
  // up > 1: the source is read through a nearest-neighbour upsampling (token u of the virtual sequence is source
  // token u / up); the window then holds SOURCE tokens and every lane computes its own row (rows repeat).
  const int up = g.up > 1 ? g.up : 1;
  const int dm = (up > 1) ? 1 : g.in_mul;  // de-interleave modulus of the LDS window
  const int lo = (up > 1) ? cw_floordiv(m_lo * g.in_mul + offmin, up) : m_lo * g.in_mul + offmin;  // first source token
  const int W = (up > 1) ? cw_floordiv(m_hi * g.in_mul + offmax, up) - lo + 1
                         : (m_hi - m_lo) * g.in_mul + (offmax - offmin) + 1;
  const int Wp = (W + dm - 1) / dm;

  void* win = cw_lds;
  unsigned char* bt = cw_lds + (size_t)Wp * dm * inner * LDW * ESZ;
  // per-tap window row, computed ONCE: the tap loop used to redo two integer divisions per tap and lane (the PMC run
  // of round 1 showed ~15 VALU + 14 SALU instructions per MFMA -- the kernels were issue-bound on index arithmetic)
  if (tid < nv) {
    const int a = s_off[tid] - offmin;
    s_row[tid] = (a % dm) * Wp + a / dm;
  }
  __syncthreads();
  // tap-invariant part of every fragment's window row: (m - m_lo) * inner + p'
  int fbase[MREP], fm[MREP];
#pragma unroll
  for (int f = 0; f < MREP; ++f) {
    // rows past the end of the sequence set are clamped (computed, never stored)
    const int mp = min(m0 + wm * (MREP * 16) + f * 16 + (lane & 15), R - 1);
    const int m = (inner > 1) ? mp / inner : mp;
    const int pin = (inner > 1) ? mp - m * inner : 0;
    fm[f] = m;
    fbase[f] = (m - m_lo) * inner + pin;
  }

  f32x4 acc[MREP][4];
#pragma unroll
  for (int f = 0; f < MREP; ++f)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const long long in_off = (long long)b * g.Tsrc * inner * g.Cin_tot + (long long)grp * g.CR;
  const float* in_b = g.in + in_off;
  const float* gate_b = g.in_gate ? g.in_gate + in_off : nullptr;
  const int c4 = (tid & 7) * 4;  // channel offset of this thread's float4 inside a chunk
  const int rslot = tid >> 3;    // 32 rows per pass

  float4 wreg[NBV];
  auto fetch_w = [&](int ti, int c0) {
    const int k = s_tap[ti];
#pragma unroll
    for (int v = 0; v < NBV; ++v) {
      const int j = rslot + 32 * v;
      const bool ok = (n0 + j < n_end) && (c0 + c4 < g.CR) && !(dbg & 4);
      const long long o = ok ? (((long long)k * g.Ntot + n0 + j) * g.CR + c0 + c4) : 0;
      float4 x = *reinterpret_cast<const float4*>(g.w + o);
      if (!ok) x = make_float4(0.f, 0.f, 0.f, 0.f);
      wreg[v] = x;
    }
  };
  auto commit_w = [&](int buf) {
    void* dst = bt + (size_t)buf * BN * LDW * ESZ;
#pragma unroll
    for (int v = 0; v < NBV; ++v) {
      const int j = rslot + 32 * v;
      cw_store4<BF16>(dst, j * LDW + c4, wreg[v].x, wreg[v].y, wreg[v].z, wreg[v].w);
    }
  };

  for (int c0 = 0; c0 < g.CR && nv > 0; c0 += CW_CK) {
    __syncthreads();  // every wave is done with the previous chunk's window and weight tiles
    fetch_w(0, c0);
    // ---- stage the window: W tokens x 32 channels, 4 rows in flight per thread
    const bool cok = (c0 + c4) < g.CR;
    const int WR = W * inner;  // window rows = (token, p') pairs, contiguous in global memory
    for (int r0 = rslot; r0 < WR; r0 += 32 * 4) {
      float4 xv[4], gv[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = r0 + 32 * u;
        const int t = lo + ((inner > 1) ? rr / inner : rr);
        ok[u] = cok && rr < WR && t >= 0 && t < g.Tsrc && !(dbg & 1);
        const long long o = ok[u] ? (((long long)lo * inner + rr) * g.Cin_tot + c0 + c4) : 0;
        xv[u] = *reinterpret_cast<const float4*>(in_b + o);
        if (gate_b) gv[u] = *reinterpret_cast<const float4*>(gate_b + o);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = r0 + 32 * u;
        if (rr >= WR) continue;
        // window row: stride-1 windows (dm == 1, most layers) store global row rr at LDS row rr -- no division
        const int rel = (inner > 1 && dm > 1) ? rr / inner : rr;
        const int pin = (inner > 1 && dm > 1) ? rr - rel * inner : 0;
        float v0 = xv[u].x, v1 = xv[u].y, v2 = xv[u].z, v3 = xv[u].w;
        if (!ok[u]) v0 = v1 = v2 = v3 = 0.f;
        if (g.in_act) {
          v0 = v0 > 0.f ? v0 : v0 * g.in_slope;
          v1 = v1 > 0.f ? v1 : v1 * g.in_slope;
          v2 = v2 > 0.f ? v2 : v2 * g.in_slope;
          v3 = v3 > 0.f ? v3 : v3 * g.in_slope;
        }
        if (gate_b && ok[u]) {
          v0 *= (gv[u].x > 0.f) ? 1.f : g.in_gate_slope;
          v1 *= (gv[u].y > 0.f) ? 1.f : g.in_gate_slope;
          v2 *= (gv[u].z > 0.f) ? 1.f : g.in_gate_slope;
          v3 *= (gv[u].w > 0.f) ? 1.f : g.in_gate_slope;
        }
        const int row = (dm == 1) ? rr : ((rel % dm) * Wp + rel / dm) * inner + pin;
        cw_store4<BF16>(win, row * LDW + c4, v0, v1, v2, v3);
      }
    }
    commit_w(0);
    __syncthreads();

    for (int ti = 0; ti < nv; ++ti) {
      if (ti + 1 < nv) fetch_w(ti + 1, c0);
      int arow[MREP];
      if (up > 1) {
        const int off = __builtin_amdgcn_readfirstlane(s_off[ti]);
#pragma unroll
        for (int f = 0; f < MREP; ++f) arow[f] = cw_floordiv(fm[f] * g.in_mul + off, up) - lo;
      } else {
        const int trow = __builtin_amdgcn_readfirstlane(s_row[ti]) * inner;
#pragma unroll
        for (int f = 0; f < MREP; ++f) arow[f] = trow + fbase[f];
      }
      const unsigned char* bcur = bt + (size_t)(ti & 1) * BN * LDW * ESZ;
      if (BF16) {
        const __bf16* Wh = reinterpret_cast<const __bf16*>(win);
        const __bf16* Bh = reinterpret_cast<const __bf16*>(bcur);
        bf16x8 af[MREP], bfr[4];
#pragma unroll
        for (int f = 0; f < MREP; ++f)
          af[f] = *reinterpret_cast<const bf16x8*>(&Wh[arow[f] * LDW + (lane >> 4) * 8]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          bfr[j] = *reinterpret_cast<const bf16x8*>(&Bh[(wn * 64 + j * 16 + (lane & 15)) * LDW + (lane >> 4) * 8]);
        if (!(dbg & 8)) {
#pragma unroll
          for (int f = 0; f < MREP; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[f], bfr[j], acc[f][j], 0, 0, 0);
        }
      } else {
        const float* Wf = reinterpret_cast<const float*>(win);
        const float* Bf = reinterpret_cast<const float*>(bcur);
#pragma unroll
        for (int ks = 0; ks < CW_CK / 4; ++ks) {
          float af[MREP], bfr[4];
#pragma unroll
          for (int f = 0; f < MREP; ++f) af[f] = Wf[arow[f] * LDW + ks * 4 + (lane >> 4)];
#pragma unroll
          for (int j = 0; j < 4; ++j) bfr[j] = Bf[(wn * 64 + j * 16 + (lane & 15)) * LDW + ks * 4 + (lane >> 4)];
#pragma unroll
          for (int f = 0; f < MREP; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[f], bfr[j], acc[f][j], 0, 0, 0);
        }
      }
      if (ti + 1 < nv) commit_w((ti + 1) & 1);
      __syncthreads();
    }
  }

  // ---- epilogue: each wave transposes its fragments through a PRIVATE 16 x 68 float LDS strip (no workgroup barrier:
  // LDS operations of one wave execute in order) so that LPR lanes write one contiguous output row segment; bias,
  // LeakyReLU, residual and LeakyReLU' gate fused.  The ablation run of round 1 (profiles/r01_conv_win_ablation.log)
  // charged 12-16 us of a 36-55 us launch to this block: per-element bias / residual loads, two barriers per
  // fragment and half of the lanes idle when a tile has 32 live output channels.
  __syncthreads();  // the strips overlay the window / weight tiles other waves may still be reading
  float* strip = reinterpret_cast<float*>(cw_lds) + wave * (16 * 68);
  const bool vec_ok = ((g.Ntot & 3) == 0) && ((g.NG & 3) == 0) && (((uintptr_t)g.out & 15) == 0) &&
                      (!g.res || ((uintptr_t)g.res & 15) == 0) && (!g.out_gate || ((uintptr_t)g.out_gate & 15) == 0) &&
                      (!g.bias || ((uintptr_t)g.bias & 15) == 0);
  // live channels of this wave's 64-column strip -> lanes per row (4 / 8 / 16) and rows per pass (16 / 8 / 4)
  const int live = min(64, n_end - (n0 + wn * 64));
  const int lpr = (live <= 16) ? 4 : ((live <= 32) ? 8 : 16);
  const int rpp = 64 / lpr;  // rows per pass
  const int lcol = (lane % lpr) * 4, lrow = lane / lpr;
  const int n = n0 + wn * 64 + lcol;
  const int cnt = min(4, n_end - n);  // <= 0: this lane has no live channel
  const bool v4 = vec_ok && cnt == 4;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (g.bias && cnt > 0) {
    if (v4) {
      const float4 t = *reinterpret_cast<const float4*>(g.bias + n);
      bv[0] = t.x, bv[1] = t.y, bv[2] = t.z, bv[3] = t.w;
    } else {
      for (int e = 0; e < cnt; ++e) bv[e] = g.bias[n + e];
    }
  }
  const long long obase = (long long)b * g.Tdst * inner * g.Ntot + n;  // rows of one batch item span < 2^31 elements
#pragma unroll
  for (int f = 0; f < MREP; ++f) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) strip[((lane >> 4) * 4 + r) * 68 + j * 16 + (lane & 15)] = acc[f][j][r];
    __builtin_amdgcn_wave_barrier();
    for (int rl = lrow; rl < 16; rl += rpp) {
      const int mp = m0 + wm * (MREP * 16) + f * 16 + rl;
      if (mp < R && cnt > 0 && !(dbg & 2)) {
        int row;  // (d * inner + pin): output row inside the batch item
        if (inner > 1) {
          const int m = mp / inner;
          row = (m * g.phases + phase) * inner + (mp - m * inner);
        } else {
          row = mp * g.phases + phase;
        }
        const long long o = obase + (long long)row * g.Ntot;
        const float4 a4 = *reinterpret_cast<const float4*>(&strip[rl * 68 + lcol]);
        float v[4] = {a4.x + bv[0], a4.y + bv[1], a4.z + bv[2], a4.w + bv[3]};
        if (g.out_act) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * g.out_slope;
        }
        if (v4) {
          if (g.res) {
            const float4 t = *reinterpret_cast<const float4*>(g.res + o);
            v[0] += t.x, v[1] += t.y, v[2] += t.z, v[3] += t.w;
          }
          if (g.out_gate) {
            const float4 t = *reinterpret_cast<const float4*>(g.out_gate + o);
            v[0] *= (t.x > 0.f) ? 1.f : g.out_gate_slope;
            v[1] *= (t.y > 0.f) ? 1.f : g.out_gate_slope;
            v[2] *= (t.z > 0.f) ? 1.f : g.out_gate_slope;
            v[3] *= (t.w > 0.f) ? 1.f : g.out_gate_slope;
          }
          f32x4 nt = {v[0], v[1], v[2], v[3]};
          __builtin_nontemporal_store(nt, reinterpret_cast<f32x4*>(g.out + o));
        } else {
          for (int e = 0; e < cnt; ++e) {
            float x = v[e];
            if (g.res) x += g.res[o + e];
            if (g.out_gate) x *= (g.out_gate[o + e] > 0.f) ? 1.f : g.out_gate_slope;
            g.out[o + e] = x;
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------------------
// Direct form of the same contract for channel counts the MFMA tiles cannot use (CR not a multiple of 4: the
// 1/2/4-channel wavelet and projection convolutions of the scale discriminators, and the input gradient of the
// Cout = 1 output convolutions): one thread per output element, taps x CR multiply-adds each.  These layers
// are bound by the activations they stream, not by arithmetic.
__global__ __launch_bounds__(256) void conv_direct_kernel(const kantts_conv_args g) {
  const long long total = (long long)g.B * g.Tdst * g.inner * g.Ntot;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int n = (int)(e % g.Ntot);
  long long r = e / g.Ntot;
  const int pi = (int)(r % g.inner);
  r /= g.inner;
  const int d = (int)(r % g.Tdst);
  const int b = (int)(r / g.Tdst);
  const int phase = d % g.phases, m = d / g.phases;
  const int grp = n / g.NG;
  const long long in_pitch = (long long)g.inner * g.Cin_tot;
  const float* in_b = g.in + ((long long)b * g.Tsrc * g.inner + pi) * g.Cin_tot + (long long)grp * g.CR;
  const float* gate_b = g.in_gate ? g.in_gate + ((long long)b * g.Tsrc * g.inner + pi) * g.Cin_tot + (long long)grp * g.CR : nullptr;
  float acc = 0.f;
  for (int k = 0; k < g.K; ++k) {
    const int u = g.in_add + phase + k * g.in_kstep;
    const int q = cw_floordiv(u, g.in_div);
    if (q * g.in_div != u) continue;
    const int tu = m * g.in_mul + q;
    const int up = g.up > 1 ? g.up : 1;
    if (tu < 0 || tu >= g.Tsrc * up) continue;
    const int t = tu / up;
    const float* wr = g.w + ((long long)k * g.Ntot + n) * g.CR;
    for (int c = 0; c < g.CR; ++c) {
      float v = in_b[(long long)t * in_pitch + c];
      if (g.in_act) v = v > 0.f ? v : v * g.in_slope;
      if (gate_b) v *= (gate_b[(long long)t * in_pitch + c] > 0.f) ? 1.f : g.in_gate_slope;
      acc += v * wr[c];
    }
  }
  if (g.bias) acc += g.bias[n];
  if (g.out_act) acc = acc > 0.f ? acc : acc * g.out_slope;
  if (g.res) acc += g.res[e];
  if (g.out_gate) acc *= (g.out_gate[e] > 0.f) ? 1.f : g.out_gate_slope;
  g.out[e] = acc;
}

template <bool BF16, int WM, int MREP>
static int cw_launch(const kantts_conv_args& g, hipStream_t st) {
  constexpr int WN = 4 / WM;
  constexpr int BQ = WM * MREP * 16;
  constexpr int BN = WN * 64;
  constexpr int LDW = BF16 ? 48 : 36;
  constexpr int ESZ = BF16 ? 2 : 4;
  // worst-case window over the phases: the valid taps of a phase span at most (K-1)*|kstep|/div + 1 offsets
  const int span = ((g.K - 1) * abs(g.in_kstep)) / g.in_div + 1;
  const int up = g.up > 1 ? g.up : 1;
  const int dm = (up > 1) ? 1 : g.in_mul;
  const int mspan = (BQ - 1) / g.inner + 1;  // tokens m a tile of BQ folded rows can straddle, minus one
  const int W = (up > 1) ? ((BQ - 1) * g.in_mul + span) / up + 3 : mspan * g.in_mul + span + 1;
  const int Wp = (W + dm - 1) / dm;
  size_t lds = (size_t)Wp * dm * g.inner * LDW * ESZ + 2 * (size_t)BN * LDW * ESZ;
  const size_t strip = 4 * 16 * 68 * sizeof(float);
  if (lds < strip) lds = strip;
  lds += CW_HDR;
  if (lds > 160 * 1024 - 1024) return KANTTS_E_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_win_kernel<BF16, WM, MREP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int mrows = (g.Tdst + g.phases - 1) / g.phases;
  const int ntpg = (g.NG + BN - 1) / BN;
  dim3 grid(g.groups * ntpg, kantts_cdiv((long long)mrows * g.inner, BQ), g.B * g.phases);
#ifdef CW_DEBUG
  kantts_conv_args gd = g;
  if (const char* e = getenv("KANTTS_CW_DBG")) gd.up = (gd.up & 0xffff) | (atoi(e) << 16);
  hipLaunchKernelGGL((conv_win_kernel<BF16, WM, MREP>), grid, dim3(CW_THREADS), lds, st, gd);
#else
  hipLaunchKernelGGL((conv_win_kernel<BF16, WM, MREP>), grid, dim3(CW_THREADS), lds, st, g);
#endif
  KANTTS_CHECK_LAUNCH();
}

extern "C" int kantts_conv_win_launch(const kantts_conv_args* a, void* stream) {
  if (!a || !a->in || !a->w || !a->out) return KANTTS_E_BADARG;
  const kantts_conv_args& g = *a;
  if (g.B < 0 || g.Tsrc < 0 || g.Tdst < 0 || g.K < 1 || g.groups < 1 || g.NG < 1 || g.CR < 1 || g.in_mul < 1 ||
      g.in_div < 1 || g.phases < 1 || g.inner < 1 || g.up < 0)
    return KANTTS_E_BADARG;
  if (g.Ntot != g.groups * g.NG || g.Cin_tot != g.groups * g.CR) return KANTTS_E_BADARG;
  if (g.B == 0 || g.Tdst == 0) return KANTTS_OK;
  if (g.CR & 3) {
    const long long total = (long long)g.B * g.Tdst * g.inner * g.Ntot;
    if (total > 0x7fffffffLL * 256) return KANTTS_E_UNSUPPORTED;
    hipLaunchKernelGGL(conv_direct_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g);
    KANTTS_CHECK_LAUNCH();
  }
  if (g.K > CW_MAXTAPS) return KANTTS_E_UNSUPPORTED;
  // float4 staging of activations and weights
  if ((g.CR & 3) || (g.Cin_tot & 3) || ((uintptr_t)g.in & 15) || ((uintptr_t)g.w & 15) ||
      (g.in_gate && ((uintptr_t)g.in_gate & 15)))
    return KANTTS_E_UNSUPPORTED;
  if ((long long)g.B * g.phases > 65535 || (g.up > 1 && g.inner > 1)) return KANTTS_E_UNSUPPORTED;
  if (g.B == 0 || g.Tdst == 0) return KANTTS_OK;
  hipStream_t st = (hipStream_t)stream;
  const long long mrows = (long long)((g.Tdst + g.phases - 1) / g.phases) * g.inner;  // folded rows per batch item
  // tile choice: the largest tile that still gives every CU about two workgroups (256 CUs); the early generator
  // stages (256 tokens x 256 channels per item) would otherwise run 128 workgroups of 128x128
  auto blocks = [&](int bq, int bn) {
    return (long long)g.groups * ((g.NG + bn - 1) / bn) * ((mrows + bq - 1) / bq) * g.B * g.phases;
  };
  bool wide = (g.NG >= 128) && (mrows >= 128);
  bool tall = mrows >= 96;
  if (wide && blocks(128, 128) < 512) wide = false;
  if (!wide && tall && blocks(128, 64) < 512 && mrows >= 64) tall = false;
  if (g.precision == 1) {
    if (wide) return cw_launch<true, 2, 4>(g, st);
    if (tall) return cw_launch<true, 4, 2>(g, st);
    return cw_launch<true, 4, 1>(g, st);
  } else if (g.precision == 0) {
    if (wide) return cw_launch<false, 2, 4>(g, st);
    if (tall) return cw_launch<false, 4, 2>(g, st);
    return cw_launch<false, 4, 1>(g, st);
  }
  return KANTTS_E_BADARG;
}
