// LSTM recurrence (H = 128) as a persistent per-sequence kernel.
//
// Replaces the time loop of torch.nn.LSTM at
//   kantts/models/sambert/adaptors.py:44-57 (2-layer duration LSTM), :109-134 (packed BiLSTM),
//   kantts/models/sambert/kantts_sambert.py:637-646 (postnet LSTM over T_mel frames).
// The input projection x @ W_ih^T + b_ih for all t is hoisted into one segmented GEMM (MFMA); this
// kernel only runs the sequential part.  One workgroup (512 threads = 8 waves) owns one
// (sequence, direction): a quad of lanes owns a cell and keeps its 4 x 128 recurrent weights in VGPRs
// (128 per lane) for the whole sequence, h_{t-1} lives in LDS (double-buffered), so a step costs
// 128 multiply-adds per lane, a few DPP moves and ONE workgroup barrier, and no HBM traffic besides
// gx[t] in / h[t] out.
// pack_padded_sequence semantics come from per-sequence lengths: a row only runs t < len (the
// reverse direction starts at len-1) and writes zeros to the padded tail.
//
// Gate order i, f, g, o (torch).  Saved for backward: post-activation gates (B,T,4H) and c (B,T,H).
#include "common.h"

#define LH 128
#define LG 512

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
// Activations of the recurrence.  All four lanes of a quad evaluate the cell update, so the step is VALU-bound (8 waves
// on 4 SIMDs): libm's tanhf (~40 instructions, and a divergent branch beside the sigmoid) made the quad layout SLOWER
// than the two-barrier one (fwd 135 -> 149 us, bwd 147 -> 228 us per call, profiles/r02_runQ_*).  tanh(x) = 2 s(2x) - 1
// shares the sigmoid's code; FAST (bf16 mode) uses v_exp_f32 / v_rcp_f32 directly (1-2 ulp), the fp32 mode keeps libm's
// expf and an IEEE division (absolute error ~1e-7 from the final subtraction, far inside the parity tolerance).
template <bool FAST>
__device__ __forceinline__ float lstm_sigmoid(float x) {
  if (FAST) return __builtin_amdgcn_rcpf(1.f + __expf(-x));
  return 1.f / (1.f + expf(-x));
}
template <bool FAST>
__device__ __forceinline__ float lstm_tanh(float x) {
  return fmaf(2.f, lstm_sigmoid<FAST>(2.f * x), -1.f);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter, i.e. it waits
// until the step's global STORES (saved gates / cell state / outputs: pure outputs, never re-read by this kernel) have
// been acknowledged by L2 -- two or three such round trips on the critical path of every time step.
__device__ __forceinline__ void lstm_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// gx: (B,T,ndir*4H) [direction d at column offset d*4H]; out: (B,T,ndir*H)
// whh: (ndir, 4H, H), bhh: (ndir, 4H); c_out: (ndir,B,T,H);
// gates_out: (ndir,B,T,H,4) -- the saved activations i, f, g, o of a cell side by side.  The tensor is private to the fwd / bwd
// pair of this file.  [round 6] Until then gate-major (ndir,B,T,4,H): a lane of the pair form stored its two gates with two
// instructions and the backward pass fetched a cell's four with four; the step's stores were measured at 0.059 of the
// 0.570 us of a forward step (profiles/r06_runLA_lstm_fwd_ablation.log).  Cell-major: one 8-byte store, one 16-byte load.
// Experiment builds only (scripts/build_lstm_abl.sh; the product Makefile never defines it): which part of a forward step
// of lstm_fwd_pair_kernel costs what -- bit 0 the step's global stores, bit 1 the transcendental activations, bit 2 the LDS
// publication of h + the barrier, bit 3 seven eighths of the dot products.  Results are wrong under any of them.
#ifndef LSTM_ABL
#define LSTM_ABL 0
#endif
typedef __bf16 lstm_bf16x2 __attribute__((ext_vector_type(2)));
union LstmPack4 {
  uint4 u;
  lstm_bf16x2 p[4];
};

// ---- quad helpers: the four lanes 4j .. 4j+3 of a wave own cell j (one lane per reduction quarter / gate)
__device__ __forceinline__ float lstm_dpp_xor1(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float lstm_dpp_xor2(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
}
template <int N>
__device__ __forceinline__ float lstm_quad_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), N * 0x55, 0xF, 0xF, false));  // quad_perm [N,N,N,N]
}

// Forward.  Round 1 / early round 2: thread r owned gate row r (128-long dot product), the 512 activations met in LDS,
// 128 threads then updated the cells, h went back through LDS: two workgroup barriers and two LDS round trips on the
// chain of every step (0.73 us per step measured, against ~0.2 us of dot products).  Now the QUAD of lanes 4j..4j+3 owns
// cell j: lane kq holds columns [32 kq, 32 kq + 32) of the four gate rows of its cell (the same 128 weights per lane),
// the four partial sums of each gate are combined by a DPP reduce-scatter (lane kq ends up with gate kq), each lane
// applies ITS gate's activation, the activations are quad-broadcast, and all four lanes update c and h redundantly:
// no LDS exchange of gates, h double-buffered in LDS, ONE barrier per step.
// BF16 = true (throughput mode): packed bf16 pairs with fp32 accumulation (v_dot2_f32_bf16); gates, cell state and
// outputs stay fp32.
template <bool BF16>
__global__ __launch_bounds__(LG) void lstm_fwd_kernel(const float* __restrict__ gx, const float* __restrict__ whh,
                                                      const float* __restrict__ bhh, const int32_t* __restrict__ lens,
                                                      float* __restrict__ out, float* __restrict__ gates_out,
                                                      float* __restrict__ c_out, int B, int T, int ndir,
                                                      int reverse_first) {
  __shared__ __attribute__((aligned(16))) float h_s[2][LH];
  __shared__ __attribute__((aligned(16))) __bf16 h_b[2][LH];
  const int tid = threadIdx.x, j = tid >> 2, kq = tid & 3;
  const int b = blockIdx.x, dir = blockIdx.y;
  const bool rev = reverse_first ? true : (dir == 1);
  const int len = lens ? min(lens[b], T) : T;
  float w[BF16 ? 1 : 4 * 32];
  lstm_bf16x2 wq[BF16 ? 4 * 16 : 1];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4* wp = reinterpret_cast<const float4*>(whh + ((long long)dir * LG + g * LH + j) * LH + kq * 32);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float4 t = wp[k];
      if (BF16) {
        wq[g * 16 + 2 * k] = (lstm_bf16x2){(__bf16)t.x, (__bf16)t.y};
        wq[g * 16 + 2 * k + 1] = (lstm_bf16x2){(__bf16)t.z, (__bf16)t.w};
      } else {
        w[g * 32 + 4 * k] = t.x;
        w[g * 32 + 4 * k + 1] = t.y;
        w[g * 32 + 4 * k + 2] = t.z;
        w[g * 32 + 4 * k + 3] = t.w;
      }
    }
  }
  // lane kq brings in gx + bias of gate kq
  const float bias = bhh ? bhh[dir * LG + kq * LH + j] : 0.f;
  if (kq == 0) {
    h_s[0][j] = 0.f;
    h_b[0][j] = (__bf16)0.f;
  }
  float c = 0.f;
  const int gx_ld = ndir * LG;  // offsets inside one sequence are 32-bit (T * ndir * 4H < 2^31: checked by the launchers)
  const float* gxb = gx + (long long)b * T * gx_ld + dir * LG + kq * LH + j;
  float* outb = out + (long long)b * T * ndir * LH + dir * LH;
  float* gob = gates_out + (((long long)dir * B + b) * T) * LG + 4 * j + kq;
  float* cob = c_out + (((long long)dir * B + b) * T) * LH;
  __syncthreads();
  // gx[t] comes from L2 / HBM (0.5-2 us) while a step is ~0.4 us: the values of the next chunk of steps are loaded --
  // unconditionally, steps clamped into the sequence -- at the top of a chunk and first touched a whole chunk later
  // (a load inside an `if`, or a ring interleaved with the steps, puts a vmcnt(0) / vmcnt(1) wait on every step).
  constexpr int PF = 8;
  float gq[PF], gn[PF];
  const int last = len > 0 ? len - 1 : 0;
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const int su = min(u, last);
    gq[u] = gxb[(rev ? last - su : su) * gx_ld];
  }
  int cur = 0;
  for (int step0 = 0; step0 < len; step0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int sn = min(step0 + PF + u, last);
      gn[u] = gxb[(rev ? last - sn : sn) * gx_ld];
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int step = step0 + u;
      if (step >= len) break;
      const int t = rev ? len - 1 - step : step;
      float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
      if (BF16) {
        const uint4* hp = reinterpret_cast<const uint4*>(&h_b[cur][kq * 32]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          LstmPack4 hv;
          hv.u = hp[k];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            p0 = __builtin_amdgcn_fdot2_f32_bf16(wq[4 * k + e], hv.p[e], p0, false);
            p1 = __builtin_amdgcn_fdot2_f32_bf16(wq[16 + 4 * k + e], hv.p[e], p1, false);
            p2 = __builtin_amdgcn_fdot2_f32_bf16(wq[32 + 4 * k + e], hv.p[e], p2, false);
            p3 = __builtin_amdgcn_fdot2_f32_bf16(wq[48 + 4 * k + e], hv.p[e], p3, false);
          }
        }
      } else {
        const float4* hp = reinterpret_cast<const float4*>(&h_s[cur][kq * 32]);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float4 hv = hp[k];
          const float hh[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            p0 = fmaf(w[4 * k + e], hh[e], p0);
            p1 = fmaf(w[32 + 4 * k + e], hh[e], p1);
            p2 = fmaf(w[64 + 4 * k + e], hh[e], p2);
            p3 = fmaf(w[96 + 4 * k + e], hh[e], p3);
          }
        }
      }
      // reduce-scatter over the quad: lane kq ends with the full pre-activation of gate kq
      const bool b0 = kq & 1, b1 = kq & 2;
      const float ka = (b0 ? p1 : p0) + lstm_dpp_xor1(b0 ? p0 : p1);  // gate (kq & 1)
      const float kb = (b0 ? p3 : p2) + lstm_dpp_xor1(b0 ? p2 : p3);  // gate (kq & 1) + 2
      const float pre = (b1 ? kb : ka) + lstm_dpp_xor2(b1 ? ka : kb) + (gq[u] + bias);
      // activation by gate: i, f, o sigmoid; g tanh
      const float sg = lstm_sigmoid<BF16>((kq == 2) ? 2.f * pre : pre);
      const float act = (kq == 2) ? fmaf(2.f, sg, -1.f) : sg;
      gob[t * LG] = act;
      const float ig = lstm_quad_bcast<0>(act), fg = lstm_quad_bcast<1>(act);
      const float gg = lstm_quad_bcast<2>(act), og = lstm_quad_bcast<3>(act);
      c = fmaf(fg, c, ig * gg);
      const float hn = og * lstm_tanh<BF16>(c);
      if (kq == 0) {
        if (BF16)
          h_b[cur ^ 1][j] = (__bf16)hn;
        else
          h_s[cur ^ 1][j] = hn;
        outb[t * ndir * LH + j] = hn;
        cob[t * LH + j] = c;
      }
      cur ^= 1;
      lstm_barrier();
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) gq[u] = gn[u];
  }
  // zero the padded tail (pad_packed_sequence) -- outputs only; saved state is never read there
  for (int tt = len; tt < T; ++tt)
    if (tid < LH) outb[(long long)tt * ndir * LH + tid] = 0.f;
}

// [round 6] A PAIR of lanes per cell (256 threads = one wave per SIMD) instead of a quad (512 threads = two waves per SIMD),
// bf16 mode -- measured 0.569 against 0.607 us per step (postnet recurrence, 612 steps: 348 against 372 us,
// profiles/r06_runT_lstm_pair_vs_quad.log; KANTTS_LSTM_PAIR=0 selects the quad kernel, which also stays the fp32 form):
// the dot products of a step are the same 512 issue cycles per SIMD either way, but everything
// else of a step -- the activation, the cell update, the stores, which every lane of a cell executes redundantly -- is issued
// by HALF as many lanes, and one wave per SIMD has no second wave's redundant work in its way.  Lane p of a pair holds columns
// [64 p, 64 p + 64) of the four gate rows of its cell (256 weights as 128 packed pairs); after the dot products one DPP
// exchange leaves lane 0 with the gates (i, f) and lane 1 with (g, o); each applies its two activations, a second exchange
// gives both lanes all four.
// Measured and refused on top of this form (profiles/r06_runX_lstm_pair_mfma_split.log): HALF of the dot products on the matrix
// pipe -- the sixteen v_mfma_f32_16x16x32_bf16 of the gates (g, o) of a wave's 32 cells (A rows permuted so that every lane
// finds its own cell in its own accumulator, B = h broadcast) issued between the 64 v_dot2 of (i, f), one MFMA per four
// v_dot2: 0.70 us per step against 0.57 -- the step is a latency chain, not an issue-bound loop, and the accumulators of
// four dependent 16-cycle MFMAs arrive later than the last v_dot2.
__global__ __launch_bounds__(256) void lstm_fwd_pair_kernel(const float* __restrict__ gx, const float* __restrict__ whh,
                                                           const float* __restrict__ bhh, const int32_t* __restrict__ lens,
                                                           float* __restrict__ out, float* __restrict__ gates_out,
                                                           float* __restrict__ c_out, int B, int T, int ndir,
                                                           int reverse_first) {
  __shared__ __attribute__((aligned(16))) __bf16 h_b[2][LH];
  const int tid = threadIdx.x, j = tid >> 1, p = tid & 1;
  const int b = blockIdx.x, dir = blockIdx.y;
  const bool rev = reverse_first ? true : (dir == 1);
  const int len = lens ? min(lens[b], T) : T;
  lstm_bf16x2 wq[4 * 32];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    // slot g of lane p holds gate g ^ 2 p: every lane KEEPS slots 0, 1 (its own gate pair) and SENDS slots 2, 3 (its
    // partner's pair) -- the exchange below needs no select
    const float4* wp = reinterpret_cast<const float4*>(whh + ((long long)dir * LG + (g ^ (2 * p)) * LH + j) * LH + p * 64);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float4 t = wp[k];
      wq[g * 32 + 2 * k] = (lstm_bf16x2){(__bf16)t.x, (__bf16)t.y};
      wq[g * 32 + 2 * k + 1] = (lstm_bf16x2){(__bf16)t.z, (__bf16)t.w};
    }
  }
  // lane p brings in gx + bias of the gates 2 p and 2 p + 1
  const float bias0 = bhh ? bhh[dir * LG + (2 * p) * LH + j] : 0.f;
  const float bias1 = bhh ? bhh[dir * LG + (2 * p + 1) * LH + j] : 0.f;
  if (p == 0) h_b[0][j] = (__bf16)0.f;
  float c = 0.f;
  const int gx_ld = ndir * LG;  // offsets inside one sequence are 32-bit (T * ndir * 4H < 2^31: checked by the launchers)
  const float* gxb = gx + (long long)b * T * gx_ld + dir * LG + (2 * p) * LH + j;
  float* outb = out + (long long)b * T * ndir * LH + dir * LH;
  float* gob = gates_out + (((long long)dir * B + b) * T) * LG + 4 * j + 2 * p;
  float* cob = c_out + (((long long)dir * B + b) * T) * LH;
  __syncthreads();
  // gx + bias of the next PF steps wait in ONE register ring: slot u is refilled, in place, by the step that has just read
  // it (the load for step + PF lands during the PF - 1 steps in between; the in-order vector-memory counter lets the next
  // reader wait for exactly that load).  Until round 6 two rings (current / next chunk) were copied into each other every PF
  // steps: 32 registers and a block of 16 loads + 16 moves in front of every eighth step, in a kernel that is out of registers
  // (the h fragments of a step were read through four registers with a full LDS round trip exposed before the first product).
  constexpr int PF = 8;
  float gq0[PF], gq1[PF];
  const int last = len > 0 ? len - 1 : 0;
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const int su = min(u, last);
    const int o = (rev ? last - su : su) * gx_ld;
    gq0[u] = gxb[o];
    gq1[u] = gxb[o + LH];
  }
  const float k1 = p ? 2.f : 1.f, k3 = p ? -1.f : 0.f;  // lane 1's first gate is tanh(g) = 2 sigmoid(2 g) - 1
  // lane 0 stores the step's output row, lane 1 its cell state (both lanes hold both): base and row pitch per lane
  float* const obase = p ? cob + j : outb + j;
  const int opitch_sel = p;  // (pitch LH for lane 1, ndir * LH for lane 0: two scalar products and one select per step)
  int cur = 0;
  float p_sink = 0.f;  // (ablation builds only: keeps the chain alive when the stores are masked)
  for (int step0 = 0; step0 < len; step0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int step = step0 + u;
      if (step < len) {  // uniform
        const int t = rev ? len - 1 - step : step;
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
        const uint4* hp = reinterpret_cast<const uint4*>(&h_b[cur][p * 64]);
        LstmPack4 hv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) hv[k].u = hp[k];  // all eight 16-byte reads in flight before the first product
#pragma unroll
        for (int k = 0; k < (LSTM_ABL & 8 ? 1 : 8); ++k) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            p0 = __builtin_amdgcn_fdot2_f32_bf16(wq[4 * k + e], hv[k].p[e], p0, false);
            p1 = __builtin_amdgcn_fdot2_f32_bf16(wq[32 + 4 * k + e], hv[k].p[e], p1, false);
            p2 = __builtin_amdgcn_fdot2_f32_bf16(wq[64 + 4 * k + e], hv[k].p[e], p2, false);
            p3 = __builtin_amdgcn_fdot2_f32_bf16(wq[96 + 4 * k + e], hv[k].p[e], p3, false);
          }
        }
        // exchange over the pair: lane 0 ends with the full pre-activations of (i, f), lane 1 with (g, o)
        const float pa = p0 + lstm_dpp_xor1(p2) + (gq0[u] + bias0);
        const float pb = p1 + lstm_dpp_xor1(p3) + (gq1[u] + bias1);
        {  // refill the slot just read: gx of step + PF (clamped to the last step of the sequence)
          const int sn = min(step + PF, last);
          const int o = (rev ? last - sn : sn) * gx_ld;
          gq0[u] = gxb[o];
          gq1[u] = gxb[o + LH];
        }
        // lane 0: sigmoid(i), sigmoid(f); lane 1: tanh(g) = 2 sigmoid(2 g) - 1, sigmoid(o)
        const float sa = (LSTM_ABL & 2) ? 0.25f * pa : lstm_sigmoid<true>(k1 * pa);
        const float a0 = fmaf(k1, sa, k3);
        const float a1 = (LSTM_ABL & 2) ? 0.25f * pb : lstm_sigmoid<true>(pb);
        if (!(LSTM_ABL & 1)) {
          *reinterpret_cast<float2*>(gob + t * LG) = make_float2(a0, a1);
        }
        const float o0 = lstm_dpp_xor1(a0), o1 = lstm_dpp_xor1(a1);
        const float fg = p ? o1 : a1, og = p ? a1 : o1;  // i . g = a0 . o0 in both lanes
        c = fmaf(fg, c, a0 * o0);
        const float hn = og * ((LSTM_ABL & 2) ? 0.5f * c : lstm_tanh<true>(c));
        if (!(LSTM_ABL & 4)) h_b[cur ^ 1][j] = (__bf16)hn;  // both lanes of the pair write the same value: no exec-mask detour
        if (!(LSTM_ABL & 1)) {
          const int off_out = t * ndir * LH, off_c = t * LH;  // scalar
          obase[opitch_sel ? off_c : off_out] = opitch_sel ? c : hn;
        }
        if (LSTM_ABL & 1) p_sink += hn;
        cur ^= 1;
        if (!(LSTM_ABL & 4)) lstm_barrier();
      }
    }
  }
  for (int tt = len; tt < T; ++tt)
    if (tid < LH) outb[(long long)tt * ndir * LH + tid] = 0.f;
  if ((LSTM_ABL & 1) && p_sink == 123.456f) outb[0] = p_sink;
}

__device__ __forceinline__ float lstm_dpp_half_mirror(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, false));  // lane i <-> 7 - i
}
__device__ __forceinline__ float lstm_dpp_ror8(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x128, 0xF, 0xF, false));  // row_ror:8, lane i <-> i ^ 8
}

// Backward through time: produces dgates_pre (ndir,B,T,4H) (gradient w.r.t. the pre-activation gates); the weight /
// input gradients are GEMMs over it (host layer).  dout: (B,T,ndir*H).
// A DPP ROW of 16 lanes owns four cells.  Phase A: lane l of the row is (cell g2(l & 7), gate l >> 2; see below): all four lanes of a
// cell carry its dh and dc and each publishes the gradient of ITS gate (one value per lane: LDS + the saved tensor).
// After ONE barrier, phase B reduces dh_prev[k] = sum_r W_hh[r][k] . dg[r] for the row's four cells: lane l covers the
// 32 gradient rows [32 l, 32 l + 32) (a 64-byte LDS read, shared by its four outputs: 128 weights per lane in registers)
// and the 16 partial sums meet through DPP adds (since round 6 a halving reduction in slot order: 5 adds for the four cells)
// -- no LDS round trip, dg double-buffered.
// History: three barriers + two LDS round trips per step 0.82 us; a quad-per-cell layout whose lanes each read a whole
// 128-row quarter of dg (16 wide LDS reads with four distinct addresses per wave: LDS-bound) 1.23 us.
template <bool BF16>
__global__ __launch_bounds__(LG) void lstm_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ whh,
                                                      const int32_t* __restrict__ lens,
                                                      const float* __restrict__ gates, const float* __restrict__ cst,
                                                      float* __restrict__ dgates, int B, int T, int ndir,
                                                      int reverse_first) {
  __shared__ __attribute__((aligned(16))) float dg_s[2][LG];
  __shared__ __attribute__((aligned(16))) __bf16 dg_b[2][LG];
  const int tid = threadIdx.x, row = tid >> 4, l = tid & 15;
  // [round 6] slot order of the halving reduction (see lstm_bwd_pair_kernel): slot s of lane l = cell s ^ g2(l & 7) with
  // g2(b) = (b0 ^ b2) + 2 (b1 ^ b2) -- g2(l ^ 1) = g2 ^ 1, g2(l ^ 2) = g2 ^ 2, g2(l ^ 7) = g2 -- so 2 + 1 + 1 + 1 = 5 DPP adds
  // leave every lane with ITS cell's dh where four all-reduces (16 adds) and a select chain did; the four lanes of a cell are
  // told apart by l >> 2, which is their gate.
  const int g2 = ((l ^ (l >> 2)) & 1) | ((((l >> 1) ^ (l >> 2)) & 1) << 1);
  const int k = row * 4 + g2, kq = l >> 2;  // phase-A identity: cell, gate
  const int b = blockIdx.x, dir = blockIdx.y;
  const bool rev = reverse_first ? true : (dir == 1);
  const int len = lens ? min(lens[b], T) : T;
  // phase-B weights: slot s = W_hh[32 l + rr][4 row + (s ^ g2)], s < 4, rr < 32
  float w[BF16 ? 1 : 4 * 32];
  lstm_bf16x2 wq[BF16 ? 4 * 16 : 1];
#pragma unroll
  for (int rr = 0; rr < 32; rr += 2) {
    const float4 w0 = *reinterpret_cast<const float4*>(whh + ((long long)dir * LG + l * 32 + rr) * LH + row * 4);
    const float4 w1 = *reinterpret_cast<const float4*>(whh + ((long long)dir * LG + l * 32 + rr + 1) * LH + row * 4);
    const float c0[4] = {w0.x, w0.y, w0.z, w0.w}, c1[4] = {w1.x, w1.y, w1.z, w1.w};
    float a0[4], a1[4];  // slot order (selects instead of a dynamic register index; once per launch)
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      const int c = sl ^ g2;
      a0[sl] = c == 0 ? c0[0] : (c == 1 ? c0[1] : (c == 2 ? c0[2] : c0[3]));
      a1[sl] = c == 0 ? c1[0] : (c == 1 ? c1[1] : (c == 2 ? c1[2] : c1[3]));
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (BF16) {
        wq[c * 16 + rr / 2] = (lstm_bf16x2){(__bf16)a0[c], (__bf16)a1[c]};
      } else {
        w[c * 32 + rr] = a0[c];
        w[c * 32 + rr + 1] = a1[c];
      }
    }
  }
  const float* doutb = dout + (long long)b * T * ndir * LH + dir * LH;
  const float* gb = gates + (((long long)dir * B + b) * T) * LG;
  const float* cb = cst + (((long long)dir * B + b) * T) * LH;
  float* dgb = dgates + (((long long)dir * B + b) * T) * LG;
  float dc = 0.f, dh_rec = 0.f;
  // time runs opposite to the forward recurrence.  The seven operands of a step (four saved gates, c_t, c_{t-1}, dout_t)
  // are loaded a CHUNK of PF steps ahead (see the forward kernel), by every lane for its cell, outside any divergent block.
  constexpr int PF = 4;
  float q_i[PF], q_f[PF], q_g[PF], q_o[PF], q_c[PF], q_cp[PF], q_d[PF];
  float n_i[PF], n_f[PF], n_g[PF], n_o[PF], n_c[PF], n_cp[PF], n_d[PF];
  const int last = len > 0 ? len - 1 : 0;
  auto fetch = [&](int step, float& vi, float& vf, float& vg, float& vo, float& vc, float& vcp, float& vd) {
    const int sc = min(step, last);
    const int t = rev ? sc : last - sc;
    const int tprev = min(max(rev ? t + 1 : t - 1, 0), T - 1);
    const float4 gv = *reinterpret_cast<const float4*>(gb + t * LG + 4 * k);
    vi = gv.x, vf = gv.y, vg = gv.z, vo = gv.w;
    vc = cb[t * LH + k];
    const float cp = cb[tprev * LH + k];
    vcp = (sc + 1 < len) ? cp : 0.f;
    vd = doutb[t * ndir * LH + k];
  };
#pragma unroll
  for (int u = 0; u < PF; ++u) fetch(u, q_i[u], q_f[u], q_g[u], q_o[u], q_c[u], q_cp[u], q_d[u]);
  int cur = 0;
  for (int step0 = 0; step0 < len; step0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) fetch(step0 + PF + u, n_i[u], n_f[u], n_g[u], n_o[u], n_c[u], n_cp[u], n_d[u]);
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int step = step0 + u;
      if (step >= len) break;
      const int t = rev ? step : len - 1 - step;
      {
        const float ig = q_i[u], fg = q_f[u], gg = q_g[u], og = q_o[u], cc = q_c[u], cprev = q_cp[u];
        const float dh = q_d[u] + dh_rec;
        const float tc = lstm_tanh<BF16>(cc);
        dc = dc + dh * og * (1.f - tc * tc);
        // lane kq publishes the gradient of gate kq: d(pre) = upstream * local derivative of the gate's activation
        const float up = kq == 0 ? dc * gg : (kq == 1 ? dc * cprev : (kq == 2 ? dc * ig : dh * tc));
        const float av = kq == 0 ? ig : (kq == 1 ? fg : (kq == 2 ? gg : og));
        const float mine = up * (kq == 2 ? (1.f - av * av) : av * (1.f - av));
        dc = dc * fg;
        if (BF16)
          dg_b[cur][kq * LH + k] = (__bf16)mine;
        else
          dg_s[cur][kq * LH + k] = mine;
        dgb[t * LG + kq * LH + k] = mine;
      }
      lstm_barrier();
      {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;  // partial dh of the row's four cells over 32 gradient rows
        if (BF16) {
          const uint4* dp = reinterpret_cast<const uint4*>(&dg_b[cur][l * 32]);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            LstmPack4 d4;
            d4.u = dp[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              a0 = __builtin_amdgcn_fdot2_f32_bf16(wq[4 * q + e], d4.p[e], a0, false);
              a1 = __builtin_amdgcn_fdot2_f32_bf16(wq[16 + 4 * q + e], d4.p[e], a1, false);
              a2 = __builtin_amdgcn_fdot2_f32_bf16(wq[32 + 4 * q + e], d4.p[e], a2, false);
              a3 = __builtin_amdgcn_fdot2_f32_bf16(wq[48 + 4 * q + e], d4.p[e], a3, false);
            }
          }
        } else {
          const float4* dp = reinterpret_cast<const float4*>(&dg_s[cur][l * 32]);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float4 d4 = dp[q];
            const float dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              a0 = fmaf(w[4 * q + e], dd[e], a0);
              a1 = fmaf(w[32 + 4 * q + e], dd[e], a1);
              a2 = fmaf(w[64 + 4 * q + e], dd[e], a2);
              a3 = fmaf(w[96 + 4 * q + e], dd[e], a3);
            }
          }
        }
        a0 += lstm_dpp_xor1(a1);
        a2 += lstm_dpp_xor1(a3);
        a0 += lstm_dpp_xor2(a2);
        a0 += lstm_dpp_half_mirror(a0);
        a0 += lstm_dpp_ror8(a0);
        dh_rec = a0;
      }
      cur ^= 1;
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      q_i[u] = n_i[u], q_f[u] = n_f[u], q_g[u] = n_g[u], q_o[u] = n_o[u];
      q_c[u] = n_c[u], q_cp[u] = n_cp[u], q_d[u] = n_d[u];
    }
  }
  // padded tail contributes nothing
  for (int tt = len; tt < T; ++tt) dgb[(long long)tt * LG + tid] = 0.f;
}

// [round 6] The pair-of-lanes form of the backward recurrence (bf16 mode; see lstm_fwd_pair_kernel): 256 threads = one wave
// per SIMD.  A DPP row of 16 lanes owns EIGHT cells.  Phase A: lane l of the row is (cell g(l & 7), gate pair l >> 3) and
// publishes the gradients of its two gates (i, f | g, o).  After the barrier, phase B: lane l covers the 32 gradient rows
// [32 l, 32 l + 32) for the row's eight cells (256 weights per lane).  Same dot-product issue per SIMD as the quad form
// (512 cycles), half the lanes for everything else.
// The 16 partial sums of each of the 8 cells meet in a HALVING reduction: 4 + 2 + 1 + 1 = 8 DPP adds and no select, where
// an all-reduce of every cell (the first form of this kernel) took 32 adds and a 7-deep select chain to pick the lane's own.
// Lane l keeps its accumulators in SLOTS: slot s belongs to cell s ^ g(l & 7), g(b) = b0 ^ 2 b1 ^ 7 b2 (a bijection of three
// bits, applied when the weights are loaded, once per launch).  The partner of each level then holds the same cell in the
// slot with one bit flipped -- xor 1: g(l ^ 1) = g(l) ^ 1 -> slot s ^ 1; xor 2 -> slot s ^ 2; row_half_mirror = xor 7:
// g(l ^ 7) = g(l) ^ 4 -> slot s ^ 4; row_ror:8 = xor 8: same g, slot 0 -- so every lane keeps the even slots, then slots 0
// and 4, then slot 0, and lanes l and l ^ 8 (the two gate pairs of cell g(l & 7)) both end with that cell's dh.
__global__ __launch_bounds__(256) void lstm_bwd_pair_kernel(const float* __restrict__ dout, const float* __restrict__ whh,
                                                           const int32_t* __restrict__ lens, const float* __restrict__ gates,
                                                           const float* __restrict__ cst, float* __restrict__ dgates, int B,
                                                           int T, int ndir, int reverse_first) {
  __shared__ __attribute__((aligned(16))) __bf16 dg_b[2][LG];
  const int tid = threadIdx.x, row = tid >> 4, l = tid & 15;
  const int g3 = (l & 1) ^ (((l >> 1) & 1) * 2) ^ (((l >> 2) & 1) * 7);  // g(l & 7)
  const int k = row * 8 + g3, gp = l >> 3;  // phase-A identity: cell, gate pair
  const int b = blockIdx.x, dir = blockIdx.y;
  const bool rev = reverse_first ? true : (dir == 1);
  const int len = lens ? min(lens[b], T) : T;
  // phase-B weights: slot s = W_hh[32 l + rr][8 row + (s ^ g3)], s < 8, rr < 32 (pairs over rr)
  lstm_bf16x2 wq[8 * 16];
#pragma unroll
  for (int rr = 0; rr < 32; rr += 2) {
    const float* r0 = whh + ((long long)dir * LG + l * 32 + rr) * LH + row * 8;
    const float4 w00 = *reinterpret_cast<const float4*>(r0), w01 = *reinterpret_cast<const float4*>(r0 + 4);
    const float4 w10 = *reinterpret_cast<const float4*>(r0 + LH), w11 = *reinterpret_cast<const float4*>(r0 + LH + 4);
    const float a0[8] = {w00.x, w00.y, w00.z, w00.w, w01.x, w01.y, w01.z, w01.w};
    const float a1[8] = {w10.x, w10.y, w10.z, w10.w, w11.x, w11.y, w11.z, w11.w};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      // slot c ^ g3 <- cell c: a static register index needs the select over the 8 slots (setup, once per launch)
#pragma unroll
      for (int sl = 0; sl < 8; ++sl)
        if ((c ^ g3) == sl) wq[sl * 16 + rr / 2] = (lstm_bf16x2){(__bf16)a0[c], (__bf16)a1[c]};
    }
  }
  const float* doutb = dout + (long long)b * T * ndir * LH + dir * LH;
  const float* gb = gates + (((long long)dir * B + b) * T) * LG;
  const float* cb = cst + (((long long)dir * B + b) * T) * LH;
  float* dgb = dgates + (((long long)dir * B + b) * T) * LG;
  float dc = 0.f, dh_rec = 0.f;
  constexpr int PF = 4;
  // two register rings used in turn (chunk A computes from ring 0 while ring 1 is being loaded, chunk B the other way
  // round): copying "next" into "current" after every chunk was 7 moves per step on the serial path.  (ONE ring refilled in
  // place, as in the forward kernel, was measured too: 0.547 against 0.525 us per step -- here the four loads and their
  // addresses land inside every step's phase A instead of in front of every fourth; profiles/r06_runLH_*.)
  // (Also measured and refused: forming everything a step derives from its SAVED operands alone -- tanh c, o (1 - tanh^2 c),
  // x (1 - x), ... -- one step ahead, so that only five operations stay behind dh_rec: 0.539 us per step with that work in
  // front of the previous barrier, 0.566 in the shadow of the LDS reads, against 0.525.  One wave per SIMD issues in order:
  // a shorter dependence chain buys nothing, the instruction count is the time; profiles/r06_runLI_*, r06_runLJ_*.)
  struct Ring {
    float i[PF], f[PF], g[PF], o[PF], c[PF], cp[PF], d[PF];
  };
  Ring r0, r1;
  const int last = len > 0 ? len - 1 : 0;
  auto fetch = [&](int step, float& vi, float& vf, float& vg, float& vo, float& vc, float& vcp, float& vd) {
    const int sc = min(step, last);
    const int t = rev ? sc : last - sc;
    const int tprev = min(max(rev ? t + 1 : t - 1, 0), T - 1);
    const float4 gv = *reinterpret_cast<const float4*>(gb + t * LG + 4 * k);
    vi = gv.x, vf = gv.y, vg = gv.z, vo = gv.w;
    vc = cb[t * LH + k];
    const float cp = cb[tprev * LH + k];
    vcp = (sc + 1 < len) ? cp : 0.f;
    vd = doutb[t * ndir * LH + k];
  };
#pragma unroll
  for (int u = 0; u < PF; ++u) fetch(u, r0.i[u], r0.f[u], r0.g[u], r0.o[u], r0.c[u], r0.cp[u], r0.d[u]);
  int cur = 0;
  auto chunk = [&](int step0, const Ring& q, Ring& n) {
#pragma unroll
    for (int u = 0; u < PF; ++u) fetch(step0 + PF + u, n.i[u], n.f[u], n.g[u], n.o[u], n.c[u], n.cp[u], n.d[u]);
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int step = step0 + u;
      if (step < len) {  // uniform
        const int t = rev ? step : len - 1 - step;
        {
          const float ig = q.i[u], fg = q.f[u], gg = q.g[u], og = q.o[u], cc = q.c[u], cprev = q.cp[u];
          const float dh = q.d[u] + dh_rec;
          const float tc = lstm_tanh<true>(cc);
          dc = dc + dh * og * (1.f - tc * tc);
          // lane gp publishes the gradients of gates 2 gp, 2 gp + 1: d(pre) = upstream * derivative of the gate's activation
          //   gp = 0: (dc g) i (1 - i), (dc c_prev) f (1 - f);  gp = 1: (dc i) (1 - g g), (dh tanh c) o (1 - o)
          // written as selects of OPERANDS around common products (as two whole expressions per side the compiler built an
          // exec-mask detour: both sides issued, plus the mask bookkeeping, on the serial path of the step)
          const float x1 = gp ? og : fg, u1 = gp ? dh * tc : dc * cprev;
          const float m1 = u1 * (x1 * (1.f - x1));
          const float v0 = gp ? fmaf(-gg, gg, 1.f) : gg * (1.f - ig);
          const float m0 = dc * ig * v0;
          dc = dc * fg;
          dg_b[cur][(2 * gp) * LH + k] = (__bf16)m0;
          dg_b[cur][(2 * gp + 1) * LH + k] = (__bf16)m1;
          float* const d0 = dgb + t * LG + (2 * gp) * LH + k;  // one address, the second store at an immediate offset
          d0[0] = m0;
          d0[LH] = m1;
        }
        lstm_barrier();
        {
          float a[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) a[c] = 0.f;
          const uint4* dp = reinterpret_cast<const uint4*>(&dg_b[cur][l * 32]);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            LstmPack4 d4;
            d4.u = dp[q];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int c = 0; c < 8; ++c) a[c] = __builtin_amdgcn_fdot2_f32_bf16(wq[c * 16 + 4 * q + e], d4.p[e], a[c], false);
          }
          // halving reduction over the row's 16 lanes (see the header): slot s of lane l = cell s ^ g(l & 7)
          a[0] += lstm_dpp_xor1(a[1]);
          a[2] += lstm_dpp_xor1(a[3]);
          a[4] += lstm_dpp_xor1(a[5]);
          a[6] += lstm_dpp_xor1(a[7]);
          a[0] += lstm_dpp_xor2(a[2]);
          a[4] += lstm_dpp_xor2(a[6]);
          a[0] += lstm_dpp_half_mirror(a[4]);
          a[0] += lstm_dpp_ror8(a[0]);
          dh_rec = a[0];
        }
        cur ^= 1;
      }
    }
  };
  for (int step0 = 0; step0 < len; step0 += 2 * PF) {
    chunk(step0, r0, r1);
    if (step0 + PF < len) chunk(step0 + PF, r1, r0);
  }
  // padded tail contributes nothing
  for (int tt = len; tt < T; ++tt) {
    dgb[(long long)tt * LG + tid] = 0.f;
    dgb[(long long)tt * LG + 256 + tid] = 0.f;
  }
}

extern "C" int kantts_lstm_fwd(const float* gx, const float* whh, const float* bhh, const int32_t* lens, float* out,
                               float* gates_save, float* c_save, int B, int T, int H, int ndir, int reverse_first,
                               int precision, void* stream) {
  if (!gx || !whh || !out || !gates_save || !c_save || B < 0 || T < 0 || ndir < 1 || ndir > 2) return KANTTS_E_BADARG;
  if (H != LH) return KANTTS_E_UNSUPPORTED;
  if ((long long)T * ndir * LG >= (1ll << 31)) return KANTTS_E_UNSUPPORTED;  // 32-bit offsets inside one sequence
  if (B == 0 || T == 0) return KANTTS_OK;
  static const char* env_pair = getenv("KANTTS_LSTM_PAIR");  // A/B switch (read once per process): 0 = the quad kernel
  if (precision == 1 && !(env_pair && atoi(env_pair) == 0))
    hipLaunchKernelGGL(lstm_fwd_pair_kernel, dim3(B, ndir), dim3(256), 0, (hipStream_t)stream, gx, whh, bhh, lens, out,
                       gates_save, c_save, B, T, ndir, reverse_first);
  else if (precision == 1)
    hipLaunchKernelGGL(lstm_fwd_kernel<true>, dim3(B, ndir), dim3(LG), 0, (hipStream_t)stream, gx, whh, bhh, lens, out,
                       gates_save, c_save, B, T, ndir, reverse_first);
  else
    hipLaunchKernelGGL(lstm_fwd_kernel<false>, dim3(B, ndir), dim3(LG), 0, (hipStream_t)stream, gx, whh, bhh, lens, out,
                       gates_save, c_save, B, T, ndir, reverse_first);
  KANTTS_CHECK_LAUNCH();
}

extern "C" int kantts_lstm_bwd(const float* dout, const float* whh, const int32_t* lens, const float* gates_save,
                               const float* c_save, float* dgates, int B, int T, int H, int ndir, int reverse_first,
                               int precision, void* stream) {
  if (!dout || !whh || !gates_save || !c_save || !dgates || B < 0 || T < 0 || ndir < 1 || ndir > 2)
    return KANTTS_E_BADARG;
  if (H != LH) return KANTTS_E_UNSUPPORTED;
  if ((long long)T * ndir * LG >= (1ll << 31)) return KANTTS_E_UNSUPPORTED;  // 32-bit offsets inside one sequence
  if (B == 0 || T == 0) return KANTTS_OK;
  static const char* env_pair = getenv("KANTTS_LSTM_PAIR_BWD");  // A/B switch (read once per process): 0 = the quad kernel
  if (precision == 1 && !(env_pair && atoi(env_pair) == 0))
    hipLaunchKernelGGL(lstm_bwd_pair_kernel, dim3(B, ndir), dim3(256), 0, (hipStream_t)stream, dout, whh, lens,
                       gates_save, c_save, dgates, B, T, ndir, reverse_first);
  else if (precision == 1)
    hipLaunchKernelGGL(lstm_bwd_kernel<true>, dim3(B, ndir), dim3(LG), 0, (hipStream_t)stream, dout, whh, lens,
                       gates_save, c_save, dgates, B, T, ndir, reverse_first);
  else
    hipLaunchKernelGGL(lstm_bwd_kernel<false>, dim3(B, ndir), dim3(LG), 0, (hipStream_t)stream, dout, whh, lens,
                       gates_save, c_save, dgates, B, T, ndir, reverse_first);
  KANTTS_CHECK_LAUNCH();
}


// ---------------------------------------------------------------------------------------------------------
// One LSTM cell update for stepwise (free-running) inference: gates = x W_ih^T + h W_hh^T + b (computed by the
// two-segment GEMM), PyTorch gate order [i | f | g | o].  VarRnnARPredictor.infer, kantts/models/sambert/adaptors.py:67-83.
__global__ __launch_bounds__(256) void lstm_cell_kernel(const float* __restrict__ gates, const float* __restrict__ c_prev,
                                                        float* __restrict__ h_out, float* __restrict__ c_out, int B, int H) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * H) return;
  const int b = t / H, j = t % H;
  const float* g = gates + (long long)b * 4 * H;
  const float gi = 1.f / (1.f + expf(-g[j]));
  const float gf = 1.f / (1.f + expf(-g[H + j]));
  const float gg = tanhf(g[2 * H + j]);
  const float go = 1.f / (1.f + expf(-g[3 * H + j]));
  const float c = gf * (c_prev ? c_prev[t] : 0.f) + gi * gg;
  c_out[t] = c;
  h_out[t] = go * tanhf(c);
}

extern "C" int kantts_lstm_cell(const float* gates, const float* c_prev, float* h_out, float* c_out, int B, int H,
                                void* stream) {
  if (!gates || !h_out || !c_out || B < 0 || H < 1) return KANTTS_E_BADARG;
  if (B == 0) return KANTTS_OK;
  hipLaunchKernelGGL(lstm_cell_kernel, dim3(kantts_cdiv((long long)B * H, 256)), dim3(256), 0, (hipStream_t)stream, gates,
                     c_prev, h_out, c_out, B, H);
  KANTTS_CHECK_LAUNCH();
}
