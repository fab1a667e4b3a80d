// LSTM recurrence (H = 128) as a persistent per-sequence kernel.
//
// Replaces the time loop of torch.nn.LSTM at
//   kantts/models/sambert/adaptors.py:44-57 (2-layer duration LSTM), :109-134 (packed BiLSTM),
//   kantts/models/sambert/kantts_sambert.py:637-646 (postnet LSTM over T_mel frames).
// The input projection x @ W_ih^T + b_ih for all t is hoisted into one segmented GEMM (MFMA); this
// kernel only runs the sequential part.  One workgroup (512 threads = 8 waves) owns one
// (sequence, direction): thread r keeps row r of W_hh (128 floats) in VGPRs for the whole
// sequence, h_{t-1} lives in LDS and is read as wave-uniform (broadcast) float4s, so a step costs
// 128 FMAs per lane + two workgroup barriers and no HBM traffic besides gx[t] in / h[t] out.
// pack_padded_sequence semantics come from per-sequence lengths: a row only runs t < len (the
// reverse direction starts at len-1) and writes zeros to the padded tail.
//
// Gate order i, f, g, o (torch).  Saved for backward: post-activation gates (B,T,4H) and c (B,T,H).
#include "common.h"

#define LH 128
#define LG 512

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter, i.e. it waits
// until the step's global STORES (saved gates / cell state / outputs: pure outputs, never re-read by this kernel) have
// been acknowledged by L2 -- two or three such round trips on the critical path of every time step.
__device__ __forceinline__ void lstm_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// gx: (B,T,ndir*4H) [direction d at column offset d*4H]; out: (B,T,ndir*H)
// whh: (ndir, 4H, H), bhh: (ndir, 4H); gates_out: (ndir,B,T,4H); c_out: (ndir,B,T,H)
typedef __bf16 lstm_bf16x2 __attribute__((ext_vector_type(2)));
union LstmPack4 {
  uint4 u;
  lstm_bf16x2 p[4];
};

// BF16 = true (throughput mode): the recurrent product h_{t-1} . W_hh[r] runs on packed bf16 pairs with fp32
// accumulation (v_dot2c_f32_bf16): 64 instead of 128 multiply-add instructions and 16 instead of 32 LDS reads per
// step and lane -- the two things the sequential loop is made of.  Gates, cell state and outputs stay fp32.
template <bool BF16>
__global__ __launch_bounds__(LG) void lstm_fwd_kernel(const float* __restrict__ gx, const float* __restrict__ whh,
                                                      const float* __restrict__ bhh, const int32_t* __restrict__ lens,
                                                      float* __restrict__ out, float* __restrict__ gates_out,
                                                      float* __restrict__ c_out, int B, int T, int ndir,
                                                      int reverse_first) {
  __shared__ __attribute__((aligned(16))) float h_s[LH];
  __shared__ __attribute__((aligned(16))) __bf16 h_b[LH];
  __shared__ float g_s[LG];
  const int r = threadIdx.x;
  const int b = blockIdx.x, dir = blockIdx.y;
  const bool rev = reverse_first ? true : (dir == 1);
  const int len = lens ? min(lens[b], T) : T;
  float w[BF16 ? 1 : LH];
  lstm_bf16x2 wq[BF16 ? LH / 2 : 1];
  {
    const float4* wp = reinterpret_cast<const float4*>(whh + ((long long)dir * LG + r) * LH);
#pragma unroll
    for (int k = 0; k < LH / 4; ++k) {
      float4 t = wp[k];
      if (BF16) {
        wq[2 * k] = (lstm_bf16x2){(__bf16)t.x, (__bf16)t.y};
        wq[2 * k + 1] = (lstm_bf16x2){(__bf16)t.z, (__bf16)t.w};
      } else {
        w[4 * k] = t.x;
        w[4 * k + 1] = t.y;
        w[4 * k + 2] = t.z;
        w[4 * k + 3] = t.w;
      }
    }
  }
  const float bias = bhh ? bhh[dir * LG + r] : 0.f;
  if (r < LH) {
    h_s[r] = 0.f;
    h_b[r] = (__bf16)0.f;
  }
  float c = 0.f;
  const long long gx_ld = (long long)ndir * LG;
  const float* gxb = gx + (long long)b * T * gx_ld + dir * LG + r;
  float* outb = out + (long long)b * T * ndir * LH + dir * LH;
  float* gob = gates_out + (((long long)dir * B + b) * T) * LG;
  float* cob = c_out + (((long long)dir * B + b) * T) * LH;
  __syncthreads();
  // The step is a chain of dependent LDS / ALU latencies (~0.3 us); the input projection gx[t] comes from L2 / HBM
  // (0.5-2 us).  Round 1 loaded it one step ahead inside an `if`: hipcc waits for a load issued in a conditional block
  // where the branch re-joins (s_waitcnt vmcnt(0)), so every step paid the full load latency (0.78 us per step over the
  // 612-step postnet sequence).  A register ring with the loads interleaved into the steps fared no better: the vector
  // memory counter retires in order and the loop back-edge makes the compiler's count conservative (vmcnt(1)).  What
  // works is CHUNKS: the gx values of the next LSTM_CH steps are loaded -- unconditionally, steps clamped into the
  // sequence -- at the top of a chunk and first touched a whole chunk later.
  constexpr int PF = 8;
  float gq[PF], gn[PF];
  const int last = len > 0 ? len - 1 : 0;
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const int su = min(u, last);
    gq[u] = gxb[(long long)(rev ? last - su : su) * gx_ld];
  }
  for (int step0 = 0; step0 < len; step0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int sn = min(step0 + PF + u, last);
      gn[u] = gxb[(long long)(rev ? last - sn : sn) * gx_ld];
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int step = step0 + u;
      if (step >= len) break;
      const int t = rev ? len - 1 - step : step;
      const float gcur = gq[u];
      float acc0 = gcur + bias, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
      if (BF16) {
        const uint4* hp = reinterpret_cast<const uint4*>(h_b);
#pragma unroll
        for (int k = 0; k < LH / 8; ++k) {
          LstmPack4 hv;
          hv.u = hp[k];
          acc0 = __builtin_amdgcn_fdot2_f32_bf16(wq[4 * k], hv.p[0], acc0, false);
          acc1 = __builtin_amdgcn_fdot2_f32_bf16(wq[4 * k + 1], hv.p[1], acc1, false);
          acc2 = __builtin_amdgcn_fdot2_f32_bf16(wq[4 * k + 2], hv.p[2], acc2, false);
          acc3 = __builtin_amdgcn_fdot2_f32_bf16(wq[4 * k + 3], hv.p[3], acc3, false);
        }
      } else {
        const float4* hp = reinterpret_cast<const float4*>(h_s);
#pragma unroll
        for (int k = 0; k < LH / 4; ++k) {
          float4 hv = hp[k];
          acc0 = fmaf(w[4 * k], hv.x, acc0);
          acc1 = fmaf(w[4 * k + 1], hv.y, acc1);
          acc2 = fmaf(w[4 * k + 2], hv.z, acc2);
          acc3 = fmaf(w[4 * k + 3], hv.w, acc3);
        }
      }
      const float pre = (acc0 + acc1) + (acc2 + acc3);
      // activation by gate block: rows [0,256) sigmoid (i,f), [256,384) tanh (g), [384,512) sigmoid (o)
      const float act = (r >= 2 * LH && r < 3 * LH) ? tanhf(pre) : sigmoidf_(pre);
      g_s[r] = act;
      gob[(long long)t * LG + r] = act;
      lstm_barrier();
      if (r < LH) {
        const float ig = g_s[r], fg = g_s[LH + r], gg = g_s[2 * LH + r], og = g_s[3 * LH + r];
        c = fmaf(fg, c, ig * gg);
        const float hn = og * tanhf(c);
        if (BF16)
          h_b[r] = (__bf16)hn;
        else
          h_s[r] = hn;
        outb[(long long)t * ndir * LH + r] = hn;
        cob[(long long)t * LH + r] = c;
      }
      lstm_barrier();
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) gq[u] = gn[u];
  }
  // zero the padded tail (pad_packed_sequence) -- outputs only; saved state is never read there
  for (int tt = len; tt < T; ++tt)
    if (r < LH) outb[(long long)tt * ndir * LH + r] = 0.f;
}

// Backward through time: produces dgates_pre (ndir,B,T,4H) (gradient w.r.t. the pre-activation
// gates); the weight / input gradients are GEMMs over it (host layer).
// dout: (B,T,ndir*H).  W_hh^T is held in registers as 4 K-slices: thread (k = tid&127, qd = tid>>7)
// keeps W_hh[qd*128 + rr][k] for rr = 0..127.
template <bool BF16>
__global__ __launch_bounds__(LG) void lstm_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ whh,
                                                      const int32_t* __restrict__ lens,
                                                      const float* __restrict__ gates, const float* __restrict__ cst,
                                                      float* __restrict__ dgates, int B, int T, int ndir,
                                                      int reverse_first) {
  __shared__ __attribute__((aligned(16))) float dg_s[LG];
  __shared__ __attribute__((aligned(16))) __bf16 dg_b[LG];
  __shared__ float part_s[4][LH];
  __shared__ float dh_s[LH];
  const int tid = threadIdx.x;
  const int b = blockIdx.x, dir = blockIdx.y;
  const bool rev = reverse_first ? true : (dir == 1);
  const int len = lens ? min(lens[b], T) : T;
  const int kcol = tid & (LH - 1), qd = tid >> 7;
  float w[BF16 ? 1 : LH];
  lstm_bf16x2 wq[BF16 ? LH / 2 : 1];
#pragma unroll
  for (int rr = 0; rr < LH; rr += 2) {
    const float w0 = whh[((long long)dir * LG + qd * LH + rr) * LH + kcol];
    const float w1 = whh[((long long)dir * LG + qd * LH + rr + 1) * LH + kcol];
    if (BF16) {
      wq[rr / 2] = (lstm_bf16x2){(__bf16)w0, (__bf16)w1};
    } else {
      w[rr] = w0;
      w[rr + 1] = w1;
    }
  }
  const float* doutb = dout + (long long)b * T * ndir * LH + dir * LH;
  const float* gb = gates + (((long long)dir * B + b) * T) * LG;
  const float* cb = cst + (((long long)dir * B + b) * T) * LH;
  float* dgb = dgates + (((long long)dir * B + b) * T) * LG;
  if (tid < LH) dh_s[tid] = 0.f;
  float dc = 0.f;
  __syncthreads();
  // time runs opposite to the forward recurrence.  The seven operands of a step (four saved gates, c_t, c_{t-1}, dout_t)
  // were loaded at the top of the step: a full L2 / HBM round trip on the critical path of every step (1.05 us per step
  // measured).  They are now loaded a CHUNK of PF steps ahead (see the forward kernel): the operands of the next chunk
  // are requested at the top of a chunk by all 512 threads (column tid & 127, step clamped into the sequence), outside
  // any divergent block, and first touched a chunk later.
  constexpr int PF = 4;
  float q_i[PF], q_f[PF], q_g[PF], q_o[PF], q_c[PF], q_cp[PF], q_d[PF];
  float n_i[PF], n_f[PF], n_g[PF], n_o[PF], n_c[PF], n_cp[PF], n_d[PF];
  const int last = len > 0 ? len - 1 : 0;
  const int col = tid & (LH - 1);
  auto fetch = [&](int step, float& vi, float& vf, float& vg, float& vo, float& vc, float& vcp, float& vd) {
    const int sc = min(step, last);
    const int t = rev ? sc : last - sc;
    const int tprev = min(max(rev ? t + 1 : t - 1, 0), T - 1);
    vi = gb[(long long)t * LG + col];
    vf = gb[(long long)t * LG + LH + col];
    vg = gb[(long long)t * LG + 2 * LH + col];
    vo = gb[(long long)t * LG + 3 * LH + col];
    vc = cb[(long long)t * LH + col];
    const float cp = cb[(long long)tprev * LH + col];
    vcp = (sc + 1 < len) ? cp : 0.f;
    vd = doutb[(long long)t * ndir * LH + col];
  };
#pragma unroll
  for (int u = 0; u < PF; ++u) fetch(u, q_i[u], q_f[u], q_g[u], q_o[u], q_c[u], q_cp[u], q_d[u]);
  for (int step0 = 0; step0 < len; step0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) fetch(step0 + PF + u, n_i[u], n_f[u], n_g[u], n_o[u], n_c[u], n_cp[u], n_d[u]);
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int step = step0 + u;
      if (step >= len) break;
      const int t = rev ? step : len - 1 - step;
      if (tid < LH) {
        const float ig = q_i[u], fg = q_f[u], gg = q_g[u], og = q_o[u], cc = q_c[u], cprev = q_cp[u];
        const float dh = q_d[u] + dh_s[tid];
        const float tc = tanhf(cc);
        const float d_o = dh * tc;
        dc = dc + dh * og * (1.f - tc * tc);
        const float d_i = dc * gg, d_g = dc * ig, d_f = dc * cprev;
        const float pi = d_i * ig * (1.f - ig), pf = d_f * fg * (1.f - fg);
        const float pg = d_g * (1.f - gg * gg), po = d_o * og * (1.f - og);
        dc = dc * fg;
        if (BF16) {
          dg_b[tid] = (__bf16)pi;
          dg_b[LH + tid] = (__bf16)pf;
          dg_b[2 * LH + tid] = (__bf16)pg;
          dg_b[3 * LH + tid] = (__bf16)po;
        } else {
          dg_s[tid] = pi;
          dg_s[LH + tid] = pf;
          dg_s[2 * LH + tid] = pg;
          dg_s[3 * LH + tid] = po;
        }
        float* dst = dgb + (long long)t * LG;
        dst[tid] = pi;
        dst[LH + tid] = pf;
        dst[2 * LH + tid] = pg;
        dst[3 * LH + tid] = po;
      }
      lstm_barrier();
      {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (BF16) {
          const uint4* dp = reinterpret_cast<const uint4*>(dg_b + qd * LH);
#pragma unroll
          for (int rr = 0; rr < LH / 8; ++rr) {
            LstmPack4 d4;
            d4.u = dp[rr];
            a0 = __builtin_amdgcn_fdot2_f32_bf16(wq[4 * rr], d4.p[0], a0, false);
            a1 = __builtin_amdgcn_fdot2_f32_bf16(wq[4 * rr + 1], d4.p[1], a1, false);
            a2 = __builtin_amdgcn_fdot2_f32_bf16(wq[4 * rr + 2], d4.p[2], a2, false);
            a3 = __builtin_amdgcn_fdot2_f32_bf16(wq[4 * rr + 3], d4.p[3], a3, false);
          }
        } else {
          const float4* dp = reinterpret_cast<const float4*>(dg_s + qd * LH);
#pragma unroll
          for (int rr = 0; rr < LH / 4; ++rr) {
            float4 d4 = dp[rr];
            a0 = fmaf(w[4 * rr], d4.x, a0);
            a1 = fmaf(w[4 * rr + 1], d4.y, a1);
            a2 = fmaf(w[4 * rr + 2], d4.z, a2);
            a3 = fmaf(w[4 * rr + 3], d4.w, a3);
          }
        }
        part_s[qd][kcol] = (a0 + a1) + (a2 + a3);
      }
      lstm_barrier();
      if (tid < LH) dh_s[tid] = (part_s[0][tid] + part_s[1][tid]) + (part_s[2][tid] + part_s[3][tid]);
      lstm_barrier();
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

extern "C" int kantts_lstm_fwd(const float* gx, const float* whh, const float* bhh, const int32_t* lens, float* out,
                               float* gates_save, float* c_save, int B, int T, int H, int ndir, int reverse_first,
                               int precision, void* stream) {
  if (!gx || !whh || !out || !gates_save || !c_save || B < 0 || T < 0 || ndir < 1 || ndir > 2) return KANTTS_E_BADARG;
  if (H != LH) return KANTTS_E_UNSUPPORTED;
  if (B == 0 || T == 0) return KANTTS_OK;
  if (precision == 1)
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
  if (B == 0 || T == 0) return KANTTS_OK;
  if (precision == 1)
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
