// CTC loss of the alignment learner (AttentionCTCLoss, kantts/train/loss.py:481-508 of the reference) and its gradient in
// ONE launch, a workgroup per utterance -- so that the Monotonic-Alignment-Search training step (sambert_16k_MAS*.yaml) can
// be captured in a hipGraph: ATen's ctc_loss copies the lengths to the host.
//
// Reference arithmetic, per utterance b with S = in_lens[b] phonemes and T = out_lens[b] mel frames:
//     logits[t, 0] = blank_logprob (a constant),  logits[t, c] = attn_logprob[b, 0, t, c - 1]   (c = 1..S)
//     lp = log_softmax(logits, classes 0..S);  target = 1, 2, ..., S;  cost_b = CTC(lp, target) / S;  cost = mean_b cost_b
// (torch.nn.CTCLoss(zero_infinity=True), reduction "mean" over a batch of one = division by the target length).
// The target's labels are all different, so the extended sequence  blank 1 blank 2 ... S blank  (2S + 1 states) allows the
// skip s - 2 -> s into EVERY non-blank state, and a class c >= 1 is emitted by exactly one state, s = 2c - 1:
//     alpha_t(s) = lp[t, l(s)] + logsumexp(alpha_{t-1}(s), alpha_{t-1}(s-1), [s odd] alpha_{t-1}(s-2))
//     beta_t(s)  = lp[t, l(s)] + logsumexp(beta_{t+1}(s),  beta_{t+1}(s+1),  [s odd] beta_{t+1}(s+2))
//     nll = -logsumexp(alpha_{T-1}(2S), alpha_{T-1}(2S-1))
//     d nll / d logits[t, c] = exp(lp[t, c]) - exp(alpha_t(2c-1) + beta_t(2c-1) + nll - lp[t, c])          (c >= 1)
// (the blank logit is a constant: no gradient; frames >= T and classes > S: zero).
//
// Layout of the launch: thread = state (up to 4 states per thread: 2S + 1 <= 1024).  The two recurrences walk T steps SIDE BY
// SIDE (alpha on threads 0..255, beta on threads 256..511), each with the
// previous step's row in LDS (double-buffered, one LDS-only barrier per step); each thread's own stream of logits is
// requested CTC_D steps ahead into a register ring, so no step waits for memory; alpha / beta rows go to a global workspace
// with fire-and-forget stores; the gradient is a third, fully parallel pass of the same workgroup.  HBM-light, latency-bound:
// ~0.15 us per step, all utterances side by side.
#include <stdlib.h>

#include "common.h"

#define CTC_THREADS 512
#define CTC_HALF 256 // threads of one recurrence (alpha: 0..255, beta: 256..511)
#define CTC_MAXS 4   // states per thread
#define CTC_D 16     // logit prefetch depth (steps)
#define CTC_NEG (-1e30f)

__device__ __forceinline__ void ctc_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// log(e^a + e^b + e^c).  The largest term contributes exactly 1: the sum is in [1, 3] and v_exp_f32 / v_log_f32 (1-2 ulp) on
// arguments of that size cost ~2e-7 absolute per step -- the same size as the rounding of the running sum itself (|alpha| grows
// to ~2000 at 612 frames: half an ulp is 6e-5).  libm's expf / logf (three + one calls of ~25 instructions on the serial path of
// every step) were half of the step time.
__device__ __forceinline__ float ctc_lse3(float a, float b, float c) {
  const float m = fmaxf(a, fmaxf(b, c));
  if (m <= 0.5f * CTC_NEG) return CTC_NEG;
  return m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
}

// BOTH recurrences at once: threads 0..255 walk alpha (DIR = +1 from t = 0), threads 256..511 walk beta (DIR = -1 from
// t = T - 1), step for step, sharing the per-step barrier (the two are independent until the gradient pass; run one after the
// other they were 0.73 ms of a 9.7 ms MAS step at 612 frames -- profiles/r06_runP_mas_step_kernel_stats_top.csv).  rows: the
// half's global workspace (T1 x NSmax); A0 / A1: the half's two LDS rows of CTC_HALF * CTC_MAXS + 4 floats (2 guard cells on
// either side hold "minus infinity").
__device__ __forceinline__ void ctc_walk_both(const float* __restrict__ lg, int T2, int T, int S, float blank, const float* lseS,
                                              float* __restrict__ rows, int NSmax, float* A0, float* A1) {
  const int tid = threadIdx.x & (CTC_HALF - 1);
  const int DIR = (threadIdx.x >= CTC_HALF) ? -1 : 1;  // (wave-uniform)
  const int NSt = 2 * S + 1;
  int nk = (NSt + CTC_HALF - 1) / CTC_HALF;  // uniform
  nk = __builtin_amdgcn_readfirstlane(nk);
  // this thread's states: s = tid + 256 k; odd s emits class (s + 1) / 2 = logits column (s - 1) / 2
  float ring[CTC_MAXS][CTC_D];
  int col[CTC_MAXS];
#pragma unroll
  for (int k = 0; k < CTC_MAXS; ++k) {
    const int s = tid + CTC_HALF * k;
    col[k] = ((s & 1) && s < NSt) ? (s - 1) >> 1 : -1;
  }
  const int t_first = DIR > 0 ? 0 : T - 1;
  auto fetch = [&](int k, int step) -> float {  // logit of this thread's state k at the step-th step of the walk
    int t = t_first + DIR * step;
    t = min(max(t, 0), T - 1);  // clamped: the load is unconditional, values past the end are never used
    return lg[(long long)t * T2 + max(col[k], 0)];
  };
#pragma unroll
  for (int k = 0; k < CTC_MAXS; ++k)
    if (k < nk) {
#pragma unroll
      for (int d = 0; d < CTC_D; ++d) ring[k][d] = fetch(k, d);
    }
  float* prev = A0;
  float* cur = A1;
  for (int st0 = 0; st0 < T; st0 += CTC_D) {
#pragma unroll
    for (int d = 0; d < CTC_D; ++d) {
      const int step = st0 + d;
      if (step < T) {  // uniform (a `break` here keeps the loop from unrolling: the ring would live in scratch)
        const int t = t_first + DIR * step;
        const float lse = lseS[t];
#pragma unroll
        for (int k = 0; k < CTC_MAXS; ++k) {
          if (k < nk) {
            const int s = tid + CTC_HALF * k;
            const float lp = (col[k] >= 0 ? ring[k][d] : blank) - lse;
            ring[k][d] = fetch(k, step + CTC_D);
            float v;
            if (step == 0) {
              const bool start = DIR > 0 ? (s <= 1) : (s >= NSt - 2);
              v = (start && s < NSt) ? lp : CTC_NEG;
            } else {
              // prev is indexed with a guard of 2 cells on either side
              const float a0 = prev[2 + s];
              const float a1 = prev[2 + s - DIR];
              const float a2 = (s & 1) ? prev[2 + s - 2 * DIR] : CTC_NEG;
              v = ctc_lse3(a0, a1, a2);
              v = (s < NSt && v > 0.5f * CTC_NEG) ? v + lp : CTC_NEG;
            }
            cur[2 + s] = v;
            if (s < NSt) rows[(long long)t * NSmax + s] = v;
          }
        }
        ctc_barrier();
        float* sw = prev;
        prev = cur;
        cur = sw;
      }
    }
  }
}

__global__ __launch_bounds__(CTC_THREADS) void ctc_attn_kernel(const kantts_ctc_args g) {
  // LDS: two alpha and two beta rows (+ guards) and the T1 row normalisers
  constexpr int ROW = CTC_HALF * CTC_MAXS + 4;
  __shared__ float As[4 * ROW];
  __shared__ float red[2];
  extern __shared__ __attribute__((aligned(16))) float lseS[];  // T1 floats
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
  const int T1 = g.T1, T2 = g.T2;
  const int S = min(max(g.in_lens[b], 0), T2), T = min(max(g.out_lens[b], 0), T1);
  const int NSmax = 2 * T2 + 1, NSt = 2 * S + 1;
  const float* lg = g.logits + (long long)b * T1 * T2;
  float* gr = g.grad + (long long)b * T1 * T2;
  float* alpha = g.ws + (long long)b * 2 * T1 * NSmax;
  float* beta = alpha + (long long)T1 * NSmax;
  if (S == 0 || T == 0) {  // torch: an empty target / input gives inf or 0 -> zero_infinity: 0 loss, 0 gradient
    for (long long i = tid; i < (long long)T1 * T2; i += CTC_THREADS) gr[i] = 0.f;
    if (tid == 0) g.loss[b] = 0.f;
    return;
  }
  // ---- phase 0: log-sum-exp of every frame's logits over the classes 0..S (a wave per frame)
  for (int t = wave; t < T; t += CTC_THREADS / 64) {
    float m = g.blank;
    for (int j = lane; j < S; j += 64) m = fmaxf(m, lg[(long long)t * T2 + j]);
    m = kantts_wave_max(m);
    float e = lane == 0 ? expf(g.blank - m) : 0.f;
    for (int j = lane; j < S; j += 64) e += expf(lg[(long long)t * T2 + j] - m);
    e = kantts_wave_sum(e);
    if (lane == 0) lseS[t] = m + logf(e);
  }
  for (int i = tid; i < 4 * ROW; i += CTC_THREADS) As[i] = CTC_NEG;
  __syncthreads();
  // ---- phase 1: alpha (threads 0..255) and beta (threads 256..511) side by side
  {
    const int half = tid >= CTC_HALF;
    ctc_walk_both(lg, T2, T, S, g.blank, lseS, half ? beta : alpha, NSmax, As + half * 2 * ROW, As + half * 2 * ROW + ROW);
  }
  // the last alpha row is in the buffer the walk left as "prev": after an even number of steps the first, odd the second (it
  // started with cur = the second)
  {
    const float* last = (T & 1) ? As + ROW : As;
    if (tid == 0) {
      const float a = last[2 + NSt - 1], c = last[2 + NSt - 2];
      red[0] = -ctc_lse3(a, c, CTC_NEG);
    }
  }
  __syncthreads();  // (also orders this workgroup's global alpha / beta stores before the reads below)
  const float nll = red[0];
  const bool finite = nll < -0.25f * CTC_NEG;  // zero_infinity: an impossible alignment (T < S) costs 0 and has no gradient
  // ---- phase 2: gradient w.r.t. the logits, scaled by grad_scale / S (the reduction of the reference)
  const float sc = g.grad_scale / (float)S;
  for (long long i = tid; i < (long long)T1 * T2; i += CTC_THREADS) {
    const int t = (int)(i / T2), j = (int)(i - (long long)t * T2);
    float v = 0.f;
    if (finite && t < T && j < S) {
      const float lp = lg[i] - lseS[t];
      const int s = 2 * j + 1;
      const float ab = alpha[(long long)t * NSmax + s] + beta[(long long)t * NSmax + s];
      const float occ = ab > 0.5f * CTC_NEG ? expf(ab + nll - lp) : 0.f;
      v = (expf(lp) - occ) * sc;
    }
    gr[i] = v;
  }
  if (tid == 0) g.loss[b] = finite ? nll / (float)S : 0.f;
}

extern "C" long long kantts_ctc_attn_workspace(int B, int T1, int T2) {
  if (B < 0 || T1 < 0 || T2 < 0) return -1;
  return (long long)B * 2 * T1 * (2 * (long long)T2 + 1);
}

extern "C" int kantts_ctc_attn(const kantts_ctc_args* a, void* stream) {
  if (!a || a->B < 0 || a->T1 < 0 || a->T2 < 0) return KANTTS_E_BADARG;
  if (a->B == 0 || a->T1 == 0 || a->T2 == 0) return KANTTS_OK;
  if (!a->logits || !a->in_lens || !a->out_lens || !a->ws || !a->loss || !a->grad) return KANTTS_E_BADARG;
  if (2 * a->T2 + 1 > CTC_HALF * CTC_MAXS) return KANTTS_E_UNSUPPORTED;  // more than 511 phonemes
  const size_t lds = (size_t)a->T1 * sizeof(float);
  if (lds > 96 * 1024) return KANTTS_E_UNSUPPORTED;                          // more than 24576 mel frames
  static bool attr_set = false;
  if (!attr_set && lds > 32 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ctc_attn_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(ctc_attn_kernel, dim3(a->B), dim3(CTC_THREADS), lds, (hipStream_t)stream, *a);
  KANTTS_CHECK_LAUNCH();
}
