// Monotonic alignment search (width 1) on the device: the integer dynamic programme behind
// KanTtsSAMBERT.binarize_attention_parallel (kantts/models/sambert/kantts_sambert.py:752-764), which the reference runs
// on the HOST (attn.cpu().numpy() -> numba b_mas / mas_width1, kantts/models/sambert/alignment.py:32-71 -> .to(device)):
// the only device->host->device crossing inside its forward pass.
//
//   log_p[0][0] = log a[0][0],  log_p[0][j>0] = -inf
//   log_p[i][j] = log a[i][j] + max(log_p[i-1][j], log_p[i-1][j-1])        ties (>=) go to j-1, exactly as the reference
//   backtrack from (To-1, Ti-1); opt[i][path(i)] = 1, and opt[0][0] = 1 (the reference's final assignment)
//
// One workgroup per utterance.  The recurrence is a chain of To dependent rows, so the kernel is latency-, not
// bandwidth-bound, and is built to keep everything off that chain:
//   1. all 1024 lanes of the workgroup take the logs in parallel into a float workspace.  log() is evaluated
//      in double and rounded to float so that the comparisons see the same values as numpy's float32 log up to its
//      (<= 1 ulp) approximation error;
//   2. ONE wave then walks the rows with the previous row of log_p in registers (column r*64+lane), neighbours by
//      DPP/shuffle, no barrier, no stores, the logs of the next 16 rows in flight; the back-pointers of a row are the
//      ballot mask of the comparison -- one 64-bit LDS word per 64 columns;
//   3. lane 0 walks the masks back through LDS (~100 cycles per frame instead of a dependent HBM/L2 load).
// Maps wider than 256 symbols or with more than 64 KB of masks use the row-parallel fallback (LDS ping-pong rows, byte
// back-pointers in a global workspace).
#include "common.h"

#define MAS_THREADS 256

__global__ __launch_bounds__(MAS_THREADS) void mas_block_kernel(const float* __restrict__ attn,
                                                                const int32_t* __restrict__ in_lens,
                                                                const int32_t* __restrict__ out_lens,
                                                                float* __restrict__ opt, uint8_t* __restrict__ ws, int To_max,
                                                                int Ti_max) {
  extern __shared__ float mas_lds[];  // 2 x Ti_max
  const int b = blockIdx.x, tid = threadIdx.x;
  const int Ti = min(in_lens[b], Ti_max), To = min(out_lens[b], To_max);
  const float* a = attn + (long long)b * To_max * Ti_max;
  float* o = opt + (long long)b * To_max * Ti_max;
  uint8_t* w = ws + (long long)b * To_max * Ti_max;
  for (long long e = tid; e < (long long)To_max * Ti_max; e += MAS_THREADS) o[e] = 0.f;
  if (Ti <= 0 || To <= 0) return;
  float* prev = mas_lds;
  float* cur = mas_lds + Ti_max;
  for (int j = tid; j < Ti; j += MAS_THREADS) {
    prev[j] = (j == 0) ? (float)log((double)a[0]) : -INFINITY;
    w[j] = 0;
  }
  __syncthreads();
  for (int i = 1; i < To; ++i) {
    for (int j = tid; j < Ti; j += MAS_THREADS) {
      float p = prev[j];
      uint8_t take = 0;
      if (j >= 1 && prev[j - 1] >= p) {
        p = prev[j - 1];
        take = 1;
      }
      cur[j] = (float)log((double)a[(long long)i * Ti_max + j]) + p;
      w[(long long)i * Ti_max + j] = take;
    }
    __syncthreads();
    float* t = prev;
    prev = cur;
    cur = t;
  }
  __syncthreads();  // back-pointers written by other lanes (global memory, same workgroup)
  if (tid == 0) {
    __threadfence_block();
    int c = Ti - 1;
    for (int i = To - 1; i >= 0; --i) {
      o[(long long)i * Ti_max + c] = 1.f;
      c -= w[(long long)i * Ti_max + c];
    }
    o[0] = 1.f;  // the reference's trailing opt[0, prev_ind[0, .]] = 1 with prev_ind[0, :] == 0
  }
}

#define MASW_THREADS 1024

template <int G, int CH>
__global__ __launch_bounds__(MASW_THREADS) void mas_wave_kernel(const float* __restrict__ attn,
                                                               const int32_t* __restrict__ in_lens,
                                                               const int32_t* __restrict__ out_lens, float* __restrict__ opt,
                                                               float* __restrict__ logs, int To_max, int Ti_max) {
  extern __shared__ unsigned long long mas_bits[];  // To x G ballot masks
  const int b = blockIdx.x, tid = threadIdx.x;
  const int Ti = min(in_lens[b], Ti_max), To = min(out_lens[b], To_max);
  const float* a = attn + (long long)b * To_max * Ti_max;
  float* o = opt + (long long)b * To_max * Ti_max;
  float* lg = logs + (long long)b * To_max * Ti_max;
  // 1. logs of the valid corner (row 0: only column 0 is reachable) by all 16 waves; the output starts as zeros
  for (int e = tid; e < To_max * Ti_max; e += MASW_THREADS) {
    int i = e / Ti_max, j = e - i * Ti_max;
    if (i < To && j < Ti) lg[e] = (i == 0 && j > 0) ? -INFINITY : (float)log((double)a[e]);
    o[e] = 0.f;
  }
  if (Ti <= 0 || To <= 0) return;
  __syncthreads();
  if (tid >= KANTTS_WAVE) return;
  const int lane = tid;
  int col[G];
  float prev[G];
#pragma unroll
  for (int r = 0; r < G; ++r) {
    col[r] = r * 64 + lane;
    prev[r] = (col[r] < Ti) ? lg[col[r]] : -INFINITY;
  }
  if (lane == 0)
    for (int r = 0; r < G; ++r) mas_bits[r] = 0ull;
  // 2. the recurrence, CH rows per trip: the next CH rows of logs are in flight while this trip's rows are consumed
  float buf[CH][G];
#pragma unroll
  for (int k = 0; k < CH; ++k)
#pragma unroll
    for (int r = 0; r < G; ++r)
      buf[k][r] = (1 + k < To && col[r] < Ti) ? lg[(long long)(1 + k) * Ti_max + col[r]] : -INFINITY;
  for (int base = 1; base < To; base += CH) {
    float nb[CH][G];
#pragma unroll
    for (int k = 0; k < CH; ++k)
#pragma unroll
      for (int r = 0; r < G; ++r)
        nb[k][r] = (base + CH + k < To && col[r] < Ti) ? lg[(long long)(base + CH + k) * Ti_max + col[r]] : -INFINITY;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int i = base + k;
      if (i < To) {
        float nv[G];
#pragma unroll
        for (int r = 0; r < G; ++r) {
          const float p0 = prev[r];
          // left neighbour: DPP wave_shr:1 (a VALU move, not an LDS bpermute -- this is the dependent chain); lane 0
          // keeps `old` = column 63 of the previous 64-column group (unused for column 0)
          const int wrap = (r > 0) ? __builtin_amdgcn_readlane(__float_as_int(prev[r > 0 ? r - 1 : 0]), 63) : 0;
          const float p1 = __int_as_float(__builtin_amdgcn_update_dpp(wrap, __float_as_int(p0), 0x138, 0xf, 0xf, false));
          const bool take = (col[r] >= 1) && (p1 >= p0);
          nv[r] = buf[k][r] + (take ? p1 : p0);
          mas_bits[(long long)i * G + r] = __ballot(take && col[r] < Ti);  // same value from every lane
        }
#pragma unroll
        for (int r = 0; r < G; ++r) prev[r] = nv[r];
      }
    }
#pragma unroll
    for (int k = 0; k < CH; ++k)
#pragma unroll
      for (int r = 0; r < G; ++r) buf[k][r] = nb[k][r];
  }
  // 3. backtrack, 64 frames per trip: lane k fetches the masks of frame top-k (one parallel LDS read), the column walks
  //    down through v_readlane on a wave-uniform scalar, lane k keeps the column of its frame and stores its 1
  int c = Ti - 1;
  for (int top = To - 1; top >= 0; top -= 64) {
    const int row = top - lane;
    unsigned int wlo[G], whi[G];
#pragma unroll
    for (int r = 0; r < G; ++r) {
      unsigned long long m = (row >= 0) ? mas_bits[(long long)row * G + r] : 0ull;
      wlo[r] = (unsigned int)m;
      whi[r] = (unsigned int)(m >> 32);
    }
    int myc = 0;
#pragma unroll
    for (int k = 0; k < 64; ++k) {
      c = __builtin_amdgcn_readfirstlane(c);
      if (lane == k) myc = c;
      const int g = c >> 6;
      unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)wlo[0], k);
      unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)whi[0], k);
#pragma unroll
      for (int r = 1; r < G; ++r) {
        unsigned int l2 = (unsigned int)__builtin_amdgcn_readlane((int)wlo[r], k);
        unsigned int h2 = (unsigned int)__builtin_amdgcn_readlane((int)whi[r], k);
        lo = (g == r) ? l2 : lo;
        hi = (g == r) ? h2 : hi;
      }
      const unsigned int word = (c & 32) ? hi : lo;
      c -= (int)((word >> (c & 31)) & 1u);
    }
    if (row >= 0) o[(long long)row * Ti_max + myc] = 1.f;
  }
  if (lane == 0) o[0] = 1.f;  // the reference's trailing opt[0, prev_ind[0, .]] = 1 with prev_ind[0, :] == 0
}

extern "C" int kantts_mas_width1(const float* attn, const int32_t* in_lens, const int32_t* out_lens, float* opt,
                                 void* workspace, int B, int To_max, int Ti_max, void* stream) {
  if (!attn || !in_lens || !out_lens || !opt || !workspace || B < 0 || To_max < 1 || Ti_max < 1) return KANTTS_E_BADARG;
  if ((long long)To_max * Ti_max > 0x7fffffffLL) return KANTTS_E_UNSUPPORTED;
  if (B == 0) return KANTTS_OK;
  hipStream_t s = (hipStream_t)stream;
  const int groups = (Ti_max + 63) / 64;
  const int G = groups <= 1 ? 1 : (groups <= 2 ? 2 : 4);
  const size_t bits = (size_t)To_max * G * sizeof(unsigned long long);
  if (groups <= 4 && bits <= 64 * 1024) {
    float* logs = (float*)workspace;
    if (G == 1)
      hipLaunchKernelGGL((mas_wave_kernel<1, 16>), dim3(B), dim3(MASW_THREADS), bits, s, attn, in_lens, out_lens, opt, logs,
                         To_max, Ti_max);
    else if (G == 2)
      hipLaunchKernelGGL((mas_wave_kernel<2, 16>), dim3(B), dim3(MASW_THREADS), bits, s, attn, in_lens, out_lens, opt, logs,
                         To_max, Ti_max);
    else
      hipLaunchKernelGGL((mas_wave_kernel<4, 8>), dim3(B), dim3(MASW_THREADS), bits, s, attn, in_lens, out_lens, opt, logs,
                         To_max, Ti_max);
    KANTTS_CHECK_LAUNCH();
  }
  if ((size_t)Ti_max * 2 * sizeof(float) > 64 * 1024) return KANTTS_E_UNSUPPORTED;
  hipLaunchKernelGGL(mas_block_kernel, dim3(B), dim3(MAS_THREADS), (size_t)Ti_max * 2 * sizeof(float), s, attn, in_lens,
                     out_lens, opt, (uint8_t*)workspace, To_max, Ti_max);
  KANTTS_CHECK_LAUNCH();
}

// =================================================================================================================
// Alignment attention of the MAS path: ConvAttention.forward after the two projections (kantts/models/sambert/
// attention.py:103-125).  The reference materialises the (B, C, T1, T2) difference tensor (0.6 GB at B=32, 600 frames,
// 100 phonemes, C=80) and runs five elementwise / reduction passes over (B, 1, T1, T2); here one workgroup owns one mel
// frame: the isotropic-Gaussian scores, log_softmax + log prior, the padding mask and the softmax never leave registers.
//
//   d[t2]       = -0.0005 * sum_c (q[t1][c] - k[t2][c])^2
//   logprob[t2] = prior ? log_softmax_{all T2}(d) + log(prior + 1e-8) : d
//   soft[t2]    = softmax over t2 < in_lens[b] of logprob ; 0 on padded text positions
#define AL_THREADS 256
#define AL_MAXR 4  // text positions per lane: T2 <= 1024

__device__ __forceinline__ float al_block_max(float v, float* red) {
  v = kantts_wave_max(v);
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
  for (int i = 1; i < AL_THREADS / 64; ++i) t = fmaxf(t, red[i]);
  return t;
}

__global__ __launch_bounds__(AL_THREADS) void align_attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                    const float* __restrict__ prior,
                                                                    const int32_t* __restrict__ in_lens,
                                                                    float* __restrict__ logprob, float* __restrict__ soft,
                                                                    int T1, int T2, int C) {
  extern __shared__ float al_lds[];  // C floats of q + 8 reduction slots
  float* qs = al_lds;
  float* red = al_lds + C;
  const int b = blockIdx.y, t1 = blockIdx.x, tid = threadIdx.x;
  const float* qr = q + ((long long)b * T1 + t1) * C;
  for (int c = tid; c < C; c += AL_THREADS) qs[c] = qr[c];
  __syncthreads();
  const int len = min(in_lens[b], T2);
  const long long row = ((long long)b * T1 + t1) * T2;
  float d[AL_MAXR];
  float mx = -INFINITY;
#pragma unroll
  for (int r = 0; r < AL_MAXR; ++r) {
    int t2 = tid + r * AL_THREADS;
    d[r] = -INFINITY;
    if (t2 < T2) {
      const float* kr = k + ((long long)b * T2 + t2) * C;
      float acc = 0.f;
      for (int c = 0; c < C; ++c) {
        float df = qs[c] - kr[c];
        acc += df * df;
      }
      d[r] = -0.0005f * acc;
      mx = fmaxf(mx, d[r]);
    }
  }
  if (prior) {
    mx = al_block_max(mx, red);
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < AL_MAXR; ++r)
      if (tid + r * AL_THREADS < T2) s += expf(d[r] - mx);
    s = kantts_block_sum(s, red);
    const float ls = logf(s);
#pragma unroll
    for (int r = 0; r < AL_MAXR; ++r) {
      int t2 = tid + r * AL_THREADS;
      if (t2 < T2) d[r] = (d[r] - mx - ls) + logf(prior[row + t2] + 1e-8f);
    }
  }
  float m2 = -INFINITY;
#pragma unroll
  for (int r = 0; r < AL_MAXR; ++r) {
    int t2 = tid + r * AL_THREADS;
    if (t2 < T2) {
      logprob[row + t2] = d[r];
      if (t2 < len) m2 = fmaxf(m2, d[r]);
    }
  }
  m2 = al_block_max(m2, red);
  float e[AL_MAXR], s2 = 0.f;
#pragma unroll
  for (int r = 0; r < AL_MAXR; ++r) {
    int t2 = tid + r * AL_THREADS;
    e[r] = (t2 < len) ? expf(d[r] - m2) : 0.f;
    s2 += e[r];
  }
  s2 = kantts_block_sum(s2, red);
#pragma unroll
  for (int r = 0; r < AL_MAXR; ++r) {
    int t2 = tid + r * AL_THREADS;
    if (t2 < T2) soft[row + t2] = e[r] / s2;
  }
}

// rows of the backward: gradient w.r.t. the Gaussian scores d
//   g_lp = d_logprob + soft * (d_soft - sum soft*d_soft);   g_d = prior ? g_lp - softmax(d) * sum g_lp : g_lp
__global__ __launch_bounds__(AL_THREADS) void align_attn_bwd_rows_kernel(const float* __restrict__ prior,
                                                                         const float* __restrict__ logprob,
                                                                         const float* __restrict__ soft,
                                                                         const float* __restrict__ d_logprob,
                                                                         const float* __restrict__ d_soft, float* __restrict__ g,
                                                                         int T1, int T2) {
  __shared__ float red[8];
  const int b = blockIdx.y, t1 = blockIdx.x, tid = threadIdx.x;
  const long long row = ((long long)b * T1 + t1) * T2;
  float sv[AL_MAXR], gv[AL_MAXR], dot = 0.f;
#pragma unroll
  for (int r = 0; r < AL_MAXR; ++r) {
    int t2 = tid + r * AL_THREADS;
    sv[r] = 0.f;
    gv[r] = 0.f;
    if (t2 < T2) {
      sv[r] = soft[row + t2];
      if (d_soft) {
        gv[r] = d_soft[row + t2];
        dot += sv[r] * gv[r];
      }
    }
  }
  if (d_soft) dot = kantts_block_sum(dot, red);
  float tot = 0.f;
#pragma unroll
  for (int r = 0; r < AL_MAXR; ++r) {
    int t2 = tid + r * AL_THREADS;
    float v = 0.f;
    if (t2 < T2) {
      v = sv[r] * (gv[r] - dot);
      if (d_logprob) v += d_logprob[row + t2];
    }
    gv[r] = v;
    tot += v;
  }
  if (prior) tot = kantts_block_sum(tot, red);
#pragma unroll
  for (int r = 0; r < AL_MAXR; ++r) {
    int t2 = tid + r * AL_THREADS;
    if (t2 < T2) {
      float v = gv[r];
      if (prior) v -= expf(logprob[row + t2] - logf(prior[row + t2] + 1e-8f)) * tot;
      g[row + t2] = v;
    }
  }
}

// dq[t1][c] = -0.001 * sum_t2 g[t1][t2] (q[t1][c] - k[t2][c])
__global__ __launch_bounds__(AL_THREADS) void align_attn_bwd_q_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                      const float* __restrict__ g, float* __restrict__ dq,
                                                                      long long total, int T1, int T2, int C) {
  long long e = (long long)blockIdx.x * AL_THREADS + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % C);
  const long long bt = e / C;  // b*T1 + t1
  const int b = (int)(bt / T1);
  const float qv = q[e];
  const float* gr = g + bt * T2;
  const float* kb = k + (long long)b * T2 * C + c;
  float acc = 0.f;
  for (int t2 = 0; t2 < T2; ++t2) acc += gr[t2] * (qv - kb[(long long)t2 * C]);
  dq[e] = -0.001f * acc;
}

// dk[t2][c] = +0.001 * sum_t1 g[t1][t2] (q[t1][c] - k[t2][c])
__global__ __launch_bounds__(AL_THREADS) void align_attn_bwd_k_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                      const float* __restrict__ g, float* __restrict__ dk,
                                                                      long long total, int T1, int T2, int C) {
  long long e = (long long)blockIdx.x * AL_THREADS + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % C);
  const long long bt = e / C;  // b*T2 + t2
  const int b = (int)(bt / T2), t2 = (int)(bt % T2);
  const float kv = k[e];
  const float* gb = g + (long long)b * T1 * T2 + t2;
  const float* qb = q + (long long)b * T1 * C + c;
  float acc = 0.f;
  for (int t1 = 0; t1 < T1; ++t1) acc += gb[(long long)t1 * T2] * (qb[(long long)t1 * C] - kv);
  dk[e] = 0.001f * acc;
}

extern "C" int kantts_align_attn_fwd(const float* q, const float* k, const float* prior, const int32_t* in_lens,
                                     float* logprob, float* soft, int B, int T1, int T2, int C, void* stream) {
  if (!q || !k || !in_lens || !logprob || !soft || B < 0 || T1 < 1 || T2 < 1 || C < 1) return KANTTS_E_BADARG;
  if (T2 > AL_THREADS * AL_MAXR || C > 4096 || B > 65535) return KANTTS_E_UNSUPPORTED;
  if (B == 0) return KANTTS_OK;
  hipLaunchKernelGGL(align_attn_fwd_kernel, dim3(T1, B), dim3(AL_THREADS), (size_t)(C + 8) * sizeof(float),
                     (hipStream_t)stream, q, k, prior, in_lens, logprob, soft, T1, T2, C);
  KANTTS_CHECK_LAUNCH();
}

extern "C" int kantts_align_attn_bwd(const float* q, const float* k, const float* prior, const float* logprob,
                                     const float* soft, const float* d_logprob, const float* d_soft, float* g_ws, float* dq,
                                     float* dk, int B, int T1, int T2, int C, void* stream) {
  if (!q || !k || !logprob || !soft || !g_ws || !dq || !dk || B < 0 || T1 < 1 || T2 < 1 || C < 1) return KANTTS_E_BADARG;
  if (T2 > AL_THREADS * AL_MAXR || B > 65535) return KANTTS_E_UNSUPPORTED;
  if (B == 0) return KANTTS_OK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(align_attn_bwd_rows_kernel, dim3(T1, B), dim3(AL_THREADS), 0, s, prior, logprob, soft, d_logprob, d_soft,
                     g_ws, T1, T2);
  long long nq = (long long)B * T1 * C, nk = (long long)B * T2 * C;
  hipLaunchKernelGGL(align_attn_bwd_q_kernel, dim3(kantts_cdiv(nq, AL_THREADS)), dim3(AL_THREADS), 0, s, q, k, g_ws, dq, nq,
                     T1, T2, C);
  hipLaunchKernelGGL(align_attn_bwd_k_kernel, dim3(kantts_cdiv(nk, AL_THREADS)), dim3(AL_THREADS), 0, s, q, k, g_ws, dk, nk,
                     T1, T2, C);
  KANTTS_CHECK_LAUNCH();
}
