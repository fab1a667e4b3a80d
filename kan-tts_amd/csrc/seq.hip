// Sequence plumbing kernels of the SAM-BERT path (all HBM/L2-bound gathers and FIRs):
//   * embedding gather-sum (+ sqrt(d) scale + sinusoid table add)  kantts_sambert.py:308-329, :62-64
//   * length regulator as an index gather + segment-sum backward    adaptors.py:15-36
//   * duration-relative positions                                   positions.py:72-90
//   * FSMN memory block: depth-wise FIR (k taps) + residual + masks fsmn.py:43-72
#include "common.h"

// ------------------------------------------------------------------------------------------------
// out[row, :] = scale * sum_k table_k[ids[row, k], :]  (+ pos[(row % T), :]); optional copy of the
// scaled sum before the positional add (the reference returns it as `ling_embedding`).
struct EmbedArgs {
  const float* tab[4];
  int ntab;
  const int64_t* ids;  // (rows, ntab)
  const float* pos;    // optional (>=T, D)
  float* out;
  float* scaled;  // optional
  int rows, T, D;
  float scale;
};

__global__ void embed_sum_kernel(const EmbedArgs a) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)a.rows * a.D;
  if (idx >= total) return;
  int row = (int)(idx / a.D), d = (int)(idx % a.D);
  float s = 0.f;
  for (int k = 0; k < a.ntab; ++k) s += a.tab[k][(long long)a.ids[(long long)row * a.ntab + k] * a.D + d];
  s *= a.scale;
  if (a.scaled) a.scaled[idx] = s;
  if (a.pos) s += a.pos[(long long)(row % a.T) * a.D + d];
  a.out[idx] = s;
}

struct EmbedBwdArgs {
  float* dtab[4];
  int ntab;
  const int64_t* ids;
  const float* dout;
  int rows, D;
  float scale;
};

// One thread per (chunk of EB_CHUNK rows, table, channel).  [round 4] The first form issued one global atomic per
// (row, table, channel): the tone / syllable-flag / word-segment / speaker / emotion tables have a handful of rows, so all
// 2 048 tokens of a batch hit the same few hundred addresses -- 70 us of serialised atomics in the middle of the captured
// step's critical path (profiles/r04_runH).  Now a thread walks its chunk with an EB_WAYS-entry accumulator cache keyed by
// the id (registers; all channels of a (chunk, table) take the same branches), and only evictions and the final flush
// reach memory: <= EB_WAYS atomics per chunk and distinct id for the small tables, at most one per row for the phoneme
// table (where the addresses differ anyway).
#define EB_CHUNK 32
#define EB_WAYS 8
__global__ __launch_bounds__(256) void embed_sum_bwd_kernel(const EmbedBwdArgs a) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int d = (int)(idx % a.D);
  const long long r = idx / a.D;
  const int k = (int)(r % a.ntab);
  const long long row0 = (r / a.ntab) * EB_CHUNK;
  if (row0 >= a.rows) return;
  float* __restrict__ tab = a.dtab[k];
  if (!tab) return;
  long long tag[EB_WAYS];
  float acc[EB_WAYS];
#pragma unroll
  for (int w = 0; w < EB_WAYS; ++w) {
    tag[w] = -1;
    acc[w] = 0.f;
  }
  int next = 0;
  const long long row1 = min((long long)a.rows, row0 + EB_CHUNK);
  // [round 5] ids and gradients of eight rows are requested before the first is used: one row per trip was one memory
  // round trip per row (0.7 us x 32 rows = 22 us per launch, three launches at the very end of the backward chain)
  for (long long rowb = row0; rowb < row1; rowb += 8) {
   long long idv[8];
   float gv[8];
#pragma unroll
   for (int u = 0; u < 8; ++u) {
     const long long rr = min(rowb + u, row1 - 1);
     idv[u] = a.ids[rr * a.ntab + k];
     gv[u] = a.dout[rr * a.D + d] * a.scale;
   }
#pragma unroll
   for (int u = 0; u < 8; ++u) {
    if (rowb + u >= row1) break;
    const long long id = idv[u];
    const float g = gv[u];
    bool hit = false;
#pragma unroll
    for (int w = 0; w < EB_WAYS; ++w) {
      const bool m = tag[w] == id;
      acc[w] += m ? g : 0.f;
      hit |= m;
    }
    if (!hit) {  // evict way `next` (uniform across the channels of this chunk and table)
#pragma unroll
      for (int w = 0; w < EB_WAYS; ++w) {
        if (w == next) {
          if (tag[w] >= 0) atomicAdd(&tab[tag[w] * a.D + d], acc[w]);
          tag[w] = id;
          acc[w] = g;
        }
      }
      next = (next + 1) & (EB_WAYS - 1);
    }
   }
  }
#pragma unroll
  for (int w = 0; w < EB_WAYS; ++w)
    if (tag[w] >= 0) atomicAdd(&tab[tag[w] * a.D + d], acc[w]);
}

extern "C" int kantts_embed_sum_fwd(const float* const* tables_host, int ntab, const int64_t* ids, const float* pos,
                                    float* out, float* scaled_out, int rows, int T, int D, float scale, void* stream) {
  if (!tables_host || ntab < 1 || ntab > 4 || !ids || !out || rows < 0 || D < 1 || T < 1) return KANTTS_E_BADARG;
  if (rows == 0) return KANTTS_OK;
  EmbedArgs a = {};
  for (int k = 0; k < ntab; ++k) a.tab[k] = tables_host[k];
  a.ntab = ntab; a.ids = ids; a.pos = pos; a.out = out; a.scaled = scaled_out; a.rows = rows; a.T = T; a.D = D;
  a.scale = scale;
  hipLaunchKernelGGL(embed_sum_kernel, dim3(kantts_cdiv((long long)rows * D, 256)), dim3(256), 0, (hipStream_t)stream, a);
  KANTTS_CHECK_LAUNCH();
}

extern "C" int kantts_embed_sum_bwd(float* const* dtables_host, int ntab, const int64_t* ids, const float* dout,
                                    int rows, int D, float scale, void* stream) {
  if (!dtables_host || ntab < 1 || ntab > 4 || !ids || !dout || rows < 0 || D < 1) return KANTTS_E_BADARG;
  if (rows == 0) return KANTTS_OK;
  EmbedBwdArgs a = {};
  for (int k = 0; k < ntab; ++k) a.dtab[k] = dtables_host[k];
  a.ntab = ntab; a.ids = ids; a.dout = dout; a.rows = rows; a.D = D; a.scale = scale;
  const long long threads = (long long)kantts_cdiv(rows, EB_CHUNK) * ntab * D;
  hipLaunchKernelGGL(embed_sum_bwd_kernel, dim3(kantts_cdiv(threads, 256)), dim3(256), 0, (hipStream_t)stream, a);
  KANTTS_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// Length-regulator index: reps = (dur + 0.5) truncated; frame t of row b belongs to token n iff
// cs[n] <= t < cs[n+1].  One wave per batch row: lane-strided tokens, sequential prefix by lane 0
// would be slow for long rows, so a wave scan is used.
//   idx  (B, Tp) int32: token index or -1      pos (B, Tp) float: t - start + 1 (t+1 when uncovered)
//   cs   (B, N+1) int32 exclusive prefix        lens (B) int64 = cs[N]
__global__ __launch_bounds__(64) void lr_index_kernel(const int64_t* __restrict__ dur_i, const float* __restrict__ dur_f,
                                                      int32_t* __restrict__ idx, float* __restrict__ pos,
                                                      int32_t* __restrict__ cs, int64_t* __restrict__ lens, int N,
                                                      int Tp) {
  const int b = blockIdx.x, lane = threadIdx.x;
  int32_t* csb = cs + (long long)b * (N + 1);
  // inclusive scan in chunks of 64 tokens
  int carry = 0;
  for (int n0 = 0; n0 < N; n0 += 64) {
    int n = n0 + lane;
    int rep = 0;
    if (n < N) {
      if (dur_i)
        rep = (int)dur_i[(long long)b * N + n];  // (int + 0.5).long() == int for ints >= 0
      else
        rep = (int)(long long)(dur_f[(long long)b * N + n] + 0.5f);
    }
    int v = rep;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      int u = __shfl_up(v, off, 64);
      if (lane >= off) v += u;
    }
    if (n < N) csb[n + 1] = carry + v;
    carry += __shfl(v, 63, 64);
  }
  if (lane == 0) {
    csb[0] = 0;
    lens[b] = carry;
  }
  __syncthreads();  // single wave: makes the cs writes visible to the lanes below (same CU)
  __threadfence_block();
  const int total = carry;
  for (int t = lane; t < Tp; t += 64) {
    int id = -1;
    float p = (float)(t + 1);
    if (t < total) {
      // binary search: largest n with cs[n] <= t
      int lo = 0, hi = N;  // cs[lo] <= t < cs[hi]
      while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (csb[mid] <= t)
          lo = mid;
        else
          hi = mid;
      }
      id = lo;
      p = (float)(t - csb[lo] + 1);
    }
    idx[(long long)b * Tp + t] = id;
    pos[(long long)b * Tp + t] = p;
  }
}

extern "C" int kantts_lr_index(const int64_t* dur_int, const float* dur_float, int32_t* idx, float* pos, int32_t* cs,
                               int64_t* lens, int B, int N, int Tp, void* stream) {
  if ((!dur_int && !dur_float) || !idx || !pos || !cs || !lens || B < 0 || N < 1 || Tp < 0) return KANTTS_E_BADARG;
  if (B == 0) return KANTTS_OK;
  hipLaunchKernelGGL(lr_index_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, dur_int, dur_float, idx, pos, cs,
                     lens, N, Tp);
  KANTTS_CHECK_LAUNCH();
}

// out[b,t,:] = x[b, idx[b,t], :] for covered, un-masked frames (t < valid[b]), else 0.
__global__ void lr_gather_kernel(const float* __restrict__ x, const int32_t* __restrict__ idx,
                                 const int64_t* __restrict__ valid, float* __restrict__ out, int B, int N, int Tp, int C,
                                 int ldo, int ooff) {
  long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)B * Tp * C;
  if (g >= total) return;
  int c = (int)(g % C);
  long long bt = g / C;
  int t = (int)(bt % Tp), b = (int)(bt / Tp);
  int id = idx[bt];
  float v = 0.f;
  if (id >= 0 && (!valid || t < (int)valid[b])) v = x[((long long)b * N + id) * C + c];
  out[bt * ldo + ooff + c] = v;
}

// dx[b,n,:] = sum_{t in [cs[n], cs[n+1]) , t < valid[b]} dout[b,t,:]
__global__ void lr_gather_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ cs,
                                     const int64_t* __restrict__ valid, float* __restrict__ dx, int B, int N, int Tp,
                                     int C, int ldo, int ooff, int accumulate) {
  long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)B * N * C;
  if (g >= total) return;
  int c = (int)(g % C);
  long long bn = g / C;
  int n = (int)(bn % N), b = (int)(bn / N);
  int s = cs[(long long)b * (N + 1) + n], e = cs[(long long)b * (N + 1) + n + 1];
  if (e > Tp) e = Tp;
  if (valid && e > (int)valid[b]) e = (int)valid[b];
  float acc = 0.f;
  for (int t = s; t < e; ++t) acc += dout[((long long)b * Tp + t) * ldo + ooff + c];
  if (accumulate)
    dx[g] += acc;
  else
    dx[g] = acc;
}

extern "C" int kantts_lr_gather_fwd(const float* x, const int32_t* idx, const int64_t* valid_lens, float* out, int B,
                                    int N, int Tp, int C, int ldo, int out_col_offset, void* stream) {
  if (!x || !idx || !out || B < 0 || N < 1 || Tp < 0 || C < 1 || ldo < C) return KANTTS_E_BADARG;
  long long total = (long long)B * Tp * C;
  if (total == 0) return KANTTS_OK;
  hipLaunchKernelGGL(lr_gather_kernel, dim3(kantts_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, idx,
                     valid_lens, out, B, N, Tp, C, ldo, out_col_offset);
  KANTTS_CHECK_LAUNCH();
}

extern "C" int kantts_lr_gather_bwd(const float* dout, const int32_t* cs, const int64_t* valid_lens, float* dx, int B,
                                    int N, int Tp, int C, int ldo, int out_col_offset, int accumulate, void* stream) {
  if (!dout || !cs || !dx || B < 0 || N < 1 || Tp < 0 || C < 1 || ldo < C) return KANTTS_E_BADARG;
  long long total = (long long)B * N * C;
  if (total == 0) return KANTTS_OK;
  hipLaunchKernelGGL(lr_gather_bwd_kernel, dim3(kantts_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dout, cs,
                     valid_lens, dx, B, N, Tp, C, ldo, out_col_offset, accumulate);
  KANTTS_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// FSMN memory block (channels-last):  xm = x * keep;  y = keep * (sum_k w[c,k] xm[t+k-lp] + xm[t]) (+ res)
// keep[b,t] = t < lens[b] (all ones when lens == NULL).  Block = one (b, 32-frame tile), thread = channel.
#define DW_TT 32
__global__ void fsmn_dwconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                       const float* __restrict__ res, const int64_t* __restrict__ lens,
                                       float* __restrict__ y, int B, int T, int C, int K, int lp) {
  const int b = blockIdx.y, t0 = blockIdx.x * DW_TT;
  const int len = lens ? (int)min((long long)lens[b], (long long)T) : T;
  const float* xb = x + (long long)b * T * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float* wc = w + (long long)c * K;
    for (int tt = 0; tt < DW_TT; ++tt) {
      const int t = t0 + tt;
      if (t >= T) break;
      float acc = 0.f;
      if (t < len) {
        for (int k = 0; k < K; ++k) {
          int ts = t + k - lp;
          if (ts >= 0 && ts < len) acc = fmaf(wc[k], xb[(long long)ts * C + c], acc);
        }
        acc += xb[(long long)t * C + c];
      }
      long long o = ((long long)b * T + t) * C + c;
      if (res) acc += res[o];
      y[o] = acc;
    }
  }
}

// dx = keep * ( sum_k w[c,k] dyk[t-k+lp] + dyk[t] ),  dyk = dy * keep
__global__ void fsmn_dwconv_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                          const int64_t* __restrict__ lens, float* __restrict__ dx, int B, int T, int C,
                                          int K, int lp) {
  const int b = blockIdx.y, t0 = blockIdx.x * DW_TT;
  const int len = lens ? (int)min((long long)lens[b], (long long)T) : T;
  const float* db = dy + (long long)b * T * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float* wc = w + (long long)c * K;
    for (int tt = 0; tt < DW_TT; ++tt) {
      const int t = t0 + tt;
      if (t >= T) break;
      float acc = 0.f;
      if (t < len) {
        for (int k = 0; k < K; ++k) {
          int ts = t - k + lp;
          if (ts >= 0 && ts < len) acc = fmaf(wc[k], db[(long long)ts * C + c], acc);
        }
        acc += db[(long long)t * C + c];
      }
      dx[((long long)b * T + t) * C + c] = acc;
    }
  }
}

// dw[c,k] += sum_{b,t<len} dy[b,t,c] * xm[b,t+k-lp,c];  block = (b, 128-frame tile), thread = channel
#define DW_WT 128
__global__ void fsmn_dwconv_bwd_dw_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                          const int64_t* __restrict__ lens, float* __restrict__ dw, int B, int T, int C,
                                          int K, int lp) {
  const int b = blockIdx.y, t0 = blockIdx.x * DW_WT;
  const int len = lens ? (int)min((long long)lens[b], (long long)T) : T;
  if (t0 >= len) return;
  const int t1 = min(t0 + DW_WT, len);
  const float* db = dy + (long long)b * T * C;
  const float* xb = x + (long long)b * T * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    for (int k = 0; k < K; ++k) {
      float acc = 0.f;
      for (int t = t0; t < t1; ++t) {
        int ts = t + k - lp;
        if (ts >= 0 && ts < len) acc = fmaf(db[(long long)t * C + c], xb[(long long)ts * C + c], acc);
      }
      atomicAdd(&dw[(long long)c * K + k], acc);
    }
  }
}

// ---- register-window versions (K == 41, every shipped FSMN): thread = channel, 16 consecutive frames
// per thread; the 56-frame input window and the 41 taps live in VGPRs, all loads of a window are
// issued back to back at clamped addresses (no branch per load), rows are contiguous in C so a
// wave reads 256 B per load.  FLIP turns the same code into the input-gradient FIR.
#define FS_K 41
#define FS_TT 16
template <bool FLIP>
__global__ __launch_bounds__(256, 2) void fsmn_fir41_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ res,
                                                        const int64_t* __restrict__ lens, float* __restrict__ y, int B,
                                                        int T, int C, int lp, int xcd_order) {
  // [round 6] XCD-aware order: a block reads a 56-frame window for its 16 output frames, so neighbouring blocks share 40 of
  // their rows -- and consecutive block ids are dealt to the 8 XCDs in turn, each with its own L2: the counters showed
  // 2.6 x the input fetched from HBM / the memory-side cache (profiles/r06_runFINAL_fetch_pmc.txt: 52 MB for 20 MB at the
  // postnet's 19 584 rows).  XCD k (= id % 8) now owns a contiguous band of (sequence, time block) pairs, walked in time order.
  int bx = blockIdx.x, by = blockIdx.y;
  if (xcd_order) {
    const int gx = gridDim.x, total = gx * gridDim.y;
    if (total >= 64) {
      const int L = by * gx + bx, k = L & 7, j = L >> 3;
      const int q = total >> 3, r = total & 7;
      const int vid = k * q + (k < r ? k : r) + j;
      by = vid / gx;
      bx = vid - by * gx;
    }
  }
  const int b = by, t0 = bx * FS_TT;
  const int len = lens ? (int)min((long long)lens[b], (long long)T) : T;
  const int lpe = FLIP ? (FS_K - 1 - lp) : lp;
  const float* xb = x + (long long)b * T * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float wk[FS_K];
#pragma unroll
    for (int k = 0; k < FS_K; ++k) wk[k] = w[(long long)c * FS_K + (FLIP ? (FS_K - 1 - k) : k)];
    float xin[FS_TT + FS_K - 1];
#pragma unroll
    for (int n = 0; n < FS_TT + FS_K - 1; ++n) {
      const int ts = t0 + n - lpe;
      const bool ok = (ts >= 0) && (ts < len);
      const float v = xb[(long long)(ok ? ts : 0) * C + c];
      xin[n] = ok ? v : 0.f;
    }
    float ctr[FS_TT];  // centre tap x[t] (the "+ input" of the memory block); re-read, it is L1-resident
#pragma unroll
    for (int o = 0; o < FS_TT; ++o) {
      const int t = t0 + o;
      const bool ok = t < len;
      const float v = xb[(long long)(ok ? t : 0) * C + c];
      ctr[o] = ok ? v : 0.f;
    }
#pragma unroll
    for (int o = 0; o < FS_TT; ++o) {
      const int t = t0 + o;
      float acc = ctr[o];
#pragma unroll
      for (int k = 0; k < FS_K; ++k) acc = fmaf(wk[k], xin[o + k], acc);
      if (t < T) {
        const long long oidx = ((long long)b * T + t) * C + c;
        float outv = (t < len) ? acc : 0.f;
        if (!FLIP && res) outv += res[oidx];  // (requesting these rows with the window was measured: 19.3 us against 15.9 --
        y[oidx] = outv;                       //  sixteen more live registers cost more than the round trips; r05_runFINAL)
      }
    }
  }
}

// weight gradient partials: block = (chunk of FS_CH frames, b, group of 64 channels); wave w of the block takes the
// w-th quarter of the chunk (two 16-frame windows), lane = channel; the four waves' 41 sums per channel meet in LDS and
// leave as ONE partial row part[(b*nchunk + chunk)][c][k].  (Until round 5 a 256-thread block walked 160 frames alone:
// 512 waves on 1024 SIMDs, ten dependent windows each -- 25.8 us at the postnet's 19 584 rows; this form has 2560 waves
// of two windows.)
#define FS_CH 128
#define FS_SUB (FS_CH / 4)
__global__ __launch_bounds__(256) void fsmn_dw41_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const int64_t* __restrict__ lens,
                                                            float* __restrict__ part, int B, int T, int C, int lp) {
  __shared__ float sm[4][FS_K][65];
  const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c0 = blockIdx.z * 64, c = c0 + lane;
  const bool cok = c < C;
  const int cc = cok ? c : 0;
  const int len = lens ? (int)min((long long)lens[b], (long long)T) : T;
  const float* xb = x + (long long)b * T * C;
  const float* db = dy + (long long)b * T * C;
  const int tbeg = chunk * FS_CH + wave * FS_SUB, tend = min(tbeg + FS_SUB, T);
  float acc[FS_K];
#pragma unroll
  for (int k = 0; k < FS_K; ++k) acc[k] = 0.f;
  for (int t0 = tbeg; t0 < tend && t0 < len; t0 += FS_TT) {
    float xin[FS_TT + FS_K - 1], dv[FS_TT];
#pragma unroll
    for (int n = 0; n < FS_TT + FS_K - 1; ++n) {
      const int ts = t0 + n - lp;
      const bool ok = cok && (ts >= 0) && (ts < len);
      const float v = xb[(long long)(ok ? ts : 0) * C + cc];
      xin[n] = ok ? v : 0.f;
    }
#pragma unroll
    for (int o = 0; o < FS_TT; ++o) {
      const int t = t0 + o;
      const bool ok = cok && (t < tend) && (t < len);
      const float v = db[(long long)(ok ? t : 0) * C + cc];
      dv[o] = ok ? v : 0.f;
    }
    // all 72 loads of the window are issued before the first product: left alone, the scheduler sinks every load to just in
    // front of its first use (`global_load; s_waitcnt vmcnt(0)` 72 times: 21 us per launch whatever the shape)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int o = 0; o < FS_TT; ++o)
#pragma unroll
      for (int k = 0; k < FS_K; ++k) acc[k] = fmaf(dv[o], xin[o + k], acc[k]);
  }
#pragma unroll
  for (int k = 0; k < FS_K; ++k) sm[wave][k][lane] = acc[k];
  __syncthreads();
  // 64 channels x 41 taps are contiguous in the partial row; consecutive threads take consecutive taps (LDS pitch 65)
  float* p = part + (((long long)b * nchunk + chunk) * C + c0) * FS_K;
  const int nout = min(64, C - c0) * FS_K;
  for (int o = threadIdx.x; o < nout; o += 256) {
    const int cl = o / FS_K, k = o - cl * FS_K;
    p[o] = (sm[0][k][cl] + sm[1][k][cl]) + (sm[2][k][cl] + sm[3][k][cl]);
  }
}

// dw[i] += sum over the partial rows, in a fixed order: thread = (output i, one of 8 interleaved slices of the rows), the
// slices meet in LDS.  (One thread per output walking all rows -- 41 workgroups, 128 dependent loads -- took 21.5 us.)
#define FS_RS 8
__global__ __launch_bounds__(64 * FS_RS) void fsmn_dw_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                    int nblk, int n) {
  __shared__ float sm[FS_RS][64];
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const bool ok = i < n;
  const long long ii = ok ? i : 0;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int p = slice;
  for (; p + 3 * FS_RS < nblk; p += 4 * FS_RS) {
    const float a0 = part[(long long)p * n + ii], a1 = part[(long long)(p + FS_RS) * n + ii];
    const float a2 = part[(long long)(p + 2 * FS_RS) * n + ii], a3 = part[(long long)(p + 3 * FS_RS) * n + ii];
    s0 += a0, s1 += a1, s2 += a2, s3 += a3;
  }
  for (; p < nblk; p += FS_RS) s0 += part[(long long)p * n + ii];
  sm[slice][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (slice == 0 && ok) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < FS_RS; ++q) s += sm[q][lane];
    dw[i] += s;
  }
}

static int fs_xcd_order() {
  static const bool off = getenv("KANTTS_FIR_NO_XCD_ORDER") != nullptr;  // A/B switch: block (time, sequence) = (x, y) as dealt
  return off ? 0 : 1;
}

extern "C" long long kantts_fsmn_dwconv_bwd_ws(int B, int T, int C, int K) {
  if (K != FS_K) return 0;
  return (long long)B * kantts_cdiv(T, FS_CH) * C * K;
}

extern "C" int kantts_fsmn_dwconv_fwd(const float* x, const float* w, const float* res, const int64_t* lens, float* y,
                                      int B, int T, int C, int K, int left_pad, void* stream) {
  if (!x || !w || !y || B < 0 || T < 0 || C < 1 || K < 1) return KANTTS_E_BADARG;
  if (B == 0 || T == 0) return KANTTS_OK;
  const int threads = C >= 256 ? 256 : (C >= 128 ? 128 : 64);
  if (K == FS_K) {
    hipLaunchKernelGGL(fsmn_fir41_kernel<false>, dim3(kantts_cdiv(T, FS_TT), B), dim3(threads), 0, (hipStream_t)stream, x, w,
                       res, lens, y, B, T, C, left_pad, fs_xcd_order());
  } else {
    hipLaunchKernelGGL(fsmn_dwconv_fwd_kernel, dim3(kantts_cdiv(T, DW_TT), B), dim3(threads), 0, (hipStream_t)stream, x, w,
                       res, lens, y, B, T, C, K, left_pad);
  }
  KANTTS_CHECK_LAUNCH();
}

extern "C" int kantts_fsmn_dwconv_bwd(const float* dy, const float* x, const float* w, const int64_t* lens, float* dx,
                                      float* dw_accum, float* workspace, long long ws_floats, int B, int T, int C, int K,
                                      int left_pad, void* stream) {
  // dx == NULL / dw_accum == NULL: only the other half (the filter gradient is a leaf of the backward graph -- the host
  // issues it on the weight-gradient stream, beside the critical path)
  if (!dy || !x || !w || (!dx && !dw_accum) || B < 0 || T < 0 || C < 1 || K < 1) return KANTTS_E_BADARG;
  if (B == 0 || T == 0) return KANTTS_OK;
  int threads = C >= 256 ? 256 : (C >= 128 ? 128 : 64);
  hipStream_t st = (hipStream_t)stream;
  if (K == FS_K) {
    const int nchunk = kantts_cdiv(T, FS_CH);
    if (dw_accum && (!workspace || ws_floats < (long long)B * nchunk * C * K)) return KANTTS_E_WORKSPACE;
    if (dx)
      hipLaunchKernelGGL(fsmn_fir41_kernel<true>, dim3(kantts_cdiv(T, FS_TT), B), dim3(threads), 0, st, dy, w,
                         (const float*)nullptr, lens, dx, B, T, C, left_pad, fs_xcd_order());
    if (dw_accum) {
      hipLaunchKernelGGL(fsmn_dw41_partial_kernel, dim3(nchunk, B, kantts_cdiv(C, 64)), dim3(256), 0, st, dy, x, lens,
                         workspace, B, T, C, left_pad);
      hipLaunchKernelGGL(fsmn_dw_reduce_kernel, dim3(kantts_cdiv(C * K, 64)), dim3(64 * FS_RS), 0, st, workspace, dw_accum,
                         B * nchunk, C * K);
    }
  } else {
    if (dx)
      hipLaunchKernelGGL(fsmn_dwconv_bwd_dx_kernel, dim3(kantts_cdiv(T, DW_TT), B), dim3(threads), 0, st, dy, w, lens, dx,
                         B, T, C, K, left_pad);
    if (dw_accum)
      hipLaunchKernelGGL(fsmn_dwconv_bwd_dw_kernel, dim3(kantts_cdiv(T, DW_WT), B), dim3(threads), 0, st, dy, x, lens,
                         dw_accum, B, T, C, K, left_pad);
  }
  KANTTS_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------ [round 5]
// Everything of a teacher-forced SAM-BERT step that depends on the batch's LENGTHS / TARGETS only, in one launch: the three
// padding masks with their clamped lengths (get_mask_from_lengths, kantts/models/utils.py:13-23; the LFR mask of
// kantts_sambert.py:736-750), the duration-position sinusoids of the regulated frames (positions.py:83-98, on the positions
// kantts_lr_index produced), the duration predictor's shifted log input (kantts_sambert.py:466-468), the attention band
// width (:981-985) and the decoder's teacher-forcing frames (:556-559).  These were ~50 stock elementwise launches of 2-5 us
// with a ~4 us dependency gap each -- 0.37 ms at the head of every captured step, before the text encoder's first kernel
// (profiles/r05_runJ_trace_*: the replayed graph runs its branches one after the other).  One flat index space; every
// element is a few integer compares, a sin / cos, a log or a copied float.
__global__ __launch_bounds__(256) void teacher_plan_kernel(const kantts_plan_args g) {
  const long long nA = (long long)g.B * g.N, nB = (long long)g.B * g.T_mel, L = g.Tp / g.r, nC = (long long)g.B * L;
  const long long nD = (long long)g.B * g.Tp * g.depth, nE = (long long)g.B * L * g.d_mel, nF = g.B;
  const long long total = nA + nB + nC + nD + nE + nF;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long k = i;
    if (k < nA) {  // input mask + shifted log durations
      const int b = (int)(k / g.N), n = (int)(k % g.N);
      const long long len = min(g.in_lens[b], (long long)g.N);
      g.in_mask[k] = n >= len ? 1 : 0;
      g.prev[k] = logf((n > 0 ? (float)g.dur[k - 1] : 0.f) + 1.f);
      continue;
    }
    k -= nA;
    if (k < nB) {  // output (mel frame) mask
      const int b = (int)(k / g.T_mel), t = (int)(k % g.T_mel);
      g.out_mask[k] = t >= min(g.out_lens[b], (long long)g.T_mel) ? 1 : 0;
      continue;
    }
    k -= nB;
    if (k < nC) {  // decoder-step (LFR) mask: ceil(len / r) valid steps
      const int b = (int)(k / L), l = (int)(k % L);
      g.lfr_mask[k] = l >= min((g.out_lens[b] + g.r - 1) / g.r, L) ? 1 : 0;
      continue;
    }
    k -= nC;
    if (k < nD) {  // duration-position sinusoid: sin on even channels, cos on odd; position 0 past the sequence's frames
      const int c = (int)(k % g.depth);
      const long long bt = k / g.depth;
      const int b = (int)(bt / g.Tp), t = (int)(bt % g.Tp);
      const long long limit = min(min(g.out_lens[b], (long long)g.T_mel), (long long)g.max_len);
      const float p = t < limit ? g.pos[bt] : 0.f;
      const float e = p / g.inv_ts[c];
      g.pos_enc[k] = (c & 1) ? cosf(e) : sinf(e);
      continue;
    }
    k -= nD;
    if (k < nE) {  // teacher forcing: go frame, then the last frame of every previous decoder step
      const int d = (int)(k % g.d_mel);
      const long long bl = k / g.d_mel;
      const int b = (int)(bl / L), l = (int)(bl % L);
      g.dec_input[k] = l > 0 ? g.mel[((long long)b * g.T_mel + (long long)l * g.r - 1) * g.d_mel + d] : 0.f;
      continue;
    }
    k -= nE;
    {  // clamped lengths in both integer widths
      const int b = (int)k;
      const long long li = min(g.in_lens[b], (long long)g.N), lo = min(g.out_lens[b], (long long)g.T_mel);
      const long long ll = min((g.out_lens[b] + g.r - 1) / g.r, L);
      g.in_l64[b] = li; g.in_l32[b] = (int32_t)li;
      g.out_l64[b] = lo; g.out_l32[b] = (int32_t)lo;
      g.lfr_l64[b] = ll; g.lfr_l32[b] = (int32_t)ll;
      g.valid[b] = min(lo, (long long)g.max_len);
    }
  }
  if (blockIdx.x == 0) {  // band width: int(max valid duration / r + 0.5)
    __shared__ float red[4];
    float m = 0.f;  // masked positions count as 0 (masked_fill(mask, 0).max())
    for (long long k = threadIdx.x; k < nA; k += blockDim.x) {
      const int b = (int)(k / g.N), n = (int)(k % g.N);
      if (n < min(g.in_lens[b], (long long)g.N)) m = fmaxf(m, (float)g.dur[k]);
    }
    m = kantts_wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
      const float v = m / (float)g.r + 0.5f;
      g.bw_val[0] = v;
      g.bw_dev[0] = (int32_t)v;
    }
  }
}

extern "C" int kantts_teacher_plan(const kantts_plan_args* gp, void* stream) {
  if (!gp) return KANTTS_E_BADARG;
  const kantts_plan_args& g = *gp;
  if (g.B < 0 || g.N < 1 || g.T_mel < 1 || g.Tp < 1 || g.r < 1 || g.Tp % g.r || g.depth < 1 || g.d_mel < 1 || g.max_len < 0)
    return KANTTS_E_BADARG;
  if (!g.in_lens || !g.out_lens || !g.dur || !g.mel || !g.pos || !g.inv_ts || !g.in_l64 || !g.in_l32 || !g.in_mask ||
      !g.out_l64 || !g.out_l32 || !g.out_mask || !g.lfr_l64 || !g.lfr_l32 || !g.lfr_mask || !g.valid || !g.pos_enc ||
      !g.prev || !g.bw_val || !g.bw_dev || !g.dec_input)
    return KANTTS_E_BADARG;
  if (g.B == 0) return KANTTS_OK;
  const long long L = g.Tp / g.r;
  const long long total = (long long)g.B * (g.N + g.T_mel + L + (long long)g.Tp * g.depth + L * g.d_mel + 1);
  int blocks = kantts_cdiv(total, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(teacher_plan_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g);
  KANTTS_CHECK_LAUNCH();
}
