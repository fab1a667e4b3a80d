// Range-limited multi-head attention for d_head = 16 (SAM-BERT: 8 heads x 16).
//
// Replaces ScaledDotProductAttention + the head split/merge permutes + the materialised masks:
//   kantts/models/sambert/__init__.py:17-29 (bmm / temperature / masked_fill(-inf) / softmax / bmm)
//   kantts/models/sambert/__init__.py:85-100 (permute+contiguous x4, mask.repeat)
//   kantts/models/sambert/kantts_sambert.py:135-166 (get_pnca_attn_mask: L x L band masks)
//
// Every mask the reference builds is an index interval [lo, hi] of allowed keys per query:
//   mode 0 (encoder, key padding)  : [0, len-1]                               for every query
//   mode 1 (PNCA x, causal band)   : [max(0, i-bw), i]            if i < len, else [0, L-1]
//   mode 2 (PNCA h, look-ahead band): [i, min(i+bw, L-1, len-1)]  if i < len, else [0, L-1]
// (rows of padded queries are fully un-masked by the reference, kantts_sambert.py:158-164; every caller
// zeroes those rows right afterwards and they carry no gradient, so unless the probabilities are
// requested the band modes skip them: context 0, no gradient -- a full-length attention per padded
// row would otherwise dominate the kernel).
// With d_head = 16 the QK^T contraction is a single MFMA k-step and the decoder bands hold
// ~6 keys, so the op is exp/IO-bound: one thread owns one query row (q, o, running softmax in
// registers), keys/values stream from L2 (wave-uniform addresses in mode 0, neighbouring rows in
// the band modes).  Nothing of size L x L is written unless the caller asks for the probabilities.
//
// Layout: q/k/v/o are addressed as ptr[(b*L + t)*ld + h*16 + d], i.e. straight out of / into the
// fused QKV projection buffers; probs (optional) is (H*B, L, L) head-major like the reference.
#include <stdlib.h>

#include "common.h"

#define DH 16

struct AttnArgs {
  const float* q;
  const float* k;
  const float* v;
  int ldq, ldk, ldv;
  float* o;
  int ldo;
  float* lse;    // (B, H, L) log-sum-exp of the scaled scores (saved for backward)
  float* probs;  // optional (H*B, L, L)
  const int32_t* lens;    // optional (B)
  const int32_t* bw_dev;  // optional device scalar band width (overrides bw)
  int B, H, L, mode, bw;
  float scale;  // 1 / sqrt(d_head)
  float drop_p;
  uint64_t seed;
  const uint64_t* seed_dev;
  // backward only
  const float* d_o;
  int lddo;
  float* dq;
  float* dk;
  float* dv;
  int lddq, lddk, lddv;
  float* dvec;  // (B, H, L): D_i = dO_i . O_i
  int accumulate_dq;
  // grid = (B, H, roles) instead of (H, B, roles): the workgroups of the 8 heads of one sequence then have linear ids
  // b + B*(h + H*z), i.e. (for B a multiple of 8) they all run on XCD b % 8 -- a 128-byte line of a (B*L, 3D) projection
  // holds the 64-byte row pieces of TWO heads, and with heads spread over the XCDs every line was filled into two L2s
  int b_first;
};
#define AT_H(a) ((a).b_first ? (int)blockIdx.y : (int)blockIdx.x)
#define AT_B(a) ((a).b_first ? (int)blockIdx.x : (int)blockIdx.y)

__device__ __forceinline__ void key_range(int mode, int i, int len, int L, int bw, int& lo, int& hi) {
  if (mode == 0) {
    lo = 0;
    hi = len - 1;
  } else if (i >= len) {
    lo = 0;
    hi = L - 1;
  } else if (mode == 1) {
    lo = max(0, i - bw);
    hi = i;
  } else {
    lo = i;
    hi = min(min(i + bw, L - 1), len - 1);
  }
}

__device__ __forceinline__ void load16(const float* p, float* r) {
  const float4* p4 = reinterpret_cast<const float4*>(p);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float4 t = p4[e];
    r[4 * e + 0] = t.x;
    r[4 * e + 1] = t.y;
    r[4 * e + 2] = t.z;
    r[4 * e + 3] = t.w;
  }
}
__device__ __forceinline__ void store16(float* p, const float* r) {
  float4* p4 = reinterpret_cast<float4*>(p);
#pragma unroll
  for (int e = 0; e < 4; ++e) p4[e] = make_float4(r[4 * e], r[4 * e + 1], r[4 * e + 2], r[4 * e + 3]);
}
__device__ __forceinline__ float dot16(const float* a, const float* b) {
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < DH; ++d) s = fmaf(a[d], b[d], s);
  return s;
}

// grid: (ceil(L/128), H, B), block 128: thread <-> query row i
__global__ __launch_bounds__(128) void attn_fwd_kernel(const AttnArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int h = blockIdx.y, b = blockIdx.z;
  if (i >= a.L) return;
  const int len = a.lens ? a.lens[b] : a.L;
  const int bw = a.bw_dev ? *a.bw_dev : a.bw;
  const uint64_t seed = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
  int lo, hi;
  key_range(a.mode, i, len, a.L, bw, lo, hi);
  if (a.mode != 0 && i >= len && !a.probs) hi = lo - 1;  // padded query: skipped (see header)
  const long long row = (long long)b * a.L + i;
  float q[DH], o[DH];
  load16(a.q + row * a.ldq + h * DH, q);
#pragma unroll
  for (int d = 0; d < DH; ++d) o[d] = 0.f;
  const float* kb = a.k + (long long)b * a.L * a.ldk + h * DH;
  const float* vb = a.v + (long long)b * a.L * a.ldv + h * DH;
  float m = -INFINITY;
  for (int j = lo; j <= hi; ++j) {
    float kk[DH];
    load16(kb + (long long)j * a.ldk, kk);
    m = fmaxf(m, dot16(q, kk) * a.scale);
  }
  float l = 0.f;
  float* prow = a.probs ? a.probs + (((long long)h * a.B + b) * a.L + i) * a.L : nullptr;
  const uint64_t rng_row = (((uint64_t)h * a.B + b) * a.L + i) * (uint64_t)a.L;
  KanttsDropSeq drop(a.drop_p, seed);  // consecutive keys share a 64-bit hash four at a time
  for (int j = lo; j <= hi; ++j) {
    float kk[DH], vv[DH];
    load16(kb + (long long)j * a.ldk, kk);
    load16(vb + (long long)j * a.ldv, vv);
    float e = expf(dot16(q, kk) * a.scale - m);
    l += e;
    float ed = e * drop.scale(rng_row + j);
#pragma unroll
    for (int d = 0; d < DH; ++d) o[d] = fmaf(ed, vv[d], o[d]);
  }
  const float inv = (hi >= lo) ? 1.f / l : 0.f;
#pragma unroll
  for (int d = 0; d < DH; ++d) o[d] *= inv;
  store16(a.o + row * a.ldo + h * DH, o);
  a.lse[((long long)b * a.H + h) * a.L + i] = (hi >= lo) ? (m + logf(l)) : 0.f;
  if (prow) {
    for (int j = 0; j < a.L; ++j) {
      float p = 0.f;
      if (j >= lo && j <= hi) {
        float kk[DH];
        load16(kb + (long long)j * a.ldk, kk);
        p = expf(dot16(q, kk) * a.scale - m) * inv * drop.scale(rng_row + j);
      }
      prow[j] = p;
    }
  }
}

// dQ (thread <-> query row); also stores D_i for the dK/dV pass.
__global__ __launch_bounds__(128) void attn_bwd_dq_kernel(const AttnArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int h = blockIdx.y, b = blockIdx.z;
  if (i >= a.L) return;
  const int len = a.lens ? a.lens[b] : a.L;
  const int bw = a.bw_dev ? *a.bw_dev : a.bw;
  const uint64_t seed = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
  int lo, hi;
  key_range(a.mode, i, len, a.L, bw, lo, hi);
  if (a.mode != 0 && i >= len) hi = lo - 1;  // padded query rows carry no gradient
  const long long row = (long long)b * a.L + i;
  float q[DH], go[DH], oo[DH], dq[DH];
  load16(a.q + row * a.ldq + h * DH, q);
  load16(a.d_o + row * a.lddo + h * DH, go);
  load16(a.o + row * a.ldo + h * DH, oo);
  const float D = dot16(go, oo);
  const long long sidx = ((long long)b * a.H + h) * a.L + i;
  const float lse = a.lse[sidx];
  a.dvec[sidx] = D;
#pragma unroll
  for (int d = 0; d < DH; ++d) dq[d] = 0.f;
  const float* kb = a.k + (long long)b * a.L * a.ldk + h * DH;
  const float* vb = a.v + (long long)b * a.L * a.ldv + h * DH;
  const uint64_t rng_row = (((uint64_t)h * a.B + b) * a.L + i) * (uint64_t)a.L;
  KanttsDropSeq drop(a.drop_p, seed);
  for (int j = lo; j <= hi; ++j) {
    float kk[DH], vv[DH];
    load16(kb + (long long)j * a.ldk, kk);
    load16(vb + (long long)j * a.ldv, vv);
    float p = expf(dot16(q, kk) * a.scale - lse);
    float dp = dot16(go, vv) * drop.scale(rng_row + j);
    float ds = p * (dp - D) * a.scale;
#pragma unroll
    for (int d = 0; d < DH; ++d) dq[d] = fmaf(ds, kk[d], dq[d]);
  }
  float* dst = a.dq + row * a.lddq + h * DH;
  if (a.accumulate_dq) {
    float old[DH];
    load16(dst, old);
#pragma unroll
    for (int d = 0; d < DH; ++d) dq[d] += old[d];
  }
  store16(dst, dq);
}

// dK, dV (thread <-> key row j): walks the queries whose interval contains j.
__global__ __launch_bounds__(128) void attn_bwd_dkv_kernel(const AttnArgs a) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int h = blockIdx.y, b = blockIdx.z;
  if (j >= a.L) return;
  const int len = a.lens ? a.lens[b] : a.L;
  const int bw = a.bw_dev ? *a.bw_dev : a.bw;
  const uint64_t seed = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
  const long long krow = (long long)b * a.L + j;
  float kk[DH], vv[DH], dk[DH], dv[DH];
  load16(a.k + krow * a.ldk + h * DH, kk);
  load16(a.v + krow * a.ldv + h * DH, vv);
#pragma unroll
  for (int d = 0; d < DH; ++d) {
    dk[d] = 0.f;
    dv[d] = 0.f;
  }
  // candidate queries: two intervals [c0,c1] (banded, valid queries) and [p0,p1] (padded queries)
  int c0, c1, p0 = 1, p1 = 0;  // padded queries contribute nothing (skipped in forward)
  if (a.mode == 0) {
    c0 = 0;
    c1 = a.L - 1;
    p0 = 1;
    p1 = 0;  // empty: mode 0 treats all queries alike
  } else {
    c0 = max(0, j - bw);
    c1 = min(j + bw, len - 1);
  }
  const float* qb = a.q + (long long)b * a.L * a.ldq + h * DH;
  const float* gb = a.d_o + (long long)b * a.L * a.lddo + h * DH;
  const long long sbase = ((long long)b * a.H + h) * a.L;
  for (int pass = 0; pass < 2; ++pass) {
    const int s = pass ? p0 : c0, e = pass ? p1 : c1;
    for (int i = s; i <= e; ++i) {
      int lo, hi;
      key_range(a.mode, i, len, a.L, bw, lo, hi);
      if (j < lo || j > hi) continue;
      float q[DH], go[DH];
      load16(qb + (long long)i * a.ldq, q);
      load16(gb + (long long)i * a.lddo, go);
      const float p = expf(dot16(q, kk) * a.scale - a.lse[sbase + i]);
      const float dsc = kantts_dropout_scale(a.drop_p, seed, ((((uint64_t)h * a.B + b) * a.L + i) * (uint64_t)a.L) + j);
      const float pd = p * dsc;
      const float dp = dot16(go, vv) * dsc;
      const float ds = p * (dp - a.dvec[sbase + i]) * a.scale;
#pragma unroll
      for (int d = 0; d < DH; ++d) {
        dv[d] = fmaf(pd, go[d], dv[d]);
        dk[d] = fmaf(ds, q[d], dk[d]);
      }
    }
  }
  store16(a.dk + krow * a.lddk + h * DH, dk);
  store16(a.dv + krow * a.lddv + h * DH, dv);
}

// ---------------------------------------------------------------------------------------------
// LDS-staged versions: one workgroup per (batch, head) stages the K/V (or Q/dO) rows of that head
// once (coalesced float4 rows, 17-float LDS row stride => conflict-free when neighbouring lanes read
// neighbouring rows, broadcast when all lanes read one row) and every thread then walks its interval
// out of LDS.  The direct-from-global kernels above are latency-bound (two dependent L2 round trips
// per key); these are the ones used whenever the head fits in 64 KB of LDS (L <= ~390).
#define AT_LD 20  // 80-byte rows: 16-byte LDS accesses, conflict-free for neighbouring rows in neighbouring lanes
#define AT_THREADS 256

// a staged row (16 floats at a 16-byte aligned LDS address) into registers: four ds_read_b128
__device__ __forceinline__ void lds16(const float* p, float* r) {
  const float4* p4 = reinterpret_cast<const float4*>(p);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float4 t = p4[e];
    r[4 * e + 0] = t.x;
    r[4 * e + 1] = t.y;
    r[4 * e + 2] = t.z;
    r[4 * e + 3] = t.w;
  }
}
__device__ __forceinline__ float dot16l(const float* a, const float* lrow) {
  float b[DH];
  lds16(lrow, b);
  return dot16(a, b);
}

__device__ __forceinline__ void stage_rows(const float* __restrict__ src, int ld, int L, float* __restrict__ dst) {
  for (int idx = threadIdx.x; idx < L * 4; idx += AT_THREADS) {
    const int row = idx >> 2, part = idx & 3;
    const float4 t = *reinterpret_cast<const float4*>(src + (long long)row * ld + part * 4);
    *reinterpret_cast<float4*>(dst + row * AT_LD + part * 4) = t;
  }
}

// Bodies of the LDS-staged kernels for head h of sequence b.  The first query / key row a thread owns is requested BEFORE
// the staging loads and the barrier, so the two global round trips overlap instead of following each other.
__device__ __forceinline__ void attn_fwd_lds_body(const AttnArgs& a, const int h, const int b, float* sm) {
  float* Ks = sm;
  float* Vs = sm + a.L * AT_LD;
  float q0[DH];
  {
    const int i0 = min((int)threadIdx.x, a.L - 1);
    load16(a.q + ((long long)b * a.L + i0) * a.ldq + h * DH, q0);
  }
  stage_rows(a.k + (long long)b * a.L * a.ldk + h * DH, a.ldk, a.L, Ks);
  stage_rows(a.v + (long long)b * a.L * a.ldv + h * DH, a.ldv, a.L, Vs);
  __syncthreads();
  const int len = a.lens ? a.lens[b] : a.L;
  const int bw = a.bw_dev ? *a.bw_dev : a.bw;
  const uint64_t seed = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
  for (int i = threadIdx.x; i < a.L; i += AT_THREADS) {
    int lo, hi;
    key_range(a.mode, i, len, a.L, bw, lo, hi);
    if (a.mode != 0 && i >= len && !a.probs) hi = lo - 1;  // padded query: skipped (see header)
    const long long row = (long long)b * a.L + i;
    float q[DH], o[DH];
    if (i == (int)threadIdx.x) {
#pragma unroll
      for (int d = 0; d < DH; ++d) q[d] = q0[d];
    } else {
      load16(a.q + row * a.ldq + h * DH, q);
    }
#pragma unroll
    for (int d = 0; d < DH; ++d) o[d] = 0.f;
    float m = -INFINITY;
    for (int j = lo; j <= hi; ++j) m = fmaxf(m, dot16l(q, Ks + j * AT_LD) * a.scale);
    float l = 0.f;
    const uint64_t rng_row = (((uint64_t)h * a.B + b) * a.L + i) * (uint64_t)a.L;
    KanttsDropSeq drop(a.drop_p, seed);
    for (int j = lo; j <= hi; ++j) {
      const float e = expf(dot16l(q, Ks + j * AT_LD) * a.scale - m);
      l += e;
      const float ed = e * drop.scale(rng_row + j);
      float vv[DH];
      lds16(Vs + j * AT_LD, vv);
#pragma unroll
      for (int d = 0; d < DH; ++d) o[d] = fmaf(ed, vv[d], o[d]);
    }
    const float inv = (hi >= lo) ? 1.f / l : 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) o[d] *= inv;
    store16(a.o + row * a.ldo + h * DH, o);
    a.lse[((long long)b * a.H + h) * a.L + i] = (hi >= lo) ? (m + logf(l)) : 0.f;
    if (a.probs) {
      float* prow = a.probs + (((long long)h * a.B + b) * a.L + i) * a.L;
      for (int j = 0; j < a.L; ++j) {
        float p = 0.f;
        if (j >= lo && j <= hi)
          p = expf(dot16l(q, Ks + j * AT_LD) * a.scale - m) * inv * drop.scale(rng_row + j);
        prow[j] = p;
      }
    }
  }
}

__global__ __launch_bounds__(AT_THREADS) void attn_fwd_lds_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  attn_fwd_lds_body(a, AT_H(a), AT_B(a), sm);
}

__device__ __forceinline__ void attn_bwd_dq_lds_body(const AttnArgs& a, const int h, const int b, float* sm) {
  float* Ks = sm;
  float* Vs = sm + a.L * AT_LD;
  float q0[DH], go0[DH], oo0[DH];
  {
    const long long r0 = (long long)b * a.L + min((int)threadIdx.x, a.L - 1);
    load16(a.q + r0 * a.ldq + h * DH, q0);
    load16(a.d_o + r0 * a.lddo + h * DH, go0);
    load16(a.o + r0 * a.ldo + h * DH, oo0);
  }
  stage_rows(a.k + (long long)b * a.L * a.ldk + h * DH, a.ldk, a.L, Ks);
  stage_rows(a.v + (long long)b * a.L * a.ldv + h * DH, a.ldv, a.L, Vs);
  __syncthreads();
  const int len = a.lens ? a.lens[b] : a.L;
  const int bw = a.bw_dev ? *a.bw_dev : a.bw;
  const uint64_t seed = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
  for (int i = threadIdx.x; i < a.L; i += AT_THREADS) {
    int lo, hi;
    key_range(a.mode, i, len, a.L, bw, lo, hi);
    if (a.mode != 0 && i >= len) hi = lo - 1;  // padded query rows carry no gradient
    const long long row = (long long)b * a.L + i;
    float q[DH], go[DH], oo[DH], dq[DH];
    if (i == (int)threadIdx.x) {
#pragma unroll
      for (int d = 0; d < DH; ++d) {
        q[d] = q0[d];
        go[d] = go0[d];
        oo[d] = oo0[d];
      }
    } else {
      load16(a.q + row * a.ldq + h * DH, q);
      load16(a.d_o + row * a.lddo + h * DH, go);
      load16(a.o + row * a.ldo + h * DH, oo);
    }
    const float D = dot16(go, oo);
    const long long sidx = ((long long)b * a.H + h) * a.L + i;
    const float lse = a.lse[sidx];
    if (a.dvec) a.dvec[sidx] = D;
#pragma unroll
    for (int d = 0; d < DH; ++d) dq[d] = 0.f;
    const uint64_t rng_row = (((uint64_t)h * a.B + b) * a.L + i) * (uint64_t)a.L;
    KanttsDropSeq drop(a.drop_p, seed);
    for (int j = lo; j <= hi; ++j) {
      float kk[DH];
      lds16(Ks + j * AT_LD, kk);
      const float p = expf(dot16(q, kk) * a.scale - lse);
      const float dp = dot16l(go, Vs + j * AT_LD) * drop.scale(rng_row + j);
      const float ds = p * (dp - D) * a.scale;
#pragma unroll
      for (int d = 0; d < DH; ++d) dq[d] = fmaf(ds, kk[d], dq[d]);
    }
    float* dst = a.dq + row * a.lddq + h * DH;
    if (a.accumulate_dq) {
      float old[DH];
      load16(dst, old);
#pragma unroll
      for (int d = 0; d < DH; ++d) dq[d] += old[d];
    }
    store16(dst, dq);
  }
}

__global__ __launch_bounds__(AT_THREADS) void attn_bwd_dq_lds_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  attn_bwd_dq_lds_body(a, AT_H(a), AT_B(a), sm);
}

// D_i = dO_i . O_i: read from dvec (written by the dq pass, which then has to run first) or, with RECOMPUTE_D, formed
// here from the O rows -- the two passes of a backward then are independent and run as roles of ONE launch
template <bool RECOMPUTE_D>
__device__ __forceinline__ void attn_bwd_dkv_lds_body(const AttnArgs& a, const int h, const int b, float* sm) {
  float* Qs = sm;
  float* Gs = sm + a.L * AT_LD;
  float* Ls = Gs + a.L * AT_LD;  // lse
  float* Ds = Ls + a.L;          // dvec
  float kk0[DH], vv0[DH];
  {
    const long long r0 = (long long)b * a.L + min((int)threadIdx.x, a.L - 1);
    load16(a.k + r0 * a.ldk + h * DH, kk0);
    load16(a.v + r0 * a.ldv + h * DH, vv0);
  }
  stage_rows(a.q + (long long)b * a.L * a.ldq + h * DH, a.ldq, a.L, Qs);
  stage_rows(a.d_o + (long long)b * a.L * a.lddo + h * DH, a.lddo, a.L, Gs);
  const long long sbase = ((long long)b * a.H + h) * a.L;
  for (int i = threadIdx.x; i < a.L; i += AT_THREADS) {
    Ls[i] = a.lse[sbase + i];
    if (RECOMPUTE_D) {
      const long long row = (long long)b * a.L + i;
      float go[DH], oo[DH];
      load16(a.d_o + row * a.lddo + h * DH, go);
      load16(a.o + row * a.ldo + h * DH, oo);
      Ds[i] = dot16(go, oo);
    } else {
      Ds[i] = a.dvec[sbase + i];
    }
  }
  __syncthreads();
  const int len = a.lens ? a.lens[b] : a.L;
  const int bw = a.bw_dev ? *a.bw_dev : a.bw;
  const uint64_t seed = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
  for (int j = threadIdx.x; j < a.L; j += AT_THREADS) {
    const long long krow = (long long)b * a.L + j;
    float kk[DH], vv[DH], dk[DH], dv[DH];
    if (j == (int)threadIdx.x) {
#pragma unroll
      for (int d = 0; d < DH; ++d) {
        kk[d] = kk0[d];
        vv[d] = vv0[d];
      }
    } else {
      load16(a.k + krow * a.ldk + h * DH, kk);
      load16(a.v + krow * a.ldv + h * DH, vv);
    }
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      dk[d] = 0.f;
      dv[d] = 0.f;
    }
    int c0, c1, p0 = 1, p1 = 0;  // padded queries contribute nothing (skipped in forward)
    if (a.mode == 0) {
      c0 = 0;
      c1 = a.L - 1;
      p0 = 1;
      p1 = 0;
    } else {
      c0 = max(0, j - bw);
      c1 = min(j + bw, len - 1);
    }
    for (int pass = 0; pass < 2; ++pass) {
      const int s0 = pass ? p0 : c0, e0 = pass ? p1 : c1;
      for (int i = s0; i <= e0; ++i) {
        int lo, hi;
        key_range(a.mode, i, len, a.L, bw, lo, hi);
        if (j < lo || j > hi) continue;
        float q[DH], go[DH];
        lds16(Qs + i * AT_LD, q);
        lds16(Gs + i * AT_LD, go);
        const float p = expf(dot16(q, kk) * a.scale - Ls[i]);
        const float dsc =
            kantts_dropout_scale(a.drop_p, seed, ((((uint64_t)h * a.B + b) * a.L + i) * (uint64_t)a.L) + j);
        const float pd = p * dsc;
        const float dp = dot16(go, vv) * dsc;
        const float ds = p * (dp - Ds[i]) * a.scale;
#pragma unroll
        for (int d = 0; d < DH; ++d) {
          dv[d] = fmaf(pd, go[d], dv[d]);
          dk[d] = fmaf(ds, q[d], dk[d]);
        }
      }
    }
    store16(a.dk + krow * a.lddk + h * DH, dk);
    store16(a.dv + krow * a.lddv + h * DH, dv);
  }
}

// must run after attn_bwd_dq_* (needs dvec)
__global__ __launch_bounds__(AT_THREADS) void attn_bwd_dkv_lds_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  attn_bwd_dkv_lds_body<false>(a, AT_H(a), AT_B(a), sm);
}

// ---------------------------------------------------------------------------------------------
// Key-padding mode (encoder self-attention: every query attends to all len keys) with P lanes per query.  With one
// thread per query an L = 64 encoder head keeps ONE wave of its workgroup busy for 2 x 64 sequential keys (29 us per
// launch, profiles/r03_runR_*); here lane `part` of a query's group takes the 4-key blocks jb = part (mod P) -- blocks, not
// single keys, so that the dropout hash, which covers 4 consecutive keys, is still evaluated once per block -- and the
// partial max / sum / context meet through DPP quad permutes.  Same masks, same dropout stream, summation order differs.
__device__ __forceinline__ float at_dpp_xor1(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float at_dpp_xor2(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
}
__device__ __forceinline__ float at_quad_sum(float v) {
  v += at_dpp_xor1(v);
  v += at_dpp_xor2(v);
  return v;
}
__device__ __forceinline__ float at_quad_max(float v) {
  v = fmaxf(v, at_dpp_xor1(v));
  v = fmaxf(v, at_dpp_xor2(v));
  return v;
}
#define AT_P 4
#define AT_QPR (AT_THREADS / AT_P)  // queries (keys) per round of the workgroup

__device__ __forceinline__ void attn_fwd_lds_quad_body(const AttnArgs& a, const int h, const int b, float* sm) {
  float* Ks = sm;
  float* Vs = sm + a.L * AT_LD;
  const int part = threadIdx.x & (AT_P - 1), iq = threadIdx.x >> 2;
  float q0[DH];
  load16(a.q + ((long long)b * a.L + min(iq, a.L - 1)) * a.ldq + h * DH, q0);
  stage_rows(a.k + (long long)b * a.L * a.ldk + h * DH, a.ldk, a.L, Ks);
  stage_rows(a.v + (long long)b * a.L * a.ldv + h * DH, a.ldv, a.L, Vs);
  __syncthreads();
  const int len = a.lens ? a.lens[b] : a.L;
  const uint64_t seed = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
  const int hi = len - 1;  // mode 0: keys [0, len - 1] for every query
  const int rounds = (a.L + AT_QPR - 1) / AT_QPR;
  for (int r = 0; r < rounds; ++r) {  // all four lanes of a group run every round (DPP needs them), stores are guarded
    const int i = r * AT_QPR + iq;
    const bool live = i < a.L;
    const long long row = (long long)b * a.L + min(i, a.L - 1);
    float q[DH], o[DH];
    if (r == 0) {
#pragma unroll
      for (int d = 0; d < DH; ++d) q[d] = q0[d];
    } else {
      load16(a.q + row * a.ldq + h * DH, q);
    }
#pragma unroll
    for (int d = 0; d < DH; ++d) o[d] = 0.f;
    float m = -INFINITY;
    for (int jb = part; jb * 4 <= hi; jb += AT_P)
      for (int j = jb * 4; j <= min(hi, jb * 4 + 3); ++j) m = fmaxf(m, dot16l(q, Ks + j * AT_LD) * a.scale);
    m = at_quad_max(m);
    float l = 0.f;
    const uint64_t rng_row = (((uint64_t)h * a.B + b) * a.L + min(i, a.L - 1)) * (uint64_t)a.L;
    KanttsDropSeq drop(a.drop_p, seed);
    for (int jb = part; jb * 4 <= hi; jb += AT_P)
      for (int j = jb * 4; j <= min(hi, jb * 4 + 3); ++j) {
        const float e = expf(dot16l(q, Ks + j * AT_LD) * a.scale - m);
        l += e;
        const float ed = e * drop.scale(rng_row + j);
        float vv[DH];
      lds16(Vs + j * AT_LD, vv);
#pragma unroll
        for (int d = 0; d < DH; ++d) o[d] = fmaf(ed, vv[d], o[d]);
      }
    l = at_quad_sum(l);
    const float inv = (hi >= 0) ? 1.f / l : 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) o[d] = at_quad_sum(o[d]) * inv;
    if (live) {
      // lane `part` stores dims [4 part, 4 part + 4): the group writes the 64-byte row piece with one 16-byte store each
      float4 w;
      w.x = part == 0 ? o[0] : part == 1 ? o[4] : part == 2 ? o[8] : o[12];
      w.y = part == 0 ? o[1] : part == 1 ? o[5] : part == 2 ? o[9] : o[13];
      w.z = part == 0 ? o[2] : part == 1 ? o[6] : part == 2 ? o[10] : o[14];
      w.w = part == 0 ? o[3] : part == 1 ? o[7] : part == 2 ? o[11] : o[15];
      *reinterpret_cast<float4*>(a.o + row * a.ldo + h * DH + part * 4) = w;
      if (part == 0) a.lse[((long long)b * a.H + h) * a.L + i] = (hi >= 0) ? (m + logf(l)) : 0.f;
      if (a.probs) {
        float* prow = a.probs + (((long long)h * a.B + b) * a.L + i) * a.L;
        for (int j = part; j < a.L; j += AT_P) {
          float p = 0.f;
          if (j <= hi) p = expf(dot16l(q, Ks + j * AT_LD) * a.scale - m) * inv * kantts_dropout_scale(a.drop_p, seed, rng_row + j);
          prow[j] = p;
        }
      }
    }
  }
}

__device__ __forceinline__ void attn_bwd_dq_lds_quad_body(const AttnArgs& a, const int h, const int b, float* sm) {
  float* Ks = sm;
  float* Vs = sm + a.L * AT_LD;
  const int part = threadIdx.x & (AT_P - 1), iq = threadIdx.x >> 2;
  float q0[DH], go0[DH], oo0[DH];
  {
    const long long r0 = (long long)b * a.L + min(iq, a.L - 1);
    load16(a.q + r0 * a.ldq + h * DH, q0);
    load16(a.d_o + r0 * a.lddo + h * DH, go0);
    load16(a.o + r0 * a.ldo + h * DH, oo0);
  }
  stage_rows(a.k + (long long)b * a.L * a.ldk + h * DH, a.ldk, a.L, Ks);
  stage_rows(a.v + (long long)b * a.L * a.ldv + h * DH, a.ldv, a.L, Vs);
  __syncthreads();
  const int len = a.lens ? a.lens[b] : a.L;
  const uint64_t seed = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
  const int hi = len - 1;
  const int rounds = (a.L + AT_QPR - 1) / AT_QPR;
  for (int r = 0; r < rounds; ++r) {
    const int i = r * AT_QPR + iq;
    const bool live = i < a.L;
    const long long row = (long long)b * a.L + min(i, a.L - 1);
    float q[DH], go[DH], oo[DH], dq[DH];
    if (r == 0) {
#pragma unroll
      for (int d = 0; d < DH; ++d) {
        q[d] = q0[d];
        go[d] = go0[d];
        oo[d] = oo0[d];
      }
    } else {
      load16(a.q + row * a.ldq + h * DH, q);
      load16(a.d_o + row * a.lddo + h * DH, go);
      load16(a.o + row * a.ldo + h * DH, oo);
    }
    const float D = dot16(go, oo);
    const long long sidx = ((long long)b * a.H + h) * a.L + min(i, a.L - 1);
    const float lse = a.lse[sidx];
    if (a.dvec && live && part == 0) a.dvec[sidx] = D;
#pragma unroll
    for (int d = 0; d < DH; ++d) dq[d] = 0.f;
    const uint64_t rng_row = (((uint64_t)h * a.B + b) * a.L + min(i, a.L - 1)) * (uint64_t)a.L;
    KanttsDropSeq drop(a.drop_p, seed);
    for (int jb = part; jb * 4 <= hi; jb += AT_P)
      for (int j = jb * 4; j <= min(hi, jb * 4 + 3); ++j) {
        float kk[DH];
        lds16(Ks + j * AT_LD, kk);
        const float p = expf(dot16(q, kk) * a.scale - lse);
        const float dp = dot16l(go, Vs + j * AT_LD) * drop.scale(rng_row + j);
        const float ds = p * (dp - D) * a.scale;
#pragma unroll
        for (int d = 0; d < DH; ++d) dq[d] = fmaf(ds, kk[d], dq[d]);
      }
#pragma unroll
    for (int d = 0; d < DH; ++d) dq[d] = at_quad_sum(dq[d]);
    if (live) {
      float4 w;
      w.x = part == 0 ? dq[0] : part == 1 ? dq[4] : part == 2 ? dq[8] : dq[12];
      w.y = part == 0 ? dq[1] : part == 1 ? dq[5] : part == 2 ? dq[9] : dq[13];
      w.z = part == 0 ? dq[2] : part == 1 ? dq[6] : part == 2 ? dq[10] : dq[14];
      w.w = part == 0 ? dq[3] : part == 1 ? dq[7] : part == 2 ? dq[11] : dq[15];
      float4* dst = reinterpret_cast<float4*>(a.dq + row * a.lddq + h * DH + part * 4);
      if (a.accumulate_dq) {
        const float4 old = *dst;
        w.x += old.x; w.y += old.y; w.z += old.z; w.w += old.w;
      }
      *dst = w;
    }
  }
}

// key j <-> group, lane `part` walks the queries i = part (mod P); D recomputed from the O rows (see attn_bwd_dkv_lds_body)
__device__ __forceinline__ void attn_bwd_dkv_lds_quad_body(const AttnArgs& a, const int h, const int b, float* sm) {
  float* Qs = sm;
  float* Gs = sm + a.L * AT_LD;
  float* Ls = Gs + a.L * AT_LD;  // lse
  float* Ds = Ls + a.L;          // dvec
  const int part = threadIdx.x & (AT_P - 1), jq = threadIdx.x >> 2;
  float kk0[DH], vv0[DH];
  {
    const long long r0 = (long long)b * a.L + min(jq, a.L - 1);
    load16(a.k + r0 * a.ldk + h * DH, kk0);
    load16(a.v + r0 * a.ldv + h * DH, vv0);
  }
  stage_rows(a.q + (long long)b * a.L * a.ldq + h * DH, a.ldq, a.L, Qs);
  stage_rows(a.d_o + (long long)b * a.L * a.lddo + h * DH, a.lddo, a.L, Gs);
  const long long sbase = ((long long)b * a.H + h) * a.L;
  for (int i = threadIdx.x; i < a.L; i += AT_THREADS) {
    Ls[i] = a.lse[sbase + i];
    const long long row = (long long)b * a.L + i;
    float go[DH], oo[DH];
    load16(a.d_o + row * a.lddo + h * DH, go);
    load16(a.o + row * a.ldo + h * DH, oo);
    Ds[i] = dot16(go, oo);
  }
  __syncthreads();
  const int len = a.lens ? a.lens[b] : a.L;
  const uint64_t seed = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
  const int rounds = (a.L + AT_QPR - 1) / AT_QPR;
  for (int r = 0; r < rounds; ++r) {
    const int j = r * AT_QPR + jq;
    const bool live = j < a.L;
    const long long krow = (long long)b * a.L + min(j, a.L - 1);
    float kk[DH], vv[DH], dk[DH], dv[DH];
    if (r == 0) {
#pragma unroll
      for (int d = 0; d < DH; ++d) {
        kk[d] = kk0[d];
        vv[d] = vv0[d];
      }
    } else {
      load16(a.k + krow * a.ldk + h * DH, kk);
      load16(a.v + krow * a.ldv + h * DH, vv);
    }
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      dk[d] = 0.f;
      dv[d] = 0.f;
    }
    // mode 0: query i sees key j iff j <= len - 1 (every query row, padded ones included, as the forward computes them)
    if (live && j <= len - 1) {
      for (int i = part; i < a.L; i += AT_P) {
        float q[DH], go[DH];
        lds16(Qs + i * AT_LD, q);
        lds16(Gs + i * AT_LD, go);
        const float p = expf(dot16(q, kk) * a.scale - Ls[i]);
        const float dsc =
            kantts_dropout_scale(a.drop_p, seed, ((((uint64_t)h * a.B + b) * a.L + i) * (uint64_t)a.L) + j);
        const float pd = p * dsc;
        const float dp = dot16(go, vv) * dsc;
        const float ds = p * (dp - Ds[i]) * a.scale;
#pragma unroll
        for (int d = 0; d < DH; ++d) {
          dv[d] = fmaf(pd, go[d], dv[d]);
          dk[d] = fmaf(ds, q[d], dk[d]);
        }
      }
    }
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      dk[d] = at_quad_sum(dk[d]);
      dv[d] = at_quad_sum(dv[d]);
    }
    if (live) {
      float4 wk, wv;
      wk.x = part == 0 ? dk[0] : part == 1 ? dk[4] : part == 2 ? dk[8] : dk[12];
      wk.y = part == 0 ? dk[1] : part == 1 ? dk[5] : part == 2 ? dk[9] : dk[13];
      wk.z = part == 0 ? dk[2] : part == 1 ? dk[6] : part == 2 ? dk[10] : dk[14];
      wk.w = part == 0 ? dk[3] : part == 1 ? dk[7] : part == 2 ? dk[11] : dk[15];
      wv.x = part == 0 ? dv[0] : part == 1 ? dv[4] : part == 2 ? dv[8] : dv[12];
      wv.y = part == 0 ? dv[1] : part == 1 ? dv[5] : part == 2 ? dv[9] : dv[13];
      wv.z = part == 0 ? dv[2] : part == 1 ? dv[6] : part == 2 ? dv[10] : dv[14];
      wv.w = part == 0 ? dv[3] : part == 1 ? dv[7] : part == 2 ? dv[11] : dv[15];
      *reinterpret_cast<float4*>(a.dk + krow * a.lddk + h * DH + part * 4) = wk;
      *reinterpret_cast<float4*>(a.dv + krow * a.lddv + h * DH + part * 4) = wv;
    }
  }
}

// Query gradient of BOTH bands of a PNCA block in one pass (the bands share the queries): K/V of the x band and of the
// memory band are staged side by side, a thread owns a query, walks one band after the other and stores the sum --
// instead of two passes into two buffers and an elementwise add over (B*L, D) per block and step.
__device__ __forceinline__ void attn_bwd_dq2_lds_body(const AttnArgs& ax, const AttnArgs& ah, const int h, const int b,
                                                      float* sm) {
  // the two bands use the SAME two LDS images one after the other (x band, barrier, memory band): the role then needs no
  // more LDS than the dk/dv roles of the launch (a launch has one dynamic LDS size: with four images resident it was
  // 55 KB for every workgroup and two workgroups per CU)
  const int L = ax.L;
  float* Ks = sm;
  float* Vs = sm + L * AT_LD;
  const int len = ax.lens ? ax.lens[b] : L;
  const uint64_t seed_off = ax.seed_dev ? *ax.seed_dev : 0ull;
  const int i = threadIdx.x;  // L <= 256 for the role (checked by the launcher): one query per thread, kept across bands
  const bool live = i < L;
  const long long row = (long long)b * L + min(i, L - 1);
  float q[DH], dq[DH];
  load16(ax.q + row * ax.ldq + h * DH, q);
#pragma unroll
  for (int d = 0; d < DH; ++d) dq[d] = 0.f;
  const long long sidx = ((long long)b * ax.H + h) * L + min(i, L - 1);
  const uint64_t rng_row = (((uint64_t)h * ax.B + b) * L + min(i, L - 1)) * (uint64_t)L;
#pragma unroll
  for (int band = 0; band < 2; ++band) {
    const AttnArgs& a = band ? ah : ax;
    float go[DH], oo[DH];
    load16(a.d_o + row * a.lddo + h * DH, go);
    load16(a.o + row * a.ldo + h * DH, oo);
    const float lse = a.lse[sidx];
    if (band) __syncthreads();  // every thread is done with the x band's images
    stage_rows(a.k + (long long)b * L * a.ldk + h * DH, a.ldk, L, Ks);
    stage_rows(a.v + (long long)b * L * a.ldv + h * DH, a.ldv, L, Vs);
    __syncthreads();
    const int bw = a.bw_dev ? *a.bw_dev : a.bw;
    int lo, hi;
    key_range(a.mode, min(i, L - 1), len, L, bw, lo, hi);
    if (!live || i >= len) hi = lo - 1;  // padded query rows carry no gradient
    const float D = dot16(go, oo);
    KanttsDropSeq drop(a.drop_p, a.seed + seed_off);
    for (int j = lo; j <= hi; ++j) {
      float kk[DH];
      lds16(Ks + j * AT_LD, kk);
      const float p = expf(dot16(q, kk) * a.scale - lse);
      const float dp = dot16l(go, Vs + j * AT_LD) * drop.scale(rng_row + j);
      const float ds = p * (dp - D) * a.scale;
#pragma unroll
      for (int d = 0; d < DH; ++d) dq[d] = fmaf(ds, kk[d], dq[d]);
    }
  }
  if (live) store16(ax.dq + row * ax.lddq + h * DH, dq);
}

// Up to four attention passes over the same (B, H) grid as ONE launch: blockIdx.z picks the pass.  A PNCA block's
// forward is two passes (causal band over x, look-ahead band over the memory), its backward four (dq and dk/dv of each
// band); an encoder block's backward two.  The passes of a group are independent (dk/dv recomputes D), so nothing orders
// them -- and a decoder block issues 2 attention launches per step instead of 6 on two streams.
#define AT_ROLE_FWD 0
#define AT_ROLE_DQ 1
#define AT_ROLE_DKV 2
#define AT_ROLE_DQ2 3  // p[z] = x band, p[3] = memory band (no block is launched for slot 3 then)
struct AttnMulti {
  AttnArgs p[4];
  int role[4];
};
template <bool QUAD>  // QUAD: key-padding problems, four lanes per query / key (separate kernel: separate register budget)
__global__ __launch_bounds__(AT_THREADS) void attn_multi_lds_kernel(const AttnMulti m) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int z = blockIdx.z;
  // fixed kernarg offsets selected by scalar branches (indexing the by-value struct with z would put it in scratch)
  AttnArgs a = m.p[0];
  int role = m.role[0];
  if (z == 1) {
    a = m.p[1];
    role = m.role[1];
  } else if (z == 2) {
    a = m.p[2];
    role = m.role[2];
  } else if (z == 3) {
    a = m.p[3];
    role = m.role[3];
  }
  if (QUAD) {
    if (role == AT_ROLE_FWD)
      attn_fwd_lds_quad_body(a, AT_H(a), AT_B(a), sm);
    else if (role == AT_ROLE_DQ)
      attn_bwd_dq_lds_quad_body(a, AT_H(a), AT_B(a), sm);
    else
      attn_bwd_dkv_lds_quad_body(a, AT_H(a), AT_B(a), sm);
  } else {
    if (role == AT_ROLE_DQ2)
      attn_bwd_dq2_lds_body(a, m.p[3], AT_H(a), AT_B(a), sm);
    else if (role == AT_ROLE_FWD)
      attn_fwd_lds_body(a, AT_H(a), AT_B(a), sm);
    else if (role == AT_ROLE_DQ)
      attn_bwd_dq_lds_body(a, AT_H(a), AT_B(a), sm);
    else
      attn_bwd_dkv_lds_body<true>(a, AT_H(a), AT_B(a), sm);
  }
}

__global__ __launch_bounds__(AT_THREADS) void attn_fwd_lds_quad_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  attn_fwd_lds_quad_body(a, AT_H(a), AT_B(a), sm);
}

static bool at_b_first() {
  static const bool head_major = getenv("KANTTS_ATTN_HEAD_MAJOR") != nullptr;  // A/B switch: the round-2 mapping
  return !head_major;
}
static inline dim3 at_grid(int H, int B, int Z) { return at_b_first() ? dim3(B, H, Z) : dim3(H, B, Z); }

static inline size_t attn_lds_bytes(int L, bool dkv) {
  return (size_t)(2 * L * AT_LD + (dkv ? 2 * L : 0)) * sizeof(float);
}

static int attn_check(AttnArgs& a) {
  a.b_first = at_b_first() ? 1 : 0;
  if (!a.q || !a.k || !a.v || !a.o || !a.lse) return KANTTS_E_BADARG;
  if (a.B < 0 || a.H < 1 || a.L < 0 || a.mode < 0 || a.mode > 2) return KANTTS_E_BADARG;
  if ((a.ldq | a.ldk | a.ldv | a.ldo) & 3) return KANTTS_E_BADARG;  // float4 row access
  return KANTTS_OK;
}

extern "C" int kantts_attn_fwd(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, float* o,
                               int ldo, float* lse, float* probs, const int32_t* lens, const int32_t* bw_dev, int bw,
                               int B, int H, int L, int d_head, int mode, float drop_p, uint64_t seed, const uint64_t* seed_dev,
                               void* stream) {
  if (d_head != DH) return KANTTS_E_UNSUPPORTED;
  AttnArgs a = {};
  a.seed_dev = seed_dev;
  a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.o = o; a.ldo = ldo; a.lse = lse;
  a.probs = probs; a.lens = lens; a.bw_dev = bw_dev; a.bw = bw; a.B = B; a.H = H; a.L = L; a.mode = mode;
  a.scale = 0.25f; a.drop_p = drop_p; a.seed = seed;
  int rc = attn_check(a);
  if (rc) return rc;
  if (B == 0 || L == 0) return KANTTS_OK;
  static const bool no_quad = getenv("KANTTS_ATTN_NO_QUAD") != nullptr;
  if (attn_lds_bytes(L, false) <= 64 * 1024 && mode == 0 && !no_quad)
    hipLaunchKernelGGL(attn_fwd_lds_quad_kernel, at_grid(H, B, 1), dim3(AT_THREADS), attn_lds_bytes(L, false),
                       (hipStream_t)stream, a);
  else if (attn_lds_bytes(L, false) <= 64 * 1024)
    hipLaunchKernelGGL(attn_fwd_lds_kernel, at_grid(H, B, 1), dim3(AT_THREADS), attn_lds_bytes(L, false), (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(kantts_cdiv(L, 128), H, B), dim3(128), 0, (hipStream_t)stream, a);
  KANTTS_CHECK_LAUNCH();
}

extern "C" int kantts_attn_bwd(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv,
                               const float* o, int ldo, const float* d_o, int lddo, const float* lse, float* dvec,
                               float* dq, float* dk, float* dv, int lddq, int lddk, int lddv, int accumulate_dq,
                               const int32_t* lens, const int32_t* bw_dev, int bw, int B, int H, int L, int d_head,
                               int mode, float drop_p, uint64_t seed, const uint64_t* seed_dev, void* stream) {
  if (d_head != DH) return KANTTS_E_UNSUPPORTED;
  AttnArgs a = {};
  a.seed_dev = seed_dev;
  a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.o = const_cast<float*>(o); a.ldo = ldo;
  a.lse = const_cast<float*>(lse); a.lens = lens; a.bw_dev = bw_dev; a.bw = bw; a.B = B; a.H = H; a.L = L;
  a.mode = mode; a.scale = 0.25f; a.drop_p = drop_p; a.seed = seed; a.d_o = d_o; a.lddo = lddo; a.dq = dq;
  a.dk = dk; a.dv = dv; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv; a.dvec = dvec; a.accumulate_dq = accumulate_dq;
  int rc = attn_check(a);
  if (rc) return rc;
  if (!d_o || !dq || !dk || !dv || !dvec || ((lddo | lddq | lddk | lddv) & 3)) return KANTTS_E_BADARG;
  if (B == 0 || L == 0) return KANTTS_OK;
  if (attn_lds_bytes(L, true) <= 64 * 1024) {
    AttnMulti m = {};
    m.p[0] = a; m.role[0] = AT_ROLE_DQ;
    m.p[1] = a; m.role[1] = AT_ROLE_DKV;
    static const bool no_quad = getenv("KANTTS_ATTN_NO_QUAD") != nullptr;
    if (mode == 0 && !no_quad)
      hipLaunchKernelGGL(attn_multi_lds_kernel<true>, at_grid(H, B, 2), dim3(AT_THREADS), attn_lds_bytes(L, true),
                         (hipStream_t)stream, m);
    else
      hipLaunchKernelGGL(attn_multi_lds_kernel<false>, at_grid(H, B, 2), dim3(AT_THREADS), attn_lds_bytes(L, true),
                         (hipStream_t)stream, m);
  } else {
    dim3 grid(kantts_cdiv(L, 128), H, B), block(128);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, block, 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, block, 0, (hipStream_t)stream, a);
  }
  KANTTS_CHECK_LAUNCH();
}

// ---------------------------------------------------------------------------------------------------------
// PNCA attention of one decoder block (MultiHeadPNCAAttention, kantts/models/sambert/__init__.py:256-306 of the
// reference): the causal band over x (K/V = columns [D, 3D) of the fused QKV projection) and the look-ahead band over the
// memory (K/V = the (B, L, 2D) memory projection) share the queries (columns [0, D) of qkv).  One launch forward, one
// backward.  Returns KANTTS_E_UNSUPPORTED when a head does not fit the LDS-staged kernels (the caller then issues
// kantts_attn_fwd / kantts_attn_bwd per band).
static void pnca_fill(AttnArgs& a, const float* qkv, const float* kv, int ldkv, int koff, float* o, float* lse,
                      const int32_t* lens, const int32_t* bw_dev, int bw, int B, int H, int L, int mode, float drop_p,
                      uint64_t seed, const uint64_t* seed_dev) {
  const int D = H * DH;
  a = AttnArgs{};
  a.q = qkv; a.ldq = 3 * D;
  a.k = kv + koff; a.v = kv + koff + D; a.ldk = a.ldv = ldkv;
  a.o = o; a.ldo = D; a.lse = lse; a.lens = lens; a.bw_dev = bw_dev; a.bw = bw; a.B = B; a.H = H; a.L = L; a.mode = mode;
  a.scale = 0.25f; a.drop_p = drop_p; a.seed = seed; a.seed_dev = seed_dev;
  a.b_first = at_b_first() ? 1 : 0;
}

extern "C" int kantts_pnca_attn_fwd(const float* qkv, const float* hkv, int ldh, float* ox, float* oh, float* lse_x, float* lse_h,
                                    const int32_t* lens, const int32_t* bw_dev, int bw_x, int bw_h, int B, int H, int L,
                                    int d_head, float drop_p, uint64_t seed_x, uint64_t seed_h, const uint64_t* seed_dev,
                                    void* stream) {
  if (d_head != DH) return KANTTS_E_UNSUPPORTED;
  if (!qkv || !hkv || !ox || !oh || !lse_x || !lse_h || B < 0 || H < 1 || L < 0 || ldh < 2 * H * DH || (ldh & 3))
    return KANTTS_E_BADARG;
  if (B == 0 || L == 0) return KANTTS_OK;
  if (attn_lds_bytes(L, false) > 64 * 1024) return KANTTS_E_UNSUPPORTED;
  const int D = H * DH;
  AttnMulti m = {};
  pnca_fill(m.p[0], qkv, qkv, 3 * D, D, ox, lse_x, lens, bw_dev, bw_x, B, H, L, 1, drop_p, seed_x, seed_dev);
  pnca_fill(m.p[1], qkv, hkv, ldh, 0, oh, lse_h, lens, bw_dev, bw_h, B, H, L, 2, drop_p, seed_h, seed_dev);
  m.role[0] = m.role[1] = AT_ROLE_FWD;
  hipLaunchKernelGGL(attn_multi_lds_kernel<false>, at_grid(H, B, 2), dim3(AT_THREADS), attn_lds_bytes(L, false),
                     (hipStream_t)stream, m);
  KANTTS_CHECK_LAUNCH();
}

// dqkv (B, L, 3D): columns [0, D) receive the query gradient, [D, 3D) the x band's key / value gradients; dhkv (B, L, 2D)
// the memory K/V gradients.  Return value 1 (instead of KANTTS_OK = 0): the query gradients of the two bands were written
// SEPARATELY -- the x band's to dqkv[..., :D], the memory band's to dqh (B, L, D) -- and the caller adds them (sequences
// of more than 256 positions: the summed form keeps one query per thread across both bands); dqh may be NULL otherwise.
extern "C" int kantts_pnca_attn_bwd(const float* qkv, const float* hkv, int ldh, const float* ox, const float* oh, const float* d_ox,
                                    const float* d_oh, const float* lse_x, const float* lse_h, float* dqkv, float* dqh,
                                    float* dhkv, const int32_t* lens, const int32_t* bw_dev, int bw_x, int bw_h, int B,
                                    int H, int L, int d_head, float drop_p, uint64_t seed_x, uint64_t seed_h,
                                    const uint64_t* seed_dev, void* stream) {
  if (d_head != DH) return KANTTS_E_UNSUPPORTED;
  if (!qkv || !hkv || !ox || !oh || !d_ox || !d_oh || !lse_x || !lse_h || !dqkv || !dhkv || B < 0 || H < 1 || L < 0 ||
      ldh < 2 * H * DH || (ldh & 3))
    return KANTTS_E_BADARG;
  if (B == 0 || L == 0) return KANTTS_OK;
  if (attn_lds_bytes(L, true) > 64 * 1024) return KANTTS_E_UNSUPPORTED;
  const int D = H * DH;
  AttnMulti m = {};
  AttnArgs x, hh;
  pnca_fill(x, qkv, qkv, 3 * D, D, const_cast<float*>(ox), const_cast<float*>(lse_x), lens, bw_dev, bw_x, B, H, L, 1, drop_p,
            seed_x, seed_dev);
  x.d_o = d_ox; x.lddo = D; x.dq = dqkv; x.dk = dqkv + D; x.dv = dqkv + 2 * D; x.lddq = x.lddk = x.lddv = 3 * D;
  pnca_fill(hh, qkv, hkv, ldh, 0, const_cast<float*>(oh), const_cast<float*>(lse_h), lens, bw_dev, bw_h, B, H, L, 2, drop_p,
            seed_h, seed_dev);
  hh.d_o = d_oh; hh.lddo = D; hh.dq = dqh; hh.lddq = D; hh.dk = dhkv; hh.dv = dhkv + D; hh.lddk = hh.lddv = 2 * D;
  static const bool no_dq2 = getenv("KANTTS_ATTN_NO_DQ2") != nullptr;
  const bool dq2 = L <= AT_THREADS && !no_dq2;  // one query per thread, held in registers across the two bands
  if (!dqh && !dq2) return KANTTS_E_BADARG;
  if (dq2) {
    // three roles: the query gradient of both bands summed in one pass (dqh is not written), dk/dv per band
    m.p[0] = x; m.role[0] = AT_ROLE_DQ2;
    m.p[1] = x; m.role[1] = AT_ROLE_DKV;
    m.p[2] = hh; m.role[2] = AT_ROLE_DKV;
    m.p[3] = hh; m.role[3] = AT_ROLE_DKV;
    hipLaunchKernelGGL(attn_multi_lds_kernel<false>, at_grid(H, B, 3), dim3(AT_THREADS), attn_lds_bytes(L, true),
                       (hipStream_t)stream, m);
    KANTTS_CHECK_LAUNCH();
  }
  m.p[0] = x; m.role[0] = AT_ROLE_DQ;
  m.p[1] = x; m.role[1] = AT_ROLE_DKV;
  m.p[2] = hh; m.role[2] = AT_ROLE_DQ;
  m.p[3] = hh; m.role[3] = AT_ROLE_DKV;
  hipLaunchKernelGGL(attn_multi_lds_kernel<false>, at_grid(H, B, 4), dim3(AT_THREADS), attn_lds_bytes(L, true),
                     (hipStream_t)stream, m);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  return 1;  // separate query gradients: the caller adds dqh onto dqkv[..., :D]
}


// ---------------------------------------------------------------------------------------------------------
// Free-running decode step (HybridAttentionDecoder.infer, kantts/models/sambert/kantts_sambert.py:208-253 with
// the K/V state of sambert/__init__.py:212-258): ONE query per sequence -- decoder position `step` -- against the
// rows [lo, hi] of a K/V buffer, the interval being the same function of (mode, step, len, band) as in training.
// The reference rebuilds two L x L masks and re-concatenates the K/V cache per layer and step; here the cache is
// a preallocated (B, L, .) buffer whose row `step` was just written by the QKV projection.
__global__ __launch_bounds__(128) void attn_decode_kernel(const AttnArgs a, int step, const int32_t* bw_seq) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.B * a.H) return;
  const int b = t / a.H, h = t % a.H;
  const int len = a.lens ? a.lens[b] : a.L;
  int lo, hi;
  key_range(a.mode, step, len, a.L, bw_seq ? bw_seq[b] : a.bw, lo, hi);
  if (a.mode != 0 && step >= len) hi = lo - 1;  // padded query: context 0 (as in the training kernels)
  float q[DH], o[DH];
  load16(a.q + (long long)b * a.ldq + h * DH, q);
#pragma unroll
  for (int d = 0; d < DH; ++d) o[d] = 0.f;
  const float* kb = a.k + (long long)b * a.L * a.ldk + h * DH;
  const float* vb = a.v + (long long)b * a.L * a.ldv + h * DH;
  float m = -INFINITY;
  for (int j = lo; j <= hi; ++j) {
    float kk[DH];
    load16(kb + (long long)j * a.ldk, kk);
    m = fmaxf(m, dot16(q, kk) * a.scale);
  }
  float l = 0.f;
  for (int j = lo; j <= hi; ++j) {
    float kk[DH], vv[DH];
    load16(kb + (long long)j * a.ldk, kk);
    load16(vb + (long long)j * a.ldv, vv);
    const float e = expf(dot16(q, kk) * a.scale - m);
    l += e;
#pragma unroll
    for (int d = 0; d < DH; ++d) o[d] = fmaf(e, vv[d], o[d]);
  }
  const float inv = (hi >= lo) ? 1.f / l : 0.f;
#pragma unroll
  for (int d = 0; d < DH; ++d) o[d] *= inv;
  store16(a.o + (long long)b * a.ldo + h * DH, o);
}

extern "C" int kantts_attn_decode(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, float* o,
                                  int ldo, const int32_t* lens, const int32_t* bw_seq, int B, int H, int L, int d_head,
                                  int mode, int step, int bw, void* stream) {
  if (!q || !k || !v || !o || B < 0 || H < 1 || L < 1 || mode < 0 || mode > 2 || step < 0 || step >= L)
    return KANTTS_E_BADARG;
  if (d_head != DH) return KANTTS_E_UNSUPPORTED;
  if ((ldq | ldk | ldv | ldo) & 3) return KANTTS_E_UNSUPPORTED;
  if (B == 0) return KANTTS_OK;
  AttnArgs a = {};
  a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.o = o; a.ldo = ldo;
  a.lens = lens; a.B = B; a.H = H; a.L = L; a.mode = mode; a.bw = bw; a.scale = 0.25f;
  hipLaunchKernelGGL(attn_decode_kernel, dim3(kantts_cdiv((long long)B * H, 128)), dim3(128), 0, (hipStream_t)stream, a,
                     step, bw_seq);
  KANTTS_CHECK_LAUNCH();
}
