// Small HBM-bound helpers of the HiFi-GAN stack.
//   weight_norm (w = g * v / ||v||, one wave-group per output row)   layers.py:29,67,105,139
//   x -> sin(x) + x                                                   hifigan.py:157
#include "common.h"

__global__ __launch_bounds__(256) void weight_norm_fwd_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                             float* __restrict__ w, int rows, int cols) {
  __shared__ float red[4];
  const int r = blockIdx.x;
  const float* vr = v + (long long)r * cols;
  float s = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) s += vr[c] * vr[c];
  s = kantts_block_sum(s, red);
  const float sc = g[r] / sqrtf(s);
  float* wr = w + (long long)r * cols;
  for (int c = threadIdx.x; c < cols; c += 256) wr[c] = vr[c] * sc;
}

// dg = sum(dw * v) / ||v||;  dv = (g / ||v||) * (dw - v * dg / ||v||)
__global__ __launch_bounds__(256) void weight_norm_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ v,
                                                             const float* __restrict__ g, float* __restrict__ dv,
                                                             float* __restrict__ dg, int rows, int cols) {
  __shared__ float red[4];
  const int r = blockIdx.x;
  const float* vr = v + (long long)r * cols;
  const float* dr = dw + (long long)r * cols;
  float s = 0.f, d = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) {
    s += vr[c] * vr[c];
    d += dr[c] * vr[c];
  }
  s = kantts_block_sum(s, red);
  d = kantts_block_sum(d, red);
  const float nrm = sqrtf(s);
  const float dgv = d / nrm;
  if (threadIdx.x == 0) dg[r] = dgv;
  const float a = g[r] / nrm, bq = dgv / nrm;
  float* o = dv + (long long)r * cols;
  for (int c = threadIdx.x; c < cols; c += 256) o[c] = a * (dr[c] - vr[c] * bq);
}

// Same reparametrisation with the WEIGHT side addressed by strides: element (row r, input channel ci, tap k) lives at
// r*rs + ci*cs + k*ks.  Tap-major (K, Cout, Cin_g) weights (rs = Cin_g, cs = 1, ks = Cout*Cin_g) are what the
// convolution kernels consume and produce, so the per-call permute().contiguous() copies of the standard layout
// disappear (they were 8 % of the HiFi-GAN training step).  v / dv keep the parameter layout (Cout, Cin_g, K).
__global__ __launch_bounds__(256) void weight_norm_strided_fwd_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                                     float* __restrict__ w, int cin, int K, long long rs,
                                                                     long long cs, long long ks) {
  __shared__ float red[4];
  const int r = blockIdx.x, cols = cin * K;
  const float* vr = v + (long long)r * cols;
  float s = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) s += vr[c] * vr[c];
  s = kantts_block_sum(s, red);
  const float sc = g[r] / sqrtf(s);
  float* wr = w + (long long)r * rs;
  for (int j = threadIdx.x; j < cols; j += 256) {  // ci fastest: coalesced on the tap-major side
    const int k = j / cin, ci = j - k * cin;
    wr[ci * cs + k * ks] = vr[ci * K + k] * sc;
  }
}

// The tap-major weight-norm forward with the two bf16 operand images of csrc/cconv.hip written in the same pass:
//   w   (K, rows, cin) fp32         (kept: the weight gradient's layout, and what the fp32-operand kernels read)
//   wf  (K, rows, cin) bf16         forward contraction
//   wd  (K, groups, cin, rows / groups) bf16   input-gradient contraction (per-tap transpose inside each group)
// One launch per convolution instead of the reparametrisation plus two or three cast / transpose copies per use
// (~480 copy launches and 2.8 ms of kernel time per HiFi-GAN step).
__global__ __launch_bounds__(256) void weight_norm_tap_images_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                                    float* __restrict__ w, __bf16* __restrict__ wf,
                                                                    __bf16* __restrict__ wd, int rows, int cin, int K,
                                                                    int groups) {
  __shared__ float red[4];
  const int r = blockIdx.x, cols = cin * K;
  const float* vr = v + (long long)r * cols;
  float s = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) s += vr[c] * vr[c];
  s = kantts_block_sum(s, red);
  const float sc = g[r] / sqrtf(s);
  const int rg = rows / groups, grp = r / rg, rl = r - grp * rg;
  for (int j = threadIdx.x; j < cols; j += 256) {  // ci fastest: coalesced on the tap-major side
    const int k = j / cin, ci = j - k * cin;
    const float val = vr[ci * K + k] * sc;
    const long long o = ((long long)k * rows + r) * cin + ci;
    w[o] = val;
    if (wf) wf[o] = (__bf16)val;
    if (wd) wd[(((long long)k * groups + grp) * cin + ci) * rg + rl] = (__bf16)val;
  }
}

extern "C" int kantts_weight_norm_tap_images(const float* v, const float* g, float* w, void* wf_bf16, void* wd_bf16, int rows,
                                             int cin, int K, int groups, void* stream) {
  if (!v || !g || !w || rows < 0 || cin < 1 || K < 1 || groups < 1 || (rows % groups)) return KANTTS_E_BADARG;
  if (rows == 0) return KANTTS_OK;
  hipLaunchKernelGGL(weight_norm_tap_images_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, v, g, w,
                     reinterpret_cast<__bf16*>(wf_bf16), reinterpret_cast<__bf16*>(wd_bf16), rows, cin, K, groups);
  KANTTS_CHECK_LAUNCH();
}

// Every weight-normed convolution of a network in one launch: a workgroup per tile of WN_R consecutive output rows of a
// layer, the layer found by bisection over the table's row0 (= tiles before the entry; uniform across the workgroup).
// Per row the same arithmetic and summation order as weight_norm_tap_images_kernel.  WN_R rows at a time so that the
// input-gradient image wd (K, groups, cin, rows / groups) -- rows fastest -- is written in 16-byte pieces: one row per
// workgroup put every 2-byte element of it on a cache line of its own (first version: 351 us per launch at 1.4 TB/s of
// useful traffic, profiles/r04_runC_gan_kernel_stats_top.csv).
#define WN_R 8
typedef __bf16 wn_bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void weight_norm_table_kernel(const float* __restrict__ flat, float* __restrict__ w,
                                                               __bf16* __restrict__ wf, __bf16* __restrict__ wd,
                                                               const kantts_wn_desc* __restrict__ tab, int ndesc) {
  __shared__ float red[4];
  __shared__ float scs[WN_R];
  const int tile = blockIdx.x;
  int lo = 0, hi = ndesc - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].row0 <= tile)
      lo = mid;
    else
      hi = mid - 1;
  }
  const kantts_wn_desc d = tab[lo];
  const int cin = d.cin, K = d.K, rows = d.rows, cols = cin * K;
  const int r0 = (tile - d.row0) * WN_R, nr = min(WN_R, rows - r0);
  const float* v0 = flat + d.v_off + (long long)r0 * cols;
  for (int i = 0; i < nr; ++i) {
    const float* vr = v0 + (long long)i * cols;
    float s = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) s += vr[c] * vr[c];
    s = kantts_block_sum(s, red);
    if (threadIdx.x == 0) scs[i] = flat[d.g_off + r0 + i] / sqrtf(s);
    __syncthreads();
  }
  const int rg = rows / d.groups;
  float* wo = w + d.w_off;
  __bf16* wfo = d.wf_off >= 0 ? wf + d.wf_off : nullptr;
  __bf16* wdo = d.wd_off >= 0 ? wd + d.wd_off : nullptr;
  // (a tile never straddles a group when there are bf16 images: rows / groups is a multiple of 8 then)
  const int grp = r0 / rg, rl0 = r0 - grp * rg;
  const bool vec = wdo && nr == WN_R && ((rg & 7) == 0);
  for (int j = threadIdx.x; j < cols; j += 256) {  // ci fastest: coalesced on the tap-major side
    const int k = j / cin, ci = j - k * cin;
    wn_bf16x8 pk;
#pragma unroll
    for (int i = 0; i < WN_R; ++i) {
      float val = 0.f;
      if (i < nr) {
        val = v0[(long long)i * cols + ci * K + k] * scs[i];
        const long long o = ((long long)k * rows + r0 + i) * cin + ci;
        wo[o] = val;
        if (wfo) wfo[o] = (__bf16)val;
        if (wdo && !vec) wdo[(((long long)k * d.groups + (r0 + i) / rg) * cin + ci) * rg + (r0 + i) % rg] = (__bf16)val;
      }
      pk[i] = (__bf16)val;
    }
    if (vec) *reinterpret_cast<wn_bf16x8*>(wdo + (((long long)k * d.groups + grp) * cin + ci) * rg + rl0) = pk;
  }
}

extern "C" int kantts_weight_norm_table(const float* flat, float* w, void* wf_bf16, void* wd_bf16,
                                        const kantts_wn_desc* table_dev, int ndesc, int total_tiles, void* stream) {
  if (!flat || !w || !table_dev || ndesc < 0 || total_tiles < 0) return KANTTS_E_BADARG;
  if (ndesc == 0 || total_tiles == 0) return KANTTS_OK;
  hipLaunchKernelGGL(weight_norm_table_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, flat, w,
                     reinterpret_cast<__bf16*>(wf_bf16), reinterpret_cast<__bf16*>(wd_bf16), table_dev, ndesc);
  KANTTS_CHECK_LAUNCH();
}

// Backward for many layers in one launch (kantts_weight_norm_table_bwd): per row the arithmetic of
// weight_norm_strided_bwd_kernel on the tap-major gradient (element (r, ci, k) at (k * rows + r) * cin + ci).
__global__ __launch_bounds__(256) void weight_norm_table_bwd_kernel(const float* __restrict__ flat, float* __restrict__ grad,
                                                                   const kantts_wn_desc* __restrict__ tab,
                                                                   const kantts_wn_bwd_args a) {
  __shared__ float red[4];
  const int tile = blockIdx.x;
  int lo = 0, hi = a.nl - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (a.tile0[mid] <= tile)
      lo = mid;
    else
      hi = mid - 1;
  }
  const kantts_wn_desc d = tab[a.desc[lo]];
  const float* dw = a.dw[lo];
  const int cin = d.cin, K = d.K, rows = d.rows, cols = cin * K;
  const int r0 = (tile - a.tile0[lo]) * WN_R, nr = min(WN_R, rows - r0);
  for (int i = 0; i < nr; ++i) {
    const int r = r0 + i;
    const float* vr = flat + d.v_off + (long long)r * cols;
    float s = 0.f, dd = 0.f;
    for (int j = threadIdx.x; j < cols; j += 256) {
      const int k = j / cin, ci = j - k * cin;
      const float vv = vr[ci * K + k];
      s += vv * vv;
      dd += dw[((long long)k * rows + r) * cin + ci] * vv;
    }
    s = kantts_block_sum(s, red);
    dd = kantts_block_sum(dd, red);
    const float nrm = sqrtf(s);
    const float gr = flat[d.g_off + r];
    const float dgv = dd / nrm;
    if (threadIdx.x == 0) grad[d.g_off + r] = dgv;
    const float aa = gr / nrm, bq = dgv / nrm;
    float* o = grad + d.v_off + (long long)r * cols;
    for (int j = threadIdx.x; j < cols; j += 256) {
      const int k = j / cin, ci = j - k * cin;
      o[ci * K + k] = aa * (dw[((long long)k * rows + r) * cin + ci] - vr[ci * K + k] * bq);
    }
  }
}

extern "C" int kantts_weight_norm_table_bwd(const float* flat, float* grad_flat, const kantts_wn_desc* table_dev,
                                            const kantts_wn_bwd_args* args, void* stream) {
  if (!flat || !grad_flat || !table_dev || !args || args->nl < 0 || args->nl > KANTTS_WN_BWD_MAX) return KANTTS_E_BADARG;
  if (args->nl == 0 || args->tile0[args->nl] == 0) return KANTTS_OK;
  for (int l = 0; l < args->nl; ++l)
    if (!args->dw[l]) return KANTTS_E_BADARG;
  hipLaunchKernelGGL(weight_norm_table_bwd_kernel, dim3(args->tile0[args->nl]), dim3(256), 0, (hipStream_t)stream, flat,
                     grad_flat, table_dev, *args);
  KANTTS_CHECK_LAUNCH();
}

__global__ __launch_bounds__(256) void weight_norm_strided_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ v,
                                                                     const float* __restrict__ g, float* __restrict__ dv,
                                                                     float* __restrict__ dg, int cin, int K, long long rs,
                                                                     long long cs, long long ks) {
  __shared__ float red[4];
  const int r = blockIdx.x, cols = cin * K;
  const float* vr = v + (long long)r * cols;
  const float* dr = dw + (long long)r * rs;
  float s = 0.f, d = 0.f;
  for (int j = threadIdx.x; j < cols; j += 256) {
    const int k = j / cin, ci = j - k * cin;
    const float vv = vr[ci * K + k];
    s += vv * vv;
    d += dr[ci * cs + k * ks] * vv;
  }
  s = kantts_block_sum(s, red);
  d = kantts_block_sum(d, red);
  const float nrm = sqrtf(s);
  const float dgv = d / nrm;
  if (threadIdx.x == 0) dg[r] = dgv;
  const float a = g[r] / nrm, bq = dgv / nrm;
  float* o = dv + (long long)r * cols;
  for (int j = threadIdx.x; j < cols; j += 256) {
    const int k = j / cin, ci = j - k * cin;
    o[ci * K + k] = a * (dr[ci * cs + k * ks] - vr[ci * K + k] * bq);
  }
}

__global__ void sinadd_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float v = x[i];
    y[i] = sinf(v) + v;
  }
}
__global__ void sinadd_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx,
                                  long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dx[i] = dy[i] * (cosf(x[i]) + 1.f);
}

extern "C" int kantts_weight_norm_fwd(const float* v, const float* g, float* w, int rows, int cols, void* stream) {
  if (!v || !g || !w || rows < 0 || cols < 1) return KANTTS_E_BADARG;
  if (rows == 0) return KANTTS_OK;
  hipLaunchKernelGGL(weight_norm_fwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, v, g, w, rows, cols);
  KANTTS_CHECK_LAUNCH();
}
extern "C" int kantts_weight_norm_bwd(const float* dw, const float* v, const float* g, float* dv, float* dg, int rows,
                                      int cols, void* stream) {
  if (!dw || !v || !g || !dv || !dg || rows < 0 || cols < 1) return KANTTS_E_BADARG;
  if (rows == 0) return KANTTS_OK;
  hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, dw, v, g, dv, dg, rows, cols);
  KANTTS_CHECK_LAUNCH();
}
extern "C" int kantts_weight_norm_strided_fwd(const float* v, const float* g, float* w, int rows, int cin, int K,
                                              long long rs, long long cs, long long ks, void* stream) {
  if (!v || !g || !w || rows < 0 || cin < 1 || K < 1) return KANTTS_E_BADARG;
  if (rows == 0) return KANTTS_OK;
  hipLaunchKernelGGL(weight_norm_strided_fwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, v, g, w, cin, K, rs, cs,
                     ks);
  KANTTS_CHECK_LAUNCH();
}
extern "C" int kantts_weight_norm_strided_bwd(const float* dw, const float* v, const float* g, float* dv, float* dg,
                                              int rows, int cin, int K, long long rs, long long cs, long long ks,
                                              void* stream) {
  if (!dw || !v || !g || !dv || !dg || rows < 0 || cin < 1 || K < 1) return KANTTS_E_BADARG;
  if (rows == 0) return KANTTS_OK;
  hipLaunchKernelGGL(weight_norm_strided_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, dw, v, g, dv, dg, cin, K,
                     rs, cs, ks);
  KANTTS_CHECK_LAUNCH();
}
// y = sin(x) + x (fp32) and, in the same pass, a = LeakyReLU(y) rounded to bf16: the operand image of the transposed
// convolution that consumes the stage input (csrc/upsample.hip); the fp32 y still feeds the repeat-upsample branch.
// ---- [round 4] mean of up to 8 tensors in one pass (+ the bf16 LeakyReLU image of the result), and its backward
// HiFi-GAN's multi-receptive-field fusion (kantts/models/hifigan/hifigan.py:160-176 of the reference: xs += block(x) over
// the stage's residual stacks, then xs / num_kernels) was two ATen adds + a division + the operand-image cast of the next
// convolution: four passes over the largest activations of the generator.  Backward: one launch writes every branch its
// own copy of g * scale (the branches consume their gradients on their own streams: no shared tensor, ops._BranchExit).
struct MeanManyArgs {
  const float* x[8];
  float* o[8];
  int n;
};
typedef __bf16 mm_bf16x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mean_many_kernel(const MeanManyArgs a, float scale, float* __restrict__ out,
                                                        __bf16* __restrict__ act, float slope, long long n4) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 s = reinterpret_cast<const float4*>(a.x[0])[i];
    for (int k = 1; k < a.n; ++k) {
      const float4 v = reinterpret_cast<const float4*>(a.x[k])[i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    s.x *= scale; s.y *= scale; s.z *= scale; s.w *= scale;
    reinterpret_cast<float4*>(out)[i] = s;
    if (act) {
      mm_bf16x4 b = {(__bf16)(s.x > 0.f ? s.x : s.x * slope), (__bf16)(s.y > 0.f ? s.y : s.y * slope),
                     (__bf16)(s.z > 0.f ? s.z : s.z * slope), (__bf16)(s.w > 0.f ? s.w : s.w * slope)};
      reinterpret_cast<mm_bf16x4*>(act)[i] = b;
    }
  }
}
__global__ __launch_bounds__(256) void scale_to_many_kernel(const MeanManyArgs a, const float* __restrict__ g, float scale,
                                                            long long n4) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 v = reinterpret_cast<const float4*>(g)[i];
    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
    for (int k = 0; k < a.n; ++k) reinterpret_cast<float4*>(a.o[k])[i] = v;
  }
}
extern "C" int kantts_mean_many(const float* const* xs_host, int n, float scale, float* out, void* act_bf16, float slope,
                                long long numel, void* stream) {
  if (!xs_host || n < 1 || n > 8 || numel < 0) return KANTTS_E_BADARG;
  if (numel == 0) return KANTTS_OK;  // (empty tensors may carry NULL pointers)
  if (!out) return KANTTS_E_BADARG;
  if ((numel & 3) || ((uintptr_t)out & 15) || ((uintptr_t)act_bf16 & 7)) return KANTTS_E_UNSUPPORTED;
  MeanManyArgs a = {};
  a.n = n;
  for (int k = 0; k < n; ++k) {
    if (!xs_host[k]) return KANTTS_E_BADARG;
    if ((uintptr_t)xs_host[k] & 15) return KANTTS_E_UNSUPPORTED;
    a.x[k] = xs_host[k];
  }
  if (numel == 0) return KANTTS_OK;
  int blocks = kantts_cdiv(numel >> 2, 256 * 4);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(mean_many_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, scale, out,
                     reinterpret_cast<__bf16*>(act_bf16), slope, numel >> 2);
  KANTTS_CHECK_LAUNCH();
}
extern "C" int kantts_scale_to_many(const float* g, float scale, float* const* outs_host, int n, long long numel,
                                    void* stream) {
  if (!outs_host || n < 1 || n > 8 || numel < 0) return KANTTS_E_BADARG;
  if (numel == 0) return KANTTS_OK;
  if (!g) return KANTTS_E_BADARG;
  if ((numel & 3) || ((uintptr_t)g & 15)) return KANTTS_E_UNSUPPORTED;
  MeanManyArgs a = {};
  a.n = n;
  for (int k = 0; k < n; ++k) {
    if (!outs_host[k]) return KANTTS_E_BADARG;
    if ((uintptr_t)outs_host[k] & 15) return KANTTS_E_UNSUPPORTED;
    a.o[k] = outs_host[k];
  }
  if (numel == 0) return KANTTS_OK;
  int blocks = kantts_cdiv(numel >> 2, 256 * 4);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(scale_to_many_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, g, scale, numel >> 2);
  KANTTS_CHECK_LAUNCH();
}

__global__ void sinadd_lrelu_kernel(const float* __restrict__ x, float* __restrict__ y, __bf16* __restrict__ a, float slope,
                                    long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float v = x[i];
    const float s = sinf(v) + v;
    y[i] = s;
    a[i] = (__bf16)(s > 0.f ? s : s * slope);
  }
}
extern "C" int kantts_sinadd_lrelu_fwd(const float* x, float* y, void* act_bf16, float slope, long long n, void* stream) {
  if (!x || !y || !act_bf16 || n < 0) return KANTTS_E_BADARG;
  if (n == 0) return KANTTS_OK;
  int blocks = kantts_cdiv(n, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(sinadd_lrelu_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y,
                     reinterpret_cast<__bf16*>(act_bf16), slope, n);
  KANTTS_CHECK_LAUNCH();
}

extern "C" int kantts_sinadd_fwd(const float* x, float* y, long long n, void* stream) {
  if (!x || !y || n < 0) return KANTTS_E_BADARG;
  if (n == 0) return KANTTS_OK;
  int blocks = kantts_cdiv(n, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(sinadd_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, n);
  KANTTS_CHECK_LAUNCH();
}
extern "C" int kantts_sinadd_bwd(const float* dy, const float* x, float* dx, long long n, void* stream) {
  if (!dy || !x || !dx || n < 0) return KANTTS_E_BADARG;
  if (n == 0) return KANTTS_OK;
  int blocks = kantts_cdiv(n, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(sinadd_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, x, dx, n);
  KANTTS_CHECK_LAUNCH();
}

// y = x * keep1(i) * keep2(i) (+ res): the FSMN encoder's two stacked dropouts and its residual add
// (kantts/models/sambert/fsmn.py:66-70,114-121: MemoryBlockV2's own dropout, the encoder's dropout on the block output,
// "memory + x") as ONE pass with regenerated masks; the same entry point is the backward (x := dy, res := NULL).  As three
// ATen kernels forward and two backward they moved 240 MB per postnet layer (saved boolean masks included) against 100.
__global__ __launch_bounds__(256) void dropout2_add_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                          float* __restrict__ y, long long n4, float p1, uint64_t seed1,
                                                          float p2, uint64_t seed2, const uint64_t* __restrict__ seed_dev) {
  const uint64_t off = seed_dev ? *seed_dev : 0ull;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    float o[4] = {v.x, v.y, v.z, v.w};
    kantts_dropout_scale4(p1, seed1 + off, (uint64_t)i * 4, o);
    kantts_dropout_scale4(p2, seed2 + off, (uint64_t)i * 4, o);
    if (res) {
      const float4 r = reinterpret_cast<const float4*>(res)[i];
      o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
    }
    reinterpret_cast<float4*>(y)[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

extern "C" int kantts_dropout2_add(const float* x, const float* res, float* y, long long n, float p1, uint64_t seed1, float p2,
                                   uint64_t seed2, const uint64_t* seed_dev, void* stream) {
  if (!x || !y || n < 0 || (n & 3) || p1 < 0.f || p1 >= 1.f || p2 < 0.f || p2 >= 1.f) return KANTTS_E_BADARG;
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)res) & 15) return KANTTS_E_BADARG;
  if (n == 0) return KANTTS_OK;
  const long long n4 = n >> 2;
  const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(dropout2_add_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, res, y, n4, p1, seed1, p2,
                     seed2, seed_dev);
  KANTTS_CHECK_LAUNCH();
}
