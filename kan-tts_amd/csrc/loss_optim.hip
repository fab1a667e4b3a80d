// Masked L1 reductions, gradient-norm and fused Adam over flat parameter arenas.
//   MelReconLoss / ProsodyReconLoss ("mae")   kantts/train/loss.py:18-37, :51-85
//   clip_grad_norm_ + Adam step               kantts/train/trainer.py:997-1004, configs/sambert_16k.yaml:54-60
// All are single-pass HBM-bound kernels; scalars that the reference pulls to the host (.item())
// stay in device memory so that the whole training step can be enqueued without a sync.
#include "common.h"

// loss[0] += sum_{b, t < lens[b], c} |target - pred| / (sum_b lens[b] * C)
// grad (optional) = sign(pred - target) / (sum lens * C) on valid rows, 0 elsewhere.
__global__ __launch_bounds__(256) void masked_l1_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                       const int64_t* __restrict__ lens, float* __restrict__ loss,
                                                       float* __restrict__ grad, int B, int T, int C) {
  __shared__ float red[4];
  long long denom_rows = 0;
  for (int b = 0; b < B; ++b) denom_rows += min((long long)lens[b], (long long)T);
  const float inv = 1.f / ((float)denom_rows * (float)C);
  const long long total = (long long)B * T * C;
  float part = 0.f;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    long long bt = g / C;
    int t = (int)(bt % T), b = (int)(bt / T);
    float gr = 0.f;
    if (t < (int)lens[b]) {
      float d = pred[g] - target[g];
      part += fabsf(d);
      gr = (d > 0.f) ? inv : ((d < 0.f) ? -inv : 0.f);
    }
    if (grad) grad[g] = gr;
  }
  part = kantts_block_sum(part, red);
  if (threadIdx.x == 0) atomicAdd(loss, part * inv);
}

// The same for C % 4 == 0, 16-byte aligned tensors and fewer than 2^31 elements (the mel losses: 32 x 612 x 80): four
// elements of one row per thread and trip (one 32-bit row / sequence split per 16 bytes instead of two 64-bit divisions
// per element -- the scalar kernel took 12 us for 1.6 M elements).  Same per-block partial sums -> one atomic per block.
__global__ __launch_bounds__(256) void masked_l1_vec4_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                            const int64_t* __restrict__ lens, float* __restrict__ loss,
                                                            float* __restrict__ grad, int B, int T, int C) {
  __shared__ float red[4];
  long long denom_rows = 0;
  for (int b = 0; b < B; ++b) denom_rows += min((long long)lens[b], (long long)T);
  const float inv = 1.f / ((float)denom_rows * (float)C);
  const unsigned total4 = (unsigned)((long long)B * T * C / 4), C4 = (unsigned)C / 4;
  float part = 0.f;
  for (unsigned q = blockIdx.x * blockDim.x + threadIdx.x; q < total4; q += gridDim.x * blockDim.x) {
    const unsigned bt = q / C4;
    const unsigned b = bt / (unsigned)T, t = bt - b * (unsigned)T;
    float4 gr = {0.f, 0.f, 0.f, 0.f};
    if ((long long)t < lens[b]) {
      const float4 p = reinterpret_cast<const float4*>(pred)[q], y = reinterpret_cast<const float4*>(target)[q];
      const float d0 = p.x - y.x, d1 = p.y - y.y, d2 = p.z - y.z, d3 = p.w - y.w;
      part += (fabsf(d0) + fabsf(d1)) + (fabsf(d2) + fabsf(d3));
      gr.x = (d0 > 0.f) ? inv : ((d0 < 0.f) ? -inv : 0.f);
      gr.y = (d1 > 0.f) ? inv : ((d1 < 0.f) ? -inv : 0.f);
      gr.z = (d2 > 0.f) ? inv : ((d2 < 0.f) ? -inv : 0.f);
      gr.w = (d3 > 0.f) ? inv : ((d3 < 0.f) ? -inv : 0.f);
    }
    if (grad) reinterpret_cast<float4*>(grad)[q] = gr;
  }
  part = kantts_block_sum(part, red);
  if (threadIdx.x == 0) atomicAdd(loss, part * inv);
}

extern "C" int kantts_masked_l1(const float* pred, const float* target, const int64_t* lens, float* loss_accum,
                                float* grad, int B, int T, int C, void* stream) {
  if (!pred || !target || !lens || !loss_accum || B < 0 || T < 0 || C < 1) return KANTTS_E_BADARG;
  long long total = (long long)B * T * C;
  if (total == 0) return KANTTS_OK;
  const bool vec = (C % 4 == 0) && total < (1ll << 31) && (((uintptr_t)pred | (uintptr_t)target | (uintptr_t)grad) & 15) == 0;
  if (vec) {
    int blocks = kantts_cdiv(total / 4, 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(masked_l1_vec4_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred, target, lens,
                       loss_accum, grad, B, T, C);
    KANTTS_CHECK_LAUNCH();
  }
  int blocks = kantts_cdiv(total, 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(masked_l1_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred, target, lens, loss_accum,
                     grad, B, T, C);
  KANTTS_CHECK_LAUNCH();
}

// ---- the five masked-L1 terms of a SAM-BERT step in one launch (kantts_masked_l1_many) ---------------------------------
struct LossManyArgs {
  kantts_loss_term t[KANTTS_LOSS_MAX_TERMS];
  int first_block[KANTTS_LOSS_MAX_TERMS + 1];
  int n;
};
__global__ __launch_bounds__(256) void masked_l1_many_kernel(const LossManyArgs a, float* __restrict__ losses) {
  __shared__ float red[4];
  __shared__ float inv_s;
  int k = 0;
  while (k + 1 < a.n && (int)blockIdx.x >= a.first_block[k + 1]) ++k;
  const kantts_loss_term q = a.t[k];
  const int blk = blockIdx.x - a.first_block[k], nblk = a.first_block[k + 1] - a.first_block[k];
  if (threadIdx.x == 0) {
    long long denom_rows = 0;
    for (int b = 0; b < q.B; ++b) denom_rows += min((long long)q.lens[b], (long long)q.T);
    inv_s = 1.f / ((float)denom_rows * (float)q.C);
  }
  __syncthreads();
  const float inv = inv_s;
  const long long total = (long long)q.B * q.T * q.C;
  const float* tf = reinterpret_cast<const float*>(q.target);
  const int64_t* ti = reinterpret_cast<const int64_t*>(q.target);
  float part = 0.f;
  // [round 5] The element loop used to split a 64-bit flat index into (row, channel) and (sequence, frame) with two 64-bit
  // divisions PER ELEMENT: 61 us for the 19 MB of a SAM-BERT step's five terms, at the end of the forward pass where
  // nothing overlaps it (profiles/r05_runPRE_forward_only_kernel_stats_top.csv).  Rows of four-channel groups with 32-bit
  // indices and 16-byte accesses when the shapes allow it (the two mel terms: C = 80); the scalar form otherwise.
  const bool vec = !q.target_log1p && (q.C & 3) == 0 && total < (1ll << 31) &&
                   ((((uintptr_t)q.pred) | ((uintptr_t)q.target) | ((uintptr_t)q.grad)) & 15) == 0;
  if (vec) {
    const unsigned c4 = (unsigned)q.C >> 2, n4 = (unsigned)(total >> 2), T = (unsigned)q.T;
    const float4* p4 = reinterpret_cast<const float4*>(q.pred);
    const float4* t4 = reinterpret_cast<const float4*>(tf);
    float4* g4 = reinterpret_cast<float4*>(q.grad);
    for (unsigned i = (unsigned)blk * 256u + threadIdx.x; i < n4; i += (unsigned)nblk * 256u) {
      const unsigned row = i / c4, b = row / T, t = row - b * T;
      float4 gr = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t < (unsigned)q.lens[b]) {
        const float4 pv = p4[i], yv = t4[i];
        const float d0 = pv.x - yv.x, d1 = pv.y - yv.y, d2 = pv.z - yv.z, d3 = pv.w - yv.w;
        part += (fabsf(d0) + fabsf(d1)) + (fabsf(d2) + fabsf(d3));
        gr.x = (d0 > 0.f) ? inv : ((d0 < 0.f) ? -inv : 0.f);
        gr.y = (d1 > 0.f) ? inv : ((d1 < 0.f) ? -inv : 0.f);
        gr.z = (d2 > 0.f) ? inv : ((d2 < 0.f) ? -inv : 0.f);
        gr.w = (d3 > 0.f) ? inv : ((d3 < 0.f) ? -inv : 0.f);
      }
      if (q.grad) g4[i] = gr;
    }
  } else {
    for (long long g = (long long)blk * 256 + threadIdx.x; g < total; g += (long long)nblk * 256) {
      const long long bt = g / q.C;
      const int t = (int)(bt % q.T), b = (int)(bt / q.T);
      float gr = 0.f;
      if (t < (int)q.lens[b]) {
        const float y = q.target_log1p ? logf((float)ti[g] + 1.f) : tf[g];
        const float d = q.pred[g] - y;
        part += fabsf(d);
        gr = (d > 0.f) ? inv : ((d < 0.f) ? -inv : 0.f);
      }
      if (q.grad) q.grad[g] = gr;
    }
  }
  part = kantts_block_sum(part, red);
  if (threadIdx.x == 0) {
    atomicAdd(losses + k, part * inv);
    atomicAdd(losses + KANTTS_LOSS_MAX_TERMS, part * inv);
  }
}

extern "C" int kantts_masked_l1_many(const kantts_loss_term* terms, int nterms, float* losses, void* stream) {
  if (!terms || !losses || nterms < 1 || nterms > KANTTS_LOSS_MAX_TERMS) return KANTTS_E_BADARG;
  LossManyArgs a = {};
  a.n = nterms;
  int nb = 0;
  for (int k = 0; k < nterms; ++k) {
    const kantts_loss_term& q = terms[k];
    if (!q.pred || !q.target || !q.lens || q.B < 0 || q.T < 0 || q.C < 1) return KANTTS_E_BADARG;
    a.t[k] = q;
    a.first_block[k] = nb;
    long long total = (long long)q.B * q.T * q.C;
    // a thread takes ~8 sixteen-byte groups; at most 256 blocks per term (every block ends with two atomics, one of them
    // onto the address all terms share)
    int blocks = kantts_cdiv(total, 8192);
    if (blocks < 1) blocks = 1;
    if (blocks > 256) blocks = 256;
    nb += blocks;
  }
  a.first_block[nterms] = nb;
  hipLaunchKernelGGL(masked_l1_many_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a, losses);
  KANTTS_CHECK_LAUNCH();
}

struct ScaleManyArgs {
  float* x[KANTTS_ELOSS_MAX_TERMS];
  long long n[KANTTS_ELOSS_MAX_TERMS];
  int count;
};
__global__ __launch_bounds__(256) void scale_many_kernel(const ScaleManyArgs a, const float* __restrict__ scale) {
  const float s = scale[0];
  float* x = a.x[blockIdx.y];
  const long long n = a.n[blockIdx.y];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) x[i] *= s;
}
extern "C" int kantts_scale_many(float* const* x, const long long* n, int count, const float* scale_dev, void* stream) {
  if (!x || !n || !scale_dev || count < 0 || count > KANTTS_ELOSS_MAX_TERMS) return KANTTS_E_BADARG;
  if (count == 0) return KANTTS_OK;
  ScaleManyArgs a = {};
  a.count = count;
  long long mx = 0;
  for (int k = 0; k < count; ++k) {
    if (!x[k] || n[k] < 0) return KANTTS_E_BADARG;
    a.x[k] = x[k];
    a.n[k] = n[k];
    if (n[k] > mx) mx = n[k];
  }
  int blocks = kantts_cdiv(mx, 1024);
  if (blocks < 1) blocks = 1;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(scale_many_kernel, dim3(blocks, count), dim3(256), 0, (hipStream_t)stream, a, scale_dev);
  KANTTS_CHECK_LAUNCH();
}

// out[0] += sum x^2
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, float* __restrict__ out, long long n) {
  __shared__ float red[4];
  float part = 0.f;
  const long long n4 = n >> 2;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 v = x4[i];
    part += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    part += x[i] * x[i];
  part = kantts_block_sum(part, red);
  if (threadIdx.x == 0) atomicAdd(out, part);
}

extern "C" int kantts_sumsq(const float* x, float* out_accum, long long n, void* stream) {
  if (!x || !out_accum || n < 0 || ((uintptr_t)x & 15)) return KANTTS_E_BADARG;
  if (n == 0) return KANTTS_OK;
  int blocks = kantts_cdiv(n, 1024);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, out_accum, n);
  KANTTS_CHECK_LAUNCH();
}

// Deterministic variant for data-parallel training: out[0] = sum x^2 with a FIXED summation order, so that every replica
// derives the bit-identical clipping factor from the bit-identical all-reduced gradient (the atomicAdd order of
// sumsq_kernel differs from run to run; replicas drifted apart by ulps in the two-process GPU test of round 2).
// Blocks store their partial sums into ws[1 + block]; the block that draws the last ticket (ws[0], reset for the next
// launch) adds them in index order.  Publication follows the agent-scope release / acquire recipe of the CDNA4 guide.
#define SUMSQ_DET_BLOCKS 1024
__global__ __launch_bounds__(256) void sumsq_det_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                       float* __restrict__ ws, long long n) {
  __shared__ float red[4];
  __shared__ int is_last;
  float part = 0.f;
  const long long n4 = n >> 2;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  // four independent loads per trip (a fixed assignment of elements to the four partial sums: still one summation order per
  // launch shape): one load in flight per thread kept a 49 MB gradient at 2 TB/s
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 v0 = x4[i], v1 = x4[i + stride], v2 = x4[i + 2 * stride], v3 = x4[i + 3 * stride];
    p0 += v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w;
    p1 += v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w;
    p2 += v2.x * v2.x + v2.y * v2.y + v2.z * v2.z + v2.w * v2.w;
    p3 += v3.x * v3.x + v3.y * v3.y + v3.z * v3.z + v3.w * v3.w;
  }
  for (; i < n4; i += stride) {
    float4 v = x4[i];
    part += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  part += (p0 + p1) + (p2 + p3);
  for (long long j = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) part += x[j] * x[j];
  part = kantts_block_sum(part, red);
  unsigned* ticket = reinterpret_cast<unsigned*>(ws);
  if (threadIdx.x == 0) {
    ws[1 + blockIdx.x] = part;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (t == gridDim.x - 1) ? 1 : 0;
    if (is_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (!is_last) return;
  float acc = 0.f;
  for (unsigned b = threadIdx.x; b < gridDim.x; b += 256) acc += ws[1 + b];  // thread t: blocks t, t+256, ... (fixed)
  acc = kantts_block_sum(acc, red);                                         // fixed tree
  if (threadIdx.x == 0) {
    out[0] = acc;
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

extern "C" int kantts_sumsq_det(const float* x, float* out, float* workspace, long long ws_floats, long long n,
                                void* stream) {
  if (!x || !out || !workspace || n < 0 || ((uintptr_t)x & 15)) return KANTTS_E_BADARG;
  if (ws_floats < SUMSQ_DET_BLOCKS + 1) return KANTTS_E_WORKSPACE;
  // (every block ends with an agent-scope release + a ticket: a quarter of the round-2 block count reads the same bytes with
  // a quarter of the fences)
  int blocks = kantts_cdiv(n > 0 ? n : 1, 4096);
  if (blocks > SUMSQ_DET_BLOCKS / 4) blocks = SUMSQ_DET_BLOCKS / 4;
  hipLaunchKernelGGL(sumsq_det_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, out, workspace, n);
  KANTTS_CHECK_LAUNCH();
}

// torch.optim.Adam (amsgrad=False) on flat fp32 buffers with optional global-norm clipping:
//   clip = max_norm > 0 ? min(1, max_norm / (sqrt(*gnorm_sq) + 1e-6)) : 1      (clip_grad_norm_)
//   g *= clip; g += wd * p; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2
//   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ m, float* __restrict__ v, long long n, float lr,
                                                  float b1, float b2, float eps, float wd, float bc1, float bc2,
                                                  const float* __restrict__ gnorm_sq, float max_norm,
                                                  const float* __restrict__ dyn) {
  if (dyn) {
    lr = dyn[0];
    bc1 = 1.f - powf(b1, dyn[1]);
    bc2 = 1.f - powf(b2, dyn[1]);
  }
  float clip = 1.f;
  if (max_norm > 0.f && gnorm_sq) clip = fminf(1.f, max_norm / (sqrtf(*gnorm_sq) + 1e-6f));
  const float step = lr / bc1;
  const float rbc2 = 1.f / sqrtf(bc2);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float gi = g[i] * clip;
    float pi = p[i];
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    float mi = fmaf(b1, m[i], (1.f - b1) * gi);
    float vi = fmaf(b2, v[i], (1.f - b2) * gi * gi);
    m[i] = mi;
    v[i] = vi;
    p[i] = pi - step * mi / (sqrtf(vi) * rbc2 + eps);
  }
}

extern "C" int kantts_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                                float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2,
                                const float* gnorm_sq, float max_norm, const float* dyn_lr_step, void* stream) {
  if (!p || !g || !m || !v || n < 0) return KANTTS_E_BADARG;
  if (n == 0) return KANTTS_OK;
  int blocks = kantts_cdiv(n, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                     weight_decay, bias_corr1, bias_corr2, gnorm_sq, max_norm, dyn_lr_step);
  KANTTS_CHECK_LAUNCH();
}

// Mean-reduced element losses of the GAN step (kantts/train/loss.py:108-256, :310):
//   mode 0: loss += scale * sum |a - b|          grad = scale * sign(a - b)      (FeatureMatchLoss / mel L1)
//   mode 1: loss += scale * sum (a - c)^2        grad = 2 * scale * (a - c)      (LSGAN real/fake targets)
__global__ __launch_bounds__(256) void elem_loss_kernel(const float* __restrict__ a, const float* __restrict__ b, float c,
                                                       int mode, float scale, float* __restrict__ loss,
                                                       float* __restrict__ grad, long long n) {
  __shared__ float red[4];
  float part = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float d = a[i] - (mode == 0 ? b[i] : c);
    float g;
    if (mode == 0) {
      part += fabsf(d);
      g = (d > 0.f) ? scale : ((d < 0.f) ? -scale : 0.f);
    } else {
      part += d * d;
      g = 2.f * scale * d;
    }
    if (grad) grad[i] = g;
  }
  part = kantts_block_sum(part, red);
  if (threadIdx.x == 0) atomicAdd(loss, part * scale);
}

// ---- [round 4] many terms of the above in one launch (include/kantts_hip.h: kantts_elem_loss_many)
struct ElossManyArgs {
  kantts_eloss_term t[KANTTS_ELOSS_MAX_TERMS];
  int first_block[KANTTS_ELOSS_MAX_TERMS + 1];
  int n;
};
__global__ __launch_bounds__(256) void elem_loss_many_kernel(const ElossManyArgs a, float* __restrict__ losses) {
  __shared__ float red[4];
  int lo = 0, hi = a.n;  // term of this block: first_block[lo] <= blockIdx.x < first_block[lo + 1]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int)blockIdx.x >= a.first_block[mid]) lo = mid; else hi = mid;
  }
  const kantts_eloss_term q = a.t[lo];
  const int blk = blockIdx.x - a.first_block[lo], nblk = a.first_block[lo + 1] - a.first_block[lo];
  float part = 0.f;
  for (long long i = (long long)blk * 256 + threadIdx.x; i < q.n; i += (long long)nblk * 256) {
    const float d = q.a[i] - (q.mode == 0 ? q.b[i] : q.target);
    float g;
    if (q.mode == 0) {
      part += fabsf(d);
      g = (d > 0.f) ? q.scale : ((d < 0.f) ? -q.scale : 0.f);
    } else {
      part += d * d;
      g = 2.f * q.scale * d;
    }
    if (q.grad) q.grad[i] = g;
  }
  part = kantts_block_sum(part, red);
  if (threadIdx.x == 0) atomicAdd(losses + q.out, part * q.scale);
}

extern "C" int kantts_elem_loss_many(const kantts_eloss_term* terms, int nterms, float* losses, void* stream) {
  if (!terms || !losses || nterms < 1 || nterms > KANTTS_ELOSS_MAX_TERMS) return KANTTS_E_BADARG;
  ElossManyArgs a = {};
  int nb = 0, k = 0;
  for (int i = 0; i < nterms; ++i) {
    const kantts_eloss_term& q = terms[i];
    if (q.n < 0 || q.mode < 0 || q.mode > 1 || q.out < 0) return KANTTS_E_BADARG;
    if (q.n == 0) continue;  // (an empty tensor's pointer may be NULL)
    if (!q.a || (q.mode == 0 && !q.b)) return KANTTS_E_BADARG;
    a.t[k] = q;
    a.first_block[k] = nb;
    int blocks = kantts_cdiv(q.n, 1024);
    if (blocks > 512) blocks = 512;
    nb += blocks;
    ++k;
  }
  if (k == 0) return KANTTS_OK;
  a.first_block[k] = nb;
  a.n = k;
  hipLaunchKernelGGL(elem_loss_many_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a, losses);
  KANTTS_CHECK_LAUNCH();
}

extern "C" int kantts_elem_loss(const float* a, const float* b, float target, int mode, float scale, float* loss_accum,
                                float* grad, long long n, void* stream) {
  if (!a || !loss_accum || n < 0 || (mode == 0 && !b) || mode < 0 || mode > 1) return KANTTS_E_BADARG;
  if (n == 0) return KANTTS_OK;
  int blocks = kantts_cdiv(n, 1024);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(elem_loss_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, target, mode, scale,
                     loss_accum, grad, n);
  KANTTS_CHECK_LAUNCH();
}
