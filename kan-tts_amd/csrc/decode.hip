// Free-running (auto-regressive) decode helpers: everything a decode step needs that depends on the step index takes
// that index from DEVICE memory, so that one decoder step is a static sequence of launches -- captured once in a
// hipGraph and replayed per step (BASELINE config 5; reference loop kantts/models/sambert/kantts_sambert.py:569-610,
// HybridAttentionDecoder.infer :208-253, MultiHeadPNCAAttention.forward under update_x_state / update_h_state
// kantts/models/sambert/__init__.py:217-306, VarRnnARPredictor.infer adaptors.py:67-83).
//
//   pnca_decode_step : per (sequence, head): append this step's K / V to the cache, then BOTH attentions of the PNCA block
//                      (causal band over the decoder's own cache, look-ahead band over the memory K / V) in one launch.
//                      The reference rebuilds two L x L masks, torch.cat's the cache and runs two bmm / softmax / bmm chains.
//   step_rows        : dst[b, step*dst_ss + e] = src[b, step*src_ss + e] -- gathers memory[:, step, :] and scatters the
//                      step's output frame without a host-side slice whose address would change per step.
//   step_rowmask     : mask[b] = step >= lens[b] (rows the reference zeroes after every sub-layer).
#include "common.h"

#define DDH 16

__device__ __forceinline__ void d_load16(const float* p, float* r) {
  const float4* p4 = reinterpret_cast<const float4*>(p);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float4 t = p4[e];
    r[4 * e + 0] = t.x; r[4 * e + 1] = t.y; r[4 * e + 2] = t.z; r[4 * e + 3] = t.w;
  }
}
__device__ __forceinline__ void d_store16(float* p, const float* r) {
  float4* p4 = reinterpret_cast<float4*>(p);
#pragma unroll
  for (int e = 0; e < 4; ++e) p4[e] = make_float4(r[4 * e], r[4 * e + 1], r[4 * e + 2], r[4 * e + 3]);
}
__device__ __forceinline__ float d_dot16(const float* a, const float* b) {
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < DDH; ++d) s = fmaf(a[d], b[d], s);
  return s;
}

// softmax(q . K[lo..hi]^T / 4) V[lo..hi] for one head; rows of the (B, L, ld) buffer, head slice at +h*16
__device__ __forceinline__ void d_attend(const float* q, const float* kb, const float* vb, int ld, int lo, int hi, float* o) {
#pragma unroll
  for (int d = 0; d < DDH; ++d) o[d] = 0.f;
  float m = -INFINITY;
  for (int j = lo; j <= hi; ++j) {
    float kk[DDH];
    d_load16(kb + (long long)j * ld, kk);
    m = fmaxf(m, d_dot16(q, kk) * 0.25f);
  }
  float l = 0.f;
  for (int j = lo; j <= hi; ++j) {
    float kk[DDH], vv[DDH];
    d_load16(kb + (long long)j * ld, kk);
    d_load16(vb + (long long)j * ld, vv);
    const float e = expf(d_dot16(q, kk) * 0.25f - m);
    l += e;
#pragma unroll
    for (int d = 0; d < DDH; ++d) o[d] = fmaf(e, vv[d], o[d]);
  }
  const float inv = (hi >= lo) ? 1.f / l : 0.f;
#pragma unroll
  for (int d = 0; d < DDH; ++d) o[d] *= inv;
}

__global__ __launch_bounds__(128) void pnca_decode_step_kernel(const float* __restrict__ qkv, int ldq, float* __restrict__ xkv,
                                                              const float* __restrict__ hkv, float* __restrict__ ox,
                                                              float* __restrict__ oh, const int32_t* __restrict__ lens,
                                                              const int32_t* __restrict__ bw_seq, int B, int H, int L,
                                                              int step_host, const int32_t* __restrict__ step_dev, int bw) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * H) return;
  const int b = t / H, h = t % H;
  const int D = H * DDH;
  const int step = step_dev ? *step_dev : step_host;
  if (step < 0 || step >= L) return;
  const int len = lens ? lens[b] : L;
  const int band = bw_seq ? bw_seq[b] : bw;
  float q[DDH], k[DDH], v[DDH], o[DDH];
  const float* row = qkv + (long long)b * ldq + h * DDH;
  d_load16(row, q);
  d_load16(row + D, k);
  d_load16(row + 2 * D, v);
  // this thread owns head h of sequence b: it appends its own slice and later reads only that slice of older rows
  float* xb = xkv + (long long)b * L * 2 * D + h * DDH;
  d_store16(xb + (long long)step * 2 * D, k);
  d_store16(xb + (long long)step * 2 * D + D, v);
  const bool live = step < len;  // padded query: context 0 (as in the training kernels and kantts_attn_decode)
  // x: causal band [max(0, step - band), step]
  d_attend(q, xb, xb + D, 2 * D, live ? max(0, step - band) : 1, live ? step : 0, o);
  d_store16(ox + (long long)b * D + h * DDH, o);
  // h: look-ahead band [step, min(step + band, L - 1, len - 1)] over the memory K / V
  const float* hb = hkv + (long long)b * L * 2 * D + h * DDH;
  d_attend(q, hb, hb + D, 2 * D, live ? step : 1, live ? min(min(step + band, L - 1), len - 1) : 0, o);
  d_store16(oh + (long long)b * D + h * DDH, o);
}

extern "C" int kantts_pnca_decode_step(const float* qkv, int ldq, float* xkv_cache, const float* hkv, float* ox, float* oh,
                                       const int32_t* lens, const int32_t* bw_seq, int B, int H, int L, int d_head, int step,
                                       const int32_t* step_dev, int bw, void* stream) {
  if (!qkv || !xkv_cache || !hkv || !ox || !oh || B < 0 || H < 1 || L < 1) return KANTTS_E_BADARG;
  if (!step_dev && (step < 0 || step >= L)) return KANTTS_E_BADARG;
  if (d_head != DDH || (ldq & 3)) return KANTTS_E_UNSUPPORTED;
  if (B == 0) return KANTTS_OK;
  hipLaunchKernelGGL(pnca_decode_step_kernel, dim3(kantts_cdiv((long long)B * H, 128)), dim3(128), 0, (hipStream_t)stream, qkv,
                     ldq, xkv_cache, hkv, ox, oh, lens, bw_seq, B, H, L, step, step_dev, bw);
  KANTTS_CHECK_LAUNCH();
}

__global__ __launch_bounds__(256) void step_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int n,
                                                       long long src_bs, long long dst_bs, long long src_ss, long long dst_ss,
                                                       int step_host, const int32_t* __restrict__ step_dev) {
  const int step = step_dev ? *step_dev : step_host;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * n) return;
  const int b = (int)(i / n), e = (int)(i % n);
  dst[b * dst_bs + step * dst_ss + e] = src[b * src_bs + step * src_ss + e];
}

extern "C" int kantts_step_rows(const float* src, float* dst, int B, int n, long long src_batch_stride,
                                long long dst_batch_stride, long long src_step_stride, long long dst_step_stride, int step,
                                const int32_t* step_dev, void* stream) {
  if (!src || !dst || B < 0 || n < 0) return KANTTS_E_BADARG;
  if (B == 0 || n == 0) return KANTTS_OK;
  hipLaunchKernelGGL(step_rows_kernel, dim3(kantts_cdiv((long long)B * n, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, B,
                     n, src_batch_stride, dst_batch_stride, src_step_stride, dst_step_stride, step, step_dev);
  KANTTS_CHECK_LAUNCH();
}

__global__ void step_rowmask_kernel(const int32_t* __restrict__ lens, uint8_t* __restrict__ mask, int B, int step_host,
                                    const int32_t* __restrict__ step_dev) {
  const int step = step_dev ? *step_dev : step_host;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) mask[b] = (step >= lens[b]) ? 1 : 0;
}

extern "C" int kantts_step_rowmask(const int32_t* lens, uint8_t* mask, int B, int step, const int32_t* step_dev, void* stream) {
  if (!lens || !mask || B < 0) return KANTTS_E_BADARG;
  if (B == 0) return KANTTS_OK;
  hipLaunchKernelGGL(step_rowmask_kernel, dim3(kantts_cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream, lens, mask, B, step,
                     step_dev);
  KANTTS_CHECK_LAUNCH();
}
