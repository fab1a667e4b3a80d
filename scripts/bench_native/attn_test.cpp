// Stand-alone check + timing of the attention entry points of csrc/attn.hip on a GPU box (no Python, no torch):
//   hipcc --offload-arch=gfx950 -O2 -c scripts/bench_native/attn_test.cpp -o /tmp/attn_test.o
//   hipcc --offload-arch=gfx950 /tmp/attn_test.o kan-tts_amd/csrc/attn.o -o scripts/bench_native/attn_test
// (attn.o is left in csrc/ by __graft_entry__.build()).
// Cases: the PNCA pair (causal band over x + look-ahead band over the memory, shared queries) and key-padding self
// attention, at the training shapes of SAM-BERT (B = 32, H = 8, d_head = 16; decoder L = 204, encoder L = 64) and a few
// odd ones.  Sampled (sequence, head, position) triples are checked against a double-precision evaluation of the
// definition in include/kantts_hip.h (dropout off), forward and backward; then the launches are timed with HIP events.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "../../include/kantts_hip.h"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

static uint64_t rng_s = 0x9e3779b9ull;
static inline float frand() {
  rng_s ^= rng_s << 13;
  rng_s ^= rng_s >> 7;
  rng_s ^= rng_s << 17;
  return (float)((rng_s >> 11) & 0xffffff) / 8388608.0f - 1.0f;
}
template <typename T>
static T* dev(const std::vector<T>& h) {
  T* d;
  CK(hipMalloc(&d, std::max<size_t>(h.size(), 1) * sizeof(T)));
  CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}
template <typename T>
static std::vector<T> host(const T* d, size_t n) {
  std::vector<T> h(n);
  CK(hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost));
  return h;
}

// allowed keys of query i (include/kantts_hip.h: mode 0 key padding, 1 causal band, 2 look-ahead band)
static void key_range(int mode, int i, int len, int L, int bw, int& lo, int& hi) {
  if (mode == 0) { lo = 0; hi = len - 1; }
  else if (i >= len) { lo = 0; hi = -1; }  // padded query rows of the band modes: skipped (context 0, no gradient)
  else if (mode == 1) { lo = std::max(0, i - bw); hi = i; }
  else { lo = i; hi = std::min(std::min(i + bw, L - 1), len - 1); }
}

struct Band {
  const float* q; int ldq;  // host pointers, element (b, t, h, d) at [(b*L + t)*ld + h*16 + d]
  const float* k; int ldk;
  const float* v; int ldv;
  int mode, bw;
};
static void ref_fwd(const Band& a, int L, int len, int b, int h, int i, double* o) {
  int lo, hi;
  key_range(a.mode, i, len, L, a.bw, lo, hi);
  for (int d = 0; d < 16; ++d) o[d] = 0;
  if (hi < lo) return;
  std::vector<double> s(hi - lo + 1);
  double m = -1e300, l = 0;
  for (int j = lo; j <= hi; ++j) {
    double acc = 0;
    for (int d = 0; d < 16; ++d) acc += (double)a.q[((size_t)b * L + i) * a.ldq + h * 16 + d] * a.k[((size_t)b * L + j) * a.ldk + h * 16 + d];
    s[j - lo] = acc * 0.25;
    m = std::max(m, s[j - lo]);
  }
  for (int j = lo; j <= hi; ++j) l += exp(s[j - lo] - m);
  for (int j = lo; j <= hi; ++j) {
    const double p = exp(s[j - lo] - m) / l;
    for (int d = 0; d < 16; ++d) o[d] += p * a.v[((size_t)b * L + j) * a.ldv + h * 16 + d];
  }
}
// dq of query i for one band given d_o (B*L, D)
static void ref_dq(const Band& a, const float* d_o, int D, int L, int len, int b, int h, int i, double* dq) {
  int lo, hi;
  key_range(a.mode, i, len, L, a.bw, lo, hi);
  for (int d = 0; d < 16; ++d) dq[d] = 0;
  if (hi < lo) return;
  const int n = hi - lo + 1;
  std::vector<double> p(n), dp(n);
  double m = -1e300, l = 0, Dv = 0;
  for (int j = lo; j <= hi; ++j) {
    double acc = 0, g = 0;
    for (int d = 0; d < 16; ++d) {
      acc += (double)a.q[((size_t)b * L + i) * a.ldq + h * 16 + d] * a.k[((size_t)b * L + j) * a.ldk + h * 16 + d];
      g += (double)d_o[((size_t)b * L + i) * D + h * 16 + d] * a.v[((size_t)b * L + j) * a.ldv + h * 16 + d];
    }
    p[j - lo] = acc * 0.25;
    dp[j - lo] = g;
    m = std::max(m, p[j - lo]);
  }
  for (int j = 0; j < n; ++j) { p[j] = exp(p[j] - m); l += p[j]; }
  for (int j = 0; j < n; ++j) { p[j] /= l; Dv += p[j] * dp[j]; }
  for (int j = lo; j <= hi; ++j) {
    const double ds = p[j - lo] * (dp[j - lo] - Dv) * 0.25;
    for (int d = 0; d < 16; ++d) dq[d] += ds * a.k[((size_t)b * L + j) * a.ldk + h * 16 + d];
  }
}



struct Timer {
  hipEvent_t e0, e1;
  Timer() { CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); }
  template <typename F>
  float us(hipStream_t st, int iters, F f) {
    f();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
  }
};

static int run_case(const char* name, int B, int H, int L, int bw, int iters) {
  const int D = H * 16;
  const size_t M = (size_t)B * L;
  std::vector<float> qkv(M * 3 * D), hkv(M * 2 * D), dox(M * D), doh(M * D);
  for (auto& x : qkv) x = frand();
  for (auto& x : hkv) x = frand();
  for (auto& x : dox) x = frand();
  for (auto& x : doh) x = frand();
  std::vector<int32_t> lens(B);
  for (int b = 0; b < B; ++b) lens[b] = L / 2 + (int)((uint32_t)(frand() * 1e6f + 2e6f) % (uint32_t)(L - L / 2 + 1));
  lens[0] = L;
  float *d_qkv = dev(qkv), *d_hkv = dev(hkv), *d_dox = dev(dox), *d_doh = dev(doh);
  int32_t* d_lens = dev(lens);
  float *ox, *oh, *lx, *lh, *dqkv, *dqh, *dhkv, *dvec;
  CK(hipMalloc(&ox, M * D * 4)); CK(hipMalloc(&oh, M * D * 4));
  CK(hipMalloc(&lx, (size_t)B * H * L * 4)); CK(hipMalloc(&lh, (size_t)B * H * L * 4)); CK(hipMalloc(&dvec, (size_t)B * H * L * 4));
  CK(hipMalloc(&dqkv, M * 3 * D * 4)); CK(hipMalloc(&dqh, M * D * 4)); CK(hipMalloc(&dhkv, M * 2 * D * 4));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  Timer T;
  int bad = 0;

  // ---- PNCA pair
  int rc = kantts_pnca_attn_fwd(d_qkv, d_hkv, 2 * D, ox, oh, lx, lh, d_lens, nullptr, bw, bw, B, H, L, 16, 0.f, 0, 0, nullptr, st);
  if (rc != 0) { printf("%-28s pnca fwd rc %d (long sequences use the per-band calls)\n", name, rc); }
  else {
    rc = kantts_pnca_attn_bwd(d_qkv, d_hkv, 2 * D, ox, oh, d_dox, d_doh, lx, lh, dqkv, dqh, dhkv, d_lens, nullptr, bw, bw, B, H,
                              L, 16, 0.f, 0, 0, nullptr, st);
    CK(hipStreamSynchronize(st));
    auto h_ox = host(ox, M * D), h_oh = host(oh, M * D), h_dqkv = host(dqkv, M * 3 * D), h_dqh = host(dqh, M * D);
    Band bx = {qkv.data(), 3 * D, qkv.data() + D, 3 * D, qkv.data() + 2 * D, 3 * D, 1, bw};
    Band bh = {qkv.data(), 3 * D, hkv.data(), 2 * D, hkv.data() + D, 2 * D, 2, bw};
    double e_o = 0, e_q = 0;
    for (int s = 0; s < 200; ++s) {
      const int b = (int)((uint32_t)(frand() * 1e6f + 2e6f) % B), h = s % H, i = (int)((uint32_t)(frand() * 1e6f + 2e6f) % L);
      double o[16], o2[16], q1[16], q2[16];
      ref_fwd(bx, L, lens[b], b, h, i, o);
      ref_fwd(bh, L, lens[b], b, h, i, o2);
      ref_dq(bx, dox.data(), D, L, lens[b], b, h, i, q1);
      ref_dq(bh, doh.data(), D, L, lens[b], b, h, i, q2);
      for (int d = 0; d < 16; ++d) {
        const size_t r = ((size_t)b * L + i);
        e_o = std::max(e_o, fabs(o[d] - h_ox[r * D + h * 16 + d]));
        e_o = std::max(e_o, fabs(o2[d] - h_oh[r * D + h * 16 + d]));
        const double got = rc == 1 ? (double)h_dqkv[r * 3 * D + h * 16 + d] + h_dqh[r * D + h * 16 + d] : h_dqkv[r * 3 * D + h * 16 + d];
        e_q = std::max(e_q, fabs(q1[d] + q2[d] - got));
      }
    }
    const float tf = T.us(st, iters, [&] { kantts_pnca_attn_fwd(d_qkv, d_hkv, 2 * D, ox, oh, lx, lh, d_lens, nullptr, bw, bw, B, H, L, 16, 0.f, 0, 0, nullptr, st); });
    const float tb = T.us(st, iters, [&] { kantts_pnca_attn_bwd(d_qkv, d_hkv, 2 * D, ox, oh, d_dox, d_doh, lx, lh, dqkv, dqh, dhkv, d_lens, nullptr, bw, bw, B, H, L, 16, 0.f, 0, 0, nullptr, st); });
    const bool ok = e_o < 2e-5 && e_q < 2e-4;
    bad += !ok;
    printf("%-28s PNCA pair  : %s  out err %.2e  dq err %.2e   fwd %7.2f us  bwd %7.2f us (rc %d)\n", name, ok ? "ok " : "BAD", e_o,
           e_q, tf, tb, rc);
  }
  // ---- key padding (encoder form) on the same qkv buffer
  rc = kantts_attn_fwd(d_qkv, d_qkv + D, d_qkv + 2 * D, 3 * D, 3 * D, 3 * D, ox, D, lx, nullptr, d_lens, nullptr, 0, B, H, L, 16, 0,
                       0.f, 0, nullptr, st);
  rc |= kantts_attn_bwd(d_qkv, d_qkv + D, d_qkv + 2 * D, 3 * D, 3 * D, 3 * D, ox, D, d_dox, D, lx, dvec, dqkv, dqkv + D,
                        dqkv + 2 * D, 3 * D, 3 * D, 3 * D, 0, d_lens, nullptr, 0, B, H, L, 16, 0, 0.f, 0, nullptr, st);
  CK(hipStreamSynchronize(st));
  if (rc != 0) { printf("%-28s key padding rc %d\n", name, rc); return bad + 1; }
  {
    auto h_ox = host(ox, M * D), h_dqkv = host(dqkv, M * 3 * D);
    Band b0 = {qkv.data(), 3 * D, qkv.data() + D, 3 * D, qkv.data() + 2 * D, 3 * D, 0, 0};
    double e_o = 0, e_q = 0;
    for (int s = 0; s < 100; ++s) {
      const int b = (int)((uint32_t)(frand() * 1e6f + 2e6f) % B), h = s % H, i = (int)((uint32_t)(frand() * 1e6f + 2e6f) % L);
      double o[16], q1[16];
      ref_fwd(b0, L, lens[b], b, h, i, o);
      ref_dq(b0, dox.data(), D, L, lens[b], b, h, i, q1);
      for (int d = 0; d < 16; ++d) {
        const size_t r = ((size_t)b * L + i);
        e_o = std::max(e_o, fabs(o[d] - h_ox[r * D + h * 16 + d]));
        e_q = std::max(e_q, fabs(q1[d] - h_dqkv[r * 3 * D + h * 16 + d]));
      }
    }
    const float tf = T.us(st, iters, [&] { kantts_attn_fwd(d_qkv, d_qkv + D, d_qkv + 2 * D, 3 * D, 3 * D, 3 * D, ox, D, lx, nullptr, d_lens, nullptr, 0, B, H, L, 16, 0, 0.f, 0, nullptr, st); });
    const float tb = T.us(st, iters, [&] { kantts_attn_bwd(d_qkv, d_qkv + D, d_qkv + 2 * D, 3 * D, 3 * D, 3 * D, ox, D, d_dox, D, lx, dvec, dqkv, dqkv + D, dqkv + 2 * D, 3 * D, 3 * D, 3 * D, 0, d_lens, nullptr, 0, B, H, L, 16, 0, 0.f, 0, nullptr, st); });
    const bool ok = e_o < 2e-5 && e_q < 2e-4;
    bad += !ok;
    printf("%-28s key padding: %s  out err %.2e  dq err %.2e   fwd %7.2f us  bwd %7.2f us\n", name, ok ? "ok " : "BAD", e_o, e_q, tf, tb);
  }
  for (void* p : {(void*)d_qkv, (void*)d_hkv, (void*)d_dox, (void*)d_doh, (void*)d_lens, (void*)ox, (void*)oh, (void*)lx, (void*)lh,
                  (void*)dvec, (void*)dqkv, (void*)dqh, (void*)dhkv})
    CK(hipFree(p));
  CK(hipStreamDestroy(st));
  return bad;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 50;
  int bad = 0;
  bad += run_case("decoder B32 L204 band 5", 32, 8, 204, 5, iters);
  bad += run_case("decoder B32 L204 band 20", 32, 8, 204, 20, iters);
  bad += run_case("encoder B32 L64", 32, 8, 64, 5, iters);
  bad += run_case("odd B3 L37 band 0", 3, 8, 37, 0, iters);
  bad += run_case("odd B5 L301 band 7", 5, 8, 301, 7, iters);
  bad += run_case("long B2 L600 band 4", 2, 8, 600, 4, iters);
  printf(bad ? "%d case(s) BAD\n" : "all ok\n", bad);
  return bad ? 1 : 0;
}
