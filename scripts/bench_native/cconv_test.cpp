// Stand-alone check + timing of csrc/cconv.hip on a GPU box (no Python, no torch):
//   hipcc --offload-arch=gfx950 -O2 scripts/bench_native/cconv_test.cpp kan-tts_amd/csrc/cconv.o -o scripts/bench_native/cconv_test
// Every case is checked on sampled outputs against a double-precision evaluation of the formula in include/kantts_hip.h
// (independent of the kernel's tiling / phase decomposition), then timed with HIP events on the launch stream.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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

static uint64_t rng_s = 0x1234567ull;
static inline uint32_t rnd() {
  rng_s ^= rng_s << 13;
  rng_s ^= rng_s >> 7;
  rng_s ^= rng_s << 17;
  return (uint32_t)(rng_s >> 11);
}
static inline float frand() { return (float)(rnd() & 0xffffff) / 8388608.0f - 1.0f; }
static inline uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fff + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}
static inline float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static int floordiv(int a, int b) {
  int q = a / b;
  if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
  return q;
}

struct Case {
  const char* name;
  int B, Tsrc, Tdst, inner, Cin, Cout, groups, K;
  int in_mul, in_add, in_kstep, in_div, phases, up;
  int flags;  // 1 bias, 2 out_act, 4 res, 8 out_gate(bf16), 16 out_gate(fp32), 32 out fp32, 64 out bf16, 128 bf_act
};

static double lrelu(double v, double s) { return v > 0 ? v : v * s; }

static int run_fwd(const Case& c, int tile, int iters) {
  const int CR = c.Cin / c.groups, NG = c.Cout / c.groups;
  const size_t n_in = (size_t)c.B * c.Tsrc * c.inner * c.Cin, n_out = (size_t)c.B * c.Tdst * c.inner * c.Cout;
  const size_t n_w = (size_t)c.K * c.Cout * CR;
  std::vector<uint16_t> h_in(n_in), h_w(n_w), h_gb(n_out);
  std::vector<float> h_bias(c.Cout), h_res(n_out), h_gf(n_out);
  for (auto& v : h_in) v = f2bf(frand());
  const float ws = 1.0f / sqrtf((float)(c.K * CR));
  for (auto& v : h_w) v = f2bf(frand() * ws);
  for (auto& v : h_bias) v = frand() * 0.1f;
  for (auto& v : h_res) v = frand();
  for (size_t i = 0; i < n_out; ++i) {
    const float q = frand();
    h_gb[i] = f2bf(q);
    h_gf[i] = q;
  }
  void *d_in, *d_w, *d_gb, *d_obf;
  float *d_bias, *d_res, *d_gf, *d_out;
  CK(hipMalloc(&d_in, n_in * 2));
  CK(hipMalloc(&d_w, n_w * 2));
  CK(hipMalloc(&d_gb, n_out * 2));
  CK(hipMalloc(&d_obf, n_out * 2));
  CK(hipMalloc(&d_bias, c.Cout * 4));
  CK(hipMalloc(&d_res, n_out * 4));
  CK(hipMalloc(&d_gf, n_out * 4));
  CK(hipMalloc(&d_out, n_out * 4));
  CK(hipMemcpy(d_in, h_in.data(), n_in * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_w, h_w.data(), n_w * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_gb, h_gb.data(), n_out * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_bias, h_bias.data(), c.Cout * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_res, h_res.data(), n_out * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_gf, h_gf.data(), n_out * 4, hipMemcpyHostToDevice));
  CK(hipMemset(d_out, 0xff, n_out * 4));
  CK(hipMemset(d_obf, 0xff, n_out * 2));

  kantts_cconv_args a;
  memset(&a, 0, sizeof(a));
  a.in = d_in;
  a.w = d_w;
  a.bias = (c.flags & 1) ? d_bias : nullptr;
  a.res = (c.flags & 4) ? d_res : nullptr;
  a.out_gate = (c.flags & 8) ? d_gb : ((c.flags & 16) ? (void*)d_gf : nullptr);
  a.out_gate_bf16 = (c.flags & 8) ? 1 : 0;
  a.out = (c.flags & 32) ? d_out : nullptr;
  a.out_bf = (c.flags & 64) ? d_obf : nullptr;
  a.B = c.B; a.Tsrc = c.Tsrc; a.Tdst = c.Tdst; a.Cin_tot = c.Cin; a.Ntot = c.Cout; a.CR = CR; a.NG = NG;
  a.groups = c.groups; a.K = c.K;
  a.in_mul = c.in_mul; a.in_add = c.in_add; a.in_kstep = c.in_kstep; a.in_div = c.in_div; a.phases = c.phases;
  a.inner = c.inner; a.up = c.up;
  a.out_act = (c.flags & 2) ? 1 : 0; a.out_slope = 0.1f;
  a.out_gate_slope = 0.25f;
  a.bf_act = (c.flags & 128) ? 1 : 0; a.bf_slope = 0.1f;
  a.tile = tile;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  int rc = kantts_cconv_launch(&a, st);
  if (rc != 0) {
    printf("%-28s tile %6d: launch rc %d\n", c.name, tile, rc);
    return rc == -2 ? 0 : 1;
  }
  CK(hipStreamSynchronize(st));
  std::vector<float> h_out(n_out);
  std::vector<uint16_t> h_obf(n_out);
  CK(hipMemcpy(h_out.data(), d_out, n_out * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h_obf.data(), d_obf, n_out * 2, hipMemcpyDeviceToHost));
  // ---- sampled check
  const int up = c.up > 1 ? c.up : 1;
  double max_err = 0, max_err_bf = 0;
  int bad = 0;
  const int nsamp = 3000;
  for (int s = 0; s < nsamp; ++s) {
    size_t e;
    if (s < 64) e = s;                       // the first outputs
    else if (s < 128) e = n_out - 1 - (s - 64);  // the last outputs
    else e = (((size_t)rnd() << 20) ^ rnd()) % n_out;
    const int n = (int)(e % c.Cout);
    size_t r = e / c.Cout;
    const int p = (int)(r % c.inner);
    r /= c.inner;
    const int d = (int)(r % c.Tdst);
    const int b = (int)(r / c.Tdst);
    const int phase = d % c.phases, m = d / c.phases;
    const int grp = n / NG;
    double acc = (c.flags & 1) ? h_bias[n] : 0.0;
    for (int k = 0; k < c.K; ++k) {
      const int u = c.in_add + phase + k * c.in_kstep;
      const int q = floordiv(u, c.in_div);
      if (q * c.in_div != u) continue;
      const int tu = m * c.in_mul + q;
      if (tu < 0 || tu >= c.Tsrc * up) continue;
      const int t = tu / up;
      const uint16_t* xr = &h_in[(((size_t)b * c.Tsrc + t) * c.inner + p) * c.Cin + (size_t)grp * CR];
      const uint16_t* wr = &h_w[((size_t)k * c.Cout + n) * CR];
      for (int ci = 0; ci < CR; ++ci) acc += (double)bf2f(xr[ci]) * (double)bf2f(wr[ci]);
    }
    double v = acc;
    if (c.flags & 2) v = lrelu(v, 0.1);
    if (c.flags & 4) v += h_res[e];
    if (c.flags & 8) v *= (bf2f(h_gb[e]) > 0) ? 1.0 : 0.25;
    if (c.flags & 16) v *= (h_gf[e] > 0) ? 1.0 : 0.25;
    if (c.flags & 32) {
      const double err = fabs(v - (double)h_out[e]);
      if (err > max_err) max_err = err;
      if (!(err <= 2e-3 + 1e-3 * fabs(v))) ++bad;
    }
    if (c.flags & 64) {
      const double vb = (c.flags & 128) ? lrelu(v, 0.1) : v;
      const double err = fabs(vb - (double)bf2f(h_obf[e]));
      if (err > max_err_bf) max_err_bf = err;
      if (!(err <= 2e-3 + 1e-2 * fabs(vb))) ++bad;
    }
  }
  // ---- timing
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) kantts_cconv_launch(&a, st);
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) kantts_cconv_launch(&a, st);
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1000.0 / iters;
  // useful flops: valid taps only ~ K / in_div
  const double flops = 2.0 * c.B * c.Tdst * c.inner * (double)c.Cout * CR * c.K / c.in_div;
  const double bytes = 2.0 * n_in + 2.0 * n_w + ((c.flags & 32) ? 4.0 : 0.0) * n_out + ((c.flags & 64) ? 2.0 : 0.0) * n_out +
                       ((c.flags & 4) ? 4.0 : 0.0) * n_out + ((c.flags & 8) ? 2.0 : 0.0) * n_out;
  printf("%-28s tile %6d: %s  err %.2e / bf %.2e   %8.1f us  %7.1f TFLOP/s  %6.0f GB/s\n", c.name, tile, bad ? "FAIL" : "ok  ",
         max_err, max_err_bf, us, flops / us * 1e-6, bytes / us * 1e-3);
  CK(hipFree(d_in)); CK(hipFree(d_w)); CK(hipFree(d_gb)); CK(hipFree(d_obf)); CK(hipFree(d_bias)); CK(hipFree(d_res));
  CK(hipFree(d_gf)); CK(hipFree(d_out));
  CK(hipStreamDestroy(st));
  return bad ? 1 : 0;
}

struct WCase {
  const char* name;
  int B, Tsrc, Tdst, inner, Cin, Cout, groups, K, stride, dil, pad, up, slices, bias, use_ws;
};

static int run_wgrad(const WCase& c, int iters) {
  const int CR = c.Cin / c.groups, NG = c.Cout / c.groups;
  const size_t n_x = (size_t)c.B * c.Tsrc * c.inner * c.Cin, n_dy = (size_t)c.B * c.Tdst * c.inner * c.Cout;
  const size_t n_w = (size_t)c.K * c.Cout * CR;
  std::vector<uint16_t> h_x(n_x), h_dy(n_dy);
  for (auto& v : h_x) v = f2bf(frand());
  for (auto& v : h_dy) v = f2bf(frand());
  void *d_x, *d_dy;
  float *d_dw, *d_db;
  CK(hipMalloc(&d_x, n_x * 2));
  CK(hipMalloc(&d_dy, n_dy * 2));
  CK(hipMalloc(&d_dw, n_w * 4));
  CK(hipMalloc(&d_db, c.Cout * 4));
  CK(hipMemcpy(d_x, h_x.data(), n_x * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_dy, h_dy.data(), n_dy * 2, hipMemcpyHostToDevice));
  CK(hipMemset(d_dw, 0, n_w * 4));
  CK(hipMemset(d_db, 0, c.Cout * 4));
  kantts_cconvw_args a;
  memset(&a, 0, sizeof(a));
  a.x = d_x; a.dy = d_dy; a.dw = d_dw; a.db = c.bias ? d_db : nullptr;
  a.B = c.B; a.Tsrc = c.Tsrc; a.Tdst = c.Tdst; a.Cin_tot = c.Cin; a.Ntot = c.Cout; a.CR = CR; a.NG = NG; a.groups = c.groups;
  a.K = c.K; a.stride = c.stride; a.dil = c.dil; a.pad = c.pad; a.inner = c.inner; a.up = c.up; a.slices = c.slices;
  float* d_ws = nullptr;
  if (c.use_ws) {
    a.ws_floats = kantts_cconv_wgrad_ws_floats(&a);
    if (a.ws_floats > 0) {
      CK(hipMalloc(&d_ws, a.ws_floats * 4));
      CK(hipMemset(d_ws, 0xff, a.ws_floats * 4));
      a.workspace = d_ws;
    }
  }
  hipStream_t st;
  CK(hipStreamCreate(&st));
  int rc = kantts_cconv_wgrad_launch(&a, st);
  if (rc != 0) {
    printf("%-28s wgrad: launch rc %d\n", c.name, rc);
    return rc == -2 ? 0 : 1;
  }
  CK(hipStreamSynchronize(st));
  std::vector<float> h_dw(n_w), h_db(c.Cout);
  CK(hipMemcpy(h_dw.data(), d_dw, n_w * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h_db.data(), d_db, c.Cout * 4, hipMemcpyDeviceToHost));
  const int up = c.up > 1 ? c.up : 1;
  int bad = 0;
  double max_rel = 0;
  const double scale = sqrt((double)c.B * c.Tdst * c.inner);  // typical magnitude of an output
  const int nsamp = 400;
  for (int s = 0; s < nsamp; ++s) {
    size_t e;
    if (s < 32) e = s;
    else if (s < 64) e = n_w - 1 - (s - 32);
    else e = (((size_t)rnd() << 20) ^ rnd()) % n_w;
    const int ci = (int)(e % CR);
    const int n = (int)((e / CR) % c.Cout);
    const int k = (int)(e / ((size_t)CR * c.Cout));
    const int grp = n / NG;
    double acc = 0;
    for (int b = 0; b < c.B; ++b)
      for (int q = 0; q < c.Tdst; ++q) {
        const int tu = q * c.stride + k * c.dil - c.pad;
        if (tu < 0 || tu >= c.Tsrc * up) continue;
        const int t = tu / up;
        for (int p = 0; p < c.inner; ++p)
          acc += (double)bf2f(h_dy[(((size_t)b * c.Tdst + q) * c.inner + p) * c.Cout + n]) *
                 (double)bf2f(h_x[(((size_t)b * c.Tsrc + t) * c.inner + p) * c.Cin + (size_t)grp * CR + ci]);
      }
    const double err = fabs(acc - (double)h_dw[e]) / scale;
    if (err > max_rel) max_rel = err;
    if (!(err <= 2e-3)) ++bad;
  }
  double max_db = 0;
  if (c.bias) {
    for (int n = 0; n < c.Cout; n += (c.Cout > 64 ? 7 : 1)) {
      double acc = 0;
      for (size_t r = 0; r < (size_t)c.B * c.Tdst * c.inner; ++r) acc += (double)bf2f(h_dy[r * c.Cout + n]);
      const double err = fabs(acc - (double)h_db[n]) / scale;
      if (err > max_db) max_db = err;
      if (!(err <= 2e-3)) ++bad;
    }
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) kantts_cconv_wgrad_launch(&a, st);
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) kantts_cconv_wgrad_launch(&a, st);
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1000.0 / iters;
  const double flops = 2.0 * c.B * c.Tdst * c.inner * (double)c.Cout * CR * c.K;
  printf("%-28s wgrad s=%d ws=%lld: %s  rel err %.2e  db %.2e   %8.1f us  %7.1f TFLOP/s\n", c.name, c.slices,
         (long long)a.ws_floats, bad ? "FAIL" : "ok  ", max_rel, max_db, us, flops / us * 1e-6);
  CK(hipFree(d_x)); CK(hipFree(d_dy)); CK(hipFree(d_dw)); CK(hipFree(d_db));
  if (d_ws) CK(hipFree(d_ws));
  CK(hipStreamDestroy(st));
  return bad ? 1 : 0;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20;
  const bool quick = argc > 2 && !strcmp(argv[2], "quick");
  int fails = 0;
  // name, B, Tsrc, Tdst, inner, Cin, Cout, groups, K, in_mul, in_add, in_kstep, in_div, phases, up, flags
  const int F = 1 | 32;            // bias, fp32 out
  const int FB = 1 | 2 | 32 | 64;  // bias, out act, both outputs
  std::vector<Case> small = {
      {"tiny k3", 2, 40, 40, 1, 32, 32, 1, 3, 1, -1, 1, 1, 1, 1, F},
      {"tiny k3 c64 res", 2, 50, 50, 1, 64, 64, 1, 3, 1, -1, 1, 1, 1, 1, F | 4 | 64 | 128},
      {"ragged c80->n136 k7", 3, 37, 37, 1, 80, 136, 1, 7, 1, -6, 1, 1, 1, 1, FB},
      {"dil5 k11 c128 causal", 2, 300, 300, 1, 128, 128, 1, 11, 1, -50, 5, 1, 1, 1, F | 4},
      {"stride3 inner5 k5", 3, 61, 21, 5, 32, 128, 1, 5, 3, -2, 1, 1, 1, 1, FB},
      {"dgrad stride3 inner5", 3, 21, 61, 5, 128, 32, 1, 5, 1, 2, -1, 3, 3, 1, F | 8},
      {"dgrad stride2 k41 g4", 2, 200, 400, 1, 128, 128, 4, 41, 1, 20, -1, 2, 2, 1, 32 | 16},
      {"grouped g4 k41 s2", 2, 400, 200, 1, 128, 128, 4, 41, 2, -20, 1, 1, 1, 1, FB},
      {"up8 k7 c64->32", 2, 30, 240, 1, 64, 32, 1, 7, 1, -6, 1, 1, 1, 8, F},
      {"dgrad of up2 (mul2)", 2, 100, 50, 1, 32, 64, 1, 8, 2, -5, 1, 1, 1, 1, F},
      {"polyphase convT 2tap", 2, 64, 64, 1, 128, 256, 1, 2, 1, 0, -1, 1, 1, 1, F | 4},
      {"k1 c256", 2, 77, 77, 1, 256, 64, 1, 1, 1, 0, 1, 1, 1, 1, F},
      {"c8 n8", 2, 33, 33, 1, 8, 8, 1, 5, 1, -2, 1, 1, 1, 1, F},
      {"inner11 tdst10", 4, 10, 10, 11, 64, 64, 1, 5, 1, -2, 1, 1, 1, 1, FB},
      // window (narrow) form: CR <= 64, one batch item per workgroup
      {"narrow c32 k11 dil5 res", 3, 300, 300, 1, 32, 32, 1, 11, 1, -50, 5, 1, 1, 1, F | 4 | 64 | 128},
      {"narrow c64->n72 k7", 2, 333, 333, 1, 64, 72, 1, 7, 1, -3, 1, 1, 1, 1, FB},
      {"narrow c24->n40 k5 s3", 3, 400, 134, 1, 24, 40, 1, 5, 3, -2, 1, 1, 1, 1, F},
      {"narrow dgrad s3 40<-24", 3, 134, 400, 1, 40, 24, 1, 5, 1, 2, -1, 3, 3, 1, 32 | 8},
      {"narrow c64 k41 g2 s4", 2, 1100, 275, 1, 128, 128, 2, 41, 4, -20, 1, 1, 1, 1, FB},
      {"narrow c32 k3 gate T70", 2, 70, 70, 1, 32, 32, 1, 3, 1, -2, 1, 1, 1, 1, 32 | 16 | 4},
      {"narrow c8 n8 k5 T100", 2, 100, 100, 1, 8, 8, 1, 5, 1, -2, 1, 1, 1, 1, F},
  };
  for (auto& c : small) {
    fails += run_fwd(c, 0, 2);
    if (!quick) {
      fails += run_fwd(c, 128128, 2);
      fails += run_fwd(c, 256064, 2);
      fails += run_fwd(c, 128064, 2);
      fails += run_fwd(c, 256032, 2);
    }
  }
  std::vector<WCase> wsmall = {
      // name, B, Tsrc, Tdst, inner, Cin, Cout, groups, K, stride, dil, pad, up, slices, bias
      {"w tiny", 2, 70, 70, 1, 64, 64, 1, 3, 1, 1, 1, 1, 1, 1, 0},
      {"w c128 k7 dil3", 3, 200, 200, 1, 128, 128, 1, 7, 1, 3, 18, 1, 1, 1, 0},
      {"w c128 k7 dil3 s3", 3, 200, 200, 1, 128, 128, 1, 7, 1, 3, 18, 1, 3, 1, 0},
      {"w c128 k7 dil3 s3 ws", 3, 200, 200, 1, 128, 128, 1, 7, 1, 3, 18, 1, 3, 1, 1},
      {"w stride3 inner5", 3, 61, 21, 5, 64, 256, 1, 5, 3, 1, 2, 1, 0, 1, 0},
      {"w ragged c72 n136", 2, 90, 90, 1, 72, 136, 1, 3, 1, 1, 1, 1, 2, 1, 0},
      {"w ragged c72 n136 ws", 2, 90, 90, 1, 72, 136, 1, 3, 1, 1, 1, 1, 2, 1, 1},
      {"w grouped g2", 2, 120, 60, 1, 128, 256, 2, 9, 2, 1, 4, 1, 0, 1, 1},
      {"w up4", 2, 50, 200, 1, 64, 64, 1, 7, 1, 1, 6, 4, 0, 0, 1},
      {"w c256 n64", 2, 130, 130, 1, 256, 64, 1, 3, 1, 1, 1, 1, 0, 1, 1},
      {"w long c64 auto ws", 4, 3000, 3000, 1, 64, 64, 1, 3, 1, 1, 1, 1, 0, 1, 1},
      {"w taps c32 k11 dil5", 3, 333, 333, 1, 32, 32, 1, 11, 1, 5, 50, 1, 0, 1, 1},
      {"w taps c32 k41 s2 g4", 2, 500, 250, 1, 128, 128, 4, 41, 2, 1, 20, 1, 0, 1, 1},
      {"w taps 32x64 k41 s4 g2", 2, 600, 150, 1, 64, 128, 2, 41, 4, 1, 20, 1, 0, 1, 1},
      {"w taps 64x32 k7", 2, 200, 200, 1, 64, 32, 1, 7, 1, 1, 3, 1, 3, 1, 0},
      {"w taps c24 n40 k5 s3", 3, 100, 34, 1, 24, 40, 1, 5, 3, 1, 2, 1, 1, 1, 0},
      {"w taps c64 k3 1 slice", 1, 70, 70, 1, 64, 64, 1, 3, 1, 1, 1, 1, 1, 0, 0},
  };
  for (auto& c : wsmall) fails += run_wgrad(c, 2);
  printf("---- small cases: %d failures\n", fails);

  // ---- bench shapes of the HiFi-GAN V1 step at batch 32 x 8192 (profiles/r02_runH_conv_shapes_cap12.log)
  std::vector<Case> big = {
      {"mpd 1024->1024 k5 p11", 64, 10, 10, 11, 1024, 1024, 1, 5, 1, -2, 1, 1, 1, 1, FB},
      {"mpd 1024->1024 k5 p2", 64, 51, 51, 2, 1024, 1024, 1, 5, 1, -2, 1, 1, 1, 1, FB},
      {"mpd 512->1024 s3 p3", 64, 102, 34, 3, 512, 1024, 1, 5, 3, -2, 1, 1, 1, 1, FB},
      {"mpd 128->512 s3 p2", 64, 456, 152, 2, 128, 512, 1, 5, 3, -2, 1, 1, 1, 1, FB},
      {"mpd dgrad 512<-1024 s3 p3", 64, 34, 102, 3, 1024, 512, 1, 5, 1, 2, -1, 3, 3, 1, 32 | 8},
      {"msd 1024->1024 k5", 64, 128, 128, 1, 1024, 1024, 1, 5, 1, -2, 1, 1, 1, 1, FB},
      {"gen res c256 k11 T256", 32, 256, 256, 1, 256, 256, 1, 11, 1, -10, 1, 1, 1, 1, F | 4},
      {"gen res c256 k3 d5", 32, 256, 256, 1, 256, 256, 1, 3, 1, -10, 5, 1, 1, 1, F},
      {"gen res c128 k11 T2048", 32, 2048, 2048, 1, 128, 128, 1, 11, 1, -10, 1, 1, 1, 1, F | 4},
      {"gen res c128 k7 bf", 32, 2048, 2048, 1, 128, 128, 1, 7, 1, -6, 1, 1, 1, 1, 1 | 2 | 64},
      {"gen res c64 k11 T4096", 32, 4096, 4096, 1, 64, 64, 1, 11, 1, -10, 1, 1, 1, 1, F | 4},
      {"gen res c64 k11 bf", 32, 4096, 4096, 1, 64, 64, 1, 11, 1, -10, 1, 1, 1, 1, 1 | 2 | 64},
      {"gen res c32 k11 T8192", 32, 8192, 8192, 1, 32, 32, 1, 11, 1, -10, 1, 1, 1, 1, F | 4},
      {"gen res c32 k11 bf", 32, 8192, 8192, 1, 32, 32, 1, 11, 1, -10, 1, 1, 1, 1, 1 | 2 | 64},
      {"gen res c32 k3 bf", 32, 8192, 8192, 1, 32, 32, 1, 3, 1, -2, 1, 1, 1, 1, 1 | 2 | 64},
      {"convT 512->256x8 poly", 32, 32, 32, 1, 512, 2048, 1, 2, 1, 0, -1, 1, 1, 1, 1 | 64},
      {"convT 256->128x8 poly", 32, 256, 256, 1, 256, 1024, 1, 2, 1, 0, -1, 1, 1, 1, 1 | 64},
      {"convT 128->64x2 poly", 32, 2048, 2048, 1, 128, 128, 1, 2, 1, 0, -1, 1, 1, 1, 1 | 64},
      {"rep up8 512->256 k7", 32, 32, 256, 1, 512, 256, 1, 7, 1, -6, 1, 1, 1, 8, 1 | 64},
      {"rep up8 256->128 k7", 32, 256, 2048, 1, 256, 128, 1, 7, 1, -6, 1, 1, 1, 8, 1 | 64},
      {"msd g16 1024->1024 k41", 32, 130, 130, 1, 1024, 1024, 16, 41, 1, -20, 1, 1, 1, 1, FB},
      {"msd g4 128->128 k41 s2", 32, 8192, 4096, 1, 128, 128, 4, 41, 2, -20, 1, 1, 1, 1, FB},
      {"msd g4 dgrad s2 B64", 64, 4096, 8192, 1, 128, 128, 4, 41, 1, 20, -1, 2, 2, 1, 32 | 8},
      {"msd packed 32->64 k41 s2", 32, 4096, 2048, 1, 128, 256, 4, 41, 2, -20, 1, 1, 1, 1, FB},
      {"gen res c64 k3 d3 bf", 32, 4096, 4096, 1, 64, 64, 1, 3, 1, -6, 3, 1, 1, 1, 1 | 2 | 64},
      {"gen res c32 k7 res+img", 32, 8192, 8192, 1, 32, 32, 1, 7, 1, -6, 1, 1, 1, 1, F | 4 | 64 | 128},
  };
  for (auto& c : big) {
    fails += run_fwd(c, 0, iters);
    const int NG = c.Cout / c.groups;
    if (c.Cin / c.groups <= 64 && c.inner == 1 && c.up <= 1) fails += run_fwd(c, c.Cout / c.groups > 32 ? 128064 : 256032, iters);
    if (!quick) {
      if (NG > 64) fails += run_fwd(c, 256064, iters);
      if (NG <= 64 && NG > 32) { fails += run_fwd(c, 128064, iters); fails += run_fwd(c, 256064, iters); }
    }
  }
  std::vector<WCase> wbig = {
      {"w mpd 1024x1024 k5 p11", 64, 10, 10, 11, 1024, 1024, 1, 5, 1, 1, 2, 1, 0, 1, 1},
      {"w mpd 1024x1024 k5 p2", 64, 51, 51, 2, 1024, 1024, 1, 5, 1, 1, 2, 1, 0, 1, 1},
      {"w mpd 512->1024 s3 p3", 64, 102, 34, 3, 512, 1024, 1, 5, 3, 1, 2, 1, 0, 1, 1},
      {"w mpd 128->512 s3 p2", 64, 456, 152, 2, 128, 512, 1, 5, 3, 1, 2, 1, 0, 1, 1},
      {"w msd 1024x1024 k5", 64, 128, 128, 1, 1024, 1024, 1, 5, 1, 1, 2, 1, 0, 1, 1},
      {"w gen c256 k11 T256", 32, 256, 256, 1, 256, 256, 1, 11, 1, 1, 10, 1, 0, 1, 1},
      {"w gen c128 k11 T2048", 32, 2048, 2048, 1, 128, 128, 1, 11, 1, 1, 10, 1, 0, 1, 1},
      {"w gen c128 k3 T2048", 32, 2048, 2048, 1, 128, 128, 1, 3, 1, 1, 2, 1, 0, 1, 1},
      {"w gen c64 k11 T4096", 32, 4096, 4096, 1, 64, 64, 1, 11, 1, 1, 10, 1, 0, 1, 1},
      {"w convT 256->1024 2tap", 32, 256, 256, 1, 256, 1024, 1, 2, 1, 1, 1, 1, 0, 1, 1},
      {"w msd g16 k41", 32, 130, 130, 1, 1024, 1024, 16, 41, 1, 1, 20, 1, 0, 1, 1},
        {"w gen c32 k11 T8192", 32, 8192, 8192, 1, 32, 32, 1, 11, 1, 1, 10, 1, 0, 1, 1},
      {"w gen c32 k3 T8192", 32, 8192, 8192, 1, 32, 32, 1, 3, 1, 1, 2, 1, 0, 1, 1},
      {"w gen c64 k3 T4096", 32, 4096, 4096, 1, 64, 64, 1, 3, 1, 1, 2, 1, 0, 1, 1},
      {"w msd g4 k41 s2 T8192", 64, 8192, 4096, 1, 128, 128, 4, 41, 2, 1, 20, 1, 0, 1, 1},
      {"w msd packed g4 k41 (32x64)", 64, 4096, 2048, 1, 128, 256, 4, 41, 2, 1, 20, 1, 0, 1, 1},
      {"w gen c128 k11 noWS", 32, 2048, 2048, 1, 128, 128, 1, 11, 1, 1, 10, 1, 0, 1, 0},
};
  for (auto& c : wbig) fails += run_wgrad(c, iters);
  printf("==== total failures: %d\n", fails);
  return fails ? 1 : 0;
}
