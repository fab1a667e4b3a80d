// Issue rate of the VALU instructions the recurrence kernels choose between (gfx950): cycles per wave64 instruction on one
// SIMD, one wave per SIMD (256 threads per workgroup, one workgroup per CU), four independent accumulator chains.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate scripts/bench_native/valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define N_IT 4096
template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float* out, unsigned seed, long long* cycles) {
  float a0 = seed * 1e-9f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a0}, p3 = {a3, a2};
  const unsigned u = seed ^ threadIdx.x;
  const bf2 wb = __builtin_bit_cast(bf2, u | 0x3f803f80u), hb = __builtin_bit_cast(bf2, (u >> 3) | 0x3f003f00u);
  const h2 wh = __builtin_bit_cast(h2, (u & 0x03ff03ffu) | 0x38003800u), hh = __builtin_bit_cast(h2, ((u >> 2) & 0x03ff03ffu) | 0x34003400u);
  float wv[8], hv[8];  // unknown to the compiler: no folding of the chains
#pragma unroll
  for (int j = 0; j < 8; ++j) { wv[j] = out[j] + 1.0001f; hv[j] = out[8 + j] + 0.9999f; }
  const float w = wv[0], h = hv[0];
  const f2 wp = {w, h}, hp = {h, w};
  const long long t0 = clock64();
#pragma unroll 1
  for (int i = 0; i < N_IT; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (KIND == 0) {  // (inline asm: left alone, the compiler pairs the chains into v_pk_fma_f32)
        asm volatile("v_fma_f32 %0, %4, %0, %5\n\tv_fma_f32 %1, %4, %1, %5\n\tv_fma_f32 %2, %4, %2, %5\n\tv_fma_f32 %3, %4, %3, %5"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(wv[j]), "v"(hv[j]));
      }
      if (KIND == 1) { a0 = __builtin_amdgcn_fdot2_f32_bf16(wb, hb, a0, false); a1 = __builtin_amdgcn_fdot2_f32_bf16(wb, hb, a1, false);
                       a2 = __builtin_amdgcn_fdot2_f32_bf16(wb, hb, a2, false); a3 = __builtin_amdgcn_fdot2_f32_bf16(wb, hb, a3, false); }
      if (KIND == 2) { a0 = __builtin_amdgcn_fdot2(wh, hh, a0, false); a1 = __builtin_amdgcn_fdot2(wh, hh, a1, false);
                       a2 = __builtin_amdgcn_fdot2(wh, hh, a2, false); a3 = __builtin_amdgcn_fdot2(wh, hh, a3, false); }
      if (KIND == 3) { p0 = __builtin_elementwise_fma(wp, hp, p0); p1 = __builtin_elementwise_fma(wp, hp, p1);
                       p2 = __builtin_elementwise_fma(wp, hp, p2); p3 = __builtin_elementwise_fma(wp, hp, p3); }
    }
  }
  const long long t1 = clock64();
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}
template <int KIND>
static void run(const char* name, float* out, long long* cyc) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_kernel<KIND>, dim3(256), dim3(256), 0, 0, out, 1u, cyc);
  hipEventRecord(e0);
  hipLaunchKernelGGL(rate_kernel<KIND>, dim3(256), dim3(256), 0, 0, out, 2u, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  long long c = 0;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = (double)N_IT * 32;
  printf("%-28s %8.3f ms  %.2f ns per instruction = %.2f cycles at 2.4 GHz; s_memtime ticks per instruction %.3f\n", name, ms,
         ms * 1e6 / n, ms * 1e6 / n * 2.4, (double)c / n);
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  run<0>("v_fma_f32", out, cyc);
  run<1>("v_dot2_f32_bf16", out, cyc);
  run<2>("v_dot2_f32_f16", out, cyc);
  run<3>("v_pk_fma_f32", out, cyc);
  return 0;
}
