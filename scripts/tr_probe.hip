// Probe of ds_read_b64_tr_b16 (gfx950): prints, for a few lanes, which LDS element each result slot holds.
// Image: s[i] = i (16-bit ints).  Test 1: lane address = 4*lane elements (contiguous image).
// Test 2: lane i of every 16-lane group points at row (i>>2) of a pitch-40 image, column group (i&3)*4, group g at column 16*g.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int mode) {
  __shared__ __attribute__((aligned(16))) short s[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) s[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  int idx = (mode == 0) ? l * 4 : ((l & 15) >> 2) * 40 + (l & 3) * 4 + (l >> 4) * 16;
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(&s[idx]));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = a[j];
}
int main() {
  short* d;
  short h[256];
  hipMalloc(&d, sizeof(h));
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  return 0;
}
