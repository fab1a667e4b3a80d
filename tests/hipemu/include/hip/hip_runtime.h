// TEST INFRASTRUCTURE -- a source-level stand-in for <hip/hip_runtime.h> that lets the kernel sources of
// kan-tts_amd/csrc/*.hip be compiled for the HOST (x86-64 clang) and executed on the CPU by tests/hipemu/runtime.cpp:
// every thread of a workgroup is a fibre, workgroups run one after another, and everything that makes lanes of a
// wavefront exchange data (shuffles, DPP, readlane, ballots, MFMA, the transposing LDS read, direct-to-LDS loads) is a
// rendezvous of the wave's 64 fibres.  Purpose: `pytest -m "not gpu"` can run the REAL kernel source against the oracle in
// a container without a GPU.  It is never built by __graft_entry__.build(), never loaded by kantts._hip (the product has
// no CPU path); only tests/test_kernel_source_on_cpu.py loads the library this produces.  What is NOT modelled: timing,
// memory coalescing, LDS banks, and wave-synchronous ordering of plain memory accesses between lanes of one wave without
// any wave-level operation in between (kernels that rely on that are listed as not emulated in tests/hipemu/README.md).
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __constant__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __align__(n) __attribute__((aligned(n)))

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct hipemu_idx { unsigned x, y, z; };
extern hipemu_idx threadIdx, blockIdx;
extern dim3 blockDim, gridDim;
static const int warpSize = 64;

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }

// ---- vector types
#define HIPEMU_VEC(T, N2, N3, N4)                                                           \
  struct N2 { T x, y; };                                                                    \
  struct N3 { T x, y, z; };                                                                 \
  struct __attribute__((aligned(sizeof(T) * 4 > 16 ? 16 : sizeof(T) * 4))) N4 { T x, y, z, w; }; \
  static inline N2 make_##N2(T x, T y) { return N2{x, y}; }                                 \
  static inline N3 make_##N3(T x, T y, T z) { return N3{x, y, z}; }                         \
  static inline N4 make_##N4(T x, T y, T z, T w) { return N4{x, y, z, w}; }
HIPEMU_VEC(float, float2, float3, float4)
HIPEMU_VEC(int, int2, int3, int4)
HIPEMU_VEC(unsigned, uint2, uint3, uint4)
HIPEMU_VEC(unsigned short, ushort2, ushort3, ushort4)
HIPEMU_VEC(short, short2, short3, short4)
#undef HIPEMU_VEC

// ---- runtime (tests/hipemu/runtime.cpp)
namespace hipemu {
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
void register_dynamic_lds(void* base, size_t bytes);
void sync_threads();
void wave_sync();
int lane();        // 0..63 within the wave
int wave_alive();  // mask helpers
uint64_t alive_mask();
// rendezvous: copy `n` bytes of this lane's payload into the wave's exchange slot, wait for the other lanes; returns the
// table of the 64 slots (64 bytes each) that stays valid until this lane's next-but-one rendezvous
const unsigned char (*post(const void* payload, int n))[64];
// same, but the lane that completes the rendezvous evaluates `fn(slots, results)` once for the whole wave; returns the
// table of the 64 result slots (16 bytes each)
const unsigned char (*post_once(const void* payload, int n, void (*fn)(const unsigned char (*)[64], unsigned char (*)[16])))[16];
}  // namespace hipemu

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
  hipemu::launch(dim3(grid), dim3(block), (size_t)(shmem), [=]() { kern(__VA_ARGS__); })

#define __syncthreads() hipemu::sync_threads()
#define KANTTS_WAVE_ORDERED() hipemu::wave_sync()  // see kan-tts_amd/csrc/common.h
#define KANTTS_OPAQUE_VGPR(x) ((void)0)            // a register-allocation hint on the device (common.h)
#define __builtin_amdgcn_s_barrier() hipemu::sync_threads()
#define __builtin_amdgcn_wave_barrier() hipemu::wave_sync()
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_fence(...) ((void)0)
// fences order memory operations that complete at once here
#define __threadfence_block() ((void)0)
#define __threadfence() ((void)0)
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_fetch_add(p, v, order, scope) hipemu_fetch_add(p, v)
#define __hip_atomic_store(p, v, order, scope) hipemu_atomic_store(p, v)
// atomics are real (relaxed) atomics of the host: one OS thread needs none, but the race-detector build (README.md) has to
// see them as what they are
template <class T>
struct hipemu_bits { typedef T type; };
template <>
struct hipemu_bits<float> { typedef unsigned type; };
template <>
struct hipemu_bits<double> { typedef unsigned long long type; };
template <class T, class F>
static inline T hipemu_atomic_rmw(T* p, F f) {
  typedef typename hipemu_bits<T>::type B;
  static_assert(sizeof(B) == sizeof(T), "atomic width");
  B* q = reinterpret_cast<B*>(p);
  B old = __atomic_load_n(q, __ATOMIC_RELAXED), neu;
  T o;
  do {
    memcpy(&o, &old, sizeof(T));
    const T n = f(o);
    memcpy(&neu, &n, sizeof(T));
  } while (!__atomic_compare_exchange_n(q, &old, neu, false, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED));
  return o;
}
template <class T, class U>
static inline T hipemu_fetch_add(T* p, U v) { return hipemu_atomic_rmw(p, [v](T o) { return (T)(o + v); }); }
template <class T, class U>
static inline T atomicAdd(T* p, U v) { return hipemu_fetch_add(p, v); }
template <class T, class U>
static inline T atomicMax(T* p, U v) { return hipemu_atomic_rmw(p, [v](T o) { return (T)v > o ? (T)v : o; }); }
template <class T, class U>
static inline T atomicExch(T* p, U v) { return hipemu_atomic_rmw(p, [v](T) { return (T)v; }); }
template <class T, class U>
static inline void hipemu_atomic_store(T* p, U v) { hipemu_atomic_rmw(p, [v](T) { return (T)v; }); }

// ---- scalar helpers
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
static inline float hipemu_expf(float x) { return expf(x); }
static inline float hipemu_logf(float x) { return logf(x); }
#define __expf hipemu_expf
#define __logf hipemu_logf
static inline float hipemu_log2f(float x) { return log2f(x); }
#define __log2f hipemu_log2f
static inline float __builtin_amdgcn_sqrtf_emu(float a) { return sqrtf(a); }
#define __builtin_amdgcn_sqrtf __builtin_amdgcn_sqrtf_emu
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __frcp_rn(float a) { return 1.0f / a; }
static inline float rsqrtf(float a) { return 1.0f / sqrtf(a); }
static inline float __builtin_amdgcn_rcpf_emu(float a) { return 1.0f / a; }
#define __builtin_amdgcn_rcpf __builtin_amdgcn_rcpf_emu
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline float __saturatef(float x) { return x < 0.f ? 0.f : (x > 1.f ? 1.f : x); }
using std::max;
using std::min;
static inline float max(float a, double b) { return fmaxf(a, (float)b); }
static inline long long min(long long a, int b) { return a < b ? a : b; }
static inline long long max(long long a, int b) { return a > b ? a : b; }
static inline long long min(int a, long long b) { return a < b ? a : b; }
static inline long long max(int a, long long b) { return a > b ? a : b; }

// ---- wave-level data exchange
template <class T>
static inline T hipemu_from_lane(const T& v, int src) {
  static_assert(sizeof(T) <= 64, "payload");
  auto slots = hipemu::post(&v, (int)sizeof(T));
  T r;
  memcpy(&r, slots[src & 63], sizeof(T));
  return r;
}
template <class T>
static inline T __shfl_xor(T v, int m, int width = 64) { return hipemu_from_lane(v, hipemu::lane() ^ m); }
template <class T>
static inline T __shfl(T v, int src, int width = 64) {
  const int l = hipemu::lane();
  return hipemu_from_lane(v, (l & ~(width - 1)) | (src & (width - 1)));
}
template <class T>
static inline T __shfl_up(T v, int d, int width = 64) {
  const int l = hipemu::lane();
  const int s = ((l & (width - 1)) >= d) ? l - d : l;
  return hipemu_from_lane(v, s);
}
template <class T>
static inline T __shfl_down(T v, int d, int width = 64) {
  const int l = hipemu::lane();
  const int s = ((l & (width - 1)) + d < width) ? l + d : l;
  return hipemu_from_lane(v, s);
}
static inline unsigned long long __ballot(int pred) {
  const int p = pred ? 1 : 0;
  auto slots = hipemu::post(&p, 4);
  unsigned long long m = 0, alive = hipemu::alive_mask();
  for (int i = 0; i < 64; ++i)
    if (((alive >> i) & 1) && *(const int*)slots[i]) m |= 1ull << i;
  return m;
}
static inline int hipemu_readlane(int v, int src) { return hipemu_from_lane(v, src); }
static inline int hipemu_readfirstlane(int v) {
  auto slots = hipemu::post(&v, 4);
  return *(const int*)slots[__builtin_ctzll(hipemu::alive_mask())];
}
#define __builtin_amdgcn_readlane hipemu_readlane
#define __builtin_amdgcn_readfirstlane hipemu_readfirstlane

// DPP controls used by the kernels: quad_perm (0x00-0xFF), row_shl/row_shr (0x101-0x11F), wave_shr:1 (0x138),
// row_ror (0x121-0x12F), row_mirror (0x140), row_half_mirror (0x141); full row / bank masks only
static inline int hipemu_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  const int l = hipemu::lane();
  auto slots = hipemu::post(&src, 4);
  int from = -1;
  if (ctrl <= 0xFF) from = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);
  else if (ctrl >= 0x101 && ctrl <= 0x10F) { const int n = ctrl & 15; from = ((l & 15) + n < 16) ? l + n : -1; }
  else if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl & 15; from = ((l & 15) >= n) ? l - n : -1; }
  else if (ctrl >= 0x121 && ctrl <= 0x12F) { const int n = ctrl & 15; from = (l & ~15) | (((l & 15) - n) & 15); }
  else if (ctrl == 0x138) from = l >= 1 ? l - 1 : -1;
  else if (ctrl == 0x140) from = (l & ~15) | (15 - (l & 15));
  else if (ctrl == 0x141) from = (l & ~7) | (7 - (l & 7));
  else abort();
  if (row_mask != 0xF || bank_mask != 0xF) abort();
  if (from < 0) return bound_ctrl ? 0 : old;
  return *(const int*)slots[from];
}
#define __builtin_amdgcn_update_dpp hipemu_update_dpp
#define __builtin_amdgcn_mov_dpp(src, ctrl, rm, bm, bc) hipemu_update_dpp((src), (src), (ctrl), (rm), (bm), (bc))

// ---- matrix cores (CDNA3/4 operand layouts: A row = lane % 16, B column = lane % 16, k block = lane / 16;
// C/D column = lane % 16, rows 4 * (lane / 16) .. + 3)
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 hipemu_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hipemu_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 hipemu_bf16x2 __attribute__((ext_vector_type(2)));
struct hipemu_mfma_bf16_payload { hipemu_bf16x8 a, b; };
static void hipemu_mfma_16x16x32_bf16_wave(const unsigned char (*slots)[64], unsigned char (*res)[16]) {
  float A[16][32], B[16][32];
  for (int l = 0; l < 64; ++l) {
    const hipemu_mfma_bf16_payload* p = (const hipemu_mfma_bf16_payload*)slots[l];
    for (int k = 0; k < 8; ++k) {
      A[l & 15][(l >> 4) * 8 + k] = (float)p->a[k];
      B[l & 15][(l >> 4) * 8 + k] = (float)p->b[k];
    }
  }
  for (int l = 0; l < 64; ++l) {
    float* out = (float*)res[l];
    const int j = l & 15;
    for (int r = 0; r < 4; ++r) {
      const int i = (l >> 4) * 4 + r;
      float acc = 0.f;
      for (int k = 0; k < 32; ++k) acc += A[i][k] * B[j][k];
      out[r] = acc;
    }
  }
}
static inline hipemu_f32x4 hipemu_mfma_16x16x32_bf16(hipemu_bf16x8 a, hipemu_bf16x8 b, hipemu_f32x4 c, int, int, int) {
  hipemu_mfma_bf16_payload p = {a, b};
  const float* d = (const float*)hipemu::post_once(&p, (int)sizeof(p), hipemu_mfma_16x16x32_bf16_wave)[hipemu::lane()];
  for (int r = 0; r < 4; ++r) c[r] += d[r];
  return c;
}
struct hipemu_mfma_f32_payload { float a, b; };
static void hipemu_mfma_16x16x4f32_wave(const unsigned char (*slots)[64], unsigned char (*res)[16]) {
  for (int l = 0; l < 64; ++l) {
    float* out = (float*)res[l];
    const int j = l & 15;
    for (int r = 0; r < 4; ++r) {
      const int i = (l >> 4) * 4 + r;
      float acc = 0.f;
      for (int kb = 0; kb < 4; ++kb)
        acc += ((const hipemu_mfma_f32_payload*)slots[i + 16 * kb])->a * ((const hipemu_mfma_f32_payload*)slots[j + 16 * kb])->b;
      out[r] = acc;
    }
  }
}
static inline hipemu_f32x4 hipemu_mfma_16x16x4f32(float a, float b, hipemu_f32x4 c, int, int, int) {
  hipemu_mfma_f32_payload p = {a, b};
  const float* d = (const float*)hipemu::post_once(&p, (int)sizeof(p), hipemu_mfma_16x16x4f32_wave)[hipemu::lane()];
  for (int r = 0; r < 4; ++r) c[r] += d[r];
  return c;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 hipemu_mfma_16x16x32_bf16
#define __builtin_amdgcn_mfma_f32_16x16x4f32 hipemu_mfma_16x16x4f32
static inline float hipemu_fdot2_bf16(hipemu_bf16x2 a, hipemu_bf16x2 b, float c, bool) {
  return c + (float)a[0] * (float)b[0] + (float)a[1] * (float)b[1];
}
#define __builtin_amdgcn_fdot2_f32_bf16 hipemu_fdot2_bf16

// ds_read_b64_tr_b16: the 16 lanes of a group each address 4 consecutive 16-bit elements; seen as a [4][16] block
// (lane = 4 * row + column / 4), lane i of the group receives column i: element j comes from lane 4 * j + i / 4,
// position i % 4
template <class PTR>
static inline hipemu_bf16x4 hipemu_ds_read_tr16_b64(PTR p) {
  hipemu_bf16x4 mine;
  memcpy(&mine, (const void*)p, 8);
  auto slots = hipemu::post(&mine, 8);
  const int l = hipemu::lane(), g = l & ~15, i = l & 15;
  hipemu_bf16x4 r;
  for (int j = 0; j < 4; ++j) r[j] = (*(const hipemu_bf16x4*)slots[g + 4 * j + (i >> 2)])[i & 3];
  return r;
}
#define __builtin_amdgcn_ds_read_tr16_b64_v4bf16 hipemu_ds_read_tr16_b64

// global_load_lds, 16 bytes per lane: the LDS base is wave-uniform (M0 = the first lane's pointer), lane L lands at
// base + 16 * L
template <class G, class L>
static inline void hipemu_global_load_lds(G src, L dst, int size, int offset, int aux) {
  if (size != 16) abort();
  const void* d = (const void*)dst;
  auto slots = hipemu::post(&d, 8);
  unsigned char* base = *(unsigned char* const*)slots[__builtin_ctzll(hipemu::alive_mask())];
  memcpy(base + offset + 16 * hipemu::lane(), (const void*)src, 16);
}
#define __builtin_amdgcn_global_load_lds hipemu_global_load_lds
