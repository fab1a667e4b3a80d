// One PNCA decoder block FORWARD as ONE launch (round 5).
//
// reference: kantts/models/sambert/__init__.py:212-348 (MultiHeadPNCAAttention.forward + PNCABlock.forward) with the
// position-wise feed-forward of :134-149 -- per block and training step
//   xn = LN(x);  [q | k | v] = xn W_qkv^T + b;  ox = band_x(q, k, v);  oh = band_h(q, hk, hv)
//   y1 = rowmask(dropout(ox W_fcx^T + oh W_fch^T + b_x + b_h) + x)
//   out = rowmask(dropout(W_2 rowmask(dropout(relu(W_1 LN(y1) + b_1))) + b_2) + y1)   (+ LN of the NEXT sub-layer's input)
// where the band masks of kantts_sambert.py:135-166 confine query i of a sequence to the keys [i - bw, i] of the decoder
// stream and [i, i + bw] of the memory stream (bw = x_band_width = h_band_width, ~5 at the benchmark batch).
//
// Until round 4 this chain was five launches (QKV contraction, both attention bands, the output contraction with the
// next LayerNorm in its epilogue, the feed-forward pair) over M = B * L = 6528 rows of 128 channels: 3.3 MB of block input,
// every launch a 5-20 us kernel whose working set sits in L2, i.e. five dependent launch latencies per block and sixty per
// decoder forward (profiles/r04_runH: exactly one kernel on the chip for 69 % of the step).  The band structure makes the
// whole block TILE-LOCAL: a workgroup that owns 32 consecutive rows needs, beyond its own rows, only
//   * the normalised rows of the PB_HX = 16 rows in front of the tile (their K / V are recomputed: 2/3 of a 16 x 384 x 128
//     contraction, 6 % of the block's work) and
//   * the memory K / V rows of the tile and of the PB_HH = 16 rows behind it (projected once for all twelve blocks by one
//     GEMM, kantts_sambert.py: HybridAttentionDecoder.forward).
// So: one workgroup of 8 waves carries its 32 rows through the whole block; q / k / v, the attention context, the
// sub-layer output y1, its LayerNorm and the 32 x 1024 hidden tile live in LDS / registers; weights are the MFMA *A*
// operand streamed from L2 in the fragment-major images of csrc/ffn_pair.hip (every weight element is used by exactly one
// wave of the workgroup).  What backward needs is still written -- qkv, both contexts, the log-sum-exps, y1 with its
// normalised rows and statistics, the hidden tile -- but nothing is read back, so those stores are off the critical path.
// Band widths above 16 are not handled here (the caller keeps the five-launch chain for them); a device-resident band
// width above 16 poisons the outputs with NaN rather than computing something else silently.
//
// Numerics: the same arithmetic as the chain (bf16 MFMA operands, fp32 accumulation, fp32 attention with the chain's key
// order, the same counter-based dropout indices), so the fused launch and the chain agree to fp32 summation order.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Ablation build only (scripts/build_pbdbg.sh, -DPB_DEBUG; never the product library): KANTTS_PB_DBG is a mask of phases to
// skip -- forward: 1 attention loops, 2 stores of what backward saves (qkv / contexts / y1 / xn1 / hid), 4 feed-forward
// phases; backward: 1 dgamma / dbeta atomics, 2 output-projection input gradient, 4 LayerNorm backward arithmetic, 8 dz store,
// 16 gate load.  Timing only: the results are wrong by construction.
#ifdef PB_DEBUG
#include <stdlib.h>
static int pb_dbg_mask() {
  static const int m = getenv("KANTTS_PB_DBG") ? atoi(getenv("KANTTS_PB_DBG")) : 0;
  return m;
}
#define PB_DBG(bit) ((dbg & (bit)) != 0)
#define PB_DBG_PARAM , const int dbg
#define PB_DBG_ARG , pb_dbg_mask()
#else
#define PB_DBG(bit) false
#define PB_DBG_PARAM
#define PB_DBG_ARG
#endif

#define PB_THREADS 512
#define PB_BM 32
#define PB_HX 16                  // rows in front of the tile whose K / V are recomputed (x band)
#define PB_HH 16                  // memory rows behind the tile (h band)
#define PB_C 128                  // model width = H * 16
#define PB_F 1024
#define PB_XP (PB_C + 16)         // bf16 row pitch of the normalised-row tiles: 288 B = 32 mod 64
#define PB_CP (2 * PB_C + 16)     // bf16 row pitch of the [ox | oh] context tile: 544 B = 32 mod 64
#define PB_TP (PB_F + 16)         // bf16 row pitch of the hidden tile
#define PB_QP (PB_C + 4)          // fp32 row pitch of the Q tile: neighbouring rows 4 banks apart
#define PB_KP (2 * PB_C + 4)      // fp32 row pitch of the K | V tiles
#define PB_DH 16

#define PB_A_BYTES (PB_BM * PB_CP * 2)                                     // 17 408: Xq (48 x 288 B) / Cs / Xs
#define PB_T_BYTES ((PB_BM * PB_QP + (PB_BM + PB_HX) * PB_KP) * 4)         // 66 816: Q + K|V tiles, later the hidden tile
#define PB_H_BYTES ((PB_BM + PB_HH) * PB_KP * 4)                           // 49 920: memory K|V rows

static_assert((PB_BM + PB_HX) * PB_XP * 2 <= PB_A_BYTES, "Xq tile");
static_assert(PB_BM * PB_TP * 2 <= PB_T_BYTES, "hidden tile");

__device__ __forceinline__ unsigned pb_pack2(float a, float b) {
  bf16x4 t = {(__bf16)a, (__bf16)b, (__bf16)0.f, (__bf16)0.f};
  return ((u32x2&)t).x;
}

__device__ __forceinline__ void pb_lds16(const float* p, float* r) {
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
__device__ __forceinline__ float pb_dot16(const float* a, const float* b) {
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < PB_DH; ++d) s = fmaf(a[d], b[d], s);
  return s;
}
__device__ __forceinline__ float pb_dot16l(const float* a, const float* lrow) {
  float b[PB_DH];
  pb_lds16(lrow, b);
  return pb_dot16(a, b);
}

// [round 6] Row tile of this workgroup.  Neighbouring 32-row tiles share their halo rows (the band of both attentions, the
// taps' rows), and consecutive workgroup ids are dealt to the 8 XCDs in turn, each with its own L2: in id order every halo row
// was fetched from HBM / the memory-side cache by two XCDs.  XCD k (= id % 8) owns a contiguous band of tiles instead
// (KANTTS_PNCA_NO_XCD_BAND=1: id order, for A/B runs).  The partial rows of dgamma / dbeta are indexed by the TILE, so the
// order of their sum does not depend on the mapping.
__device__ __forceinline__ int pb_tile(int xcd_band) {
  const int L = blockIdx.x, total = gridDim.x;
  if (!xcd_band || total < 64) return L;
  const int k = L & 7, j = L >> 3, q = total >> 3, r = total & 7;
  return k * q + (k < r ? k : r) + j;
}
static int pb_xcd_band() {
  static const bool off = getenv("KANTTS_PNCA_NO_XCD_BAND") != nullptr;
  return off ? 0 : 1;
}

__global__ __launch_bounds__(PB_THREADS) void pnca_block_fwd_kernel(const kantts_pnca_block_args g, const int xcd_band PB_DBG_PARAM) {
  __shared__ __attribute__((aligned(16))) unsigned char As[PB_A_BYTES];
  __shared__ __attribute__((aligned(16))) unsigned char Tr[PB_T_BYTES];
  __shared__ __attribute__((aligned(16))) float Hs[(PB_BM + PB_HH) * PB_KP];
  __shared__ __attribute__((aligned(16))) float B1s[PB_F];
  __shared__ __attribute__((aligned(16))) float St[512];
  __bf16* Xq = reinterpret_cast<__bf16*>(As);   // (48, PB_XP): normalised rows m0 - 16 .. m0 + 31
  __bf16* Cs = reinterpret_cast<__bf16*>(As);   // (32, PB_CP): [ox | oh] of the tile, bf16
  __bf16* Xs = reinterpret_cast<__bf16*>(As);   // (32, PB_XP): LayerNorm(y1) of the tile
  float* Qs = reinterpret_cast<float*>(Tr);                              // (32, PB_QP)
  float* KVs = reinterpret_cast<float*>(Tr) + PB_BM * PB_QP;             // (48, PB_KP): [k | v] of rows m0 - 16 .. m0 + 31
  __bf16* Ts = reinterpret_cast<__bf16*>(Tr);                            // (32, PB_TP): hidden tile

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kg = lane >> 4;
  const int M = g.B * g.L, L = g.L;
  const int tile_id = pb_tile(xcd_band);
  const int m0 = tile_id * PB_BM;
  const uint64_t seed_off = g.seed_dev ? *g.seed_dev : 0ull;
  const __bf16* __restrict__ wq = reinterpret_cast<const __bf16*>(g.wqkv);
  const __bf16* __restrict__ wfx = reinterpret_cast<const __bf16*>(g.wfcx);
  const __bf16* __restrict__ wfh = reinterpret_cast<const __bf16*>(g.wfch);
  const __bf16* __restrict__ w1 = reinterpret_cast<const __bf16*>(g.w1);
  const __bf16* __restrict__ w2 = reinterpret_cast<const __bf16*>(g.w2);
  const float* dummy = reinterpret_cast<const float*>(g.wqkv);  // any mapped, 16-byte aligned address
  const float4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // ------------------------------------------------------------------------------------------------ global loads, in the
  // order they are consumed (the vector-memory counter retires in order): normalised rows, memory rows, QKV weights
  u32x4 xq[2];
  {
    const __bf16* xn = reinterpret_cast<const __bf16*>(g.xn);
#pragma unroll
    for (int it = 0; it < 2; ++it) {  // 48 rows x 16 chunks of 16 B: 1.5 per thread (clamped source, masked below)
      const int id = min(tid + PB_THREADS * it, (PB_BM + PB_HX) * 16 - 1);
      const long long src = max(0ll, min((long long)m0 - PB_HX + (id >> 4), (long long)M - 1));
      xq[it] = *reinterpret_cast<const u32x4*>(xn + src * PB_C + (id & 15) * 8);
    }
  }
  float4 hreg[6];  // memory K | V rows m0 .. m0 + 47 (clamped to M - 1): 48 x 64 chunks of 16 B
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const int id = tid + PB_THREADS * it;
    const long long row = min((long long)m0 + (id >> 6), (long long)M - 1);
    hreg[it] = *reinterpret_cast<const float4*>(g.hkv + row * g.ldh + (id & 63) * 4);
  }
  float4 b1reg = zero4;  // bias of the feed-forward's first contraction -> LDS (threads 0..255)
  if (tid < PB_F / 4 && g.bias1) b1reg = *reinterpret_cast<const float4*>(g.bias1 + tid * 4);
  u32x4 wqf[3][4];  // wave: output row blocks 3 wave + a (16 channels each) x 4 reduction blocks of 32
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      wqf[a][kk] = *reinterpret_cast<const u32x4*>(wq + ((long long)((wave * 3 + a) * 4 + kk)) * 512 + lane * 8);
  // this lane's slice of the residual stream and of the epilogue vectors: tokens b * 16 + li, channels n0 .. n0 + 3
  const int nq = wave & 3, kh = wave >> 2;
  const int n0 = nq * 32 + kh * 16 + kg * 4;
  float4 xres[2];
  bool rz[2], live[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const long long m = min((long long)m0 + b * 16 + li, (long long)M - 1);
    live[b] = m0 + b * 16 + li < M;
    xres[b] = *reinterpret_cast<const float4*>(g.x + m * PB_C + n0);
    const uint8_t q = *(g.rowmask ? g.rowmask + m : reinterpret_cast<const uint8_t*>(dummy));
    rz[b] = g.rowmask && q != 0;
  }
  float4 bfc = *reinterpret_cast<const float4*>(g.bfcx ? g.bfcx + n0 : dummy);
  if (!g.bfcx) bfc = zero4;
  {
    float4 t = *reinterpret_cast<const float4*>(g.bfch ? g.bfch + n0 : dummy);
    if (!g.bfch) t = zero4;
    bfc.x += t.x; bfc.y += t.y; bfc.z += t.z; bfc.w += t.w;
  }
  // the staged operands go to LDS before the first contraction (their registers are then free for it)
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int id = tid + PB_THREADS * it;
    if (id < (PB_BM + PB_HX) * 16) {
      const long long src = (long long)m0 - PB_HX + (id >> 4);
      const u32x4 v = (src >= 0 && src < M) ? xq[it] : (u32x4){0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(&Xq[(id >> 4) * PB_XP + (id & 15) * 8]) = v;
    }
  }
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const int id = tid + PB_THREADS * it;
    *reinterpret_cast<float4*>(&Hs[(id >> 6) * PB_KP + (id & 63) * 4]) = hreg[it];
  }
  if (tid < PB_F / 4) *reinterpret_cast<float4*>(&B1s[tid * 4]) = b1reg;
  __syncthreads();  // Xq, memory rows complete

  // ------------------------------------------------------------------------------------------------ [q | k | v] of 48 rows
  {
    f32x4 acc[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) acc[a][tb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 bf[3];
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(&Xq[(tb * 16 + li) * PB_XP + kk * 32 + kg * 8]);
        bf[tb] = (bf16x8&)v;
      }
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int tb = 0; tb < 3; ++tb)
          acc[a][tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((const bf16x8&)wqf[a][kk], bf[tb], acc[a][tb], 0, 0, 0);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int c0 = (wave * 3 + a) * 16 + kg * 4;  // 4 consecutive output channels of [q | k | v]
      float4 bq = *reinterpret_cast<const float4*>(g.bqkv ? g.bqkv + c0 : dummy);
      if (!g.bqkv) bq = zero4;
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) {
        const int j = tb * 16 + li;  // tile row: global row m0 - 16 + j
        const f32x4 v = {acc[a][tb][0] + bq.x, acc[a][tb][1] + bq.y, acc[a][tb][2] + bq.z, acc[a][tb][3] + bq.w};
        if (c0 < PB_C) {
          if (tb > 0) *reinterpret_cast<f32x4*>(&Qs[(j - PB_HX) * PB_QP + c0]) = v;
        } else {
          *reinterpret_cast<f32x4*>(&KVs[j * PB_KP + (c0 - PB_C)]) = v;
        }
        const long long m = (long long)m0 - PB_HX + j;
        if (tb > 0 && m < M && g.qkv && !PB_DBG(2)) *reinterpret_cast<f32x4*>(g.qkv + m * (3 * PB_C) + c0) = v;
      }
    }
  }
  // weights of the output contraction (row block nq * 2 + kh; reduction blocks 0..3 over ox = fc_x, 4..7 over oh = fc_h) and
  // the first LayerNorm's vectors: requested now, consumed after the attention phase
  const float4 l1g = *reinterpret_cast<const float4*>(g.ln1_gamma + n0), l1b = *reinterpret_cast<const float4*>(g.ln1_beta + n0);
  u32x4 wff[8];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    wff[kk] = *reinterpret_cast<const u32x4*>(wfx + ((long long)((nq * 2 + kh) * 4 + kk)) * 512 + lane * 8);
    wff[4 + kk] = *reinterpret_cast<const u32x4*>(wfh + ((long long)((nq * 2 + kh) * 4 + kk)) * 512 + lane * 8);
  }
  // the feed-forward weight stream starts now: two units of phase 1 in flight across the attention phase (four would not
  // fit the register file beside the attention's working set), the other two are requested right after it
  //   phase 1, step s (chunk of 256 hidden units): rows s*256 + wave*32 + {0, 16}, 4 k-blocks of 32 each
  //   phase 2, unit u: rows (wave & 3)*32 + {0, 16}, k-blocks (wave >> 2)*16 + u*4 + {0..3}
  u32x4 ring[4][8];
  auto load1 = [&](u32x4* w, int s) {
    const __bf16* p = w1 + ((long long)(s * 16 + wave * 2) * 4) * 512 + lane * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = *reinterpret_cast<const u32x4*>(p + j * 512);
  };
  auto load2 = [&](u32x4* w, int u) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const __bf16* p = w2 + ((long long)(nq * 2 + a) * (PB_F >> 5) + kh * 16 + u * 4) * 512 + lane * 8;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) w[a * 4 + kk] = *reinterpret_cast<const u32x4*>(p + kk * 512);
    }
  };
  load1(ring[0], 0);
  load1(ring[1], 1);
  __syncthreads();  // Q, K | V and the memory rows are in LDS; Xq is dead

  // ------------------------------------------------------------------------------------------------ both attention bands
  // thread <-> (band, head, row): 16 consecutive lanes are 16 rows of one head (row pitch = 4 banks: conflict-free)
  {
    const int band = tid >> 8, head = (tid >> 5) & 7, row = tid & 31;
    const long long m = (long long)m0 + row;
    const bool valid = m < M;
    const int b = valid ? (int)(m / L) : 0, i = valid ? (int)(m - (long long)b * L) : 0;
    const int len = g.lens ? g.lens[b] : L;
    const int bw = g.bw_dev ? *g.bw_dev : (band ? g.bw_h : g.bw_x);
    int lo, hi;
    if (!valid || i >= len) {  // padded query rows are skipped (csrc/attn.hip header): context 0
      lo = 0;
      hi = -1;
    } else if (band == 0) {
      lo = max(0, i - bw);
      hi = i;
    } else {
      lo = i;
      hi = min(min(i + bw, L - 1), len - 1);
    }
    if (PB_DBG(1)) hi = lo - 1;
    // key j of the sequence <-> tile row: x band KVs[row + 16 - (i - j)], memory band Hs[row + (j - i)]
    const float* ktile = (band ? Hs : KVs) + head * PB_DH;
    const int r0 = band ? row - i : row + PB_HX - i;  // tile row of key j is r0 + j
    float q[PB_DH], o[PB_DH];
    pb_lds16(&Qs[row * PB_QP + head * PB_DH], q);
#pragma unroll
    for (int d = 0; d < PB_DH; ++d) o[d] = 0.f;
    float mx = -INFINITY;
    for (int j = lo; j <= hi; ++j) mx = fmaxf(mx, pb_dot16l(q, ktile + (r0 + j) * PB_KP) * 0.25f);
    float l = 0.f;
    const uint64_t rng_row = (((uint64_t)head * g.B + b) * L + i) * (uint64_t)L;
    KanttsDropSeq drop(g.att_p, (band ? g.seed_h : g.seed_x) + seed_off);
    for (int j = lo; j <= hi; ++j) {
      const float e = expf(pb_dot16l(q, ktile + (r0 + j) * PB_KP) * 0.25f - mx);
      l += e;
      const float ed = e * drop.scale(rng_row + j);
      float vv[PB_DH];
      pb_lds16(ktile + (r0 + j) * PB_KP + PB_C, vv);
#pragma unroll
      for (int d = 0; d < PB_DH; ++d) o[d] = fmaf(ed, vv[d], o[d]);
    }
    const float inv = (hi >= lo) ? 1.f / l : 0.f;
#pragma unroll
    for (int d = 0; d < PB_DH; ++d) o[d] *= inv;
    if (bw > PB_HX) {  // a band this launch cannot hold: never a silently different result
#pragma unroll
      for (int d = 0; d < PB_DH; ++d) o[d] = __builtin_nanf("");
    }
    if (valid && !PB_DBG(2)) {
      float* od = (band ? g.oh : g.ox) + m * PB_C + head * PB_DH;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        *reinterpret_cast<float4*>(od + 4 * e) = make_float4(o[4 * e], o[4 * e + 1], o[4 * e + 2], o[4 * e + 3]);
      (band ? g.lse_h : g.lse_x)[((long long)b * (PB_C / PB_DH) + head) * L + i] = (hi >= lo) ? (mx + logf(l)) : 0.f;
    }
    u32x4 p0 = {pb_pack2(o[0], o[1]), pb_pack2(o[2], o[3]), pb_pack2(o[4], o[5]), pb_pack2(o[6], o[7])};
    u32x4 p1 = {pb_pack2(o[8], o[9]), pb_pack2(o[10], o[11]), pb_pack2(o[12], o[13]), pb_pack2(o[14], o[15])};
    *reinterpret_cast<u32x4*>(&Cs[row * PB_CP + band * PB_C + head * PB_DH]) = p0;
    *reinterpret_cast<u32x4*>(&Cs[row * PB_CP + band * PB_C + head * PB_DH + 8]) = p1;
  }
  load1(ring[2], 2);
  load1(ring[3], 3);
  float4 bs2 = *reinterpret_cast<const float4*>(g.bias2 ? g.bias2 + n0 : dummy);
  if (!g.bias2) bs2 = zero4;
  const bool ln2 = g.ln2_out != nullptr;
  const float4 l2g = *reinterpret_cast<const float4*>(ln2 ? g.ln2_gamma + n0 : dummy);
  const float4 l2b = *reinterpret_cast<const float4*>(ln2 ? g.ln2_beta + n0 : dummy);
  __syncthreads();  // context tile complete; Q / K / V / memory rows are dead

  // ------------------------------------------------------------------------------------------------ y1 = fc_x(ox) + fc_h(oh)
  float y1v[2][4];
  {
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(&Cs[(b * 16 + li) * PB_CP + kk * 32 + kg * 8]);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((const bf16x8&)wff[kk], (const bf16x8&)v, acc[b], 0, 0, 0);
      }
    }
    const uint64_t sdf = g.fc_seed + seed_off;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const long long m = (long long)m0 + b * 16 + li;
      float* o = y1v[b];
      o[0] = acc[b][0] + bfc.x; o[1] = acc[b][1] + bfc.y; o[2] = acc[b][2] + bfc.z; o[3] = acc[b][3] + bfc.w;
      if (g.fc_p > 0.f) kantts_dropout_scale4(g.fc_p, sdf, (uint64_t)m * (uint64_t)PB_C + (uint64_t)n0, o);
      o[0] += xres[b].x; o[1] += xres[b].y; o[2] += xres[b].z; o[3] += xres[b].w;
      if (rz[b]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = 0.f;
      }
      if (live[b] && g.y1 && !PB_DBG(2)) *reinterpret_cast<f32x4*>(g.y1 + m * PB_C + n0) = (f32x4){o[0], o[1], o[2], o[3]};
    }
  }
  // LayerNorm(128) of a token: its channels sit in 4 lanes (kg) of each of the 8 waves -> two-pass statistics through LDS
  auto layer_norm_rows = [&](const auto& v, float (&mu)[2], float (&rs)[2], float eps) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      float ps[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = pass ? v[b][r] - mu[b] : v[b][r];
          t += pass ? d * d : d;
        }
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        ps[b] = t;
      }
      if (kg == 0) {
        St[pass * 256 + (wave * 2 + 0) * 16 + li] = ps[0];
        St[pass * 256 + (wave * 2 + 1) * 16 + li] = ps[1];
      }
      __syncthreads();
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += St[pass * 256 + (w * 2 + b) * 16 + li];
        if (pass)
          rs[b] = 1.0f / sqrtf(t * (1.f / 128.f) + eps);
        else
          mu[b] = t * (1.f / 128.f);
      }
    }
  };
  {
    float mu[2], rs[2];
    layer_norm_rows(y1v, mu, rs, g.ln1_eps);  // (its barriers also retire every wave's reads of the context tile)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const long long m = (long long)m0 + b * 16 + li;
      const float* o = y1v[b];
      const float z0 = (o[0] - mu[b]) * rs[b] * l1g.x + l1b.x, z1 = (o[1] - mu[b]) * rs[b] * l1g.y + l1b.y;
      const float z2 = (o[2] - mu[b]) * rs[b] * l1g.z + l1b.z, z3 = (o[3] - mu[b]) * rs[b] * l1g.w + l1b.w;
      const u32x2 pk = {pb_pack2(z0, z1), pb_pack2(z2, z3)};
      *reinterpret_cast<u32x2*>(&Xs[(b * 16 + li) * PB_XP + n0]) = pk;
      if (!live[b]) continue;
      if (g.xn1 && !PB_DBG(2)) *reinterpret_cast<u32x2*>(reinterpret_cast<__bf16*>(g.xn1) + m * PB_C + n0) = pk;
      if (wave == 0 && kg == 0 && g.mean1) {
        g.mean1[m] = mu[b];
        g.rstd1[m] = rs[b];
      }
    }
  }
  __syncthreads();  // normalised rows of the tile complete

  // ------------------------------------------------------------------------------------------------ feed-forward, phase 1
  // (csrc/ffn_pair.hip: T^T[f][tok] = W1[f][:] . X[tok][:], f in chunks of 256; no workgroup barrier inside the phase)
  const int crow = lane >> 2, ccol = (lane & 3) * 8;
  {
    f32x4 acc[2][2];
    const uint64_t sd1 = g.drop1_seed + seed_off;
    auto mfma1 = [&](const u32x4* w, int c) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        bf16x8 bf[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(&Xs[(b * 16 + li) * PB_XP + kk * 32 + kg * 8]);
          bf[b] = (bf16x8&)v;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((const bf16x8&)w[a * 4 + kk], bf[b], acc[a][b], 0, 0, 0);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int f0 = c * 256 + wave * 32 + a * 16 + kg * 4;
        const float4 bs = *reinterpret_cast<const float4*>(&B1s[f0]);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int tok = b * 16 + li;
          const long long m = (long long)m0 + tok;
          float o[4] = {acc[a][b][0] + bs.x, acc[a][b][1] + bs.y, acc[a][b][2] + bs.z, acc[a][b][3] + bs.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = fmaxf(o[r], 0.f);
          if (g.drop1_p > 0.f) kantts_dropout_scale4(g.drop1_p, sd1, (uint64_t)m * (uint64_t)PB_F + (uint64_t)f0, o);
          if (rz[b]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = 0.f;
          }
          const u32x2 pk = {pb_pack2(o[0], o[1]), pb_pack2(o[2], o[3])};
          *reinterpret_cast<u32x2*>(&Ts[tok * PB_TP + f0]) = pk;
        }
      }
      // the wave's own 32 columns of the hidden tile -> HBM (a wave's LDS operations execute in order: no barrier)
      KANTTS_WAVE_ORDERED();
      if (g.hid && !PB_DBG(2)) {
        __bf16* tp = reinterpret_cast<__bf16*>(g.hid);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int i = crow + 16 * it;
          const u32x4 v = *reinterpret_cast<const u32x4*>(&Ts[i * PB_TP + c * 256 + wave * 32 + ccol]);
          if (m0 + i < M) *reinterpret_cast<u32x4*>(tp + ((long long)m0 + i) * PB_F + c * 256 + wave * 32 + ccol) = v;
        }
      }
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // PB_F / 256 = 4 steps; the tail pulls in the four units of phase 2
      if (!PB_DBG(4)) mfma1(ring[j], j);
      load2(ring[j], j);
    }
  }
  __syncthreads();  // hidden tile complete

  // ------------------------------------------------------------------------------------------------ feed-forward, phase 2
  float ov[2][4];
  {
    f32x4 acc2[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc2[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (PB_DBG(4)) continue;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        bf16x8 bf[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(&Ts[(b * 16 + li) * PB_TP + (kh * 16 + u * 4 + kk) * 32 + kg * 8]);
          bf[b] = (bf16x8&)v;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc2[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((const bf16x8&)ring[u][a * 4 + kk], bf[b], acc2[a][b], 0, 0, 0);
      }
    }
    // the two halves of the reduction meet through LDS (the hidden tile is dead): wave (nq, kh) finishes row block a = kh
    __syncthreads();
    float* Ex = reinterpret_cast<float*>(Tr);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const f32x4 snd = kh ? acc2[0][b] : acc2[1][b];
      *reinterpret_cast<f32x4*>(&Ex[(((nq * 2 + (kh ^ 1)) * 2 + b) * 64 + lane) * 4]) = snd;
    }
    __syncthreads();
    const uint64_t sd2 = g.drop2_seed + seed_off;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(&Ex[(((nq * 2 + kh) * 2 + b) * 64 + lane) * 4]);
      const f32x4 s = (kh ? acc2[1][b] : acc2[0][b]) + v;
      const long long m = (long long)m0 + b * 16 + li;
      float* o = ov[b];
      o[0] = s[0] + bs2.x; o[1] = s[1] + bs2.y; o[2] = s[2] + bs2.z; o[3] = s[3] + bs2.w;
      if (g.drop2_p > 0.f) kantts_dropout_scale4(g.drop2_p, sd2, (uint64_t)m * (uint64_t)PB_C + (uint64_t)n0, o);
      o[0] += y1v[b][0]; o[1] += y1v[b][1]; o[2] += y1v[b][2]; o[3] += y1v[b][3];
      if (rz[b]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = 0.f;
      }
      if (live[b]) *reinterpret_cast<f32x4*>(g.out + m * PB_C + n0) = (f32x4){o[0], o[1], o[2], o[3]};
    }
  }
  if (ln2) {  // LayerNorm of the sub-layer that consumes the block's output (the next block's attention, or the stack's)
    float mu[2], rs[2];
    layer_norm_rows(ov, mu, rs, g.ln2_eps);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const long long m = (long long)m0 + b * 16 + li;
      if (!live[b]) continue;
      const float* o = ov[b];
      const float z0 = (o[0] - mu[b]) * rs[b] * l2g.x + l2b.x, z1 = (o[1] - mu[b]) * rs[b] * l2g.y + l2b.y;
      const float z2 = (o[2] - mu[b]) * rs[b] * l2g.z + l2b.z, z3 = (o[3] - mu[b]) * rs[b] * l2g.w + l2b.w;
      if (g.ln2_out_bf16) {
        const u32x2 pk = {pb_pack2(z0, z1), pb_pack2(z2, z3)};
        *reinterpret_cast<u32x2*>(reinterpret_cast<__bf16*>(g.ln2_out) + m * PB_C + n0) = pk;
      } else {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.ln2_out) + m * PB_C + n0) = (f32x4){z0, z1, z2, z3};
      }
      if (wave == 0 && kg == 0) {
        g.ln2_mean[m] = mu[b];
        g.ln2_rstd[m] = rs[b];
      }
    }
  }
}

// ================================================================================================================
// The ROW-LOCAL half of the block's BACKWARD as one launch: from the gradient of the block output down to the gradients of
// the two attention contexts --
//   dz  = gate_{hid > 0}(dropout_2(dy) W_2) / (1 - p_1)          (feed-forward, hidden pre-activation; kept for dW_1)
//   dh  = dz W_1                                                  (gradient of the normalised rows LN1(y1), bf16 like them)
//   g1  = rowmask(LN1'(dh; y1) + dy)                              (gradient of y1: LayerNorm backward + residual branch)
//   [d_ox | d_oh] = dropout_fc(g1) [W_fcx | W_fch]               (input gradient of the output projection)
// -- i.e. kantts_ffn_pair (backward form) + kantts_ln128_bwd_rows + two kantts_bgemm_nt launches of the chain
// (ops_bf16._FusedFFNB / _LayerNorm128 / _FusedLinearB .backward), with the same arithmetic: bf16 MFMA operands, dh rounded to
// bf16 before the LayerNorm backward, fp32 everywhere else.  dgamma / dbeta of LN1 leave as 256 atomics per workgroup.
// What crosses rows -- the attention backward, whose key gradients collect queries from neighbouring tiles -- stays a
// launch of its own (csrc/attn.hip), followed by the QKV input gradient with the first LayerNorm's backward in its
// epilogue (csrc/gemm_bf16.hip): a block's backward is 3 launches instead of 7.
__global__ __launch_bounds__(PB_THREADS) void pnca_block_bwd_kernel(const kantts_pnca_block_bwd_args g, const int xcd_band PB_DBG_PARAM) {
  __shared__ __attribute__((aligned(16))) __bf16 Xs[PB_BM * PB_XP];   // dropout_2(dy) tile, later dropout_fc(g1)
  __shared__ __attribute__((aligned(16))) __bf16 Ts[PB_BM * PB_TP];   // gate tile -> dz tile
  __shared__ __attribute__((aligned(16))) float St[512];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kg = lane >> 4;
  const int M = g.M;
  const int tile_id = pb_tile(xcd_band);
  const int m0 = tile_id * PB_BM;
  const uint64_t seed_off = g.seed_dev ? *g.seed_dev : 0ull;
  const __bf16* __restrict__ wt2 = reinterpret_cast<const __bf16*>(g.wt2);  // W2^T (F x 128)
  const __bf16* __restrict__ wt1 = reinterpret_cast<const __bf16*>(g.wt1);  // W1^T (128 x F)
  const __bf16* __restrict__ wxt = reinterpret_cast<const __bf16*>(g.wfcxT);
  const __bf16* __restrict__ wht = reinterpret_cast<const __bf16*>(g.wfchT);
  const float* dummy = reinterpret_cast<const float*>(g.wt2);
  const int nq = wave & 3, kh = wave >> 2;
  const int n0 = nq * 32 + kh * 16 + kg * 4;

  u32x4 ring[4][8];
  auto load1 = [&](u32x4* w, int s) {  // rows s*256 + wave*32 + {0, 16} of W2^T, 4 k-blocks
    const __bf16* p = wt2 + ((long long)(s * 16 + wave * 2) * 4) * 512 + lane * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = *reinterpret_cast<const u32x4*>(p + j * 512);
  };
  auto load2 = [&](u32x4* w, int u) {  // rows nq*32 + {0, 16} of W1^T, k-blocks kh*16 + u*4 + {0..3}
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const __bf16* p = wt1 + ((long long)(nq * 2 + a) * (PB_F >> 5) + kh * 16 + u * 4) * 512 + lane * 8;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) w[a * 4 + kk] = *reinterpret_cast<const u32x4*>(p + kk * 512);
    }
  };
#pragma unroll
  for (int j = 0; j < 4; ++j) load1(ring[j], j);

  // ---- dy tile -> dropout_2 -> bf16 (masked rows and rows past M are zero)
  {
    const int j = tid >> 4, ch = tid & 15;
    const long long src = (long long)m0 + j;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (src < M && !(g.rowmask && g.rowmask[src])) {
      const float* p = g.dy + src * PB_C + ch * 8;
      float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
      if (g.drop2_p > 0.f) {
        const uint64_t base = (uint64_t)src * (uint64_t)PB_C + (uint64_t)(ch * 8);
        const uint64_t sd = g.drop2_seed + seed_off;
        float lo4[4] = {a.x, a.y, a.z, a.w}, hi4[4] = {b.x, b.y, b.z, b.w};
        kantts_dropout_scale4(g.drop2_p, sd, base, lo4);
        kantts_dropout_scale4(g.drop2_p, sd, base + 4, hi4);
        a = make_float4(lo4[0], lo4[1], lo4[2], lo4[3]);
        b = make_float4(hi4[0], hi4[1], hi4[2], hi4[3]);
      }
      v.x = pb_pack2(a.x, a.y);
      v.y = pb_pack2(a.z, a.w);
      v.z = pb_pack2(b.x, b.y);
      v.w = pb_pack2(b.z, b.w);
    }
    *reinterpret_cast<u32x4*>(&Xs[j * PB_XP + ch * 8]) = v;
  }
  // ---- gate (the saved hidden activation) into the cells the gradient tile will overwrite
  {
    const __bf16* gp = reinterpret_cast<const __bf16*>(g.hid);
    u32x4 gq[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int id = tid + PB_THREADS * it;
      const long long row = PB_DBG(16) ? 0ll : min((long long)m0 + (id >> 7), (long long)M - 1);
      gq[it] = *reinterpret_cast<const u32x4*>(gp + row * PB_F + (id & 127) * 8);
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int id = tid + PB_THREADS * it;
      *reinterpret_cast<u32x4*>(&Ts[(id >> 7) * PB_TP + (id & 127) * 8]) = gq[it];
    }
  }
  // what the LayerNorm backward needs from memory, requested before the weight stream is consumed
  float4 y1r[2], dyr[2];
  float mu[2], rs[2];
  bool rz[2], live[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const long long m = min((long long)m0 + b * 16 + li, (long long)M - 1);
    live[b] = m0 + b * 16 + li < M;
    y1r[b] = *reinterpret_cast<const float4*>(g.y1 + m * PB_C + n0);
    dyr[b] = *reinterpret_cast<const float4*>(g.dy + m * PB_C + n0);
    mu[b] = g.mean1[m];
    rs[b] = g.rstd1[m];
    const uint8_t q = *(g.rowmask ? g.rowmask + m : reinterpret_cast<const uint8_t*>(dummy));
    rz[b] = g.rowmask && q != 0;
  }
  const float4 gm = *reinterpret_cast<const float4*>(g.ln1_gamma + n0);
  __syncthreads();  // dy tile and gate tile complete

  // ---- phase 1: dz^T[f][tok] = W2^T[f][:] . dy_d[tok][:], gated by the saved activation
  const int crow = lane >> 2, ccol = (lane & 3) * 8;
  {
    f32x4 acc[2][2];
    auto mfma1 = [&](const u32x4* w, int c) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        bf16x8 bf[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(&Xs[(b * 16 + li) * PB_XP + kk * 32 + kg * 8]);
          bf[b] = (bf16x8&)v;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((const bf16x8&)w[a * 4 + kk], bf[b], acc[a][b], 0, 0, 0);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int f0 = c * 256 + wave * 32 + a * 16 + kg * 4;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int tok = b * 16 + li;
          const u32x2 q = *reinterpret_cast<const u32x2*>(&Ts[tok * PB_TP + f0]);
          const float gv[4] = {__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16),
                               __uint_as_float(q.y & 0xffff0000u)};
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (gv[r] > 0.f) ? acc[a][b][r] * g.alpha1 : 0.f;
          const u32x2 pk = {pb_pack2(o[0], o[1]), pb_pack2(o[2], o[3])};
          *reinterpret_cast<u32x2*>(&Ts[tok * PB_TP + f0]) = pk;
        }
      }
      KANTTS_WAVE_ORDERED();
      __bf16* tp = reinterpret_cast<__bf16*>(g.dz);
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int i = crow + 16 * it;
        const u32x4 v = *reinterpret_cast<const u32x4*>(&Ts[i * PB_TP + c * 256 + wave * 32 + ccol]);
        if (m0 + i < M && !PB_DBG(8)) *reinterpret_cast<u32x4*>(tp + ((long long)m0 + i) * PB_F + c * 256 + wave * 32 + ccol) = v;
      }
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mfma1(ring[j], j);
      load2(ring[j], j);
    }
  }
  __syncthreads();  // dz tile complete

  // ---- phase 2: dh^T[n][tok] = W1^T[n][:] . dz[tok][:]
  float dh[2][4];
  {
    f32x4 acc2[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc2[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        bf16x8 bf[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(&Ts[(b * 16 + li) * PB_TP + (kh * 16 + u * 4 + kk) * 32 + kg * 8]);
          bf[b] = (bf16x8&)v;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc2[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((const bf16x8&)ring[u][a * 4 + kk], bf[b], acc2[a][b], 0, 0, 0);
      }
    }
    __syncthreads();  // every wave is done with the dz tile: its cells carry the exchange of the two reduction halves
    float* Ex = reinterpret_cast<float*>(Ts);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const f32x4 snd = kh ? acc2[0][b] : acc2[1][b];
      *reinterpret_cast<f32x4*>(&Ex[(((nq * 2 + (kh ^ 1)) * 2 + b) * 64 + lane) * 4]) = snd;
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(&Ex[(((nq * 2 + kh) * 2 + b) * 64 + lane) * 4]);
      const f32x4 s = (kh ? acc2[1][b] : acc2[0][b]) + v;
      // the chain stores dh in the dtype of the normalised rows (bf16) and the LayerNorm backward reads that
      const unsigned p01 = pb_pack2(s[0], s[1]), p23 = pb_pack2(s[2], s[3]);
      dh[b][0] = __uint_as_float(p01 << 16); dh[b][1] = __uint_as_float(p01 & 0xffff0000u);
      dh[b][2] = __uint_as_float(p23 << 16); dh[b][3] = __uint_as_float(p23 & 0xffff0000u);
      if (!live[b]) dh[b][0] = dh[b][1] = dh[b][2] = dh[b][3] = 0.f;
    }
  }
  // weights of the output projection's input gradient: rows 32 wave + {0, 16} of [W_fcx^T ; W_fch^T] (256 x 128)
  u32x4 wcf[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int rb = wave * 2 + a;  // row block of the 256 context channels: 0..7 = fc_x, 8..15 = fc_h
    const __bf16* p = (rb < 8 ? wxt : wht) + ((long long)((rb & 7) * 4)) * 512 + lane * 8;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) wcf[a][kk] = *reinterpret_cast<const u32x4*>(p + kk * 512);
  }

  // ---- LayerNorm backward of LN1 at y1 (csrc/norm.hip: ln128_bwd_kernel) + residual branch + row mask
  float g1v[2][4];
  {
    const float gmv[4] = {gm.x, gm.y, gm.z, gm.w};
    float xh[2][4], gg[2][4], pg[4] = {0.f, 0.f, 0.f, 0.f}, pb[4] = {0.f, 0.f, 0.f, 0.f};
    float ps1[2], ps2[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float xv[4] = {y1r[b].x, y1r[b].y, y1r[b].z, y1r[b].w};
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xh[b][r] = (xv[r] - mu[b]) * rs[b];
        gg[b][r] = dh[b][r] * gmv[r];
        s1 += gg[b][r];
        s2 += gg[b][r] * xh[b][r];
        pg[r] += dh[b][r] * xh[b][r];
        pb[r] += dh[b][r];
      }
      s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
      ps1[b] = s1;
      ps2[b] = s2;
    }
    if (kg == 0) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        St[(wave * 2 + b) * 16 + li] = ps1[b];
        St[256 + (wave * 2 + b) * 16 + li] = ps2[b];
      }
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        s1 += St[(w * 2 + b) * 16 + li];
        s2 += St[256 + (w * 2 + b) * 16 + li];
      }
      s1 *= (1.f / 128.f);
      s2 *= (1.f / 128.f);
      const float dres[4] = {dyr[b].x, dyr[b].y, dyr[b].z, dyr[b].w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float o = rs[b] * (gg[b][r] - s1 - xh[b][r] * s2) + dres[r];
        if (rz[b]) o = 0.f;
        g1v[b][r] = o;
      }
      const long long m = (long long)m0 + b * 16 + li;
      if (live[b]) *reinterpret_cast<f32x4*>(g.g1 + m * PB_C + n0) = (f32x4){g1v[b][0], g1v[b][1], g1v[b][2], g1v[b][3]};
      // dropout_fc(g1) -> bf16 tile (the dy tile is dead: every wave passed the barrier after phase 1)
      float o[4] = {g1v[b][0], g1v[b][1], g1v[b][2], g1v[b][3]};
      if (g.fc_p > 0.f) kantts_dropout_scale4(g.fc_p, g.fc_seed + seed_off, (uint64_t)m * (uint64_t)PB_C + (uint64_t)n0, o);
      if (!live[b]) o[0] = o[1] = o[2] = o[3] = 0.f;
      const u32x2 pk = {pb_pack2(o[0], o[1]), pb_pack2(o[2], o[3])};
      *reinterpret_cast<u32x2*>(&Xs[(b * 16 + li) * PB_XP + n0]) = pk;
    }
    // dgamma / dbeta of LN1: sum over the tile's 32 tokens (16 lanes x 2) -> this workgroup's row of the workspace; the
    // caller sums the rows (kantts_rows_sum_accum, off the critical path: the sums are parameter gradients).  One atomic per
    // channel and workgroup instead -- 204 workgroups on the same 256 addresses -- cost 18 of the launch's 35 us, a ticket
    // counter with agent-scope fences (the last workgroup adds the rows) even more: the release writes back an L2 full of
    // this launch's own output (profiles/r05_runF_pnca_block_ablation.log, r05_runG_*).
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) {
        pg[r] += __shfl_xor(pg[r], off, 64);
        pb[r] += __shfl_xor(pb[r], off, 64);
      }
    }
    if (li == 0 && !PB_DBG(1)) {
      float* part = g.ws + (long long)tile_id * (2 * PB_C);
      *reinterpret_cast<f32x4*>(part + n0) = (f32x4){pg[0], pg[1], pg[2], pg[3]};
      *reinterpret_cast<f32x4*>(part + PB_C + n0) = (f32x4){pb[0], pb[1], pb[2], pb[3]};
    }
  }
  __syncthreads();  // dropout_fc(g1) tile complete

  // ---- [d_ox | d_oh]^T[c][tok] = [W_fcx^T ; W_fch^T][c][:] . g1_d[tok][:]
  {
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (PB_DBG(2)) return;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 bf[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(&Xs[(b * 16 + li) * PB_XP + kk * 32 + kg * 8]);
        bf[b] = (bf16x8&)v;
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((const bf16x8&)wcf[a][kk], bf[b], acc[a][b], 0, 0, 0);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int c0 = (wave * 2 + a) * 16 + kg * 4;  // context channel: [0, 128) = ox, [128, 256) = oh
      float* dst = c0 < PB_C ? g.d_ox + c0 : g.d_oh + (c0 - PB_C);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const long long m = (long long)m0 + b * 16 + li;
        if (m < M) *reinterpret_cast<f32x4*>(dst + m * PB_C) = acc[a][b];
      }
    }
  }
}

// ================================================================================================================
// The CROSS-ROW half of the block's backward as one launch: both attention bands' backward, the input gradient of the QKV
// projection and the backward of the block's first LayerNorm --
//   dq_i  = sum_{j in x band(i)} ds_ij k_j + sum_{j in h band(i)} ds_ij hk_j          ds = p (dp - D) / sqrt(16)
//   dk_j, dv_j  (decoder stream) from the queries i in [j, j + bw];   dhk_j, dhv_j (memory) from the queries i in [j - bw, j]
//   dxn = [dq | dk | dv] W_qkv  (rounded to bf16, as the chain hands it over);   dx = rowmask(LN0'(dxn; x) + g1)
// -- i.e. kantts_pnca_attn_bwd (csrc/attn.hip: three roles over whole sequences staged per (batch, head)) followed by
// kantts_bgemm_nt_lnbwd (csrc/gemm_bf16.hip), same arithmetic.  The band confines everything a 32-row tile needs to the 16
// rows either side of it: k | v of the rows in front and the memory k | v behind for the query gradients; q, dO, D = dO . O and
// the log-sum-exps of the rows behind (decoder band) / in front (memory band) for the key / value gradients, whose
// probabilities are recomputed (as the separate launch does).  Stage 1 (thread = (band, head, row)) forms dq; the inputs of
// stage 2 (thread = (stream, head, key row)) are fetched into registers meanwhile and take over stage 1's LDS; the 32 x 384
// gradient tile then is the B operand of the contraction with W_qkv^T streamed from L2, and its LayerNorm backward epilogue
// is the one of pnca_block_bwd_kernel (dgamma / dbeta as partial rows).
#define PB2_QROWS (PB_BM + PB_HX + PB_HH)   // 64 query rows: 16 in front (memory band), 16 behind (decoder band)
#define PB2_GP (3 * PB_C + 16)              // bf16 pitch of the [dq | dk | dv] tile: 800 B = 32 mod 64

__global__ __launch_bounds__(PB_THREADS) void pnca_attn_qkv_bwd_kernel(const kantts_pnca_attn_bwd_args g, const int xcd_band PB_DBG_PARAM) {
  // R: stage 1 = [k|v rows m0-16 .. m0+31 | memory k|v rows m0 .. m0+47]; stage 2 = [q rows m0-16 .. m0+47 | dOx rows m0 ..
  // m0+47 | dOh rows m0-16 .. m0+31]; then the bf16 gradient tile
  __shared__ __attribute__((aligned(16))) float R[2 * (PB_BM + PB_HX) * PB_KP];
  __shared__ __attribute__((aligned(16))) float DQs[PB_BM * PB_QP];
  __shared__ __attribute__((aligned(16))) float Lx[(PB_BM + PB_HX) * 8], Dx[(PB_BM + PB_HX) * 8];  // rows m0 .. m0+47
  __shared__ __attribute__((aligned(16))) float Lh[(PB_BM + PB_HX) * 8], Dh[(PB_BM + PB_HX) * 8];  // rows m0-16 .. m0+31
  __shared__ __attribute__((aligned(16))) float St[512];
  float* KVs = R;
  float* HKs = R + (PB_BM + PB_HX) * PB_KP;
  float* Qs = R;                                   // (64, PB_QP)
  float* Gx = R + PB2_QROWS * PB_QP;               // (48, PB_QP)
  float* Gh = Gx + (PB_BM + PB_HX) * PB_QP;        // (48, PB_QP)
  __bf16* Tg = reinterpret_cast<__bf16*>(R);       // (32, PB2_GP)
  static_assert((PB2_QROWS + 2 * (PB_BM + PB_HX)) * PB_QP <= 2 * (PB_BM + PB_HX) * PB_KP, "stage 2 fits stage 1's region");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kg = lane >> 4;
  const int L = g.L, H = PB_C / PB_DH;
  const long long M = (long long)g.B * L;
  const int tile_id = pb_tile(xcd_band);
  const int m0 = tile_id * PB_BM;
  const uint64_t seed_off = g.seed_dev ? *g.seed_dev : 0ull;
  const int bw_x = g.bw_dev ? *g.bw_dev : g.bw_x, bw_h = g.bw_dev ? *g.bw_dev : g.bw_h;
  const __bf16* __restrict__ wt = reinterpret_cast<const __bf16*>(g.wqkvT);  // W_qkv^T (128 x 384)
  const float* dummy = reinterpret_cast<const float*>(g.wqkvT);

  // ---- stage-1 operands: k | v of the tile and the 16 rows in front, memory k | v of the tile and the 16 rows behind
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    float4 t4[6];
#pragma unroll
    for (int it = 0; it < 6; ++it) {
      const int id = tid + PB_THREADS * it, j = id >> 6, c = (id & 63) * 4;
      const long long mk = max(0ll, min((long long)m0 - PB_HX + j, M - 1)), mh = min((long long)m0 + j, M - 1);
      t4[it] = half ? *reinterpret_cast<const float4*>(g.hkv + mh * g.ldh + c)
                    : *reinterpret_cast<const float4*>(g.qkv + mk * (3 * PB_C) + PB_C + c);
    }
#pragma unroll
    for (int it = 0; it < 6; ++it) {
      const int id = tid + PB_THREADS * it, j = id >> 6, c = (id & 63) * 4;
      *reinterpret_cast<float4*>(&(half ? HKs : KVs)[j * PB_KP + c]) = t4[it];
    }
  }
  // ---- D = dO . O and the log-sum-exps: decoder band for rows m0 .. m0+47, memory band for rows m0-16 .. m0+31.
  // Four lanes per (row, head), one float4 of dO and of O each.
  for (int task = tid; task < 2 * (PB_BM + PB_HX) * 8 * 4; task += PB_THREADS) {
    const int part = task & 3, head = (task >> 2) & 7, jr = (task >> 5) % (PB_BM + PB_HX), band = task / (32 * (PB_BM + PB_HX));
    const long long m = (long long)m0 + jr - (band ? PB_HX : 0);
    const bool ok = m >= 0 && m < M;
    const long long mc = ok ? m : 0;
    const float4 a = *reinterpret_cast<const float4*>((band ? g.d_oh : g.d_ox) + mc * PB_C + head * PB_DH + part * 4);
    const float4 o = *reinterpret_cast<const float4*>((band ? g.oh : g.ox) + mc * PB_C + head * PB_DH + part * 4);
    float d = a.x * o.x + a.y * o.y + a.z * o.z + a.w * o.w;
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    if (part == 0) {
      const int b = (int)(mc / L), i = (int)(mc - (long long)b * L);
      (band ? Dh : Dx)[jr * 8 + head] = ok ? d : 0.f;
      (band ? Lh : Lx)[jr * 8 + head] = ok ? (band ? g.lse_h : g.lse_x)[((long long)b * H + head) * L + i] : 0.f;
    }
  }
  // ---- stage-2 operands into registers (they take over the stage-1 region after the barrier that ends stage 1)
  // (ext-vector registers, not float4 structs: arrays of the struct type stay memory objects across the long stage-1 code
  // and the compiler parks them in LDS -- 32 KB for 512 threads -- or in scratch)
  f32x4 pq[4], pgx[3], pgh[3];
#pragma unroll
  for (int it = 0; it < 4; ++it) {  // q rows m0-16 .. m0+47: 64 x 32 chunks
    const int id = tid + PB_THREADS * it, j = id >> 5, c = (id & 31) * 4;
    const long long m = max(0ll, min((long long)m0 - PB_HX + j, M - 1));
    pq[it] = *reinterpret_cast<const f32x4*>(g.qkv + m * (3 * PB_C) + c);
  }
#pragma unroll
  for (int it = 0; it < 3; ++it) {  // dOx rows m0 .. m0+47, dOh rows m0-16 .. m0+31: 48 x 32 chunks each
    const int id = tid + PB_THREADS * it, j = id >> 5, c = (id & 31) * 4;
    const long long mx = min((long long)m0 + j, M - 1), mh = max(0ll, min((long long)m0 - PB_HX + j, M - 1));
    pgx[it] = *reinterpret_cast<const f32x4*>(g.d_ox + mx * PB_C + c);
    pgh[it] = *reinterpret_cast<const f32x4*>(g.d_oh + mh * PB_C + c);
  }
  // stage-1 / stage-2 thread coordinates: (band | stream, head, row); 16 consecutive lanes = 16 rows of one head
  const int band = tid >> 8, head = (tid >> 5) & 7, row = tid & 31;
  const long long m = (long long)m0 + row;
  const bool valid = m < M;
  const int b = valid ? (int)(m / L) : 0, i = valid ? (int)(m - (long long)b * L) : 0;
  const int len = g.lens ? g.lens[b] : L;
  const long long mc = valid ? m : 0;
  // own query, output gradient (stage 1) and own key / value rows (stage 2), from memory
  float q[PB_DH], go[PB_DH], kk[PB_DH], vv[PB_DH];
  {
    const float* qp = g.qkv + mc * (3 * PB_C) + head * PB_DH;
    const float* gp = (band ? g.d_oh : g.d_ox) + mc * PB_C + head * PB_DH;
    const float* kp = band ? g.hkv + mc * g.ldh + head * PB_DH : qp + PB_C;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float4 a = *reinterpret_cast<const float4*>(qp + 4 * e), c = *reinterpret_cast<const float4*>(gp + 4 * e);
      const float4 k4 = *reinterpret_cast<const float4*>(kp + 4 * e), v4 = *reinterpret_cast<const float4*>(kp + PB_C + 4 * e);
      q[4 * e] = a.x; q[4 * e + 1] = a.y; q[4 * e + 2] = a.z; q[4 * e + 3] = a.w;
      go[4 * e] = c.x; go[4 * e + 1] = c.y; go[4 * e + 2] = c.z; go[4 * e + 3] = c.w;
      kk[4 * e] = k4.x; kk[4 * e + 1] = k4.y; kk[4 * e + 2] = k4.z; kk[4 * e + 3] = k4.w;
      vv[4 * e] = v4.x; vv[4 * e + 1] = v4.y; vv[4 * e + 2] = v4.z; vv[4 * e + 3] = v4.w;
    }
  }
  __syncthreads();  // stage-1 operands, D and log-sum-exps are in LDS

  // ---------------------------------------------------------------------------------------- stage 1: query gradients
  float dq[PB_DH];
#pragma unroll
  for (int d = 0; d < PB_DH; ++d) dq[d] = 0.f;
  {
    const int bw = band ? bw_h : bw_x;
    int lo = 0, hi = -1;
    if (valid && i < len) {
      if (band == 0) { lo = max(0, i - bw); hi = i; }
      else { lo = i; hi = min(min(i + bw, L - 1), len - 1); }
    }
    if (PB_DBG(1)) hi = lo - 1;
    const float* ktile = (band ? HKs : KVs) + head * PB_DH;
    const int r0 = band ? row - i : row + PB_HX - i;
    const float D = band ? Dh[(row + PB_HX) * 8 + head] : Dx[row * 8 + head];
    const float lse = band ? Lh[(row + PB_HX) * 8 + head] : Lx[row * 8 + head];
    const uint64_t rng_row = (((uint64_t)head * g.B + b) * L + i) * (uint64_t)L;
    KanttsDropSeq drop(g.att_p, (band ? g.seed_h : g.seed_x) + seed_off);
    for (int j = lo; j <= hi; ++j) {
      float kj[PB_DH];
      pb_lds16(ktile + (r0 + j) * PB_KP, kj);
      const float p = expf(pb_dot16(q, kj) * 0.25f - lse);
      const float dp = pb_dot16l(go, ktile + (r0 + j) * PB_KP + PB_C) * drop.scale(rng_row + j);
      const float ds = p * (dp - D) * 0.25f;
#pragma unroll
      for (int d = 0; d < PB_DH; ++d) dq[d] = fmaf(ds, kj[d], dq[d]);
    }
    if (bw > PB_HX) {
#pragma unroll
      for (int d = 0; d < PB_DH; ++d) dq[d] = __builtin_nanf("");
    }
  }
  if (band == 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      *reinterpret_cast<float4*>(&DQs[row * PB_QP + head * PB_DH + 4 * e]) = make_float4(dq[4 * e], dq[4 * e + 1], dq[4 * e + 2], dq[4 * e + 3]);
  }
  __syncthreads();  // stage 1 done with its region; the memory band's query gradients are in DQs
  if (band == 0) {  // decoder band + memory band (the chain's summed form adds them in this order)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float4 o = *reinterpret_cast<const float4*>(&DQs[row * PB_QP + head * PB_DH + 4 * e]);
      *reinterpret_cast<float4*>(&DQs[row * PB_QP + head * PB_DH + 4 * e]) =
          make_float4(dq[4 * e] + o.x, dq[4 * e + 1] + o.y, dq[4 * e + 2] + o.z, dq[4 * e + 3] + o.w);
    }
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int id = tid + PB_THREADS * it, j = id >> 5, c = (id & 31) * 4;
    *reinterpret_cast<f32x4*>(&Qs[j * PB_QP + c]) = pq[it];
  }
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const int id = tid + PB_THREADS * it, j = id >> 5, c = (id & 31) * 4;
    *reinterpret_cast<f32x4*>(&Gx[j * PB_QP + c]) = pgx[it];
    *reinterpret_cast<f32x4*>(&Gh[j * PB_QP + c]) = pgh[it];
  }
  // what the LayerNorm backward epilogue needs (tokens b2 * 16 + li, channels n0 .. n0 + 3)
  const int n0 = wave * 16 + kg * 4;
  float4 xr[2], gr[2];
  float mu[2], rs[2];
  bool rz[2], live[2];
#pragma unroll
  for (int b2 = 0; b2 < 2; ++b2) {
    const long long mm = min((long long)m0 + b2 * 16 + li, M - 1);
    live[b2] = (long long)m0 + b2 * 16 + li < M;
    xr[b2] = *reinterpret_cast<const float4*>(g.x + mm * PB_C + n0);
    gr[b2] = *reinterpret_cast<const float4*>(g.dres ? g.dres + mm * PB_C + n0 : dummy);
    if (!g.dres) gr[b2] = make_float4(0.f, 0.f, 0.f, 0.f);
    mu[b2] = g.mean0[mm];
    rs[b2] = g.rstd0[mm];
    const uint8_t zq = *(g.zero_rows ? g.zero_rows + mm : reinterpret_cast<const uint8_t*>(dummy));
    rz[b2] = g.zero_rows && zq != 0;
  }
  const float4 gm = *reinterpret_cast<const float4*>(g.ln0_gamma + n0);
  u32x4 wf[12];  // W_qkv^T row block `wave` (16 output channels), 12 reduction blocks of 32
#pragma unroll
  for (int kk2 = 0; kk2 < 12; ++kk2) wf[kk2] = *reinterpret_cast<const u32x4*>(wt + ((long long)(wave * 12 + kk2)) * 512 + lane * 8);
  __syncthreads();  // stage-2 operands in LDS; DQs holds the summed query gradients

  // ---------------------------------------------------------------------------------------- stage 2: key / value gradients
  // `band` now names the stream whose key row this thread owns: 0 = decoder k | v (queries behind), 1 = memory (queries in front)
  float dk[PB_DH], dv[PB_DH];
#pragma unroll
  for (int d = 0; d < PB_DH; ++d) dk[d] = dv[d] = 0.f;
  {
    const int bw = band ? bw_h : bw_x;
    int lo = 0, hi = -1;  // query positions i' of this sequence that see key i
    if (valid) {
      if (band == 0) { lo = i; hi = min(min(i + bw, L - 1), len - 1); }
      else if (i <= len - 1) { lo = max(0, i - bw); hi = i; }  // (i <= L - 1 always)
    }
    if (PB_DBG(1)) hi = lo - 1;
    const uint64_t seed = (band ? g.seed_h : g.seed_x) + seed_off;
    const float* Gt = band ? Gh : Gx;
    for (int ip = lo; ip <= hi; ++ip) {
      const int qrow = row + PB_HX + (ip - i);                 // Qs row of query ip
      const int grow = band ? row + PB_HX + (ip - i) : row + (ip - i);
      float qi[PB_DH], gi[PB_DH];
      pb_lds16(&Qs[qrow * PB_QP + head * PB_DH], qi);
      pb_lds16(&Gt[grow * PB_QP + head * PB_DH], gi);
      const float lse = band ? Lh[grow * 8 + head] : Lx[grow * 8 + head];
      const float D = band ? Dh[grow * 8 + head] : Dx[grow * 8 + head];
      const float p = expf(pb_dot16(qi, kk) * 0.25f - lse);
      const float dsc = kantts_dropout_scale(g.att_p, seed, ((((uint64_t)head * g.B + b) * L + ip) * (uint64_t)L) + i);
      const float pd = p * dsc;
      const float dp = pb_dot16(gi, vv) * dsc;
      const float ds = p * (dp - D) * 0.25f;
#pragma unroll
      for (int d = 0; d < PB_DH; ++d) {
        dv[d] = fmaf(pd, gi[d], dv[d]);
        dk[d] = fmaf(ds, qi[d], dk[d]);
      }
    }
    if (bw > PB_HX) {
#pragma unroll
      for (int d = 0; d < PB_DH; ++d) dk[d] = dv[d] = __builtin_nanf("");
    }
  }
  if (band == 1 && valid) {  // memory stream: gradients of this block's memory projection rows
    float* dst = g.dhkv + m * g.lddh + head * PB_DH;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      *reinterpret_cast<float4*>(dst + 4 * e) = make_float4(dk[4 * e], dk[4 * e + 1], dk[4 * e + 2], dk[4 * e + 3]);
      *reinterpret_cast<float4*>(dst + PB_C + 4 * e) = make_float4(dv[4 * e], dv[4 * e + 1], dv[4 * e + 2], dv[4 * e + 3]);
    }
  }
  float dqs[PB_DH];
  pb_lds16(&DQs[row * PB_QP + head * PB_DH], dqs);  // (summed query gradient of (row, head); only the memory-stream threads use it)
  __syncthreads();  // every thread is done with the stage-2 operands: the region becomes the bf16 gradient tile
  {
    // [dq | dk | dv] of the tile: fp32 to memory (the weight gradient of the projection reads it), bf16 to LDS
    auto emit = [&](const float (&v)[PB_DH], int col) {
      if (valid && g.dqkv) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          *reinterpret_cast<float4*>(g.dqkv + m * (3 * PB_C) + col + 4 * e) = make_float4(v[4 * e], v[4 * e + 1], v[4 * e + 2], v[4 * e + 3]);
      }
      const u32x4 p0 = {pb_pack2(v[0], v[1]), pb_pack2(v[2], v[3]), pb_pack2(v[4], v[5]), pb_pack2(v[6], v[7])};
      const u32x4 p1 = {pb_pack2(v[8], v[9]), pb_pack2(v[10], v[11]), pb_pack2(v[12], v[13]), pb_pack2(v[14], v[15])};
      *reinterpret_cast<u32x4*>(&Tg[row * PB2_GP + col]) = valid ? p0 : (u32x4){0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(&Tg[row * PB2_GP + col + 8]) = valid ? p1 : (u32x4){0u, 0u, 0u, 0u};
    };
    if (band == 0) {  // decoder stream threads hold dk, dv of (row, head)
      emit(dk, PB_C + head * PB_DH);
      emit(dv, 2 * PB_C + head * PB_DH);
    } else {          // memory stream threads carry the summed query gradient of (row, head): saved in dqs below
      emit(dqs, head * PB_DH);
    }
  }
  __syncthreads();  // gradient tile complete

  // ---------------------------------------------------------------------------------------- dxn = dqkv . W_qkv, LayerNorm backward
  float dh[2][4];
  {
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kk2 = 0; kk2 < 12; ++kk2) {
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(&Tg[(b2 * 16 + li) * PB2_GP + kk2 * 32 + kg * 8]);
        acc[b2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((const bf16x8&)wf[kk2], (const bf16x8&)v, acc[b2], 0, 0, 0);
      }
    }
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2) {  // the chain hands the normalised rows' gradient over in bf16
      const unsigned p01 = pb_pack2(acc[b2][0], acc[b2][1]), p23 = pb_pack2(acc[b2][2], acc[b2][3]);
      dh[b2][0] = __uint_as_float(p01 << 16); dh[b2][1] = __uint_as_float(p01 & 0xffff0000u);
      dh[b2][2] = __uint_as_float(p23 << 16); dh[b2][3] = __uint_as_float(p23 & 0xffff0000u);
      if (!live[b2]) dh[b2][0] = dh[b2][1] = dh[b2][2] = dh[b2][3] = 0.f;
    }
  }
  {
    const float gmv[4] = {gm.x, gm.y, gm.z, gm.w};
    float xh[2][4], gg[2][4], pg[4] = {0.f, 0.f, 0.f, 0.f}, pb[4] = {0.f, 0.f, 0.f, 0.f};
    float ps1[2], ps2[2];
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2) {
      const float xv[4] = {xr[b2].x, xr[b2].y, xr[b2].z, xr[b2].w};
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xh[b2][r] = (xv[r] - mu[b2]) * rs[b2];
        gg[b2][r] = dh[b2][r] * gmv[r];
        s1 += gg[b2][r];
        s2 += gg[b2][r] * xh[b2][r];
        pg[r] += dh[b2][r] * xh[b2][r];
        pb[r] += dh[b2][r];
      }
      s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
      ps1[b2] = s1;
      ps2[b2] = s2;
    }
    if (kg == 0) {
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) {
        St[(wave * 2 + b2) * 16 + li] = ps1[b2];
        St[256 + (wave * 2 + b2) * 16 + li] = ps2[b2];
      }
    }
    __syncthreads();
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        s1 += St[(w * 2 + b2) * 16 + li];
        s2 += St[256 + (w * 2 + b2) * 16 + li];
      }
      s1 *= (1.f / 128.f);
      s2 *= (1.f / 128.f);
      const float dres[4] = {gr[b2].x, gr[b2].y, gr[b2].z, gr[b2].w};
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o[r] = rs[b2] * (gg[b2][r] - s1 - xh[b2][r] * s2) + dres[r];
        if (rz[b2]) o[r] = 0.f;
      }
      const long long mm = (long long)m0 + b2 * 16 + li;
      if (live[b2]) *reinterpret_cast<f32x4*>(g.dx + mm * PB_C + n0) = (f32x4){o[0], o[1], o[2], o[3]};
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) {
        pg[r] += __shfl_xor(pg[r], off, 64);
        pb[r] += __shfl_xor(pb[r], off, 64);
      }
    }
    if (li == 0) {
      float* part = g.ws + (long long)tile_id * (2 * PB_C);
      *reinterpret_cast<f32x4*>(part + n0) = (f32x4){pg[0], pg[1], pg[2], pg[3]};
      *reinterpret_cast<f32x4*>(part + PB_C + n0) = (f32x4){pb[0], pb[1], pb[2], pb[3]};
    }
  }
}

static bool pb_aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

extern "C" int kantts_pnca_block_fwd(const kantts_pnca_block_args* gp, void* stream) {
  if (!gp) return KANTTS_E_BADARG;
  const kantts_pnca_block_args& g = *gp;
  if (!g.x || !g.xn || !g.hkv || !g.wqkv || !g.wfcx || !g.wfch || !g.w1 || !g.w2 || !g.ln1_gamma || !g.ln1_beta || !g.ox ||
      !g.oh || !g.lse_x || !g.lse_h || !g.out || g.B < 0 || g.L < 0)
    return KANTTS_E_BADARG;
  if (g.H != PB_C / PB_DH || g.C != PB_C || g.F != PB_F) return KANTTS_E_UNSUPPORTED;
  if (g.ldh < 2 * PB_C || (g.ldh & 3)) return KANTTS_E_UNSUPPORTED;
  if (!g.bw_dev && (g.bw_x > PB_HX || g.bw_h > PB_HH || g.bw_x < 0 || g.bw_h < 0)) return KANTTS_E_UNSUPPORTED;
  if (g.ln2_out && (!g.ln2_gamma || !g.ln2_beta || !g.ln2_mean || !g.ln2_rstd)) return KANTTS_E_BADARG;
  if ((g.mean1 == nullptr) != (g.rstd1 == nullptr)) return KANTTS_E_BADARG;
  const void* al[] = {g.x, g.xn, g.hkv, g.wqkv, g.bqkv, g.wfcx, g.wfch, g.bfcx, g.bfch, g.ln1_gamma, g.ln1_beta, g.w1, g.w2,
                      g.bias1, g.bias2, g.ln2_gamma, g.ln2_beta, g.qkv, g.ox, g.oh, g.y1, g.xn1, g.hid, g.out, g.ln2_out};
  for (const void* p : al)
    if (p && !pb_aligned16(p)) return KANTTS_E_UNSUPPORTED;
  const long long M = (long long)g.B * g.L;
  if (M == 0) return KANTTS_OK;
  hipLaunchKernelGGL(pnca_block_fwd_kernel, dim3(kantts_cdiv(M, PB_BM)), dim3(PB_THREADS), 0, (hipStream_t)stream, g, pb_xcd_band() PB_DBG_ARG);
  KANTTS_CHECK_LAUNCH();
}

extern "C" long long kantts_pnca_block_bwd_ws_floats(int M) {
  return (long long)kantts_cdiv(M > 0 ? M : 1, PB_BM) * (2 * PB_C);
}

extern "C" int kantts_pnca_block_bwd(const kantts_pnca_block_bwd_args* gp, void* stream) {
  if (!gp) return KANTTS_E_BADARG;
  const kantts_pnca_block_bwd_args& g = *gp;
  if (!g.dy || !g.hid || !g.y1 || !g.mean1 || !g.rstd1 || !g.ln1_gamma || !g.wt2 || !g.wt1 || !g.wfcxT || !g.wfchT || !g.dz ||
      !g.g1 || !g.d_ox || !g.d_oh || g.M < 0)
    return KANTTS_E_BADARG;
  if (g.C != PB_C || g.F != PB_F) return KANTTS_E_UNSUPPORTED;
  if (!g.ws || g.ws_floats < kantts_pnca_block_bwd_ws_floats(g.M)) return KANTTS_E_WORKSPACE;
  const void* al[] = {g.dy, g.hid, g.y1, g.ln1_gamma, g.wt2, g.wt1, g.wfcxT, g.wfchT, g.dz, g.g1, g.d_ox, g.d_oh, g.ws};
  for (const void* p : al)
    if (!pb_aligned16(p)) return KANTTS_E_UNSUPPORTED;
  if (g.M == 0) return KANTTS_OK;
  hipLaunchKernelGGL(pnca_block_bwd_kernel, dim3(kantts_cdiv(g.M, PB_BM)), dim3(PB_THREADS), 0, (hipStream_t)stream, g, pb_xcd_band() PB_DBG_ARG);
  KANTTS_CHECK_LAUNCH();
}

// dst0[c] += sum_r src[r][c] (c < split), dst1[c - split] += sum_r src[r][c] (c >= split) for up to KANTTS_ROWSUM_MAX
// problems in one launch: the partial rows launches left in their workspaces (pnca_block_bwd_kernel, bgemm_nt_lnb_kernel: one
// row of 128 dgamma + 128 dbeta sums per workgroup) summed in a fixed order.  Workgroup = 32 columns x 8 row lanes of one
// problem; a lane's loads are independent (the first version walked 51 dependent loads per lane: 15 us for 200 KB).
__global__ __launch_bounds__(256) void rows_sum_many_kernel(const kantts_rowsum_args g) {
  __shared__ float red[8][32];
  const int p = blockIdx.y, cl = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl, rows = g.rows[p], cols = g.cols;
  const float* src = g.src[p];
  float t = 0.f;
  if (c < cols) {
#pragma unroll 8
    for (int r = part; r < rows; r += 8) t += src[(long long)r * cols + c];
  }
  red[part][cl] = t;
  __syncthreads();
  if (part == 0 && c < cols) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) a += red[w][cl];
    float* d = c < g.split ? g.dst0[p] + c : g.dst1[p] + (c - g.split);
    *d += a;
  }
}

extern "C" int kantts_rows_sum_many(const kantts_rowsum_args* gp, void* stream) {
  if (!gp) return KANTTS_E_BADARG;
  const kantts_rowsum_args& g = *gp;
  if (g.n < 0 || g.n > KANTTS_ROWSUM_MAX || g.cols < 0 || g.split < 0 || g.split > g.cols) return KANTTS_E_BADARG;
  for (int i = 0; i < g.n; ++i)
    if (!g.src[i] || g.rows[i] < 0 || (g.split > 0 && !g.dst0[i]) || (g.split < g.cols && !g.dst1[i])) return KANTTS_E_BADARG;
  if (g.n == 0 || g.cols == 0) return KANTTS_OK;
  hipLaunchKernelGGL(rows_sum_many_kernel, dim3(kantts_cdiv(g.cols, 32), g.n), dim3(256), 0, (hipStream_t)stream, g);
  KANTTS_CHECK_LAUNCH();
}

extern "C" int kantts_pnca_attn_qkv_bwd(const kantts_pnca_attn_bwd_args* gp, void* stream) {
  if (!gp) return KANTTS_E_BADARG;
  const kantts_pnca_attn_bwd_args& g = *gp;
  if (!g.qkv || !g.hkv || !g.ox || !g.oh || !g.d_ox || !g.d_oh || !g.lse_x || !g.lse_h || !g.wqkvT || !g.x || !g.mean0 ||
      !g.rstd0 || !g.ln0_gamma || !g.dhkv || !g.dx || !g.ws || g.B < 0 || g.L < 0)
    return KANTTS_E_BADARG;
  if (g.H != PB_C / PB_DH || g.C != PB_C) return KANTTS_E_UNSUPPORTED;
  if (g.ldh < 2 * PB_C || (g.ldh & 3) || g.lddh < 2 * PB_C || (g.lddh & 3)) return KANTTS_E_UNSUPPORTED;
  if (!g.bw_dev && (g.bw_x > PB_HX || g.bw_h > PB_HH || g.bw_x < 0 || g.bw_h < 0)) return KANTTS_E_UNSUPPORTED;
  const long long M = (long long)g.B * g.L;
  if (g.ws_floats < kantts_pnca_block_bwd_ws_floats((int)M)) return KANTTS_E_WORKSPACE;
  const void* al[] = {g.qkv, g.hkv, g.ox, g.oh, g.d_ox, g.d_oh, g.wqkvT, g.x, g.dres, g.ln0_gamma, g.dqkv, g.dhkv, g.dx, g.ws};
  for (const void* p : al)
    if (p && !pb_aligned16(p)) return KANTTS_E_UNSUPPORTED;
  if (M == 0) return KANTTS_OK;
  hipLaunchKernelGGL(pnca_attn_qkv_bwd_kernel, dim3(kantts_cdiv(M, PB_BM)), dim3(PB_THREADS), 0, (hipStream_t)stream, g, pb_xcd_band() PB_DBG_ARG);
  KANTTS_CHECK_LAUNCH();
}
