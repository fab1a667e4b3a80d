// Fused framed STFT -> magnitude -> mel filterbank -> dB -> normalise, one pass over the waveform.
//
// Replaces (reference) kantts/utils/audio_torch.py:155-186 (MelSpectrogram.forward: torch.stft +
// power + sqrt + matmul(513x80) + clamp + 20 log10 + symmetric normalise) and :8-31 (stft magnitude).
// HBM-bound by construction: algorithmic traffic is hop*4 B read + n_mels*4 B written per frame
// (1344 B at hop 256 / 80 mels); the windowed frame, the complex spectrum (4 KB/frame if it went
// through HBM as with a library FFT) and the 513-bin magnitude never leave the CU.
//   * one workgroup (256 threads) walks MEL_FB consecutive frames of one utterance;
//   * real FFT of size N as a complex radix-2 Stockham FFT of size N/2 in LDS (ping-pong, twiddles
//     from a host-built fp64->fp32 table), then the even/odd split post-pass;
//   * the mel filterbank is applied in its sparse (triangular support) form straight from LDS;
//   * each thread owns one mel channel and stores MEL_FB consecutive frames (contiguous in the
//     (B, n_mels, frames) output) at once.
#include <atomic>

#include "common.h"

#define MEL_FB 4
#define MEL_THREADS 256

struct MelArgs {
  const float* wav;
  int B, T, n_fft, log2m, hop, frames, pad_mode;  // pad_mode 0: zeros ("constant"), 1: reflect
  const float* window;                            // (n_fft) window already centred/padded to n_fft
  const float2* tw;                               // (n_fft/2) exp(-2 pi i t / n_fft)
  float eps_power;                                // clamp on re^2+im^2 before sqrt
  // mel mode
  const int32_t* mel_start;  // (n_mels) first bin of the support
  const int32_t* mel_len;    // (n_mels)
  const int32_t* mel_off;    // (n_mels) offset into mel_w
  const float* mel_w;        // packed non-zero weights
  int n_mels;
  float eps_mel;
  float* out_mel;  // (B, n_mels, frames) normalised, or NULL
  float* out_mag;  // (B, frames, n_fft/2+1), or NULL
  // dB normalisation: S = 20 log10(max(mel, 1e-5)) - ref_db;  symmetric: clip(2*max_norm*(S-min_db)/(-min_db) - max_norm,
  // +-max_norm) (spectral_normalize_torch);  else clip(max_norm*(S-min_db)/(-min_db), 0, max_norm) (dsp._normalize)
  float ref_db, min_db, max_norm;
  int symmetric;
  int mel_fm;  // out_mel (and dmel of the backward) is (B, frames, n_mels) instead of (B, n_mels, frames)
};

__device__ __forceinline__ float mel_normalise(const MelArgs& a, float mel) {
  const float db = 20.f * log10f(fmaxf(mel, 1e-5f)) - a.ref_db;
  const float u = (db - a.min_db) / (-a.min_db);
  if (a.symmetric) return fminf(fmaxf(2.f * a.max_norm * u - a.max_norm, -a.max_norm), a.max_norm);
  return fminf(fmaxf(a.max_norm * u, 0.f), a.max_norm);
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

__global__ __launch_bounds__(MEL_THREADS) void melspec_kernel(const MelArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int M = a.n_fft >> 1;
  float2* buf0 = reinterpret_cast<float2*>(smem);
  float2* buf1 = buf0 + M;
  float* amp = reinterpret_cast<float*>(buf1 + M);  // M + 1 magnitudes
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int f0 = blockIdx.x * MEL_FB;
  const float* x = a.wav + (long long)b * a.T;
  float melv[MEL_FB];
#pragma unroll
  for (int q = 0; q < MEL_FB; ++q) melv[q] = 0.f;

  for (int q = 0; q < MEL_FB; ++q) {
    const int f = f0 + q;
    if (f >= a.frames) break;  // uniform across the block
    const int start = f * a.hop - M;  // centre padding of n_fft/2
    // ---- windowed frame, packed as z[n] = x[2n] + i x[2n+1]
    for (int n = tid; n < M; n += MEL_THREADS) {
      float v[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        int s = start + 2 * n + e;
        float xv = 0.f;
        if (a.pad_mode == 1) {
          if (s < 0) s = -s;
          if (s >= a.T) s = 2 * (a.T - 1) - s;
          xv = (s >= 0 && s < a.T) ? x[s] : 0.f;
        } else if (s >= 0 && s < a.T) {
          xv = x[s];
        }
        v[e] = xv * a.window[2 * n + e];
      }
      buf0[n] = make_float2(v[0], v[1]);
    }
    __syncthreads();
    // ---- complex Stockham radix-2 FFT of size M
    float2* src = buf0;
    float2* dst = buf1;
    for (int s = 0; s < a.log2m; ++s) {
      const int Ns = 1 << s;
      for (int j = tid; j < (M >> 1); j += MEL_THREADS) {
        const int k = j & (Ns - 1);
        // exp(-2 pi i k / (2 Ns)) = exp(-2 pi i (k * M/(2Ns)) / M) = tw_N[2 * k * M / (2 Ns)]
        const float2 w = a.tw[(k * (M >> (s + 1))) << 1];
        const float2 u = src[j];
        const float2 t = cmul(src[j + (M >> 1)], w);
        const int d = ((j >> s) << (s + 1)) + k;
        dst[d] = make_float2(u.x + t.x, u.y + t.y);
        dst[d + Ns] = make_float2(u.x - t.x, u.y - t.y);
      }
      __syncthreads();
      float2* tmp = src;
      src = dst;
      dst = tmp;
    }
    // ---- split post-pass: X[k] = E + w_N^k O,  E = (Z[k] + conj Z[M-k]) / 2,  O = -i (Z[k] - conj Z[M-k]) / 2
    for (int k = tid; k <= M; k += MEL_THREADS) {
      float re, im;
      if (k == 0 || k == M) {
        const float2 z0 = src[0];
        re = (k == 0) ? (z0.x + z0.y) : (z0.x - z0.y);
        im = 0.f;
      } else {
        const float2 zk = src[k];
        const float2 zc = src[M - k];  // conj applied below
        const float er = 0.5f * (zk.x + zc.x), ei = 0.5f * (zk.y - zc.y);
        const float dr = zk.x - zc.x, di = zk.y + zc.y;      // Z[k] - conj(Z[M-k])
        const float orr = 0.5f * di, oi = -0.5f * dr;        // -i/2 * (dr + i di)
        const float2 w = a.tw[k];
        re = er + (orr * w.x - oi * w.y);
        im = ei + (orr * w.y + oi * w.x);
      }
      const float mag = sqrtf(fmaxf(re * re + im * im, a.eps_power));
      amp[k] = mag;
      if (a.out_mag) a.out_mag[((long long)b * a.frames + f) * (M + 1) + k] = mag;
    }
    __syncthreads();
    // ---- sparse mel filterbank + dB + symmetric normalisation
    if (a.out_mel && tid < a.n_mels) {
      const int st = a.mel_start[tid], ln = a.mel_len[tid];
      const float* w = a.mel_w + a.mel_off[tid];
      float acc = 0.f;
      for (int i = 0; i < ln; ++i) acc = fmaf(amp[st + i], w[i], acc);
      acc = fmaxf(acc, a.eps_mel);
      melv[q] = mel_normalise(a, acc);
    }
    __syncthreads();  // amp / buffers are reused by the next frame
  }
  if (a.out_mel && tid < a.n_mels) {
    if (a.mel_fm) {
#pragma unroll
      for (int q = 0; q < MEL_FB; ++q)
        if (f0 + q < a.frames) a.out_mel[((long long)b * a.frames + f0 + q) * a.n_mels + tid] = melv[q];
    } else {
      float* o = a.out_mel + ((long long)b * a.n_mels + tid) * a.frames + f0;
#pragma unroll
      for (int q = 0; q < MEL_FB; ++q)
        if (f0 + q < a.frames) o[q] = melv[q];
    }
  }
}


// ------------------------------------------------------------------------------------------------
// [round 4] Register-resident form for n_fft = 1024 (M = n_fft / 2 = 512 = 8 x 8 x 8 complex points).
//
// melspec_kernel above is instruction-issue / LDS-latency bound: nine radix-2 passes, every one a round trip of the whole
// frame through LDS plus a twiddle load from global memory and a workgroup barrier (343 us for 67 584 frames of n_fft 1024
// = 2 800 CU cycles per frame against ~200 of arithmetic).  A first attempt to fix that -- one wave per frame, radix-4
// Stockham passes in private LDS buffers -- measured no better (325 us, profiles/r04_runG_melspec_wave_kernel_ab.log): it
// still moved the frame through LDS five times.  Here ONE WAVE owns TWO consecutive frames and they live in its REGISTERS
// (8 complex values per lane and frame):
//   pass 1   lane L holds z[L + 64 a], a < 8: a radix-8 DFT over a in registers (no exchange), twiddle W_M^(L k0);
//   pass 2/3 the 8 independent 64-point DFTs across the lanes as 8 x 8: exchange through a padded (bank-conflict free)
//            wave-private LDS buffer, radix-8 DFT in registers, twiddle W_64^(c k1), exchange, radix-8 DFT;
//   split    natural-order write, every lane reads the 4 pairs (Z[k], Z[M - k]) and emits BOTH |X[k]| and |X[M - k]|
//            (E +- w^k O) into the wave's magnitude rows in LDS;
//   mel      the sparse filterbank cut into chunks of 8 weights (table + weights staged once per workgroup in LDS),
//            lane i reduces chunks i, i + 64, ..., then lane c adds channel c's chunk sums, dB + normalise.
// Three LDS exchanges per frame instead of ten, no workgroup barrier inside the frame loop (a wave's LDS operations
// execute in order), twiddles / window / split twiddles are per-lane constants held in registers across all the frames a
// wave walks (persistent grid).  The two frames are the two halves of every packed fp32 operation (v_pk_add / v_pk_mul /
// v_pk_fma_f32: real and imaginary parts are separate registers, frame A in .x and frame B in .y), so a complex multiply,
// a rotation by -i or a butterfly needs no lane or half swizzle at all, and one 16-byte LDS cell carries one complex value of
// both frames.  A first version with one frame per wave and (re, im) in the two halves spent a third of its instructions on
// half swaps and selects: 168 us against this form's time in profiles/r04_runN_melspec_register_form.log.
// n_fft = 2048 (16 values per lane and frame) was tried in the one-frame form: 298 VGPRs, one wave per SIMD, 1 018 us against
// the radix-2 kernel's 835 -- that size stays on melspec_kernel.
typedef float mf2 __attribute__((ext_vector_type(2)));
typedef float mf4 __attribute__((ext_vector_type(4)));
struct mc {  // one complex value of two frames
  mf2 re, im;
};
__device__ __forceinline__ mc operator+(mc a, mc b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ mc operator-(mc a, mc b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ mc mc_mul(mc a, mf2 w) { return {a.re * w.x - a.im * w.y, a.re * w.y + a.im * w.x}; }
__device__ __forceinline__ mc mc_mul_mi(mc a) { return {a.im, -a.re}; }  // * (-i)
__device__ __forceinline__ void mc_dft4(mc& c0, mc& c1, mc& c2, mc& c3) {
  const mc s0 = c0 + c2, s1 = c0 - c2, s2 = c1 + c3, s3 = mc_mul_mi(c1 - c3);
  c0 = s0 + s2;
  c1 = s1 + s3;
  c2 = s0 - s2;
  c3 = s1 - s3;
}
// in-place forward DFT of 8 values, natural order in and out
__device__ __forceinline__ void mc_dft8(mc* v) {
  const float h = 0.70710678118654752f;
  mc a0 = v[0] + v[4], a1 = v[1] + v[5], a2 = v[2] + v[6], a3 = v[3] + v[7];
  mc b0 = v[0] - v[4], b1 = v[1] - v[5], b2 = v[2] - v[6], b3 = v[3] - v[7];
  b1 = {(b1.re + b1.im) * h, (b1.im - b1.re) * h};   // * W8^1 = (1 - i) / sqrt 2
  b2 = mc_mul_mi(b2);                                // * W8^2 = -i
  b3 = {(b3.im - b3.re) * h, (b3.re + b3.im) * -h};  // * W8^3 = (-1 - i) / sqrt 2
  mc_dft4(a0, a1, a2, a3);
  mc_dft4(b0, b1, b2, b3);
  v[0] = a0; v[2] = a1; v[4] = a2; v[6] = a3;
  v[1] = b0; v[3] = b1; v[5] = b2; v[7] = b3;
}
__device__ __forceinline__ mf4 mc_pack(mc a) { return (mf4){a.re.x, a.re.y, a.im.x, a.im.y}; }
__device__ __forceinline__ mc mc_unpack(mf4 q) { return {(mf2){q.x, q.y}, (mf2){q.z, q.w}}; }
// a wave's own LDS writes are visible to its later reads (in-order LDS pipeline); this only pins the compiler's order
__device__ __forceinline__ void mel_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// dB + normalisation of mel_normalise with the hardware's log2 / a multiplication by the reciprocal range (1-2 ulp each:
// < 1e-6 in normalised units, the parity bound is 1e-4)
__device__ __forceinline__ float mel_normalise_fast(const MelArgs& a, float mel, float inv_range) {
  const float db = 6.0205999132796239f * __log2f(fmaxf(mel, 1e-5f)) - a.ref_db;
  const float u = (db - a.min_db) * inv_range;
  if (a.symmetric) return fminf(fmaxf(2.f * a.max_norm * u - a.max_norm, -a.max_norm), a.max_norm);
  return fminf(fmaxf(a.max_norm * u, 0.f), a.max_norm);
}

#define MELR_WAVES 4    // waves per workgroup, each with its own pair of frames
#define MELR_CH 8       // filterbank weights per chunk
#define MELR_CHMAX 256  // chunks the table holds (else: per-channel loop over global weights, as melspec_kernel)
#define MELR_SLOTS 2    // mel channels per lane (n_mels <= 128; more: melspec_kernel)
#define MELR_M 512
#define MELR_XB (8 * 72 + 8)  // 16-byte cells of the exchange buffer; the magnitudes reuse it: bin k at cell k + (k >> 3) (<= 583)

// one waveform sample of a frame that reaches over an end of the utterance: zero or reflect padding, branch-free
__device__ __forceinline__ float melr_edge_sample(const MelArgs& a, const float* __restrict__ x, int s) {
  if (a.pad_mode == 1) {
    s = s < 0 ? -s : s;
    s = s >= a.T ? 2 * (a.T - 1) - s : s;
  }
  const float xv = x[min(max(s, 0), a.T - 1)];
  return (s >= 0 && s < a.T) ? xv : 0.f;
}
// raw samples 2 (lane + 64 r) + {0, 1} of frame f, r < 8, into half `H` (0: frame A, 1: frame B) of v[r]; zeros for a
// frame beyond the utterance (the window is applied when the pair is processed: the fetch runs one pair ahead)
template <int H>
__device__ __forceinline__ void melr_load(const MelArgs& a, const float* __restrict__ x, int f, int lane, mc (&v)[8]) {
  const int start = f * a.hop - MELR_M;
  const bool interior = f < a.frames && start >= 0 && start + 2 * MELR_M <= a.T;
  if (interior) {
    const float* xs = x + start + 2 * lane;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      v[r].re[H] = xs[128 * r];
      v[r].im[H] = xs[128 * r + 1];
    }
  } else {
    const float live = f < a.frames ? 1.f : 0.f;
    const int s0 = start + 2 * lane;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      v[r].re[H] = melr_edge_sample(a, x, s0 + 128 * r) * live;
      v[r].im[H] = melr_edge_sample(a, x, s0 + 128 * r + 1) * live;
    }
  }
}

__global__ __launch_bounds__(64 * MELR_WAVES, 3) void melspec_reg_kernel(const MelArgs a, int items, int pairs_per_row) {
  constexpr int M = MELR_M;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // workgroup-shared: filterbank weights, chunk table, channel -> first chunk;  then per wave: exchange buffer, magnitudes
  float* lds_w = smem;
  int* tab = reinterpret_cast<int*>(lds_w + MELR_CHMAX * MELR_CH);
  int* cfirst = tab + MELR_CHMAX;  // n_mels + 1 entries (<= 129)
  mf2* lds_win = reinterpret_cast<mf2*>(cfirst + 132);  // window[2 n], window[2 n + 1]
  mf2* lds_tws = lds_win + MELR_M;                      // W_N^k, k < M / 2 (split pass)
  float* wave_base = reinterpret_cast<float*>(lds_tws + MELR_M / 2);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  mf4* xbuf = reinterpret_cast<mf4*>(wave_base + (size_t)wv * (4 * MELR_XB));
  mf2* amp = reinterpret_cast<mf2*>(xbuf);       // magnitude of bin k, both frames: low half of cell k + (k >> 3), see the split
  mf2* part = reinterpret_cast<mf2*>(xbuf) + 1;  // chunk sum i: high half of cell i (part[2 i])

  // ---- once per workgroup: the sparse filterbank cut into chunks over ALIGNED blocks of MELR_CH bins, zero padded:
  //      chunk i = weights of bins 8 tab[i] .. 8 tab[i] + 7 (slots 0-3 of all chunks, then slots 4-7: 16-byte reads at a
  //      16-byte lane stride), channel c = chunks cfirst[c] .. cfirst[c + 1] - 1
  bool mel_fast = false;
  int nch_total = 0;
  for (int n = tid; n < MELR_M; n += 64 * MELR_WAVES) lds_win[n] = (mf2){a.window[2 * n], a.window[2 * n + 1]};
  for (int k = tid; k < MELR_M / 2; k += 64 * MELR_WAVES) lds_tws[k] = (mf2){a.tw[k].x, a.tw[k].y};
  if (a.out_mel) {
    if (tid == 0) {
      int first = 0;
      for (int c = 0; c < a.n_mels; ++c) {
        cfirst[c] = first;
        const int st = a.mel_start[c], ln = a.mel_len[c];
        if (ln > 0) first += ((st + ln - 1) >> 3) - (st >> 3) + 1;
      }
      cfirst[a.n_mels] = first;
    }
    __syncthreads();
    nch_total = cfirst[a.n_mels];
    mel_fast = nch_total <= MELR_CHMAX;
    if (mel_fast) {  // uniform
      for (int i = tid; i < MELR_CHMAX * MELR_CH; i += 64 * MELR_WAVES) lds_w[i] = 0.f;
      for (int i = tid; i < MELR_CHMAX; i += 64 * MELR_WAVES) tab[i] = 0;
      __syncthreads();
      for (int c = tid; c < a.n_mels; c += 64 * MELR_WAVES) {
        const int st = a.mel_start[c], ln = a.mel_len[c], fi = cfirst[c], blk0 = st >> 3;
        const float* w = a.mel_w + a.mel_off[c];
        for (int i = 0; i < ln; ++i) {
          const int bin = st + i, ci = fi + (bin >> 3) - blk0, t = bin & 7;
          lds_w[(t >> 2) * (4 * MELR_CHMAX) + 4 * ci + (t & 3)] = w[i];
        }
        if (ln > 0)
          for (int q = blk0; q <= (st + ln - 1) >> 3; ++q) tab[fi + q - blk0] = q;
      }
    }
  }
  __syncthreads();

  // ---- per-lane constants
  const int c8 = lane & 7, hi8 = lane >> 3;
  mf2 tw1[8];  // W_M^(lane k0)                               (after pass 1)
  mf2 tw2[8];  // W_64^(c k1) = W_M^(8 c k1), c = lane & 7    (after pass 2)
  auto twM = [&](int x) -> mf2 {  // W_M^x = W_N^(2x), the table holds W_N^t for t < M
    int t = 2 * x;
    const bool neg = t >= M;
    if (neg) t -= M;
    const float2 w = a.tw[t];
    return neg ? (mf2){-w.x, -w.y} : (mf2){w.x, w.y};
  };
#pragma unroll
  for (int k0 = 0; k0 < 8; ++k0) tw1[k0] = twM(lane * k0);
#pragma unroll
  for (int k1 = 0; k1 < 8; ++k1) tw2[k1] = twM(8 * c8 * k1);
  const float inv_range = 1.f / (-a.min_db);
  const int nit = (nch_total + 63) >> 6;

  const int wave_global = blockIdx.x * MELR_WAVES + wv, wave_count = gridDim.x * MELR_WAVES;
  mc v[8];
  if (wave_global < items) {
    const int b = wave_global / pairs_per_row, fa = (wave_global - b * pairs_per_row) * 2;
    melr_load<0>(a, a.wav + (long long)b * a.T, fa, lane, v);
    melr_load<1>(a, a.wav + (long long)b * a.T, fa + 1, lane, v);
  }
#pragma unroll 1
  for (int item = wave_global; item < items; item += wave_count) {
    const int b = item / pairs_per_row, fa = (item - b * pairs_per_row) * 2;  // frames fa, fa + 1 (the second may not exist)
    // the (cos, sin) pairs stay ONE register pair each: opaque per iteration, so the compiler cannot hoist a (cos, cos) /
    // (sin, sin) splat of every constant out of the loop (52 more VGPRs: 204 against the budget of 168 for three waves
    // per SIMD); inside the loop the broadcast is an operand modifier of v_pk_mul / v_pk_fma (op_sel)
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      KANTTS_OPAQUE_VGPR(tw1[k]);
      KANTTS_OPAQUE_VGPR(tw2[k]);
    }
    // the samples of this pair were fetched (raw) while the previous pair's filterbank ran; window them now
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const mf2 w = lds_win[lane + 64 * r];
      v[r].re *= w.x;
      v[r].im *= w.y;
    }
    // ---- pass 1: DFT over a (registers), twiddle, exchange 1: cell (k0, L) at 72 k0 + L
    mc_dft8(v);
#pragma unroll
    for (int k0 = 0; k0 < 8; ++k0) xbuf[72 * k0 + lane] = mc_pack((k0 == 0) ? v[0] : mc_mul(v[k0], tw1[k0]));
    mel_wave_sync();
    // ---- pass 2: lane (k0, c) = 8 k0 + c takes b = 0..7; DFT over b, twiddle W_64^(c k1), exchange 2: cell (k0, k1, c)
    //      at 65 c + k0 + 8 k1 (ds_write_b128 is serviced in groups of 8 contiguous lanes over 32 banks: 65 c mod 8 = c)
#pragma unroll
    for (int bb = 0; bb < 8; ++bb) v[bb] = mc_unpack(xbuf[72 * hi8 + 8 * bb + c8]);
    mel_wave_sync();
    mc_dft8(v);
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) xbuf[65 * c8 + hi8 + 8 * k1] = mc_pack((k1 == 0) ? v[0] : mc_mul(v[k1], tw2[k1]));
    mel_wave_sync();
    // ---- pass 3: lane l = k0 + 8 k1 takes c = 0..7; DFT over c -> k2; X[k], k = l + 64 k2, to cell k
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) v[cc] = mc_unpack(xbuf[65 * cc + lane]);
    mel_wave_sync();
    mc_dft8(v);
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) xbuf[lane + 64 * k2] = mc_pack(v[k2]);
    mel_wave_sync();
    // ---- split: pairs (k, M - k), k = lane + 64 j < M / 2.  The magnitudes of both frames go IN PLACE into the low half
    //      of cells k and M - k (each of these cells is read by this lane only; |X[M]| takes the spare cell M), so a wave
    //      needs no second LDS array: 9 KB per wave, three workgroups per CU
    const bool has_b = fa + 1 < a.frames;
    float* omag = a.out_mag ? a.out_mag + ((long long)b * a.frames + fa) * (M + 1) : nullptr;
    {
      mf4 zkq[4], zmq[4];
      mf2 tws[4];  // W_N^k, k = lane + 64 j
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        zkq[j] = xbuf[lane + 64 * j];
        zmq[j] = xbuf[(M - lane - 64 * j) & (M - 1)];
        tws[j] = lds_tws[lane + 64 * j];
      }
      const mf4 zhq = xbuf[M / 2];
      mel_wave_sync();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = lane + 64 * j;
        const mc zk = mc_unpack(zkq[j]), zm = mc_unpack(zmq[j]);
        const mc e = {(zk.re + zm.re) * 0.5f, (zk.im - zm.im) * 0.5f};
        const mc o = {(zk.im + zm.im) * 0.5f, (zm.re - zk.re) * 0.5f};  // -i/2 (Z[k] - conj Z[M-k])
        const mc t = mc_mul(o, tws[j]);
        const mc x1 = e + t, x2 = e - t;
        const mf2 p1 = x1.re * x1.re + x1.im * x1.im, p2 = x2.re * x2.re + x2.im * x2.im;
        const mf2 m1 = {__builtin_amdgcn_sqrtf(fmaxf(p1.x, a.eps_power)), __builtin_amdgcn_sqrtf(fmaxf(p1.y, a.eps_power))};
        const mf2 m2 = {__builtin_amdgcn_sqrtf(fmaxf(p2.x, a.eps_power)), __builtin_amdgcn_sqrtf(fmaxf(p2.y, a.eps_power))};
        amp[2 * (k + (k >> 3))] = m1;
        amp[2 * (M - k + ((M - k) >> 3))] = m2;
        if (omag) {
          omag[k] = m1.x;
          omag[M - k] = m2.x;
          if (has_b) {
            omag[M + 1 + k] = m1.y;
            omag[M + 1 + M - k] = m2.y;
          }
        }
      }
      if (lane == 0) {
        const mc z = mc_unpack(zhq);
        const mf2 p = z.re * z.re + z.im * z.im;
        const mf2 mh = {__builtin_amdgcn_sqrtf(fmaxf(p.x, a.eps_power)), __builtin_amdgcn_sqrtf(fmaxf(p.y, a.eps_power))};
        amp[2 * (M / 2 + M / 16)] = mh;
        if (omag) {
          omag[M / 2] = mh.x;
          if (has_b) omag[M + 1 + M / 2] = mh.y;
        }
      }
    }
    // this lane's chunks (lane + 64 it) and channels (lane + 64 sl): first bin of each chunk, chunk range of each channel
    // (from LDS, not held in registers across pairs: the register budget of three waves per SIMD is spent on the twiddles)
    int ch_bin[MELR_CHMAX / 64], ch_first[MELR_SLOTS], ch_end[MELR_SLOTS];
    int ln = lane;
    KANTTS_OPAQUE_VGPR(ln);  // addresses derived from the lane are recomputed here, not hoisted out of the loop into registers
    if (mel_fast) {
#pragma unroll
      for (int it = 0; it < MELR_CHMAX / 64; ++it) ch_bin[it] = tab[ln + 64 * it];
#pragma unroll
      for (int sl = 0; sl < MELR_SLOTS; ++sl) {
        const int c = min(ln + 64 * sl, a.n_mels - 1);
        ch_first[sl] = cfirst[c];
        ch_end[sl] = cfirst[c + 1];
      }
    }
    // the next pair's samples: issued here, first touched at the top of the next iteration (the filterbank below hides
    // the memory latency; v is dead from the split pass on)
    {
      const int nitem = item + wave_count;
      if (nitem < items) {
        const int nb = nitem / pairs_per_row, nfa = (nitem - nb * pairs_per_row) * 2;
        melr_load<0>(a, a.wav + (long long)nb * a.T, nfa, lane, v);
        melr_load<1>(a, a.wav + (long long)nb * a.T, nfa + 1, lane, v);
      }
    }
    if (lane < MELR_CH - 1) amp[2 * (M + M / 8 + 1 + lane)] = (mf2){0.f, 0.f};  // bins M + 1 .. M + 7 of the last block: weight 0, must be finite
    mel_wave_sync();
    // ---- sparse mel filterbank + dB + normalisation
    if (a.out_mel) {
      mf2 melv[MELR_SLOTS];
      if (mel_fast) {
#pragma unroll
        for (int it = 0; it < MELR_CHMAX / 64; ++it) {
          if (it < nit) {  // uniform
            const int i = ln + 64 * it;
            const mf4 w0 = reinterpret_cast<const mf4*>(lds_w)[i], w1 = reinterpret_cast<const mf4*>(lds_w)[MELR_CHMAX + i];
            const mf2* ap = amp + 2 * 9 * ch_bin[it];  // block q: cells 9 q .. 9 q + 7 (lanes on neighbouring blocks: 144 B apart)
            mf2 acc = ap[0] * w0.x;
            acc += ap[2] * w0.y;
            acc += ap[4] * w0.z;
            acc += ap[6] * w0.w;
            acc += ap[8] * w1.x;
            acc += ap[10] * w1.y;
            acc += ap[12] * w1.z;
            acc += ap[14] * w1.w;
            part[2 * i] = acc;
          }
        }
        mel_wave_sync();
#pragma unroll
        for (int sl = 0; sl < MELR_SLOTS; ++sl) {
          mf2 acc = {0.f, 0.f};
          const mf2* pp = part + 2 * ch_first[sl];
          const int cnt = (ln + 64 * sl < a.n_mels) ? ch_end[sl] - ch_first[sl] : 0;
#pragma unroll
          for (int i = 0; i < 4; ++i) {  // a lane's first four reads are always inside the wave's cells
            const mf2 p = pp[2 * i];
            acc += (i < cnt) ? p : (mf2){0.f, 0.f};
          }
          for (int i = 4; i < cnt; ++i) acc += pp[2 * i];
          melv[sl] = acc;
        }
        mel_wave_sync();  // the cells are rewritten by the next pair of frames
      } else {
#pragma unroll
        for (int sl = 0; sl < MELR_SLOTS; ++sl) {
          const int c = lane + 64 * sl;
          mf2 acc = {0.f, 0.f};
          if (c < a.n_mels) {
            const int st = a.mel_start[c], ln = a.mel_len[c];
            const float* w = a.mel_w + a.mel_off[c];
            for (int i = 0; i < ln; ++i) acc += amp[2 * (st + i + ((st + i) >> 3))] * w[i];
          }
          melv[sl] = acc;
        }
        mel_wave_sync();
      }
#pragma unroll
      for (int sl = 0; sl < MELR_SLOTS; ++sl) {
        const int c = lane + 64 * sl;
        if (c < a.n_mels) {
          // frame-major output: a wave's 80 channels of a frame are 320 contiguous bytes; channel-major: 8-byte stores
          // into n_mels rows (partial 32-byte sectors: 3 x the algorithmic write traffic, profiles/r04_runFINAL_melspec_pmc)
          float* o = a.mel_fm ? a.out_mel + ((long long)b * a.frames + fa) * a.n_mels + c
                              : a.out_mel + ((long long)b * a.n_mels + c) * a.frames + fa;
          const long long step = a.mel_fm ? a.n_mels : 1;
          o[0] = mel_normalise_fast(a, fmaxf(melv[sl].x, a.eps_mel), inv_range);
          if (has_b) o[step] = mel_normalise_fast(a, fmaxf(melv[sl].y, a.eps_mel), inv_range);
        }
      }
    }
  }
}

// The launchers consult no environment variable (the ABI is stateless apart from these two explicit knobs, which exist for
// sweeps and for the tests that must reach the persistent-grid loop and the radix-2 kernel): kantts_melspec_tuning sets them.
static std::atomic<int> mel_grid_cap{0}, mel_generic_only{0};

extern "C" int kantts_melspec_tuning(int grid_cap, int generic_only) {
  if (grid_cap < 0) return KANTTS_E_BADARG;
  mel_grid_cap.store(grid_cap, std::memory_order_relaxed);
  mel_generic_only.store(generic_only ? 1 : 0, std::memory_order_relaxed);
  return KANTTS_OK;
}

static int melspec_reg_launch(const MelArgs& a, hipStream_t st) {
  const int pairs = kantts_cdiv(a.frames, 2);
  const long long items_ll = (long long)a.B * pairs;
  if (items_ll > 0x7fffffffLL) return KANTTS_E_UNSUPPORTED;
  const int items = (int)items_ll;
  const size_t lds = ((size_t)MELR_CHMAX * MELR_CH + MELR_CHMAX + 132 + 3 * MELR_M + (size_t)MELR_WAVES * (4 * MELR_XB)) * sizeof(float);
  // persistent grid: every workgroup resident (three per CU: 168 VGPRs, 52.8 KB of LDS), waves stride over the pairs of frames
  const int cap_set = mel_grid_cap.load(std::memory_order_relaxed);  // kantts_melspec_tuning (sweeps / tests); 0 = default
  const int cap = cap_set > 0 ? cap_set : 768;
  int grid = kantts_cdiv(items, MELR_WAVES);
  if (grid > cap) grid = cap;
  hipLaunchKernelGGL(melspec_reg_kernel, dim3(grid), dim3(64 * MELR_WAVES), lds, st, a, items, pairs);
  KANTTS_CHECK_LAUNCH();
}

extern "C" int kantts_melspec_norm_fwd(const float* wav, int B, int T, int n_fft, int hop, int frames, int pad_mode,
                                       const float* window, const float* twiddle, float eps_power,
                                       const int32_t* mel_start, const int32_t* mel_len, const int32_t* mel_off,
                                       const float* mel_w, int n_mels, float eps_mel, float ref_level_db,
                                       float min_level_db, float max_norm, int symmetric, float* out_mel, float* out_mag,
                                       void* stream);
extern "C" int kantts_melspec_norm_fwd_fm(const float* wav, int B, int T, int n_fft, int hop, int frames, int pad_mode,
                                          const float* window, const float* twiddle, float eps_power,
                                          const int32_t* mel_start, const int32_t* mel_len, const int32_t* mel_off,
                                          const float* mel_w, int n_mels, float eps_mel, float ref_level_db,
                                          float min_level_db, float max_norm, int symmetric, int mel_frame_major,
                                          float* out_mel, float* out_mag, void* stream);

extern "C" int kantts_melspec_fwd(const float* wav, int B, int T, int n_fft, int hop, int frames, int pad_mode,
                                  const float* window, const float* twiddle, float eps_power,
                                  const int32_t* mel_start, const int32_t* mel_len, const int32_t* mel_off,
                                  const float* mel_w, int n_mels, float eps_mel, float* out_mel, float* out_mag,
                                  void* stream) {
  // MelSpectrogram.forward's fixed normalisation: ref 20 dB, floor -100 dB, symmetric +-4
  return kantts_melspec_norm_fwd(wav, B, T, n_fft, hop, frames, pad_mode, window, twiddle, eps_power, mel_start, mel_len,
                                 mel_off, mel_w, n_mels, eps_mel, 20.f, -100.f, 4.f, 1, out_mel, out_mag, stream);
}

extern "C" int kantts_melspec_norm_fwd(const float* wav, int B, int T, int n_fft, int hop, int frames, int pad_mode,
                                       const float* window, const float* twiddle, float eps_power,
                                       const int32_t* mel_start, const int32_t* mel_len, const int32_t* mel_off,
                                       const float* mel_w, int n_mels, float eps_mel, float ref_level_db,
                                       float min_level_db, float max_norm, int symmetric, float* out_mel, float* out_mag,
                                       void* stream) {
  return kantts_melspec_norm_fwd_fm(wav, B, T, n_fft, hop, frames, pad_mode, window, twiddle, eps_power, mel_start, mel_len,
                                    mel_off, mel_w, n_mels, eps_mel, ref_level_db, min_level_db, max_norm, symmetric, 0,
                                    out_mel, out_mag, stream);
}

extern "C" int kantts_melspec_norm_fwd_fm(const float* wav, int B, int T, int n_fft, int hop, int frames, int pad_mode,
                                          const float* window, const float* twiddle, float eps_power,
                                          const int32_t* mel_start, const int32_t* mel_len, const int32_t* mel_off,
                                          const float* mel_w, int n_mels, float eps_mel, float ref_level_db,
                                          float min_level_db, float max_norm, int symmetric, int mel_frame_major,
                                          float* out_mel, float* out_mag, void* stream) {
  if (!(min_level_db < 0.f) || !(max_norm > 0.f)) return KANTTS_E_BADARG;
  if (!wav || !window || !twiddle || B < 0 || T < 1 || n_fft < 8 || hop < 1 || frames < 0) return KANTTS_E_BADARG;
  if (n_fft & (n_fft - 1)) return KANTTS_E_UNSUPPORTED;
  if (!out_mel && !out_mag) return KANTTS_E_BADARG;
  if (out_mel && (!mel_start || !mel_len || !mel_off || !mel_w || n_mels < 1 || n_mels > MEL_THREADS))
    return KANTTS_E_BADARG;
  if (pad_mode == 1 && T <= n_fft / 2) return KANTTS_E_BADARG;  // reflect needs pad < T (torch.stft rule)
  if (B == 0 || frames == 0) return KANTTS_OK;
  MelArgs a = {};
  a.wav = wav; a.B = B; a.T = T; a.n_fft = n_fft; a.hop = hop; a.frames = frames; a.pad_mode = pad_mode;
  a.window = window; a.tw = reinterpret_cast<const float2*>(twiddle); a.eps_power = eps_power;
  a.mel_start = mel_start; a.mel_len = mel_len; a.mel_off = mel_off; a.mel_w = mel_w; a.n_mels = n_mels;
  a.eps_mel = eps_mel; a.out_mel = out_mel; a.out_mag = out_mag;
  a.ref_db = ref_level_db; a.min_db = min_level_db; a.max_norm = max_norm; a.symmetric = symmetric;
  a.mel_fm = mel_frame_major ? 1 : 0;
  int m = n_fft >> 1, l2 = 0;
  while ((1 << l2) < m) ++l2;
  a.log2m = l2;
  const bool generic_only = mel_generic_only.load(std::memory_order_relaxed) != 0;  // kantts_melspec_tuning (A/B, tests)
  if (!generic_only && n_fft == 2 * MELR_M && (!out_mel || n_mels <= 64 * MELR_SLOTS)) return melspec_reg_launch(a, (hipStream_t)stream);
  size_t lds = (size_t)m * 2 * sizeof(float2) + (size_t)(m + 1) * sizeof(float);
  if (lds > 160 * 1024) return KANTTS_E_UNSUPPORTED;
  hipLaunchKernelGGL(melspec_kernel, dim3(kantts_cdiv(frames, MEL_FB), B), dim3(MEL_THREADS), lds, (hipStream_t)stream, a);
  KANTTS_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// Backward of the mel path (needed by MelSpectrogramLoss on generated audio, kantts/train/loss.py:259-311):
// d wav += window * Re( sum_{k<=N/2} G[k] e^{+2 pi i k n / N} ),  G[k] = d amp[k] * X[k] / amp[k]  (0 where the
// power clamp is active), d amp = melmat (sparse) * d mel, d mel from the dB / clamp chain.  One workgroup per
// frame: the forward spectrum is recomputed in LDS (cheaper than saving 4 KB/frame), the adjoint transform is
// one N-point complex Stockham FFT (of conj G, conjugated back), overlap-add into d wav with atomics.
struct MelBwdArgs {
  MelArgs f;
  const float* dmel;  // (B, n_mels, frames)
  const float* dmag;  // alternative input: (B, frames, n_fft/2+1) gradient of the STFT magnitude itself (then dmel = NULL)
  float* dwav;        // (B, T) accumulated
};

__global__ __launch_bounds__(MEL_THREADS) void melspec_bwd_kernel(const MelBwdArgs ba) {
  const MelArgs& a = ba.f;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = a.n_fft, M = N >> 1;
  float2* buf0 = reinterpret_cast<float2*>(smem);       // N complex
  float2* buf1 = buf0 + N;                              // N complex
  float2* X = buf1 + N;                                 // M + 1 spectrum bins
  float* damp = reinterpret_cast<float*>(X + M + 1);    // M + 1
  const int tid = threadIdx.x;
  const int b = blockIdx.y, f = blockIdx.x;
  const float* x = a.wav + (long long)b * a.T;
  const int start = f * a.hop - M;
  // ---- forward spectrum of this frame (same steps as melspec_kernel)
  for (int n = tid; n < M; n += MEL_THREADS) {
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      int s = start + 2 * n + e;
      float xv = 0.f;
      if (a.pad_mode == 1) {
        if (s < 0) s = -s;
        if (s >= a.T) s = 2 * (a.T - 1) - s;
        xv = (s >= 0 && s < a.T) ? x[s] : 0.f;
      } else if (s >= 0 && s < a.T) {
        xv = x[s];
      }
      v[e] = xv * a.window[2 * n + e];
    }
    buf0[n] = make_float2(v[0], v[1]);
  }
  __syncthreads();
  float2* src = buf0;
  float2* dst = buf1;
  for (int s = 0; s < a.log2m; ++s) {
    const int Ns = 1 << s;
    for (int j = tid; j < (M >> 1); j += MEL_THREADS) {
      const int k = j & (Ns - 1);
      const float2 w = a.tw[(k * (M >> (s + 1))) << 1];
      const float2 u = src[j];
      const float2 t = cmul(src[j + (M >> 1)], w);
      const int d = ((j >> s) << (s + 1)) + k;
      dst[d] = make_float2(u.x + t.x, u.y + t.y);
      dst[d + Ns] = make_float2(u.x - t.x, u.y - t.y);
    }
    __syncthreads();
    float2* tmp = src;
    src = dst;
    dst = tmp;
  }
  for (int k = tid; k <= M; k += MEL_THREADS) {
    float re, im;
    if (k == 0 || k == M) {
      const float2 z0 = src[0];
      re = (k == 0) ? (z0.x + z0.y) : (z0.x - z0.y);
      im = 0.f;
    } else {
      const float2 zk = src[k];
      const float2 zc = src[M - k];
      const float er = 0.5f * (zk.x + zc.x), ei = 0.5f * (zk.y - zc.y);
      const float dr = zk.x - zc.x, di = zk.y + zc.y;
      const float orr = 0.5f * di, oi = -0.5f * dr;
      const float2 w = a.tw[k];
      re = er + (orr * w.x - oi * w.y);
      im = ei + (orr * w.y + oi * w.x);
    }
    X[k] = make_float2(re, im);
    damp[k] = ba.dmag ? ba.dmag[((long long)b * a.frames + f) * (M + 1) + k] : 0.f;
  }
  __syncthreads();
  // ---- d mel -> d amp (scatter over each filter's support)
  if (!ba.dmag && tid < a.n_mels) {
    const int st = a.mel_start[tid], ln = a.mel_len[tid];
    const float* w = a.mel_w + a.mel_off[tid];
    float acc = 0.f;
    for (int i = 0; i < ln; ++i) {
      const float2 z = X[st + i];
      acc = fmaf(sqrtf(fmaxf(z.x * z.x + z.y * z.y, a.eps_power)), w[i], acc);
    }
    const float g = a.mel_fm ? ba.dmel[((long long)b * a.frames + f) * a.n_mels + tid]
                             : ba.dmel[((long long)b * a.n_mels + tid) * a.frames + f];
    const float melc = fmaxf(acc, a.eps_mel);
    const float db = 20.f * log10f(fmaxf(melc, 1e-5f)) - 20.f;
    const float nv = 8.f * ((db + 100.f) / 100.f) - 4.f;
    float gm = 0.f;
    if (nv > -4.f && nv < 4.f && melc > 1e-5f && acc > a.eps_mel) gm = g * 0.08f * 8.685889638065035f / melc;  // 20/ln(10)
    if (gm != 0.f)
      for (int i = 0; i < ln; ++i) atomicAdd(&damp[st + i], gm * w[i]);
  }
  __syncthreads();
  // ---- conj(G) into the N-point buffer (zero above N/2)
  for (int k = tid; k < N; k += MEL_THREADS) {
    float2 gq = make_float2(0.f, 0.f);
    if (k <= M) {
      const float2 z = X[k];
      const float p = z.x * z.x + z.y * z.y;
      if (p > a.eps_power) {
        const float sc = damp[k] / sqrtf(p);
        gq = make_float2(sc * z.x, -sc * z.y);
      }
    }
    buf0[k] = gq;
  }
  __syncthreads();
  src = buf0;
  dst = buf1;
  for (int s = 0; s <= a.log2m; ++s) {  // log2(N) stages
    const int Ns = 1 << s;
    for (int j = tid; j < M; j += MEL_THREADS) {
      const int k = j & (Ns - 1);
      const float2 w = a.tw[k * (M >> s)];  // exp(-2 pi i k / (2 Ns)) = tw_N[k * N / (2 Ns)]
      const float2 u = src[j];
      const float2 t = cmul(src[j + M], w);
      const int d = ((j >> s) << (s + 1)) + k;
      dst[d] = make_float2(u.x + t.x, u.y + t.y);
      dst[d + Ns] = make_float2(u.x - t.x, u.y - t.y);
    }
    __syncthreads();
    float2* tmp = src;
    src = dst;
    dst = tmp;
  }
  // y[n] = conj(FFT(conj G))[n];  d x_w[n] = Re y[n] = Re FFT(conj G)[n]
  float* dx = ba.dwav + (long long)b * a.T;
  for (int n = tid; n < N; n += MEL_THREADS) {
    const float v = src[n].x * a.window[n];
    int s = start + n;
    if (a.pad_mode == 1) {
      if (s < 0) s = -s;
      if (s >= a.T) s = 2 * (a.T - 1) - s;
    }
    if (s >= 0 && s < a.T && v != 0.f) atomicAdd(&dx[s], v);
  }
}

extern "C" int kantts_melspec_bwd_fm(const float* wav, const float* dmel, int B, int T, int n_fft, int hop, int frames,
                                     int pad_mode, const float* window, const float* twiddle, float eps_power,
                                     const int32_t* mel_start, const int32_t* mel_len, const int32_t* mel_off,
                                     const float* mel_w, int n_mels, float eps_mel, int mel_frame_major, float* dwav_accum,
                                     void* stream);
extern "C" int kantts_melspec_bwd(const float* wav, const float* dmel, int B, int T, int n_fft, int hop, int frames,
                                  int pad_mode, const float* window, const float* twiddle, float eps_power,
                                  const int32_t* mel_start, const int32_t* mel_len, const int32_t* mel_off,
                                  const float* mel_w, int n_mels, float eps_mel, float* dwav_accum, void* stream) {
  return kantts_melspec_bwd_fm(wav, dmel, B, T, n_fft, hop, frames, pad_mode, window, twiddle, eps_power, mel_start, mel_len,
                               mel_off, mel_w, n_mels, eps_mel, 0, dwav_accum, stream);
}
extern "C" int kantts_melspec_bwd_fm(const float* wav, const float* dmel, int B, int T, int n_fft, int hop, int frames,
                                     int pad_mode, const float* window, const float* twiddle, float eps_power,
                                     const int32_t* mel_start, const int32_t* mel_len, const int32_t* mel_off,
                                     const float* mel_w, int n_mels, float eps_mel, int mel_frame_major, float* dwav_accum,
                                     void* stream) {
  if (!wav || !dmel || !window || !twiddle || !dwav_accum || !mel_start || !mel_len || !mel_off || !mel_w)
    return KANTTS_E_BADARG;
  if (B < 0 || T < 1 || n_fft < 8 || hop < 1 || frames < 0 || n_mels < 1 || n_mels > MEL_THREADS) return KANTTS_E_BADARG;
  if (n_fft & (n_fft - 1)) return KANTTS_E_UNSUPPORTED;
  if (B == 0 || frames == 0) return KANTTS_OK;
  MelBwdArgs ba = {};
  MelArgs& a = ba.f;
  a.wav = wav; a.B = B; a.T = T; a.n_fft = n_fft; a.hop = hop; a.frames = frames; a.pad_mode = pad_mode;
  a.window = window; a.tw = reinterpret_cast<const float2*>(twiddle); a.eps_power = eps_power;
  a.mel_start = mel_start; a.mel_len = mel_len; a.mel_off = mel_off; a.mel_w = mel_w; a.n_mels = n_mels;
  a.eps_mel = eps_mel;
  a.ref_db = 20.f; a.min_db = -100.f; a.max_norm = 4.f; a.symmetric = 1;
  a.mel_fm = mel_frame_major ? 1 : 0;
  ba.dmel = dmel; ba.dwav = dwav_accum;
  int m = n_fft >> 1, l2 = 0;
  while ((1 << l2) < m) ++l2;
  a.log2m = l2;
  size_t lds = (size_t)n_fft * 2 * sizeof(float2) + (size_t)(m + 1) * (sizeof(float2) + sizeof(float));
  if (lds > 64 * 1024) return KANTTS_E_UNSUPPORTED;
  hipLaunchKernelGGL(melspec_bwd_kernel, dim3(frames, B), dim3(MEL_THREADS), lds, (hipStream_t)stream, ba);
  KANTTS_CHECK_LAUNCH();
}

// Backward of the STFT magnitude (kantts/utils/audio_torch.py:8-31: sqrt(clamp(re^2 + im^2, eps))) -- the gradient path
// of MultiResolutionSTFTLoss (kantts/train/loss.py:312-441).  Same kernel as the mel backward with the d-amplitude taken
// straight from dmag (B, frames, n_fft/2+1).
extern "C" int kantts_stft_mag_bwd(const float* wav, const float* dmag, int B, int T, int n_fft, int hop, int frames,
                                   int pad_mode, const float* window, const float* twiddle, float eps_power,
                                   float* dwav_accum, void* stream) {
  if (!wav || !dmag || !window || !twiddle || !dwav_accum) return KANTTS_E_BADARG;
  if (B < 0 || T < 1 || n_fft < 8 || hop < 1 || frames < 0) return KANTTS_E_BADARG;
  if (n_fft & (n_fft - 1)) return KANTTS_E_UNSUPPORTED;
  if (B == 0 || frames == 0) return KANTTS_OK;
  MelBwdArgs ba = {};
  MelArgs& a = ba.f;
  a.wav = wav; a.B = B; a.T = T; a.n_fft = n_fft; a.hop = hop; a.frames = frames; a.pad_mode = pad_mode;
  a.window = window; a.tw = reinterpret_cast<const float2*>(twiddle); a.eps_power = eps_power;
  a.n_mels = 0;
  ba.dmel = nullptr; ba.dmag = dmag; ba.dwav = dwav_accum;
  int m = n_fft >> 1, l2 = 0;
  while ((1 << l2) < m) ++l2;
  a.log2m = l2;
  size_t lds = (size_t)n_fft * 2 * sizeof(float2) + (size_t)(m + 1) * (sizeof(float2) + sizeof(float));
  if (lds > 64 * 1024) return KANTTS_E_UNSUPPORTED;
  hipLaunchKernelGGL(melspec_bwd_kernel, dim3(frames, B), dim3(MEL_THREADS), lds, (hipStream_t)stream, ba);
  KANTTS_CHECK_LAUNCH();
}
