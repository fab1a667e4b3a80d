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
    float* o = a.out_mel + ((long long)b * a.n_mels + tid) * a.frames + f0;
#pragma unroll
    for (int q = 0; q < MEL_FB; ++q)
      if (f0 + q < a.frames) o[q] = melv[q];
  }
}


// [round 4] A second form of this kernel -- one WAVE per frame with private ping-pong buffers, radix-4 Stockham passes and
// no workgroup barrier inside a frame, window / twiddles / mel weights staged in LDS, 16 frames per workgroup -- was
// written, passed every parity test (9.5e-7 against the oracle) and was measured against this one on one box
// (profiles/r04_runG_melspec_wave_kernel_ab.log): 325 against 343 us at 67 584 frames of n_fft 1024, 35 against 25 us at
// the benchmark's 1 056 frames, 812 against 518 us at n_fft 2048.  The barriers were not the bound: the kernel is
// instruction-issue / LDS-latency bound per wave (address arithmetic and LDS instructions around ~10 flops per butterfly),
// and the private buffers cut the resident waves per CU from 32 to 8.  Removed; what would pay is a register-resident
// radix-8 FFT (three passes, two LDS exchanges per frame) -- DESIGN.md section 8.

extern "C" int kantts_melspec_norm_fwd(const float* wav, int B, int T, int n_fft, int hop, int frames, int pad_mode,
                                       const float* window, const float* twiddle, float eps_power,
                                       const int32_t* mel_start, const int32_t* mel_len, const int32_t* mel_off,
                                       const float* mel_w, int n_mels, float eps_mel, float ref_level_db,
                                       float min_level_db, float max_norm, int symmetric, float* out_mel, float* out_mag,
                                       void* stream);

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
  int m = n_fft >> 1, l2 = 0;
  while ((1 << l2) < m) ++l2;
  a.log2m = l2;
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
    const float g = ba.dmel[((long long)b * a.n_mels + tid) * a.frames + f];
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

extern "C" int kantts_melspec_bwd(const float* wav, const float* dmel, int B, int T, int n_fft, int hop, int frames,
                                  int pad_mode, const float* window, const float* twiddle, float eps_power,
                                  const int32_t* mel_start, const int32_t* mel_len, const int32_t* mel_off,
                                  const float* mel_w, int n_mels, float eps_mel, float* dwav_accum, void* stream) {
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
