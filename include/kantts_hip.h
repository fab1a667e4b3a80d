/* libkantts_hip.so -- C ABI of the MI355X (gfx950) kernels behind the KAN-TTS hot path.
 *
 * The reference (modelscope/KAN-TTS) has no FFI: every "kernel" is a stock ATen op called from
 * the kantts/models Python modules.  Each entry point below names the reference call sites (file:line under
 * /root/reference) whose arithmetic it replaces.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - plain C, no C++ types / exceptions cross this boundary; every function returns int:
 *     0 = KANTTS_OK, negative = KANTTS_E_* (bad arguments), positive = hipError_t.
 *   - every pointer is a DEVICE pointer unless the parameter name ends in _host; the caller
 *     (PyTorch's caching allocator in the Python host layer) owns all buffers incl. workspaces;
 *     the library never allocates device memory and keeps no references after returning.
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); stateless and re-entrant.
 *   - tensors are contiguous row-major fp32 unless stated; lengths/indices are int32 or int64
 *     as stated; "tokens" are rows of a (B, T, C) channels-last activation, row = b*T + t.
 */
#ifndef KANTTS_HIP_H
#define KANTTS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KANTTS_OK 0
#define KANTTS_E_BADARG (-1)
#define KANTTS_E_UNSUPPORTED (-2)
#define KANTTS_E_WORKSPACE (-3)

/* Library/ABI version and target ISA string ("gfx950"). */
int kantts_abi_version(void);
const char* kantts_target_arch(void);

/* ------------------------------------------------------------------------------------------
 * Segmented GEMM:  C[i][j] (op)= epilogue( alpha * sum_seg sum_tap sum_kk A(i,kk) * B(j,kk) )
 *
 * One kernel covers every dense contraction of the path (channels-last activations):
 *   nn.Linear fwd / dgrad / wgrad      kantts/models/sambert/__init__.py:82,102,217,242,287,297
 *   Conv1d k=1 and k=3 (im2col-free: one segment with ntaps shifted row windows)
 *                                      kantts/models/sambert/__init__.py:115-127, fsmn.py:14-29
 *   concatenated inputs (torch.cat + Linear) as several segments
 *                                      kantts/models/sambert/kantts_sambert.py:173-174, adaptors.py:54
 *   sums of projections (fc_x + fc_h)  kantts/models/sambert/__init__.py:287-299
 *   HiFi-GAN dilated Conv1d / polyphase ConvTranspose1d (taps + strided C)
 *                                      kantts/models/hifigan/layers.py:82-88,153-161
 * A(i,kk) = a[i*a_is + kk*a_ks] and B(j,kk) = b[j*b_js + kk*b_ks + tap*b_tap]; a token shift
 * (conv tap) moves the index that is a token (row of a (B,T,C) tensor) by s = shift0+tap*step and
 * yields 0 when (tok % T) + s falls outside [0, T).  Arithmetic: precision 0 = fp32 MFMA
 * (v_mfma_f32_16x16x4_f32, exact f32 FMA chain), 1 = bf16 MFMA inputs with fp32 accumulate
 * (v_mfma_f32_16x16x32_bf16), 2 = scalar fp32 reference kernel (debug).
 */
#define KANTTS_GEMM_MAX_SEG 4
typedef struct kantts_gemm_seg {
  const float* a;      /* A operand */
  const float* a_gate; /* optional, same addressing as A: A(i,kk) is used only where gate > 0
                          (ReLU / zeroed-row backward)                                           */
  const float* b;      /* B operand */
  int64_t a_is, a_ks;  /* A strides over i and kk (elements) */
  int64_t b_js, b_ks;  /* B strides over j and kk */
  int64_t b_tap;       /* B element offset per tap */
  int32_t klen;        /* reduction length per tap */
  int32_t ntaps;       /* >= 1 */
  int32_t a_tok_axis;  /* 0: A has no token shift, 1: i is the token index, 2: kk is */
  int32_t a_shift0, a_shift_step;
  int32_t b_tok_axis;  /* 0: none, 2: kk is the token index of B */
  int32_t b_shift0, b_shift_step;
  float a_drop_p;      /* > 0: A(i,kk) *= dropout keep-scale regenerated from (a_drop_seed, element
                          offset of A) -- backward of an epilogue dropout, A = dY contiguous (M,N) */
  uint64_t a_drop_seed;
  /* Extended token map (strided / transposed / period-folded / nearest-upsampled convolutions of the
   * HiFi-GAN stack, kantts/models/hifigan/{layers,hifigan}.py).  All zero = the plain map above.
   * A token tok of a (B, Tq, inner) domain reads source row ((b*Tsrc) + t)*inner + pi with
   * t = q*mul + shift; when div > 1, t must be a multiple of div and t /= div; the element is 0
   * unless 0 <= t < Tsrc*up; finally t /= up (nearest-neighbour upsampling of the source). */
  int32_t a_inner, a_Tq, a_Tsrc, a_mul, a_div, a_up;
  int32_t b_inner, b_Tq, b_Tsrc, b_mul, b_div, b_up;
  float a_slope;       /* a_act = 1: A value x -> x > 0 ? x : a_slope * x (LeakyReLU fused on load) */
  int32_t a_act;
  float b_slope;
  int32_t b_act;
  float a_gate_slope;  /* where the gate is <= 0, A is multiplied by this (0: hard gate; LeakyReLU backward) */
  int32_t a_mode, b_mode; /* staging layout hints, see csrc/gemm.hip: 0 scalar/lanes along k, 1 scalar/lanes
                          along rows, 2 float4 along k (needs unit k stride, klen % 4 == 0, 16-byte aligned
                          rows, no token map on kk), 3 float4 along rows (unit row stride, extent % 4 == 0,
                          no token map on the row index).  0 is always valid. */
} kantts_gemm_seg;

typedef struct kantts_gemm_args {
  kantts_gemm_seg seg[KANTTS_GEMM_MAX_SEG];
  int32_t nseg;
  int32_t M, N;        /* extent of i and j */
  int32_t T;           /* tokens per sequence for token shifts (ignored when no shift) */
  float* c;
  int64_t c_is, c_js;  /* C strides */
  const float* bias;   /* over j, optional: v = (acc + bias[j] + bias2[j]) * alpha */
  const float* bias2;  /* optional second bias (sum of two projections) */
  const float* res;    /* optional residual added after the activation */
  int64_t r_is, r_js;
  const uint8_t* rowmask; /* optional, over i: rows with mask != 0 are written as 0 */
  const uint8_t* kmask;   /* optional, over kk (same for all segments): kk with mask != 0 skipped */
  float* a_rowsum;     /* optional: a_rowsum[i] += sum_kk A(i,kk) of segment 0 (bias gradient) */
  float alpha;
  int32_t relu;        /* 1: v = max(v, 0) before residual */
  int32_t accumulate;  /* 0: C = v, 1: atomicAdd(C, v) (required when splitk > 1) */
  int32_t splitk;      /* >= 1: reduction tiles are dealt round-robin to gridDim.z slices */
  int32_t precision;   /* 0 fp32 MFMA, 1 bf16 MFMA, 2 scalar reference */
  float drop_p;        /* dropout applied to v after the activation, before the residual */
  uint64_t drop_seed;  /* mask element (i,j) = rng(drop_seed + *seed_dev, i*N + j) */
  const uint64_t* seed_dev; /* optional device word added to every dropout seed of this launch (lets a
                          captured hipGraph draw fresh masks on every replay) */
  /* grouped convolutions: gridDim.z = groups * splitk; group g offsets every operand */
  int32_t groups;      /* 0/1 = none */
  int64_t a_gs, b_gs, c_gs, bias_gs, r_gs; /* element offsets per group for A, B, C(+gate), bias, res */
  float out_slope;     /* out_act = 1: LeakyReLU(out_slope) on v instead of / after relu=0 */
  int32_t out_act;
  const float* gate;   /* optional, addressed like C: v *= (gate > 0 ? 1 : gate_slope)  (backward through
                          an input-side LeakyReLU) */
  float gate_slope;
  int32_t z_taps;      /* > 0: gridDim.z additionally enumerates z_taps taps of segment 0 (each z-slab handles one
                          tap and writes C + tap * c_tap): one launch for all taps of a conv weight gradient */
  int64_t c_tap;
} kantts_gemm_args;

int kantts_gemm_seg_launch(const kantts_gemm_args* args_host, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm (eps inside sqrt, biased variance).  Replaces nn.LayerNorm(eps=1e-6) at
 * kantts/models/sambert/__init__.py:63,130,198 and kantts/models/sambert/kantts_sambert.py:58,128.
 * x,y,dx,dy: (M,C); mean,rstd: (M) saved for backward; dgamma/dbeta are ACCUMULATED (atomicAdd). */
int kantts_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                         float* rstd, int M, int C, float eps, void* stream);
int kantts_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                         const float* rstd, float* dx, float* dgamma_accum, float* dbeta_accum, int M, int C,
                         void* stream);

/* ------------------------------------------------------------------------------------------
 * Range-limited multi-head attention, d_head = 16.  Replaces ScaledDotProductAttention and the mask
 * tensors (kantts/models/sambert/__init__.py:17-29,85-100; kantts_sambert.py:135-166).
 * q/k/v/o/d*: element (b,t,h,d) at ptr[(b*L+t)*ld + h*16 + d].  mode 0: key padding (encoder),
 * 1: PNCA x (causal band [i-bw, i]), 2: PNCA h (look-ahead band [i, i+bw]); lens (B) int32 = valid
 * positions per sequence (NULL: all); bw_dev: optional device scalar overriding bw.
 * lse: (B,H,L) saved; probs: optional (H*B, L, L) post-dropout probabilities (reference layout).
 * Band modes: rows of padded queries (t >= lens[b]) are un-masked in the reference but zeroed by every
 * caller; they are computed only when probs is requested, otherwise their context is 0, and they never
 * receive / produce gradients.
 * Backward writes dq (or adds to it when accumulate_dq), dk, dv; dvec (B,H,L) is scratch. */
int kantts_attn_fwd(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, float* o, int ldo,
                    float* lse, float* probs, const int32_t* lens, const int32_t* bw_dev, int bw, int B, int H,
                    int L, int d_head, int mode, float drop_p, uint64_t seed, const uint64_t* seed_dev,
                    void* stream);
int kantts_attn_bwd(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, const float* o,
                    int ldo, const float* d_o, int lddo, const float* lse, float* dvec, float* dq, float* dk,
                    float* dv, int lddq, int lddk, int lddv, int accumulate_dq, const int32_t* lens,
                    const int32_t* bw_dev, int bw, int B, int H, int L, int d_head, int mode, float drop_p,
                    uint64_t seed, const uint64_t* seed_dev, void* stream);

/* Both attentions of a PNCA block (MultiHeadPNCAAttention.forward, kantts/models/sambert/__init__.py:256-306: the causal
 * band over x and the look-ahead band over the memory share the queries) as ONE launch forward and ONE backward.
 * qkv (B,L,3D) = fused x projection [q | k_x | v_x], hkv = memory projection [k_h | v_h] with row pitch ldh >= 2D (the
 * twelve blocks' projections are columns of one (B,L,12*2D) GEMM output), D = H*16;
 * ox / oh (B,L,D) contexts, lse_x / lse_h (B,H,L) saved.  Backward: dqkv (B,L,3D) = [dq | dk_x | dv_x] with dq the SUM of
 * both bands' query gradients, dhkv (B,L,2D) = [dk_h | dv_h]; every output is written, none accumulated.  For L > 256
 * (more queries than a workgroup has threads) the backward returns 1 instead of KANTTS_OK: dqkv's
 * first D columns then hold the x band's query gradient only, dqh (B,L,D, required in that case) the memory band's, and
 * the caller adds them.  Same masks, dropout streams (seed_x / seed_h) and padded-row rules as kantts_attn_fwd / _bwd with
 * mode 1 / mode 2.  KANTTS_E_UNSUPPORTED if a head's rows do not fit in LDS (L > ~390): use the per-band calls. */
int kantts_pnca_attn_fwd(const float* qkv, const float* hkv, int ldh, float* ox, float* oh, float* lse_x, float* lse_h,
                         const int32_t* lens, const int32_t* bw_dev, int bw_x, int bw_h, int B, int H, int L, int d_head,
                         float drop_p, uint64_t seed_x, uint64_t seed_h, const uint64_t* seed_dev, void* stream);
int kantts_pnca_attn_bwd(const float* qkv, const float* hkv, int ldh, const float* ox, const float* oh, const float* d_ox,
                         const float* d_oh, const float* lse_x, const float* lse_h, float* dqkv, float* dqh, float* dhkv,
                         const int32_t* lens, const int32_t* bw_dev, int bw_x, int bw_h, int B, int H, int L, int d_head,
                         float drop_p, uint64_t seed_x, uint64_t seed_h, const uint64_t* seed_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * LSTM recurrence, H = 128 (time loop of torch.nn.LSTM: kantts/models/sambert/adaptors.py:44-57,
 * :109-134; kantts_sambert.py:637-646).  gx (B,T,ndir*4H) = x W_ih^T + b_ih (from the GEMM);
 * whh (ndir,4H,H), bhh (ndir,4H) or NULL; lens (B) int32 or NULL = pack_padded_sequence lengths;
 * out (B,T,ndir*H); gates_save (ndir*B*T*4H floats; private to the fwd / bwd pair: since round 6 a cell's four activations
 * side by side, (ndir,B,T,H,4)) and c_save (ndir,B,T,H) are kept for backward.
 * Backward returns dgates (ndir,B,T,4H) = gradient w.r.t. the pre-activation gates.
 * precision 0: fp32 recurrent products; 1: h / dgates and W_hh rounded to bf16 for the recurrent product only
 * (packed v_dot2c_f32_bf16, fp32 accumulate) -- gates, cell state, outputs stay fp32. */
int kantts_lstm_fwd(const float* gx, const float* whh, const float* bhh, const int32_t* lens, float* out,
                    float* gates_save, float* c_save, int B, int T, int H, int ndir, int reverse_first,
                    int precision, void* stream);
int kantts_lstm_bwd(const float* dout, const float* whh, const int32_t* lens, const float* gates_save,
                    const float* c_save, float* dgates, int B, int T, int H, int ndir, int reverse_first,
                    int precision, void* stream);

/* ------------------------------------------------------------------------------------------
 * Embedding gather-sum: out[row] = scale * sum_k table_k[ids[row,k]] (+ pos[row % T]);
 * scaled_out (optional) receives the value before the positional add.  tables_host is a HOST array
 * of ntab (<=4) device pointers.  kantts/models/sambert/kantts_sambert.py:308-329, :62-64.
 * Backward scatter-adds scale*dout into dtables (NULL entries skipped). */
int kantts_embed_sum_fwd(const float* const* tables_host, int ntab, const int64_t* ids, const float* pos,
                         float* out, float* scaled_out, int rows, int T, int D, float scale, void* stream);
int kantts_embed_sum_bwd(float* const* dtables_host, int ntab, const int64_t* ids, const float* dout, int rows,
                         int D, float scale, void* stream);
/* [round 5] The length- / target-only preparation of a teacher-forced SAM-BERT step in one launch (csrc/seq.hip;
 * kantts/models/utils.py:13-23, kantts_sambert.py:466-468, 556-559, 736-750, 981-985, positions.py:83-98):
 *   in / out / lfr: lengths clamped to N / T_mel / Tp/r as int64 and int32, and masks (uint8, 1 = padding) of shapes (B, N),
 *     (B, T_mel), (B, Tp/r) -- lfr lengths are ceil(out_lens / r);  valid[b] = min(out length, max_len);
 *   pos_enc (B, Tp, depth): sin (even channels) / cos (odd) of pos_masked / inv_ts[c], pos = kantts_lr_index's, masked to 0
 *     from frame min(out length, max_len) on;  prev (B, N) = log(dur[b, n-1] + 1) (0 for n = 0);
 *   bw_val = max over valid (b, n) of dur / r + 0.5 (fp32), bw_dev = its truncation (int32);
 *   dec_input (B, Tp/r, d_mel): row 0 zeros, row l = mel[b, l*r - 1]. */
typedef struct kantts_plan_args {
  const int64_t* in_lens;
  const int64_t* out_lens;
  const int64_t* dur;
  const float* mel;
  const float* pos;
  const float* inv_ts;
  int32_t B, N, T_mel, Tp, max_len, r, d_mel, depth;
  int64_t* in_l64;
  int32_t* in_l32;
  uint8_t* in_mask;
  int64_t* out_l64;
  int32_t* out_l32;
  uint8_t* out_mask;
  int64_t* lfr_l64;
  int32_t* lfr_l32;
  uint8_t* lfr_mask;
  int64_t* valid;
  float* pos_enc;
  float* prev;
  float* bw_val;
  int32_t* bw_dev;
  float* dec_input;
} kantts_plan_args;
int kantts_teacher_plan(const kantts_plan_args* args, void* stream);

/* ------------------------------------------------------------------------------------------
 * Length regulator (kantts/models/sambert/adaptors.py:15-36, positions.py:72-90) in index form.
 * kantts_lr_index: reps = trunc(dur + 0.5) (dur_int (B,N) int64 or dur_float (B,N) fp32);
 *   idx (B,Tp) int32 token per frame or -1; pos (B,Tp) 1-based position inside the token (t+1 when
 *   uncovered); cs (B,N+1) int32 exclusive prefix sums; lens (B) int64 totals.  Bit-exact.
 * kantts_lr_gather_fwd: out[b,t, off:off+C] = x[b, idx[b,t], :] (0 when idx<0 or t >= valid_lens[b]);
 *   out rows have stride ldo (lets three regulators write one concatenated buffer).
 * kantts_lr_gather_bwd: deterministic segment sum over [cs[n], cs[n+1]). */
int kantts_lr_index(const int64_t* dur_int, const float* dur_float, int32_t* idx, float* pos, int32_t* cs,
                    int64_t* lens, int B, int N, int Tp, void* stream);
int kantts_lr_gather_fwd(const float* x, const int32_t* idx, const int64_t* valid_lens, float* out, int B, int N,
                         int Tp, int C, int ldo, int out_col_offset, void* stream);
int kantts_lr_gather_bwd(const float* dout, const int32_t* cs, const int64_t* valid_lens, float* dx, int B, int N,
                         int Tp, int C, int ldo, int out_col_offset, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * FSMN memory block (kantts/models/sambert/fsmn.py:43-72), channels-last (B,T,C), w (C,K):
 *   xm = x*keep; y = keep*(sum_k w[c,k]*xm[t+k-left_pad] + xm[t]) (+ res); keep[b,t] = t < lens[b].
 * Backward: dx (w.r.t. x) and dw_accum (atomicAdd); d(res) = dy.  dx or dw_accum may be NULL: only the other gradient is
 * computed (two calls on two streams: the filter gradient runs beside the backward pass's critical path). */
int kantts_fsmn_dwconv_fwd(const float* x, const float* w, const float* res, const int64_t* lens, float* y, int B,
                           int T, int C, int K, int left_pad, void* stream);
int kantts_fsmn_dwconv_bwd(const float* dy, const float* x, const float* w, const int64_t* lens, float* dx,
                           float* dw_accum, float* workspace, long long ws_floats, int B, int T, int C, int K,
                           int left_pad, void* stream);
/* floats of caller-owned workspace the backward needs (weight-gradient partials; 0 when none) */
long long kantts_fsmn_dwconv_bwd_ws(int B, int T, int C, int K);

/* ------------------------------------------------------------------------------------------
 * Masked L1 ("mae") reduction: loss_accum[0] += sum_{t<lens[b]} |target-pred| / (sum(lens)*C);
 * grad (optional, (B,T,C)) = d loss / d pred.  kantts/train/loss.py:18-37, :51-85. */
int kantts_masked_l1(const float* pred, const float* target, const int64_t* lens, float* loss_accum, float* grad,
                     int B, int T, int C, void* stream);

/* Mean-style element losses of the GAN step (kantts/train/loss.py:108-256,310): mode 0: loss_accum += scale*sum|a-b|,
 * grad = scale*sign(a-b); mode 1: loss_accum += scale*sum (a-target)^2, grad = 2*scale*(a-target).  grad optional. */
int kantts_elem_loss(const float* a, const float* b, float target, int mode, float scale, float* loss_accum,
                     float* grad, long long n, void* stream);

/* out_accum[0] += sum x^2 (global gradient norm, torch.nn.utils.clip_grad_norm_) */
int kantts_sumsq(const float* x, float* out_accum, long long n, void* stream);

/* torch.optim.Adam step on flat fp32 arenas with optional global-norm clipping read from device
 * memory (gnorm_sq = sum of squares of all gradients): kantts/train/trainer.py:997-1004. */
int kantts_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                     float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2,
                     const float* gnorm_sq, float max_norm, const float* dyn_lr_step, void* stream);
/* dyn_lr_step (optional, device, 2 floats {lr, step}): when given, lr and the bias corrections
 * 1 - beta^step are taken from device memory (hipGraph replay with a changing schedule). */

/* ------------------------------------------------------------------------------------------
 * Fused framed STFT -> |.| -> sparse mel filterbank -> 20 log10 - 20 -> clamp(8(x+100)/100-4, +-4).
 * Replaces MelSpectrogram.forward (kantts/utils/audio_torch.py:155-186) and, with out_mag only,
 * stft() (:8-31).  wav (B,T); window (n_fft) = hann(win_length) centred in n_fft; twiddle (n_fft/2
 * complex pairs) = exp(-2 pi i t / n_fft); pad_mode 0 = zeros (MelSpectrogram), 1 = reflect (stft);
 * frames = 1 + T / hop.  The mel basis is passed in support form: filter m covers bins
 * [mel_start[m], mel_start[m]+mel_len[m]) with weights mel_w[mel_off[m] ...].
 * out_mel (B, n_mels, frames) and/or out_mag (B, frames, n_fft/2+1). */
int kantts_melspec_fwd(const float* wav, int B, int T, int n_fft, int hop, int frames, int pad_mode,
                       const float* window, const float* twiddle, float eps_power, const int32_t* mel_start,
                       const int32_t* mel_len, const int32_t* mel_off, const float* mel_w, int n_mels,
                       float eps_mel, float* out_mel, float* out_mag, void* stream);

/* The same pipeline with the dB normalisation as parameters: S = 20 log10(max(mel, 1e-5)) - ref_level_db;
 * symmetric: clip(2*max_norm*(S - min_level_db)/(-min_level_db) - max_norm, +-max_norm); else clip(max_norm*(S - min_level_db)
 * /(-min_level_db), 0, max_norm).  Serves the offline feature extractor melspectrogram()
 * (kantts/preprocess/audio_processor/core/dsp.py:165-201: eps_power = 0, eps_mel = 1e-5, ref 20, min -100, max_norm 1,
 * asymmetric); kantts_melspec_fwd is this function at (20, -100, 4, symmetric). */
int kantts_melspec_norm_fwd(const float* wav, int B, int T, int n_fft, int hop, int frames, int pad_mode,
                            const float* window, const float* twiddle, float eps_power, const int32_t* mel_start,
                            const int32_t* mel_len, const int32_t* mel_off, const float* mel_w, int n_mels,
                            float eps_mel, float ref_level_db, float min_level_db, float max_norm, int symmetric,
                            float* out_mel, float* out_mag, void* stream);

/* [round 4] The same two functions with the mel tensor FRAME-major when mel_frame_major != 0: out_mel / dmel are
 * (B, frames, n_mels) instead of (B, n_mels, frames) -- a frame's channels are one contiguous store (the channel-major
 * layout makes every store a partial sector: 3 x the algorithmic write traffic); the host layer hands the reference's
 * (B, n_mels, frames) shape out as a transposed view.  mel_frame_major == 0: identical to the functions above / below. */
int kantts_melspec_norm_fwd_fm(const float* wav, int B, int T, int n_fft, int hop, int frames, int pad_mode,
                               const float* window, const float* twiddle, float eps_power, const int32_t* mel_start,
                               const int32_t* mel_len, const int32_t* mel_off, const float* mel_w, int n_mels, float eps_mel,
                               float ref_level_db, float min_level_db, float max_norm, int symmetric, int mel_frame_major,
                               float* out_mel, float* out_mag, void* stream);
int kantts_melspec_bwd_fm(const float* wav, const float* dmel, int B, int T, int n_fft, int hop, int frames, int pad_mode,
                          const float* window, const float* twiddle, float eps_power, const int32_t* mel_start,
                          const int32_t* mel_len, const int32_t* mel_off, const float* mel_w, int n_mels, float eps_mel,
                          int mel_frame_major, float* dwav_accum, void* stream);

/* [round 5] Measurement aid, not part of the model: a launch that reads read_bytes from src and writes write_bytes to dst
 * once each (16-byte chunks, whole chip) -- the copy roof a kernel with those algorithmic bytes is judged against
 * (bench.py: upsampling.copy_roof_us). */
int kantts_copy_roof(const void* src, long long read_bytes, void* dst, long long write_bytes, void* stream);

/* [round 5] The two knobs of the mel-STFT launchers, explicit instead of environment variables read on every launch: grid_cap
 * > 0 caps the persistent grid of the register-resident n_fft 1024 kernel (0 = default, 768 workgroups) -- sweeps, and the
 * tests that must reach its several-pairs-per-wave loop with small inputs; generic_only != 0 routes every size to the radix-2
 * kernel (A/B, tests).  Process-global; the only state the library keeps. */
int kantts_melspec_tuning(int grid_cap, int generic_only);

/* Backward of the mel path of kantts_melspec_fwd (MelSpectrogramLoss on generated audio, kantts/train/loss.py:
 * 259-311): dwav_accum (B,T) += d loss / d wav given dmel (B, n_mels, frames).  The spectrum is recomputed. */
int kantts_melspec_bwd(const float* wav, const float* dmel, int B, int T, int n_fft, int hop, int frames,
                       int pad_mode, const float* window, const float* twiddle, float eps_power,
                       const int32_t* mel_start, const int32_t* mel_len, const int32_t* mel_off, const float* mel_w,
                       int n_mels, float eps_mel, float* dwav_accum, void* stream);

/* The five masked-L1 terms of a SAM-BERT training step (MelReconLoss on the decoder and postnet mels, ProsodyReconLoss on
 * log-duration / pitch / energy: kantts/train/loss.py:7-85) in ONE launch: term k over pred (B, T, C) against target
 * (float, or -- target_log1p -- int64 values v read as log(v + 1): the duration targets) on rows t < lens[b];
 * losses[k] += mean |target - pred|, losses[KANTTS_LOSS_MAX_TERMS] += the same (the step's total); grad[k] (optional) =
 * d term k / d pred as kantts_masked_l1 writes it.  losses must be zeroed by the caller. */
#define KANTTS_LOSS_MAX_TERMS 5
typedef struct kantts_loss_term {
  const float* pred;
  const void* target;
  const int64_t* lens;
  float* grad;
  int32_t B, T, C, target_log1p;
} kantts_loss_term;
int kantts_masked_l1_many(const kantts_loss_term* terms, int nterms, float* losses, void* stream);
/* x_k *= *scale_dev for up to KANTTS_ELOSS_MAX_TERMS tensors in one launch (the backward of the call above and of
 * kantts_elem_loss_many: the upstream gradient of a loss sum is a device scalar). */
int kantts_scale_many(float* const* x, const long long* n, int count, const float* scale_dev, void* stream);

/* [round 4] The mean-reduced element losses of a GAN step in ONE launch per criterion (kantts/train/loss.py:108-256: the
 * feature-matching loss alone is ~48 l1_loss calls per step -- one per discriminator layer --, the adversarial criteria 8 /
 * 16 mse_loss calls; each was a zero-fill, a reduction launch and an addition).  Term k adds into losses[out]:
 *   mode 0: scale * sum |a - b|      grad (optional) = scale * sign(a - b)
 *   mode 1: scale * sum (a - target)^2      grad = 2 * scale * (a - target)
 * exactly as kantts_elem_loss does for one term.  terms: HOST array of nterms <= KANTTS_ELOSS_MAX_TERMS; losses: device
 * accumulators, zeroed by the caller. */
#define KANTTS_ELOSS_MAX_TERMS 64
typedef struct kantts_eloss_term {
  const float* a;
  const float* b;  /* mode 0 only */
  float* grad;     /* may be NULL */
  long long n;
  float target, scale;
  int32_t mode, out;
} kantts_eloss_term;
int kantts_elem_loss_many(const kantts_eloss_term* terms, int nterms, float* losses, void* stream);

/* ------------------------------------------------------------------------------------------
 * weight_norm reparametrisation w = g * v / ||v|| per output row (torch.nn.utils.weight_norm, dim=0):
 * kantts/models/hifigan/layers.py:29,67,105,139, hifigan.py:224,332.  v,w,dw,dv: (rows, cols); g,dg: (rows). */
int kantts_weight_norm_fwd(const float* v, const float* g, float* w, int rows, int cols, void* stream);
int kantts_weight_norm_bwd(const float* dw, const float* v, const float* g, float* dv, float* dg, int rows,
                           int cols, void* stream);

/* The same reparametrisation writing / reading the weight through strides: element (row r, input channel ci, tap k) at
 * r*rs + ci*cs + k*ks (tap-major (K, Cout, Cin_g): rs = Cin_g, cs = 1, ks = Cout*Cin_g); v / dv are (rows, cin, K). */
int kantts_weight_norm_strided_fwd(const float* v, const float* g, float* w, int rows, int cin, int K, long long rs,
                                   long long cs, long long ks, void* stream);
int kantts_weight_norm_strided_bwd(const float* dw, const float* v, const float* g, float* dv, float* dg, int rows,
                                   int cin, int K, long long rs, long long cs, long long ks, void* stream);

/* Tap-major weight norm with the bf16 operand images of kantts_cconv_launch in the same pass (v (rows, cin, K)):
 * w (K, rows, cin) fp32; wf_bf16 (K, rows, cin) = bf16(w) (optional); wd_bf16 (K, groups, cin, rows / groups) = the
 * per-tap transpose inside each group, the weight of the input-gradient contraction (optional). */
int kantts_weight_norm_tap_images(const float* v, const float* g, float* w, void* wf_bf16, void* wd_bf16, int rows, int cin,
                                  int K, int groups, void* stream);

/* The same for EVERY weight-normed convolution of a network in ONE launch (the network's parameters live in one flat
 * fp32 arena; images are rebuilt once per optimizer step, not once per layer and forward pass).  Table entry (device
 * memory): v at flat + v_off (rows, cin, K), g at flat + g_off (rows); outputs at w + w_off (K, rows, cin) fp32,
 * wf_bf16 + wf_off (K, rows, cin) and wd_bf16 + wd_off (K, groups, cin, rows / groups) -- wf_off / wd_off < 0: that layer
 * has no bf16 images; row0 = number of 8-row TILES, ceil(rows / 8) each, of all earlier entries (the launch has one
 * workgroup per tile; entries are found by bisection over row0); total_tiles = their sum.  Offsets of the bf16 images must
 * be multiples of 8 elements.  kantts/models/hifigan/layers.py:29,67, hifigan.py:224,332. */
typedef struct kantts_wn_desc {
  int64_t v_off, g_off, w_off, wf_off, wd_off;
  int32_t rows, cin, K, groups, row0, pad_;
} kantts_wn_desc;
int kantts_weight_norm_table(const float* flat, float* w, void* wf_bf16, void* wd_bf16, const kantts_wn_desc* table_dev,
                             int ndesc, int total_tiles, void* stream);

/* Backward of the same reparametrisation for up to KANTTS_WN_BWD_MAX layers of one network in one launch: layer l of the
 * launch is table entry desc[l], its tap-major weight gradient (K, rows, cin) fp32 is dw[l]; dv (rows, cin, K) and dg
 * (rows) are written into the network's flat GRADIENT arena at the entry's v_off / g_off (the parameters' own offsets).
 * tile0[l] = 8-row tiles of the layers before l in THIS launch (tile0[nl] = their total = the grid). */
#define KANTTS_WN_BWD_MAX 64
typedef struct kantts_wn_bwd_args {
  const float* dw[KANTTS_WN_BWD_MAX];
  int32_t desc[KANTTS_WN_BWD_MAX];
  int32_t tile0[KANTTS_WN_BWD_MAX + 1];
  int32_t nl;
} kantts_wn_bwd_args;
int kantts_weight_norm_table_bwd(const float* flat, float* grad_flat, const kantts_wn_desc* table_dev,
                                 const kantts_wn_bwd_args* args, void* stream);

/* y = sin(x) + x and its backward dx = dy * (cos(x) + 1)  (kantts/models/hifigan/hifigan.py:157) */
int kantts_sinadd_fwd(const float* x, float* y, long long n, void* stream);
int kantts_sinadd_bwd(const float* dy, const float* x, float* dx, long long n, void* stream);

/* Channels-last 1-D convolution with an LDS-resident input window (csrc/conv_win.hip): the im2col-free
 * kernel behind Conv1d / CausalConv1d (kantts/models/hifigan/layers.py:15-91) and their input gradients.
 *   for phase in [0, phases), m in [0, ceil((Tdst - phase) / phases)):      d = m*phases + phase
 *     out[b,d,n] = post( bias[n] + sum_{k : u_k % in_div == 0} sum_c pre(in[b, m*in_mul + u_k/in_div, g*CR + c]) * w[k][n][c] )
 *     u_k = in_add + phase + k*in_kstep;  taps whose source token falls outside [0, Tsrc) contribute 0;
 *     g = n / NG is the group of output channel n (Ntot = groups*NG outputs, Cin_tot = groups*CR inputs).
 *   pre(v)  : LeakyReLU(in_slope) when in_act, then v *= (in_gate > 0 ? 1 : in_gate_slope) when in_gate
 *   post(v) : LeakyReLU(out_slope) when out_act, + res, then *= (out_gate > 0 ? 1 : out_gate_slope)
 * forward : in_mul = stride, in_add = -pad, in_kstep = dilation, in_div = 1, phases = 1
 * dgrad   : in = dy, in_mul = 1, in_add = pad, in_kstep = -dilation, in_div = phases = stride
 * w is tap-major (K, Ntot, CR) fp32.  CR % 4 != 0 (1/2-channel layers) runs a direct one-thread-per-output kernel;
 * otherwise KANTTS_E_UNSUPPORTED when pointers are not 16-byte aligned or K > 64 -- callers then use
 * kantts_gemm_seg_launch. */
typedef struct {
  const float* in;
  const float* in_gate;
  const float* w;
  float* out;
  const float* bias;
  const float* res;
  const float* out_gate;
  int B, Tsrc, Tdst, Cin_tot, Ntot, CR, NG, groups, K;
  int in_mul, in_add, in_kstep, in_div, phases;
  int inner; /* folded axis between time and channels (MPD period): in is (B, Tsrc, inner, Cin_tot), out (B, Tdst, inner, Ntot) */
  int up;    /* > 1: `in` is read through a nearest-neighbour upsampling: token u of the rule above is source token u / up */
  float in_slope;
  int in_act;
  float in_gate_slope;
  float out_slope;
  int out_act;
  float out_gate_slope;
  int precision; /* 0 fp32 MFMA, 1 bf16 MFMA (fp32 accumulate) */
} kantts_conv_args;
int kantts_conv_win_launch(const kantts_conv_args* args, void* stream);

/* Weight / bias gradient of the same convolutions (csrc/conv_wgrad.hip), accumulated (+=) with fp32 atomics:
 *   dw[k][n][c] += sum_{b,p,q} gate(dy[b,q,p,n]) * act(x[b, q*stride + k*dil - pad, p, g*CR + c]),   g = n / NG
 *   db[n]       += sum_{b,p,q} gate(dy[b,q,p,n])                                   (db may be NULL)
 * x is (B, Tsrc, inner, Cin_tot), dy / dy_gate are (B, Tdst, inner, Ntot), dw is tap-major (K, Ntot, CR).
 * gate(v) = v * (dy_gate > 0 ? 1 : dy_gate_slope) when dy_gate is given; act = LeakyReLU(x_slope) when x_act.
 * CR or NG not a multiple of 4 runs a direct kernel (up must be 1); otherwise KANTTS_E_UNSUPPORTED when a
 * pointer is not 16-byte aligned. */
typedef struct {
  const float* x;
  const float* dy;
  const float* dy_gate;
  float* dw;
  float* db;
  int B, Tsrc, Tdst, Cin_tot, Ntot, CR, NG, groups, K;
  int stride, dil, pad, inner;
  int up; /* > 1: x is read through a nearest-neighbour upsampling (virtual token u = source token u / up) */
  float x_slope;
  int x_act;
  float dy_gate_slope;
  int precision; /* 0 fp32 MFMA, 1 bf16 MFMA (fp32 accumulate) */
} kantts_convw_args;
int kantts_conv_wgrad_launch(const kantts_convw_args* args, void* stream);

/* Single-input-channel convolutions (first layer of the HiFi-GAN discriminators; csrc/conv_c1.hip), fp32 streaming kernels.
 * x / dx are (B, Tsrc, inner), y (= dy for the gradients) and gate are (B, Tdst, inner, Cout), w / dw are (Cout, K) row-major.
 *   mode 0  y[b,q,p,n]   = act( bias[n] + sum_k x[b, q*stride + k*dil - pad, p] * w[n][k] )      act = LeakyReLU(out_slope) when out_act
 *   mode 1  dx[b,t,p]   += sum_k sum_n gate(y[b,q,p,n]) * w[n][k]   at t = q*stride + k*dil - pad   (dx must start at zero)
 *   mode 2  dw[n][k]    += sum_{b,q,p} gate(y[b,q,p,n]) * x[b, q*stride + k*dil - pad, p];   db[n] += sum gate(y)   (db may be NULL)
 * gate(v) = v * (gate > 0 ? 1 : gate_slope) when gate is given.
 * KANTTS_E_UNSUPPORTED unless K <= 16, Cout % 4 == 0, Cout <= 256 and 256 % Cout == 0. */
typedef struct {
  const float* x;
  float* dx;
  float* y;
  const float* gate;
  const float* w;
  const float* bias;
  float* dw;
  float* db;
  int B, Tsrc, Tdst, Cout, K, stride, dil, pad, inner;
  float out_slope;
  int out_act;
  float gate_slope;
  void* y_bf16; /* [round 4] mode 0, optional: the bf16 image of y (after the output activation) for the convolution that
                   consumes it -- the second layer of every sub-discriminator otherwise starts with a cast pass over y */
} kantts_conv_c1_args;
int kantts_conv_c1_launch(const kantts_conv_c1_args* args, int mode, void* stream);

/* [round 4] Convolutions with ONE output channel (csrc/conv_n1.hip): the conv_post layers of the generator
 * (kantts/models/hifigan/hifigan.py:178-180: Conv1d(32 -> 1, k = 7) after leaky_relu(x, 0.01)), of the period discriminators
 * (:238-262, Conv2d(1024 -> 1, (3, 1))) and of the scale discriminators (:373-402).  A dot product per output position:
 *   mode 0  y[b,q,p] = bias + sum_k sum_c w[k][c] act(x[b, q*stride + k*dil - pad, p, c])
 *   mode 1  dx[b,t,p,c] = act'(x) sum_k w[k][c] dy[b, (t + pad - k*dil) / stride, p]   (y holds dy; dx fully written)
 *   mode 2  dw[k][c] += sum act(x) dy,  db[0] += sum dy   (y holds dy; dw / db zeroed or accumulating, caller's choice)
 * x (B, Tsrc, inner, Cin) fp32, y (B, Tdst, inner); the weight and its gradient are addressed as w[k * w_ks + c * w_cs]
 * (tap-major (K, 1, Cin): w_ks = Cin, w_cs = 1; parameter layout (1, Cin, K): w_ks = 1, w_cs = K).  act = LeakyReLU(in_slope)
 * when in_act.  KANTTS_E_UNSUPPORTED unless Cin is a power of two in [4, 1024], K <= 8 and (Cin / 4 / min(64, Cin / 4)) * K <= 16. */
typedef struct {
  const float* x;
  float* dx;
  float* y;
  const float* w;
  const float* bias;
  float* dw;
  float* db;
  int B, Tsrc, Tdst, Cin, K, stride, dil, pad, inner;
  int w_ks, w_cs;
  float in_slope;
  int in_act;
} kantts_conv_n1_args;
int kantts_conv_n1_launch(const kantts_conv_n1_args* args, int mode, void* stream);

/* Free-running inference steps.
 * kantts_attn_decode: one query per sequence (decoder position `step`) against rows [lo, hi] of a (B, L, .) K/V
 *   buffer; the interval is the training kernels' function of (mode, step, len, bw) (HybridAttentionDecoder.infer,
 *   kantts/models/sambert/kantts_sambert.py:208-253; K/V state sambert/__init__.py:212-258).  q / o hold B rows.
 *   bw_seq (optional, B entries) gives every sequence its own band width: the reference infers one utterance at a
 *   time with the band of THAT utterance, which a batch can only reproduce per sequence.
 * kantts_lstm_cell: gates (B, 4H) in PyTorch order [i|f|g|o] -> h, c (VarRnnARPredictor.infer, adaptors.py:67-83). */
int kantts_attn_decode(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, float* o, int ldo,
                       const int32_t* lens, const int32_t* bw_seq, int B, int H, int L, int d_head, int mode, int step,
                       int bw, void* stream);
int kantts_lstm_cell(const float* gates, const float* c_prev, float* h_out, float* c_out, int B, int H, void* stream);

/* Monotonic alignment search, width 1 (csrc/mas.hip): replaces the host round trip of binarize_attention_parallel
 * (kantts/models/sambert/kantts_sambert.py:752-764 -> numba b_mas, alignment.py:32-71).  attn / opt are (B, To_max, Ti_max)
 * (the reference's (B, 1, mel, text) with the singleton squeezed); opt receives the 0/1 hard alignment of the valid
 * (out_lens[b] x in_lens[b]) corner and zeros elsewhere; workspace: B*To_max*Ti_max*4 bytes (float logs, or byte
 * back-pointers on the wide-map fallback). */
int kantts_mas_width1(const float* attn, const int32_t* in_lens, const int32_t* out_lens, float* opt, void* workspace,
                      int B, int To_max, int Ti_max, void* stream);

/* Alignment attention of the MAS path (csrc/mas.hip): the part of ConvAttention.forward after the projections
 * (kantts/models/sambert/attention.py:103-125).  q (B,T1,C) mel-side / k (B,T2,C) text-side encodings, channels last;
 * prior (B,T1,T2) or NULL; in_lens[b] = unpadded text length (positions >= it are masked out of `soft`).
 * logprob = attn_logprob, soft = attn of the reference, both (B,T1,T2) (= (B,1,T1,T2)).  T2 <= 1024.
 * bwd: d_logprob / d_soft may be NULL; g_ws: B*T1*T2 floats of workspace; dq / dk are overwritten. */
int kantts_align_attn_fwd(const float* q, const float* k, const float* prior, const int32_t* in_lens, float* logprob,
                          float* soft, int B, int T1, int T2, int C, void* stream);
int kantts_align_attn_bwd(const float* q, const float* k, const float* prior, const float* logprob, const float* soft,
                          const float* d_logprob, const float* d_soft, float* g_ws, float* dq, float* dk, int B, int T1,
                          int T2, int C, void* stream);

/* ------------------------------------------------------------------------------------------
 * Contractions with bf16 operands in HBM (round 2; csrc/gemm_bf16.hip).  Same reference call sites as the segmented
 * GEMM above (nn.Linear / Conv1d k=1,3 forward, input gradient, weight gradient:
 * kantts/models/sambert/__init__.py:82,102,115-127,217,242,287,297; fsmn.py:14-29; kantts_sambert.py:173-174),
 * used when the numerics mode is bf16: weights come from the parameter arena's bf16 shadow, activations that only feed
 * contractions are stored bf16 by their producers.  All extents / leading dimensions are multiples of 8 elements and
 * all base pointers 16-byte aligned (KANTTS_E_UNSUPPORTED otherwise: the caller falls back to the segmented GEMM).
 *
 * kantts_bgemm_nt:  C[i][j] = epilogue( sum_s sum_k A_s[tok(i) + a_shift_s][k] * B_s(j, k) )
 *   A_s: (M, klen_s) rows of bf16 (or fp32 when a_f32: rounded to bf16 once per tile), row stride lda.
 *   B_s: b_kn = 0: (N, klen_s) k-contiguous, row stride ldb (forward: the weight);
 *        b_kn = 1: (klen_s, N) n-contiguous, row stride ldb (input gradient through the same weight buffer).
 *   a_shift: conv tap -- row i reads token i + a_shift of its own sequence of T tokens, zero outside [0, T).
 *   epilogue: v = (acc + bias[j] + bias2[j]) * alpha; relu; dropout(drop_p; element i*N + j of drop_seed + *seed_dev);
 *             v += res[i][j] (fp32); v = gate[i][j] > 0 ? v : 0 (ReLU backward on a saved activation); rowmask[i] -> 0;
 *             stored as bf16 (c_bf16) or fp32.
 *   a_drop_p > 0 (a_f32 only): A is an incoming gradient through an epilogue dropout of the forward pass; element
 *             (i, k) is multiplied by the keep-scale regenerated from (a_drop_seed + *seed_dev, i * a_drop_ld + k).
 */
#define KANTTS_BGEMM_MAX_SEG 12
typedef struct kantts_bgemm_seg {
  const void* a;
  const void* b;
  int64_t lda, ldb;
  int32_t klen;
  int32_t a_shift;
} kantts_bgemm_seg;

typedef struct kantts_bgemm_args {
  kantts_bgemm_seg seg[KANTTS_BGEMM_MAX_SEG];
  int32_t nseg, M, N, T;
  int32_t a_f32, b_kn;
  void* c;
  int64_t ldc;
  int32_t c_bf16;
  int32_t relu;
  const float* bias;
  const float* bias2;
  float alpha;
  float drop_p;
  uint64_t drop_seed;
  const uint64_t* seed_dev;
  const float* res;
  int64_t ldr;
  const void* gate;
  int64_t ldg;
  int32_t gate_bf16;
  float a_drop_p;
  uint64_t a_drop_seed;
  int64_t a_drop_ld;
  const uint8_t* rowmask;
  /* optional: LayerNorm(128) of the output rows in the epilogue (N == 128, fp32 c): the pre-LN sub-layer that consumes c
   * (kantts/models/sambert/__init__.py:130-131, 198) then needs no launch of its own.  ln_out (M,128) bf16 / fp32,
   * ln_mean / ln_rstd (M) as kantts_ln128_fwd writes them (its backward is unchanged). */
  const float* ln_gamma;
  const float* ln_beta;
  void* ln_out;
  int32_t ln_out_bf16;
  float ln_eps;
  float* ln_mean;
  float* ln_rstd;
} kantts_bgemm_args;
int kantts_bgemm_nt(const kantts_bgemm_args* args, void* stream);

/* kantts_bgemm_nt with the BACKWARD of a LayerNorm(128) as its epilogue.  The contraction's result (N == 128, b_kn) is
 * the gradient dy of a LayerNorm's output -- the input gradient of the projection that consumes the normalised rows of a
 * pre-LN sub-layer (kantts/models/sambert/__init__.py:63-64, 89-91, 198-199) -- and what the launch writes is what
 * kantts_ln128_bwd_rows makes of it: dx (M,128) fp32 (+ dres; rows of zero_rows written as 0) and the dgamma / dbeta
 * accumulations.  dy is rounded to bf16 first when args->c_bf16 is set (the value the two-launch form passes through
 * memory) and is itself stored only if args->c is not NULL.  args->ln_out / gate / relu / drop_p must be unset. */
typedef struct kantts_lnbwd_args {
  const float* x;        /* (M,128) the LayerNorm's input */
  const float* gamma;    /* (128) */
  const float* mean;     /* (M) as kantts_ln128_fwd wrote them */
  const float* rstd;
  const float* dres;     /* optional (M,128): gradient of the residual branch that by-passes the normalisation */
  const uint8_t* zero_rows; /* optional (M) */
  float* dx;             /* (M,128) */
  float* dgamma_accum;   /* (128), += */
  float* dbeta_accum;
  float* part_rows;      /* [round 5] optional (ceil(M / 32), 256): when set, workgroup w writes its 128 dgamma + 128 dbeta sums
                          * to row w INSTEAD of adding them to the accumulators (256 atomics per workgroup on the same 256
                          * addresses serialise in L2); the caller adds the rows: kantts_rows_sum_accum */
} kantts_lnbwd_args;
int kantts_bgemm_nt_lnbwd(const kantts_bgemm_args* args, const kantts_lnbwd_args* ln, void* stream);

/* kantts_bgemm_tn:  c[n*c_ns + k*c_ks + tap*c_ts] += alpha * sum_m A[m][n] * B[tok(m) + shift0 + tap*shift_step][k]
 *                   db[n] += alpha * sum_m A[m][n]                                   (optional)
 * Weight / bias gradients: A = gradient of the layer output (M, N), B = layer input (M, K), tokens m are the reduction
 * axis; fp32 atomics into a pre-zeroed gradient that may already be in the parameter's own layout (strides c_*).
 * A / B are bf16 or fp32 (a_f32 / b_f32); a_drop_* as above with the logical index m*N + n.  slices = 0 lets the
 * library choose the token split. */
typedef struct kantts_bgemm_tn_args {
  const void* a;
  const void* b;
  int64_t lda, ldb;
  int32_t M, N, K, T;
  int32_t a_f32, b_f32;
  int32_t ntaps, shift0, shift_step, slices;
  float* c;
  int64_t c_ns, c_ks, c_ts;
  float* db;
  float alpha;
  float a_drop_p;
  uint64_t a_drop_seed;
  const uint64_t* seed_dev;
} kantts_bgemm_tn_args;
int kantts_bgemm_tn(const kantts_bgemm_tn_args* args, void* stream);

/* Up to KANTTS_TN_MAX_GROUP weight gradients of ONE shape (shape / dtypes / strides / alpha / dropout probability from
 * *shape; operand, output, bias-gradient pointers and dropout seeds per problem, arrays in HOST memory) in a single
 * launch.  The host layer defers the weight gradients of a backward pass and issues them grouped by shape. */
#define KANTTS_TN_MAX_GROUP 16
int kantts_bgemm_tn_grouped(const kantts_bgemm_tn_args* shape, int nprob, const void* const* a_host, const void* const* b_host,
                            float* const* c_host, float* const* db_host, const uint64_t* a_drop_seed_host, void* stream);

/* fp32 -> bf16 (round to nearest even) over n elements (n % 8 == 0, 16-byte aligned): the parameter arena's shadow. */
int kantts_cast_f32_bf16(const float* src, void* dst_bf16, long long n, void* stream);

/* Conv1d weights (N, Cin, KT) fp32 at src + src_off -> tap-major (KT, N, Cin) bf16 at dst + dst_off, one table entry
 * per weight (table in device memory), one launch for all of them (kantts/models/sambert/__init__.py:115-127). */
typedef struct kantts_tapmajor_desc {
  int64_t src_off, dst_off;
  int32_t N, Cin, KT, pad_;
} kantts_tapmajor_desc;
int kantts_tapmajor_bf16(const float* src, void* dst_bf16, const kantts_tapmajor_desc* table_dev, int ndesc,
                         int blocks_per_desc, void* stream);

/* dz = (y > 0) ? dy * scale : 0 over n elements (n % 8 == 0), bf16 output: ReLU (+ dropout: scale = 1/(1-p), y is the
 * post-dropout activation) backward of a fused linear whose output was stored for the gate
 * (kantts/models/sambert/__init__.py:40-49,141-142). */
int kantts_relu_gate_bf16(const void* dy, int dy_bf16, const void* y, int y_bf16, void* dz_bf16, float scale, long long n,
                          void* stream);

/* kantts_ffn_pair: the two contractions of a position-wise feed-forward block in one launch
 * (kantts/models/sambert/__init__.py:134-149; forward and, with `gate`, the input-gradient pass).
 *   phase 1   t[m][f] = epi1( sum_tap x[m + tap - pad][:] . w1[tap][f][:] )          m < M, f < F, reduction K1 = 128
 *             forward  (gate == NULL): epi1 = rowmask1( dropout_{drop1}( relu?( . + bias1 ) ) )
 *             backward (gate != NULL): epi1 = gate[m][f] > 0 ? alpha1 * . : 0                      (KT == 1 only)
 *   phase 2   y[m][n] = rowmask2( dropout_{drop2}( t[m][:] . w2[n][:] + bias2 ) + res[m][n] )      n < N = 128
 * x: (M, 128) bf16 or fp32 (x_f32; fp32 rows may carry a regenerated dropout xdrop_* with index m*128 + k and are zeroed
 * where xrowmask[m] != 0).  w1: the (KT*F, 128) matrix [tap*F + f][k], w2: the (128, F) matrix [n][f], both bf16 in the
 * FRAGMENT-MAJOR layout of kantts_fragmajor_bf16 -- the backward pass hands in the transposed weights (w1 := W2^T as
 * (F, 128), w2 := W1^T as (128, F)).  t_out (optional): bf16 (M, F) row-major, the intermediate the weight gradients need.
 * Taps do not cross sequence boundaries: rows are B sequences of T tokens.  Dropout indices: m*F + f (drop1), m*128 + n
 * (drop2); seeds are offset by *seed_dev (graph replay).  F = 1024.  KT2 = 3 (backward form only, M % T == 0): phase 2
 * sums three taps of the intermediate, w2 = three (128, F) images -- the input gradient of a k = 3 first convolution.  Returns KANTTS_E_UNSUPPORTED for other shapes: the
 * caller falls back to two kantts_bgemm_nt launches. */
typedef struct kantts_ffn_args {
  const void* x;
  int64_t ldx;
  int32_t x_f32;
  int32_t M, T, K1, F, N, KT, pad;
  const void* w1;
  const void* w2;
  const float* bias1;
  const float* bias2;
  int32_t relu;
  float alpha1;
  float drop1_p;
  float drop2_p;
  float xdrop_p;
  uint64_t drop1_seed;
  uint64_t drop2_seed;
  uint64_t xdrop_seed;
  const uint64_t* seed_dev;
  const void* gate;
  const uint8_t* rowmask1;
  const uint8_t* rowmask2;
  const uint8_t* xrowmask;
  void* t_out;
  const float* res;
  int64_t ldr;
  void* y;
  int64_t ldy;
  int32_t y_bf16;
  int32_t KT2, s2_first, s2_step; /* taps of phase 2 (0 / 1: none): y[m] = sum_t t[m + s2_first + t*s2_step] . w2[t]^T */
  /* optional (forward form, KT2 <= 1): LayerNorm(128) of the output rows in the epilogue, as kantts_bgemm_args ln_* --
   * the block's output feeds the next block's pre-LN attention sub-layer (kantts/models/sambert/__init__.py:63, 198) */
  const float* ln_gamma;
  const float* ln_beta;
  void* ln_out;
  int32_t ln_out_bf16;
  float ln_eps;
  float* ln_mean;
  float* ln_rstd;
} kantts_ffn_args;
int kantts_ffn_pair(const kantts_ffn_args* args, void* stream);

/* [round 5] One PNCA decoder block forward as ONE launch (csrc/pnca_block.hip; reference
 * kantts/models/sambert/__init__.py:212-348 with the feed-forward of :134-149; band masks kantts_sambert.py:135-166):
 *   [q|k|v] = xn Wqkv^T + bqkv;  ox / oh = x-band / memory-band attention (csrc/attn.hip conventions, 8 heads x 16);
 *   y1 = rowmask(dropout_fc(ox Wfcx^T + oh Wfch^T + bfcx + bfch) + x);  xn1 = LN1(y1);
 *   hid = rowmask(dropout_1(relu(xn1 W1^T + b1)));  out = rowmask(dropout_2(hid W2^T + b2) + y1);  ln2_out = LN2(out).
 * x (M, 128) fp32 block input and xn (M, 128) bf16 its LayerNorm (M = B * L rows); hkv: this block's memory K | V rows,
 * fp32, row pitch ldh >= 256 floats; wqkv (384 x 128), wfcx / wfch (128 x 128), w1 (1024 x 128), w2 (128 x 1024):
 * fragment-major bf16 images (kantts_fragmajor_bf16).  Written for the backward pass (each optional unless noted): qkv
 * (M, 384) fp32, ox / oh (M, 128) fp32 (required), lse_x / lse_h (B, 8, L) (required), y1 fp32, xn1 bf16 + mean1 / rstd1,
 * hid (M, 1024) bf16; out (M, 128) fp32 (required); ln2_* as the ln_* fields of kantts_ffn_args.  Dropout seeds are offset
 * by *seed_dev; indices are those of the separate launches (kantts_pnca_attn_fwd, kantts_bgemm_nt, kantts_ffn_pair), so the
 * fused launch and the chain draw the same masks.  bw_dev (device scalar) overrides bw_x / bw_h.  Band widths above 16 are
 * not supported: KANTTS_E_UNSUPPORTED when known on the host, NaN outputs when only the device knows. */
typedef struct kantts_pnca_block_args {
  const float* x;
  const void* xn;
  const float* hkv;
  int64_t ldh;
  int32_t B, L, H, C, F;
  const int32_t* lens;
  const int32_t* bw_dev;
  int32_t bw_x, bw_h;
  const uint8_t* rowmask;
  const void* wqkv;
  const float* bqkv;
  const void* wfcx;
  const void* wfch;
  const float* bfcx;
  const float* bfch;
  const float* ln1_gamma;
  const float* ln1_beta;
  float ln1_eps;
  const void* w1;
  const void* w2;
  const float* bias1;
  const float* bias2;
  float att_p, fc_p, drop1_p, drop2_p;
  uint64_t seed_x, seed_h, fc_seed, drop1_seed, drop2_seed;
  const uint64_t* seed_dev;
  float* qkv;
  float* ox;
  float* oh;
  float* lse_x;
  float* lse_h;
  float* y1;
  void* xn1;
  float* mean1;
  float* rstd1;
  void* hid;
  float* out;
  const float* ln2_gamma;
  const float* ln2_beta;
  void* ln2_out;
  int32_t ln2_out_bf16;
  float ln2_eps;
  float* ln2_mean;
  float* ln2_rstd;
} kantts_pnca_block_args;
int kantts_pnca_block_fwd(const kantts_pnca_block_args* args, void* stream);

/* [round 5] The row-local half of a PNCA block's backward as one launch (csrc/pnca_block.hip): kantts_ffn_pair (backward
 * form) + kantts_ln128_bwd_rows + the two input-gradient launches of the output projection, same arithmetic
 * (kantts/models/sambert/__init__.py:134-149, 286-306 differentiated):
 *   dz = gate_{hid > 0}(dropout_2(dy) W2) * alpha1 (bf16, (M, 1024); the weight gradient of W1 reads it);
 *   dh = dz W1 (rounded to bf16);  g1 = rowmask(LN1'(dh; y1, mean1, rstd1, gamma1) + dy) (fp32 (M, 128));
 *   d_ox = dropout_fc(g1) Wfcx, d_oh = dropout_fc(g1) Wfch (fp32 (M, 128)).  The gradients of LN1's gamma / beta leave as
 *   PARTIAL ROWS: workgroup w writes 128 dgamma sums and 128 dbeta sums of its 32 rows to ws[w*256 ..]; the caller adds the
 *   rows (kantts_rows_sum_accum) -- same-address atomics from 204 workgroups cost half of the launch.
 * dy: gradient of the block output (rows of rowmask are read as zero).  wt2 = W2^T (1024 x 128), wt1 = W1^T (128 x 1024),
 * wfcxT / wfchT = Wfcx^T / Wfch^T (128 x 128): fragment-major bf16 images.  Dropout indices as in the forward launches. */
typedef struct kantts_pnca_block_bwd_args {
  const float* dy;
  const void* hid;
  const float* y1;
  const float* mean1;
  const float* rstd1;
  const float* ln1_gamma;
  const uint8_t* rowmask;
  int32_t M, C, F;
  const void* wt2;
  const void* wt1;
  const void* wfcxT;
  const void* wfchT;
  float alpha1, drop2_p, fc_p;
  uint64_t drop2_seed, fc_seed;
  const uint64_t* seed_dev;
  void* dz;
  float* g1;
  float* d_ox;
  float* d_oh;
  float* ws;            /* caller-owned, >= kantts_pnca_block_bwd_ws_floats(M) floats, 16-byte aligned: ceil(M / 32) rows */
  long long ws_floats;  /* of [128 dgamma | 128 dbeta] partial sums, every element written */
} kantts_pnca_block_bwd_args;
int kantts_pnca_block_bwd(const kantts_pnca_block_bwd_args* args, void* stream);
long long kantts_pnca_block_bwd_ws_floats(int M);
/* [round 5] The cross-row half of a PNCA block's backward as one launch (csrc/pnca_block.hip): kantts_pnca_attn_bwd +
 * kantts_bgemm_nt_lnbwd of the chain, same arithmetic and conventions (kantts/models/sambert/__init__.py:212-306
 * differentiated; band masks kantts_sambert.py:135-166):
 *   dqkv (M, 384) fp32 = [query | key | value] gradients of both attention bands (the memory band's query gradient added);
 *   dhkv: gradient of this block's memory K | V rows, row pitch lddh;
 *   dx (M, 128) = rowmask_{zero_rows}(LN0'(bf16(dqkv Wqkv); x, mean0, rstd0, gamma0) + dres);
 *   ws: ceil(M / 32) partial rows of [128 dgamma0 | 128 dbeta0] (kantts_rows_sum_many).
 * wqkvT: fragment-major bf16 image of Wqkv^T (128 x 384).  Band widths above 16: KANTTS_E_UNSUPPORTED / NaN as the forward. */
typedef struct kantts_pnca_attn_bwd_args {
  const float* qkv;
  const float* hkv;
  int64_t ldh;
  const float* ox;
  const float* oh;
  const float* d_ox;
  const float* d_oh;
  const float* lse_x;
  const float* lse_h;
  int32_t B, L, H, C;
  const int32_t* lens;
  const int32_t* bw_dev;
  int32_t bw_x, bw_h;
  float att_p;
  uint64_t seed_x, seed_h;
  const uint64_t* seed_dev;
  const void* wqkvT;
  const float* x;
  const float* mean0;
  const float* rstd0;
  const float* ln0_gamma;
  const float* dres;
  const uint8_t* zero_rows;
  float* dqkv;
  float* dhkv;
  int64_t lddh;
  float* dx;
  float* ws;
  long long ws_floats;
} kantts_pnca_attn_bwd_args;
int kantts_pnca_attn_qkv_bwd(const kantts_pnca_attn_bwd_args* args, void* stream);

/* For each of n problems: dst0[c] += sum_r src[r*cols + c] for c < split, dst1[c - split] += ... for c >= split (fixed
 * summation order) -- the partial rows of kantts_pnca_block_bwd / kantts_bgemm_nt_lnbwd, many of them in one launch. */
#define KANTTS_ROWSUM_MAX 32
typedef struct kantts_rowsum_args {
  int32_t n, cols, split;
  int32_t rows[KANTTS_ROWSUM_MAX];
  const float* src[KANTTS_ROWSUM_MAX];
  float* dst0[KANTTS_ROWSUM_MAX];
  float* dst1[KANTTS_ROWSUM_MAX];
} kantts_rowsum_args;
int kantts_rows_sum_many(const kantts_rowsum_args* args, void* stream);

/* Fragment-major bf16 images of weight matrices, a table of them in one launch (the parameter arena's per-step refresh).
 * Entry: the (R, K) matrix with element (r, k) = src[src_off + r*sr + k*sk] (fp32; any orientation of the master weight)
 * is written to dst + dst_off so that every 16 x 32 block (r/16, k/32) is 1 KB in the order one A-operand load of
 * v_mfma_f32_16x16x32_bf16 reads it: dst[((r/16)*(K/32) + k/32)*512 + (((k%32)/8)*16 + r%16)*8 + k%8].
 * R % 16 == 0, K % 32 == 0. */
typedef struct kantts_fragmajor_desc {
  int64_t src_off, dst_off;
  int64_t sr, sk;
  int32_t R, K;
} kantts_fragmajor_desc;
int kantts_fragmajor_bf16(const float* src, void* dst_bf16, const kantts_fragmajor_desc* table_dev, int ndesc,
                          int blocks_per_desc, void* stream);

/* nn.LayerNorm(128, eps) forward / backward, 16 lanes per row; the output (and the incoming gradient) may be bf16 because
 * LayerNorm outputs only feed contractions (kantts/models/sambert/__init__.py:63,130,198; kantts_sambert.py:58,128). */
int kantts_ln128_fwd(const float* x, const float* gamma, const float* beta, void* y, int y_bf16, float* mean, float* rstd,
                     int M, float eps, void* stream);
/* dres (optional, fp32 (M,128)): gradient of a residual branch taken from the same x; dx = LN-gradient + dres in one pass. */
int kantts_ln128_bwd(const void* dy, int dy_bf16, const float* x, const float* gamma, const float* mean, const float* rstd,
                     const float* dres, float* dx, float* dgamma_accum, float* dbeta_accum, int M, void* stream);
/* ... and zero_rows (optional, uint8 (M)): rows of dx written as 0.  A pre-LN sub-layer's input x is the output of the
 * previous sub-layer, which zeroes the padded rows of its output (kantts/models/sambert/__init__.py:177-178, 340-341) and
 * therefore of its incoming gradient: when this LayerNorm is the only consumer of x, that masking is this kernel's store
 * instead of a separate pass over the gradient. */
int kantts_ln128_bwd_rows(const void* dy, int dy_bf16, const float* x, const float* gamma, const float* mean,
                          const float* rstd, const float* dres, float* dx, float* dgamma_accum, float* dbeta_accum,
                          const unsigned char* zero_rows, int M, void* stream);

/* Backward of kantts_melspec_fwd's magnitude output (the reference's stft(), kantts/utils/audio_torch.py:8-31:
 * sqrt(clamp(re^2 + im^2, eps_power))): dwav_accum (B,T) += d loss / d wav given dmag (B, frames, n_fft/2+1).  Gradient
 * path of STFTLoss / MultiResolutionSTFTLoss (kantts/train/loss.py:312-441); zero where the clamp is active. */
int kantts_stft_mag_bwd(const float* wav, const float* dmag, int B, int T, int n_fft, int hop, int frames, int pad_mode,
                        const float* window, const float* twiddle, float eps_power, float* dwav_accum, void* stream);

/* out[0] = sum x^2 with a fixed summation order (bit-reproducible: data-parallel replicas must derive the same
 * clipping factor from the same all-reduced gradient).  workspace: >= 1025 floats, zero before the first call (word 0 is
 * a ticket counter the kernel resets itself).  kantts/train/trainer.py:997-1004 (clip_grad_norm_). */
int kantts_sumsq_det(const float* x, float* out, float* workspace, long long ws_floats, long long n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Free-running decode with the step index in device memory (csrc/decode.hip; BASELINE config 5): one decoder step is a
 * static launch sequence that a hipGraph replays per step.  `step_dev` (int32 device scalar) overrides `step` when set.
 *
 * kantts_pnca_decode_step -- MultiHeadPNCAAttention under update_x_state / update_h_state
 *   (kantts/models/sambert/__init__.py:217-306): qkv (B, 3*H*16) rows = this step's [q | k | v]; k / v are appended to
 *   xkv_cache (B, L, 2*H*16) at row `step`; ox / oh (B, H*16) = causal-band attention over the cache and look-ahead-band
 *   attention over the memory projections hkv (B, L, 2*H*16).  Padded queries (step >= lens[b]) give 0.
 * kantts_step_rows -- dst[b*dst_bs + step*dst_ss + e] = src[b*src_bs + step*src_ss + e], e < n  (memory[:, step, :] gather,
 *   output-frame scatter; kantts_sambert.py:589-603).
 * kantts_step_rowmask -- mask[b] = step >= lens[b]. */
int kantts_pnca_decode_step(const float* qkv, int ldq, float* xkv_cache, const float* hkv, float* ox, float* oh,
                            const int32_t* lens, const int32_t* bw_seq, int B, int H, int L, int d_head, int step,
                            const int32_t* step_dev, int bw, void* stream);
int kantts_step_rows(const float* src, float* dst, int B, int n, long long src_batch_stride, long long dst_batch_stride,
                     long long src_step_stride, long long dst_step_stride, int step, const int32_t* step_dev, void* stream);
int kantts_step_rowmask(const int32_t* lens, uint8_t* mask, int B, int step, const int32_t* step_dev, void* stream);

/* [round 4] HiFi-GAN multi-receptive-field fusion (kantts/models/hifigan/hifigan.py:160-176: xs += resblock(x) over the
 * num_kernels residual stacks of a stage, then xs / num_kernels): out = scale * sum_k xs[k] in one pass, optionally with the
 * bf16 image of LeakyReLU(out, slope) that the next convolution reads (act_bf16 may be NULL).  xs_host / outs_host: HOST
 * arrays of n <= 8 DEVICE pointers; numel % 4 == 0, 16-byte aligned buffers.
 * kantts_scale_to_many: its backward -- outs[k] = scale * g for every k (each branch its own gradient buffer). */
int kantts_mean_many(const float* const* xs_host, int n, float scale, float* out, void* act_bf16, float slope,
                     long long numel, void* stream);
int kantts_scale_to_many(const float* g, float scale, float* const* outs_host, int n, long long numel, void* stream);

/* ------------------------------------------------------------------------------------------
 * HiFi-GAN upsampling as an HBM stream (csrc/upsample.hip): CausalConvTranspose1d with kernel 2*S, stride S
 * (kantts/models/hifigan/layers.py:125-165, hifigan.py:67-80,160) in polyphase form on bf16 activations:
 *   out[(b*T + t)*S + r][co] = bias[co] + sum_{j=0,1} sum_ci lrelu(x[b*T + t - j][ci]) * w[ci][co][r + j*S]  (+ res)
 * x (B*T, Cin) bf16; wp (S*Cout, 2*Cin) bf16 = the weight with rows permuted for 16-byte stores: row mt*16 + rho,
 * mt = (r*(Cout/32) + cb)*2 + h, holds output channel cb*32 + (rho >> 2)*8 + h*4 + (rho & 3) of phase r; column j*Cin + ci
 * holds w[ci][co][r + j*S].  out / res (B*T*S, Cout) bf16 (out_bf16) or fp32.  in_slope = 1 for pre-activated input.
 * Shapes: (Cin, Cout, S) in {(128, 64, 2), (64, 32, 2)}; wider layers are contractions (kantts_bgemm_nt, two segments). */
int kantts_upsample_stream(const void* x_bf16, const void* wp_bf16, const float* bias, const void* res, void* out, int B,
                           int T, int Cin, int Cout, int S, float in_slope, int out_bf16, void* stream);
/* y = sin(x) + x (hifigan.py:157) and act = bf16(LeakyReLU(y, slope)) in one pass. */
int kantts_sinadd_lrelu_fwd(const float* x, float* y, void* act_bf16, float slope, long long n, void* stream);

/* y[i] = x[i] * keep_{p1,seed1}(i) * keep_{p2,seed2}(i) (+ res[i]): two stacked dropouts and a residual add in one pass
 * with regenerated masks -- FsmnEncoderV2 / MemoryBlockV2 (kantts/models/sambert/fsmn.py:66-70,114-121).  The backward
 * is the same call with x := dy, res := NULL.  n % 4 == 0, 16-byte aligned; seeds are offset by *seed_dev. */
int kantts_dropout2_add(const float* x, const float* res, float* y, long long n, float p1, uint64_t seed1, float p2,
                        uint64_t seed2, const uint64_t* seed_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * bf16 convolution contractions for the HiFi-GAN layers (csrc/cconv.hip, round 3): the same token rule as
 * kantts_conv_win_launch, but both operands are bf16 IN MEMORY (activations pre-activated by their producer, weights a
 * bf16 tap-major image) and are copied global -> LDS by `global_load_lds` (16 bytes per lane, no VGPR staging, no
 * conversion); a tile's rows run across batch items, padding / phase / group edges are lanes whose source address is a
 * 16-byte block of zeros.  Replaces Conv1d / CausalConv1d / the (k,1) Conv2d of the period discriminators / the
 * polyphase form of ConvTranspose1d and their input gradients (kantts/models/hifigan/layers.py:15-165,
 * hifigan.py:200-267,305-407) for channel counts that are multiples of 8.
 *   for phase in [0, phases), m in [0, ceil((Tdst - phase) / phases)):      d = m*phases + phase
 *     v[b,d,p,n] = bias[n] + sum_{k : u_k % in_div == 0} sum_c in[b, (m*in_mul + u_k/in_div) / up, p, g*CR + c] * w[k][n][c]
 *     u_k = in_add + phase + k*in_kstep;  virtual source tokens outside [0, Tsrc*up) contribute 0;  g = n / NG
 *     v = LeakyReLU(v, out_slope) when out_act;  v += res;  v *= (out_gate > 0 ? 1 : out_gate_slope)
 *     (res_after_gate: the residual is added after the gate instead -- identity path of a gated input gradient)
 *     out[b,d,p,n] = v (fp32, optional);  out_bf[b,d,p,n] = bf16(bf_act ? LeakyReLU(v, bf_slope) : v) (optional)
 * in (B, Tsrc, inner, Cin_tot) bf16;  w (K, Ntot, CR) bf16;  bias / res fp32;  out_gate bf16 (out_gate_bf16) or fp32.
 * KANTTS_E_UNSUPPORTED unless CR % 8 == 0, NG % 8 == 0, K <= 64, phases <= 8 and all pointers are 16-byte aligned. */
typedef struct {
  const void* in;
  const void* w;
  const float* bias;
  const float* res;
  const void* out_gate;
  float* out;
  void* out_bf;
  int B, Tsrc, Tdst, Cin_tot, Ntot, CR, NG, groups, K;
  int in_mul, in_add, in_kstep, in_div, phases;
  int inner;
  int up;
  float out_slope;
  int out_act;
  float out_gate_slope;
  int out_gate_bf16;
  float bf_slope;
  int bf_act;
  int tile; /* 0 = automatic; else BM*1000 + BN of a compiled tile (bench / tests) */
  int res_after_gate;
} kantts_cconv_args;
int kantts_cconv_launch(const kantts_cconv_args* args, void* stream);

/* Weight / bias gradients of the same convolutions from bf16 operands (csrc/cconv.hip):
 *   dw[k][n][c] += sum_{b,p,q} dy[b,q,p,n] * x[b, (q*stride + k*dil - pad) / up, p, g*CR + c];   db[n] += sum dy[b,q,p,n]
 * x (B, Tsrc, inner, Cin_tot) bf16 (already activated), dy (B, Tdst, inner, Ntot) bf16 (already gated); dw tap-major
 * (K, Ntot, CR) fp32, db fp32 or NULL.  Every workgroup owns one (tap, 128 x 128) output tile and walks a contiguous
 * slice of the tokens; `slices` > 1 splits the token axis (0 = automatic: 1 whenever the output tiles alone fill the
 * chip).  The slices' partial tiles go to `workspace` (caller-owned, >= kantts_cconv_wgrad_ws_floats(args) floats, 16-byte
 * aligned, contents irrelevant) and are summed into dw by a second kernel in a fixed order; without a workspace they
 * meet in fp32 atomics.  dw / db must be zero (or hold a value to accumulate onto) before the call. */
typedef struct {
  const void* x;
  const void* dy;
  float* dw;
  float* db;
  int B, Tsrc, Tdst, Cin_tot, Ntot, CR, NG, groups, K;
  int stride, dil, pad, inner, up;
  int slices;
  float* workspace;
  long long ws_floats;
} kantts_cconvw_args;
int kantts_cconv_wgrad_launch(const kantts_cconvw_args* args, void* stream);
long long kantts_cconv_wgrad_ws_floats(const kantts_cconvw_args* args);

/* dst = bf16(f(src)) over a flat fp32 buffer, n % 8 == 0: the bf16 operand images the kernels above read.
 *   gate == NULL:  f(v) = act ? LeakyReLU(v, slope) : v                  (activated input image)
 *   gate != NULL:  f(v) = v * (gate > 0 ? 1 : slope)                     (gated output gradient; gate fp32 or bf16) */
int kantts_act_cast_bf16(const float* src, const void* gate, int gate_bf16, void* dst, int act, float slope, long long n,
                         void* stream);

/* ------------------------------------------------------------------------------------------
 * Batch assembly on the device (csrc/batching.hip): padding / cropping of ragged utterances resident in
 * HBM -- Padder and the collate functions of kantts/datasets/dataset.py:34-85, 278-311, 690-827.
 *   out[b][t][c] = t < len[b] ? src[(row_off[b] + start[b] + t) * C + c] : pad[c]     b < B, t < Tmax, c < C
 * transpose != 0 writes out[b][c][t] instead (the vocoder's mel crop, (frames, C) -> (C, frames)).  row_off (B) int64 =
 * first row of utterance b in the flat (rows, C) source; start (B) int32 or NULL = crop offset in rows; len (B) int32 =
 * rows to copy (<= Tmax); pad (C) or NULL (zeros). */
int kantts_ragged_rows_f32(const float* src, const int64_t* row_off, const int32_t* start, const int32_t* len,
                           const float* pad, float* out, int B, int Tmax, int C, int transpose, void* stream);
int kantts_ragged_rows_i64(const int64_t* src, const int64_t* row_off, const int32_t* start, const int32_t* len,
                           const int64_t* pad, int64_t* out, int B, int Tmax, int C, int transpose, void* stream);

/* Launch-shape knobs for sweeps and tests -- they never change a result.  tn_tile: output tile of kantts_bgemm_tn* as
 * BN * 1000 + BK (64128 / 128128 / 64256 / 128256; anything else = the library's rule; the code + 1, e.g. 64129, selects
 * that tile WITHOUT the XCD-aware workgroup mapping of round 6 -- the 3-D grid of rounds 2-5, for A/B runs); tn_slices: token slices of the
 * same launches (0 = the rule); c1_wgrad_wgs: workgroup cap of the persistent weight-gradient launch of kantts_conv_c1_launch
 * (0 = 256).  The host layer maps KANTTS_TN_TILE / KANTTS_TN_SLICES / KANTTS_C1_WGRAD_WGS onto this call; the library reads
 * no environment variable for them (until round 5 it did, per launch). */
int kantts_launch_tuning(int tn_tile, int tn_slices, int c1_wgrad_wgs);

/* ------------------------------------------------------------------------------------------
 * [round 5] Free-running inference loops as ONE launch each (csrc/ar_infer.hip; SURVEY 8 row e1, bf16 mode).
 *
 * kantts_pnca_decode_run: every step of the free-running mel decoder for every sequence of a batch -- the loop of
 * kantts/models/sambert/kantts_sambert.py:569-610 around HybridAttentionDecoder.infer (:208-253; PNCA state updates
 * kantts/models/sambert/__init__.py:217-306).  One workgroup owns one sequence and walks its steps; per step the
 * matrix-vector products stream the bf16 weights from L2 as MFMA A operands (every column of the B operand is the
 * sequence's vector), activations stay in LDS, the decoder's K / V cache and the output frames are the only HBM writes.
 * Shapes fixed by the kernel: d_model 128, 8 heads x 16, feed-forward width 1024, prenet (d_mel -> 256 -> 256 -> 128).
 *   w : bf16 blob; every matrix (out, in) has its input width padded to a multiple of 128 (zeros) and is stored
 *       fragment-major like the images of kantts_fragmajor_bf16: element (16 t + i, 32 kb + 8 q + e) at
 *       (((t * in/32 + kb) * 4 + q) * 16 + i) * 8 + e -- the 1 KB one MFMA A operand needs is contiguous:
 *       P1 256 x pad(d_mel) | P2 256 x 256 | P3 128 x 256 | IN 128 x pad(d_mem + 128)  [columns: memory, prenet]
 *       per layer: QKV 384 x 128 | FC 128 x 256 [fc_x | fc_h] | W1 1024 x 128 | W2 128 x 1024
 *       OUT pad16(d_out) x 128
 *   f : fp32 blob: b_P1 256 | b_P2 256 | b_P3 128 | b_IN 128 |
 *       per layer: ln0 gamma 128, beta 128 | b_QKV 384 | b_FC 128 (= fc_x.bias + fc_h.bias) | ln1 gamma 128, beta 128 |
 *                  b_W1 1024 | b_W2 128
 *       final ln gamma 128, beta 128 | b_OUT pad16(d_out)
 *   (kantts_pnca_decode_blob_sizes reports both element counts.)
 *   memory (B, L, d_mem) fp32; hkv (B, L, n_layer * 256) fp32: the memory K | V projection of layer i at columns
 *   [256 i, 256 i + 256); xkv (n_layer, B, L, 256) fp32 workspace (the decoder's own K | V cache, contents irrelevant);
 *   out (B, L, d_out); lens (B) steps per sequence (rows at and after lens[b] are the reference's masked rows: x = 0);
 *   bw_seq (B) band width per sequence or NULL (then `bw`).  Band widths above 127: KANTTS_E_UNSUPPORTED (bw) / NaN in
 *   `out` (device-side bw_seq).  Frame fed back: the last d_mel values of a step's output. */
typedef struct kantts_decode_args {
  const void* w;
  const float* f;
  const float* memory;
  const float* hkv;
  float* xkv;
  float* out;
  const int32_t* lens;
  const int32_t* bw_seq;
  int B, L, d_mem, d_mel, d_out, n_layer, bw;
  float in_scale; /* sqrt(d_model): the input projection's alpha */
  float eps;      /* of every LayerNorm */
} kantts_decode_args;
int kantts_pnca_decode_run(const kantts_decode_args* args, void* stream);
int kantts_pnca_decode_blob_sizes(int d_mel, int d_mem, int d_out, int n_layer, long long* w_elems, long long* f_elems);

/* kantts_dur_ar_run: the free-running duration predictor (VarRnnARPredictor.infer, kantts/models/sambert/adaptors.py:67-83):
 * token i consumes the prediction of token i - 1 through prenet (1 -> 128 -> 128) -> 2 LSTM cells (H = 128) -> Linear -> ReLU.
 * One workgroup per sequence walks its tokens.  The part of the first cell's gate pre-activations that depends on the
 * conditioning only is one GEMM the caller runs before:  gc (B, T, 512) = cond . W_ih0[:, 128:]^T + b_ih0 + b_hh0.
 *   w : bf16 blob (fragment-major matrices, as above): P2 128 x 128 | G0 512 x 256 [W_ih0[:, :128] | W_hh0] |
 *       G1 512 x 256 [W_ih1 | W_hh1]
 *   f : fp32 blob: w_P1 128 | b_P1 128 | b_P2 128 | b_G1 512 (= b_ih1 + b_hh1) | w_fc 128 | b_fc 1 | 0 0 0
 *       (the one-input prenet layer and the one-row output layer are evaluated in fp32)
 *   out (B, T): ReLU(fc(h1)) per token, 0 at and after lens[b] (lens NULL: every sequence has T tokens). */
typedef struct kantts_durar_args {
  const void* w;
  const float* f;
  const float* gc;
  float* out;
  const int32_t* lens;
  int B, T;
} kantts_durar_args;
int kantts_dur_ar_run(const kantts_durar_args* args, void* stream);
/* kantts_dur_ar_run_f32: the same loop in fp32 arithmetic (products are fp32 FMAs in k order) -- what inference uses for
 * the duration predictor in EVERY precision mode, so that the index tensors derived from its output (durations, regulated
 * lengths, band widths: kantts/models/sambert/kantts_sambert.py:455-460,989-993) equal the reference's bit for bit.
 *   w : fp32 blob, the same three matrices, each (N, K) stored k-chunk-major: element (n, k) at ((k / 4) * N + n) * 4 + k % 4
 *   f, gc, out, lens: as kantts_dur_ar_run. */
int kantts_dur_ar_run_f32(const kantts_durar_args* args, void* stream);

/* kantts_ctc_attn: AttentionCTCLoss (kantts/train/loss.py:481-508) and its gradient in one launch, a workgroup per utterance
 * (replaces torch.nn.CTCLoss, whose ATen implementation copies the lengths to the host: the MAS training step could not be
 * captured).  Per utterance b: classes 0..S (S = in_lens[b]) with logit[t, 0] = `blank` (a constant) and logit[t, c] =
 * logits[b, t, c - 1]; log_softmax over those classes; CTC with the target 1..S over the first out_lens[b] frames;
 * zero_infinity semantics.
 *   logits (B, T1, T2) fp32 (attn_logprob[:, 0]); in_lens / out_lens (B) int32 on the device;
 *   ws: workspace of kantts_ctc_attn_workspace(B, T1, T2) floats (alpha and beta rows; contents irrelevant);
 *   loss (B): nll_b / S_b (0 for an impossible alignment or an empty utterance);
 *   grad (B, T1, T2): grad_scale * d loss[b] / d logits (zero at frames >= out_lens[b] and columns >= in_lens[b]) -- the
 *   caller passes grad_scale = 1 / B for the reference's mean over the batch and multiplies by the incoming gradient.
 * Limits: T2 <= 511 phonemes, T1 <= 24576 frames (KANTTS_E_UNSUPPORTED beyond). */
typedef struct kantts_ctc_args {
  const float* logits;
  const int32_t* in_lens;
  const int32_t* out_lens;
  float* ws;
  float* loss;
  float* grad;
  int B, T1, T2;
  float blank;
  float grad_scale;
} kantts_ctc_args;
long long kantts_ctc_attn_workspace(int B, int T1, int T2);
int kantts_ctc_attn(const kantts_ctc_args* args, void* stream);

/* kantts_enc_attn_fwd: the attention sub-layer of an encoder FFT block (MultiHeadSelfAttention.forward,
 * kantts/models/sambert/__init__.py:52-106 inside FFTBlock.forward :152-184) as ONE launch, a workgroup per sequence:
 *   qkv = xn . W_qkv^T + b;  8-head attention over the keys [0, lens[b]) of every query (+ attention dropout);
 *   y1 = rowmask(dropout(context . W_fc^T + b_fc) + x);  xn1 = LayerNorm(y1)  (the feed-forward sub-layer's, optional)
 * -- what kantts_bgemm_nt + kantts_attn_fwd (mode 0) + kantts_bgemm_nt with its LayerNorm epilogue compute, with the same
 * dropout streams (element index (row, channel) for the projection; ((head * B + b) * L + query) * L + key for the attention).
 *   x (M, 128) fp32, M = B * L; xn (M, 128) bf16: LayerNorm of x (the caller's); lens (B) or NULL; rowmask (M) or NULL;
 *   wqkv / wfc: fragment-major bf16 images (kantts_fragmajor_bf16) of the (384, 128) / (128, 128) weights;
 *   written for the backward pass: qkv (M, 384) fp32, o (M, 128) contexts, lse (B, 8, L), y1 (M, 128), xn1 (M, 128) bf16 or
 *   fp32 (xn1_bf16) with mean1 / rstd1 (M).  d_model 128, 8 heads of 16, L <= 128 (KANTTS_E_UNSUPPORTED beyond). */
typedef struct kantts_enc_attn_args {
  const float* x;
  const void* xn;
  const int32_t* lens;
  const uint8_t* rowmask;
  const void* wqkv;
  const float* bqkv;
  const void* wfc;
  const float* bfc;
  const float* ln1_gamma;
  const float* ln1_beta;
  float ln1_eps;
  float att_p, fc_p;
  uint64_t att_seed, fc_seed;
  const uint64_t* seed_dev;
  float* qkv;
  float* o;
  float* lse;
  float* y1;
  void* xn1;
  int xn1_bf16;
  float* mean1;
  float* rstd1;
  int B, L;
} kantts_enc_attn_args;
int kantts_enc_attn_fwd(const kantts_enc_attn_args* args, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KANTTS_HIP_H */
