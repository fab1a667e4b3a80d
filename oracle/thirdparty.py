"""TEST INFRASTRUCTURE -- restatements of arithmetic that lives in un-vendored third-party
dependencies of the reference (absent from /root/reference).

The reference holds no test / golden vector for any of these and the packages are not installed in this image, so they
are restated from their published algorithms.  What pins them:

* ``librosa.filters.mel``  -- librosa 0.9.2 (environment.yaml:11); call sites
  kantts/utils/audio_torch.py:125-131 and kantts/preprocess/audio_processor/core/dsp.py:135-139.
  Slaney mel scale (htk=False), Slaney area normalisation, float32 result (n_mels, 1+n_fft//2).
  PINNED (round 3) against an independent implementation that IS installed:
  ``transformers.audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney")`` -- <= 2e-9 at every configuration the
  shipped yamls use (tests/test_independent_pins.py); self-consistency KATs in tests/test_thirdparty_kat.py.
* ``librosa.stft`` -- librosa 0.9.2; call site kantts/preprocess/audio_processor/core/dsp.py:8-9.  ``librosa_stft`` below
  restates the published algorithm (center=True with ZERO padding of n_fft // 2 [0.9.x default pad_mode="constant"],
  periodic Hann window of win_length centred in n_fft, frames = 1 + len(y) // hop, float64 window x frames -> rfft ->
  stored at the complex type matching the input: complex64 for float32 PCM).  PINNED against ``scipy.signal.stft``
  (tests/test_independent_pins.py).  oracle/ref_harness.py serves it to the reference as ``librosa.stft`` so that the
  reference's OWN ``dsp.melspectrogram`` / ``AudioProcessor.mel_extract`` can be run and recorded
  (tests/golden/dsp_melspec.pt, made by oracle/make_golden.py::dsp_melspec_case).
* ``pytorch_wavelets.DWT1DForward(wave="db3", J=1, mode="zero")`` -- unpinned git master
  (environment.yaml:64) + pywavelets 1.3.0; call site kantts/models/hifigan/hifigan.py:445-448,469-471.
  Filter taps pinned by the closed-form Daubechies D6 coefficients (tests/test_independent_pins.py).  Padding /
  decimation phase: the PUBLISHED definition followed is PyWavelets' single-level ``dwt(x, 'db3', mode='zero')``
  (pywt docs, "Signal extension modes" + ``dwt_coeff_len``): the signal is extended by zeros, fully convolved with
  the decomposition filter and the ODD-indexed samples of the full convolution are kept:
      cA[n] = (x * dec_lo)[2n + 1],  cD[n] = (x * dec_hi)[2n + 1],  n = 0 .. floor((N + 5) / 2) - 1,
  i.e. cA[n] = sum_k dec_lo[k] x[2n + 1 - k] with x = 0 outside [0, N).  pytorch_wavelets' ``afb1d`` reaches the same
  samples by padding p // 2 = 4 zeros on the left (+1 on the right for odd N) and a stride-2 cross-correlation with the
  reversed filters.  Pinned by a KAT against ``numpy.convolve`` (independent full convolution) and hand-computed
  values for N = 7 and N = 8 (tests/test_thirdparty_kat.py); no wavelet package is installed here, so the KAT pins
  the restatement to the published definition, not to a run of the package.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- mel filterbank
def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    if f.ndim:
        big = f >= min_log_hz
        mels[big] = min_log_mel + np.log(f[big] / min_log_hz) / logstep
    elif f >= min_log_hz:
        mels = min_log_mel + np.log(f / min_log_hz) / logstep
    return mels


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    if m.ndim:
        big = m >= min_log_mel
        freqs[big] = min_log_hz * np.exp(logstep * (m[big] - min_log_mel))
    elif m >= min_log_mel:
        freqs = min_log_hz * np.exp(logstep * (m - min_log_mel))
    return freqs


def librosa_mel(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, **_unused):
    """Slaney-style mel filterbank, float32 (n_mels, 1 + n_fft // 2)."""
    if fmax is None:
        fmax = float(sr) / 2
    n_freq = 1 + n_fft // 2
    fftfreqs = np.linspace(0.0, float(sr) / 2, n_freq, endpoint=True)
    mel_pts = np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2)
    mel_f = _mel_to_hz(mel_pts)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, n_freq), dtype=np.float32)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2 : n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis].astype(np.float32)
    return weights


# --------------------------------------------------------------------------- librosa.stft
def librosa_stft(y, n_fft=2048, hop_length=None, win_length=None, window="hann", center=True, dtype=None,
                 pad_mode="constant", **_unused):
    """librosa 0.9.2 ``stft``: (1 + n_fft // 2, 1 + len(y) // hop) complex, complex64 for float32 input."""
    assert window == "hann" and center and pad_mode == "constant"
    y = np.asarray(y)
    win_length = n_fft if win_length is None else win_length
    hop_length = win_length // 4 if hop_length is None else hop_length
    if dtype is None:
        dtype = np.complex64 if y.dtype == np.float32 else np.complex128
    n = np.arange(win_length)
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)  # scipy.signal.get_window("hann", fftbins=True)
    left = (n_fft - win_length) // 2
    wpad = np.zeros(n_fft)
    wpad[left:left + win_length] = w
    yp = np.pad(y, (n_fft // 2, n_fft // 2), mode="constant")
    frames = 1 + (len(yp) - n_fft) // hop_length
    idx = np.arange(n_fft)[:, None] + hop_length * np.arange(frames)[None, :]
    return np.fft.rfft(wpad[:, None] * yp[idx], axis=0).astype(dtype)


# --------------------------------------------------------------------------- db3 DWT
DB3_DEC_LO = [
    0.035226291882100656,
    -0.08544127388224149,
    -0.13501102001039084,
    0.4598775021193313,
    0.8068915093133388,
    0.3326705529509569,
]
DB3_DEC_HI = [((-1.0) ** (k + 1)) * DB3_DEC_LO[5 - k] for k in range(6)]


def dwt_db3_zero(x, dec_lo=None, dec_hi=None):
    """Single-level db3 analysis, zero padding.  x: (B, C, N) -> (lo, hi) each (B, C, floor((N+5)/2)).

    y[n] = sum_k dec[k] * xpad[2n + 5 - k]  (true convolution, stride 2)."""
    lo = torch.tensor(DB3_DEC_LO if dec_lo is None else dec_lo, dtype=x.dtype, device=x.device)
    hi = torch.tensor(DB3_DEC_HI if dec_hi is None else dec_hi, dtype=x.dtype, device=x.device)
    B, C, N = x.shape
    L = 6
    outsize = (N + L - 1) // 2
    p = 2 * (outsize - 1) - N + L
    if p % 2 == 1:
        x = F.pad(x, (0, 1))
    pad = p // 2
    # cross-correlation with the reversed filters == convolution with the filters
    w = torch.stack([lo.flip(0), hi.flip(0)], dim=0)[:, None, :]  # (2,1,6)
    w = w.repeat(C, 1, 1)
    y = F.conv1d(x, w, padding=pad, stride=2, groups=C)  # (B, 2C, out)
    y = y.view(B, C, 2, -1)
    return y[:, :, 0].contiguous(), y[:, :, 1].contiguous()


class DWT1DForward(nn.Module):
    """Stub-compatible stand-in for pytorch_wavelets.DWT1DForward (db3, J=1, zero mode only)."""

    def __init__(self, J=1, wave="db3", mode="zero"):
        super().__init__()
        assert J == 1 and wave == "db3" and mode == "zero"
        # pytorch_wavelets registers its filters as buffers h0/h1 (that is why the reference sets
        # broadcast_buffers=False, kantts/models/__init__.py:69-70); shapes (1,1,6) reversed filters.
        self.register_buffer("h0", torch.tensor(DB3_DEC_LO[::-1]).view(1, 1, 6))
        self.register_buffer("h1", torch.tensor(DB3_DEC_HI[::-1]).view(1, 1, 6))

    def forward(self, x):
        lo, hi = dwt_db3_zero(x)
        return lo, [hi]


def wav_load(path, sr=None):
    """Stand-in for librosa.load on PCM files that are already at ``sr`` (no resampling here): int16 / 32768 as float32,
    which is what librosa / soundfile return for 16-bit PCM."""
    from scipy.io import wavfile

    rate, data = wavfile.read(path)
    if sr is not None and rate != sr:
        raise ValueError("stand-in loader: %s is at %d Hz, asked for %d Hz" % (path, rate, sr))
    if data.dtype == np.int16:
        data = (data / 32768.0).astype(np.float32)
    return data.astype(np.float32), rate
