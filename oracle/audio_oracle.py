"""TEST INFRASTRUCTURE -- CPU restatement of the reference's mel-STFT feature extractor.

Restates kantts/utils/audio_torch.py (torch path) with explicit framing + rFFT so that nothing
depends on ``torch.stft`` defaults.  The mel basis comes from oracle/thirdparty.py (librosa restatement; pinned against
transformers.audio_utils.mel_filter_bank, the framing against scipy.signal.stft: tests/test_independent_pins.py).  Only tests/, smoke()
and bench.py's cpu_baseline leg may import this file.
"""
import torch
import torch.nn.functional as F

from thirdparty import librosa_mel


def _frames(x, n_fft, hop, pad_mode):
    """center=True framing of torch.stft: pad n_fft//2 both sides, frames = 1 + T // hop."""
    x = F.pad(x[:, None, :], (n_fft // 2, n_fft // 2), mode=pad_mode)[:, 0]
    return x.unfold(1, n_fft, hop)  # (B, frames, n_fft)


def _window(win_length, n_fft, dtype):
    w = torch.hann_window(win_length, periodic=True, dtype=dtype)
    left = (n_fft - win_length) // 2
    return F.pad(w, (left, n_fft - win_length - left))


def stft_magnitude(x, fft_size, hop_size, win_length, clamp=1e-7, pad_mode="reflect"):
    """``stft`` of kantts/utils/audio_torch.py:8-31: |STFT| with clamp(re^2+im^2, 1e-7) -> sqrt,
    result (B, frames, fft_size//2+1).  torch.stft default pad_mode is "reflect"."""
    fr = _frames(x, fft_size, hop_size, pad_mode) * _window(win_length, fft_size, x.dtype)
    spec = torch.fft.rfft(fr, n=fft_size, dim=-1)
    return torch.sqrt(torch.clamp(spec.real ** 2 + spec.imag ** 2, min=clamp))


def mel_basis(fs, fft_size, num_mels, fmin, fmax):
    """melmat (n_freq, n_mels) float32 -- kantts/utils/audio_torch.py:125-132 stores the transpose."""
    fmin = 0 if fmin is None else fmin
    fmax = fs / 2 if fmax is None else fmax
    return torch.from_numpy(librosa_mel(sr=fs, n_fft=fft_size, n_mels=num_mels, fmin=fmin, fmax=fmax).T.copy()).float()


def mel_spectrogram(x, fs=22050, fft_size=1024, hop_size=256, win_length=None, num_mels=80,
                    fmin=80, fmax=7600, eps=1e-10, melmat=None):
    """MelSpectrogram.forward, kantts/utils/audio_torch.py:155-186 (+ spectral_normalize_torch :42-57):
    zero-padded centred STFT -> sqrt(clamp(power, eps)) -> @ melmat -> clamp(eps)
    -> 20*log10(clamp(., 1e-5)) - 20 -> clamp(8*((.+100)/100) - 4, -4, 4).  Returns (B, n_mels, frames).
    (``log_base`` is ignored by the reference, audio_torch.py:183-186.)"""
    if x.dim() == 3:
        x = x.reshape(-1, x.shape[2])
    win_length = fft_size if win_length is None else win_length
    fr = _frames(x, fft_size, hop_size, "constant") * _window(win_length, fft_size, x.dtype)
    spec = torch.fft.rfft(fr, n=fft_size, dim=-1)
    amp = torch.sqrt(torch.clamp(spec.real ** 2 + spec.imag ** 2, min=eps))
    if melmat is None:
        melmat = mel_basis(fs, fft_size, num_mels, fmin, fmax).to(x.dtype)
    mel = torch.clamp(amp @ melmat, min=eps)
    db = 20 * torch.log10(torch.clamp(mel, min=1e-5)) - 20.0
    out = torch.clamp(2 * 4.0 * ((db + 100.0) / 100.0) - 4.0, min=-4.0, max=4.0)
    return out.transpose(1, 2)


def dsp_stft_magnitude(y, n_fft, hop_length, win_length):
    """|librosa.stft(y, n_fft, hop_length, win_length)| under librosa 0.9.2's defaults (center=True, pad_mode="constant",
    periodic Hann window centred in n_fft): (frames, bins) float64.  Pinned against scipy.signal.stft by
    tests/test_independent_pins.py."""
    import numpy as np

    y = np.asarray(y, dtype=np.float64)
    pad = n_fft // 2
    yp = np.pad(y, (pad, pad))
    frames = 1 + len(y) // hop_length
    n = np.arange(win_length)
    w = 0.5 - 0.5 * np.cos(2 * np.pi * n / win_length)
    left = (n_fft - win_length) // 2
    wpad = np.zeros(n_fft)
    wpad[left:left + win_length] = w
    idx = np.arange(n_fft)[None, :] + hop_length * np.arange(frames)[:, None]
    return np.abs(np.fft.rfft(yp[idx] * wpad, n=n_fft, axis=-1))


def dsp_melspectrogram(y, sample_rate, n_fft=1024, hop_length=256, win_length=1024, n_mels=80, max_norm=1.0,
                       min_level_db=-100, ref_level_db=20, fmin=50, fmax=8000, symmetric=False, preemphasize=False):
    """``melspectrogram`` of kantts/preprocess/audio_processor/core/dsp.py:165-201 in float64 numpy:
    librosa.stft(y, n_fft, hop_length, win_length) [librosa 0.9.2: center=True with zero padding, periodic Hann window
    centred in n_fft -- framing / window / padding pinned against scipy.signal.stft, the mel basis against
    transformers.audio_utils.mel_filter_bank]
    -> |D| -> librosa mel basis (:135-139) -> 20 log10(max(1e-5, .)) - ref_level_db (:20, :190) -> _normalize (:66-75)
    -> transpose to (frames, n_mels)."""
    import numpy as np

    y = np.asarray(y, dtype=np.float64)
    if preemphasize:
        y = np.concatenate([y[:1], y[1:] - 0.98 * y[:-1]])
    mag = dsp_stft_magnitude(y, n_fft, hop_length, win_length)  # (frames, bins)
    basis = librosa_mel(sr=sample_rate, n_fft=n_fft, n_mels=n_mels, fmin=fmin, fmax=fmax).astype(np.float64)
    S = 20 * np.log10(np.maximum(1e-5, mag @ basis.T)) - ref_level_db
    u = (S - min_level_db) / (-min_level_db)
    if symmetric:
        return np.clip(2 * max_norm * u - max_norm, -max_norm, max_norm)
    return np.clip(max_norm * u, 0, max_norm)
