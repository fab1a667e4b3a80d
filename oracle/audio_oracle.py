"""TEST INFRASTRUCTURE -- CPU restatement of the reference's mel-STFT feature extractor.

Restates kantts/utils/audio_torch.py (torch path) with explicit framing + rFFT so that nothing
depends on ``torch.stft`` defaults.  The mel basis comes from oracle/thirdparty.py (librosa
restatement, PARITY UNPINNED by the reference -- see that file's header).  Only tests/, smoke()
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
