"""Offline mel-spectrogram feature extractor on the MI355X mel-STFT kernel.

``melspectrogram`` has the signature and result layout of the reference's numpy/librosa implementation
(kantts/preprocess/audio_processor/core/dsp.py:165-201): librosa.stft (centre-padded with zeros, periodic Hann window
of ``win_length`` centred in ``n_fft``) -> magnitude -> Slaney mel basis -> 20 log10(max(1e-5, .)) - ref_level_db ->
clip(max_norm * (S - min_level_db) / -min_level_db, 0, max_norm) (or the symmetric variant) -> (frames, n_mels).
Here the whole chain is ONE kernel launch per batch of waveforms (framing, FFT in LDS, sparse mel filterbank, dB and
normalisation fused; the complex spectrum never reaches HBM); the reference runs it per utterance on 16 CPU workers.
The rest of the reference module (wav I/O, silence trimming, Griffin-Lim) is outside the hot path.
"""
import numpy as np
import torch

from kantts.utils.audio_torch import _launch, _support_form, slaney_mel_basis

_basis_cache = {}


def _mel_support(sample_rate, n_fft, fmin, fmax, n_mels, device):
    key = (sample_rate, n_fft, fmin, fmax, n_mels, str(device))
    hit = _basis_cache.get(key)
    if hit is None:
        assert fmax <= sample_rate // 2
        melmat = slaney_mel_basis(sr=sample_rate, n_fft=n_fft, n_mels=n_mels, fmin=fmin, fmax=fmax)  # (n_mels, bins)
        hit = tuple(t.to(device) for t in _support_form(torch.from_numpy(melmat.T.copy()).float()))
        _basis_cache[key] = hit
    return hit


def preemphasis(wav, k=0.98, preemphasize=False):
    if not preemphasize:
        return wav
    out = wav.clone()
    out[..., 1:] -= k * wav[..., :-1]
    return out


def melspectrogram_batch(wavs, sample_rate, n_fft=1024, hop_length=256, win_length=1024, n_mels=80, max_norm=1.0,
                         min_level_db=-100, ref_level_db=20, fmin=50, fmax=8000, symmetric=False, preemphasize=False):
    """wavs: (B, T) float tensor on the device -> (B, 1 + T // hop_length, n_mels)."""
    x = preemphasis(wavs.float(), preemphasize=preemphasize)
    mel, _ = _launch(x, n_fft, hop_length, win_length, "hann", 0, 0.0,
                     mel=_mel_support(sample_rate, n_fft, fmin, fmax, n_mels, x.device), eps_mel=1e-5,
                     norm=(ref_level_db, min_level_db, max_norm, symmetric))
    return mel.transpose(1, 2)


def melspectrogram(y, sample_rate, n_fft=1024, hop_length=256, win_length=1024, n_mels=80, max_norm=1.0,
                   min_level_db=-100, ref_level_db=20, fmin=50, fmax=8000, symmetric=False, preemphasize=False):
    """numpy (T,) -> numpy (frames, n_mels), like the reference."""
    dev = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
    x = torch.from_numpy(np.ascontiguousarray(y, dtype=np.float32)).to(dev)[None, :]
    out = melspectrogram_batch(x, sample_rate, n_fft, hop_length, win_length, n_mels, max_norm, min_level_db,
                               ref_level_db, fmin, fmax, symmetric, preemphasize)
    return out[0].cpu().numpy()
