"""Self-consistency known-answer tests for the third-party arithmetic the reference never pins
(librosa mel basis, pytorch_wavelets db3 DWT).  Independent pins (scipy STFT, transformers' mel filterbank,
closed-form db3 taps) live in tests/test_independent_pins.py; the DWT's padding / phase convention stays "parity unpinned"."""
import numpy as np
import torch

import thirdparty as TP


def test_mel_basis_shape_support_and_norm():
    m = TP.librosa_mel(sr=22050, n_fft=1024, n_mels=80, fmin=80, fmax=7600)
    assert m.shape == (80, 513) and m.dtype == np.float32
    assert (m >= 0).all()
    freqs = np.linspace(0, 11025, 513)
    peaks = freqs[m.argmax(axis=1)]
    assert (np.diff(peaks) > 0).all()            # monotone centre frequencies
    assert peaks[0] > 80 and peaks[-1] < 7600
    assert m[:, freqs < 80].sum() == 0 and m[:, freqs > 7600].sum() == 0
    # Slaney area normalisation: integral of each triangle over Hz is ~1 (discretised)
    area = (m * (freqs[1] - freqs[0])).sum(axis=1)
    assert np.allclose(area[20:], 1.0, atol=0.08)
    # Slaney scale is linear below 1 kHz: equal spacing of the low filters
    low = peaks[peaks < 900]
    assert np.allclose(np.diff(low), np.diff(low).mean(), atol=freqs[1] - freqs[0] + 1e-6)


def test_db3_filters_orthonormal():
    lo = np.array(TP.DB3_DEC_LO)
    hi = np.array(TP.DB3_DEC_HI)
    assert abs(lo.sum() - np.sqrt(2)) < 1e-10 and abs((lo ** 2).sum() - 1) < 1e-10
    assert abs(hi.sum()) < 1e-10 and abs((hi ** 2).sum() - 1) < 1e-10
    assert abs((lo * hi).sum()) < 1e-10
    for s in (2, 4):
        assert abs((lo[s:] * lo[:-s]).sum()) < 1e-10


def test_dwt_lengths_and_energy():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 1, 8192, generator=g, dtype=torch.float64)
    lo, hi = TP.dwt_db3_zero(x)
    assert lo.shape[-1] == 4098 and hi.shape[-1] == 4098
    e_in, e_out = float((x ** 2).sum()), float((lo ** 2).sum() + (hi ** 2).sum())
    assert abs(e_in - e_out) < 1e-9 * e_in       # orthonormal transform with zero extension
    lo2, _ = TP.dwt_db3_zero(lo)
    assert lo2.shape[-1] == 2051
