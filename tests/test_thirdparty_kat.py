"""Self-consistency known-answer tests for the third-party arithmetic the reference never pins
(librosa mel basis, pytorch_wavelets db3 DWT).  Independent pins (scipy STFT, transformers' mel filterbank,
closed-form db3 taps) live in tests/test_independent_pins.py; the DWT's padding / decimation-phase convention is pinned
below to PyWavelets' published definition (odd samples of the zero-extended full convolution) by hand-computed values
for N = 7, 8 and by numpy.convolve, for the oracle's restatement and for the product's kernel."""
import numpy as np
import pytest
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


# ---- the DWT's zero-padding / decimation-phase convention ---------------------------------------------------------------
# Published definition followed (PyWavelets docs, single-level ``dwt(x, 'db3', mode='zero')``; ``dwt_coeff_len`` =
# floor((N + L - 1) / 2)): zero extension, FULL convolution with the decomposition filter, keep the ODD-indexed samples:
#     cA[n] = (x * dec_lo)[2n + 1] = sum_k dec_lo[k] x[2n + 1 - k],   x = 0 outside [0, N).
# Values below for x = 1, 2, .., N are hand-computable from the six taps, e.g. for both N:
#     cA[0] = lo[0] x[1] + lo[1] x[0] = 0.0352262919 * 2 - 0.0854412739 * 1 = -0.0149886901
#     cA[1] = lo[0] x[3] + lo[1] x[2] + lo[2] x[1] + lo[3] x[0] = 0.1409051675 - 0.2563238216 - 0.2700220400 + 0.4598775021
#           = 0.0744368080
# and for N = 8 the last one has a single term inside the signal's support pair: cA[5] = lo[4] x[7] + lo[5] x[6]... (k with
# 0 <= 11 - k < 8: k = 4, 5) = 0.8068915093 * 8 + 0.3326705530 * 7 = 8.7838259452.
KAT_DB3 = {
    7: ([-0.014988690118040174, 0.07443680798022781, 2.5701933797754615, 5.116810169464847, 9.723844335470137,
         2.328693870656698],
        [0.141550403411425, 0.03522629188713533, 0.0, 2.661364423600697, -0.2562980373687836, 0.2465840431747046]),
    8: ([-0.014988690118040174, 0.07443680798022781, 2.5701933797754615, 5.398620504521652, 8.64375617538701,
         8.783825945163407],
        [0.141550403411425, 0.03522629188713533, 0.0, 0.0, -3.9353180543234343, 0.9301142342326365]),
}


def test_dwt_hand_values_by_the_definition():
    lo, hi = TP.DB3_DEC_LO, TP.DB3_DEC_HI
    assert abs((lo[0] * 2 + lo[1] * 1) - KAT_DB3[7][0][0]) < 1e-12
    assert abs((lo[0] * 4 + lo[1] * 3 + lo[2] * 2 + lo[3] * 1) - KAT_DB3[8][0][1]) < 1e-12
    assert abs((lo[4] * 8 + lo[5] * 7) - KAT_DB3[8][0][5]) < 1e-12
    assert abs((lo[5] * 7) - KAT_DB3[7][0][5]) < 1e-12                 # N = 7: 2n + 1 - k = 6 only for k = 5
    assert abs((hi[4] * 8 + hi[5] * 7) - KAT_DB3[8][1][5]) < 1e-12


def test_dwt_pad_and_phase_convention_kat_n7_n8():
    for N, (ca, cd) in KAT_DB3.items():
        x = torch.arange(1, N + 1, dtype=torch.float64).view(1, 1, N)
        lo, hi = TP.dwt_db3_zero(x)
        assert lo.shape[-1] == (N + 5) // 2 == len(ca)
        assert np.abs(lo[0, 0].numpy() - np.array(ca)).max() < 1e-11, N
        assert np.abs(hi[0, 0].numpy() - np.array(cd)).max() < 1e-11, N


def test_dwt_equals_odd_samples_of_the_full_convolution():
    """numpy.convolve is the independent full convolution; every length class (odd / even, shorter than the filter)."""
    rng = np.random.default_rng(11)
    for N in (1, 2, 5, 6, 7, 8, 31, 64, 257, 8192):
        x = rng.standard_normal(N)
        lo, hi = TP.dwt_db3_zero(torch.from_numpy(x).view(1, 1, N))
        assert np.abs(lo[0, 0].numpy() - np.convolve(x, TP.DB3_DEC_LO)[1::2]).max() < 1e-12, N
        assert np.abs(hi[0, 0].numpy() - np.convolve(x, TP.DB3_DEC_HI)[1::2]).max() < 1e-12, N


def _product_dwt_vs_definition(dev):
    from kantts.models.hifigan.hifigan import DWT1DForward

    m = DWT1DForward(J=1, wave="db3").to(dev)
    rng = np.random.default_rng(12)
    for N, (ca, cd) in KAT_DB3.items():
        x = torch.arange(1, N + 1, dtype=torch.float32).view(1, 1, N).to(dev)
        yl, (yh,) = m(x)
        assert np.abs(yl[0, 0].cpu().numpy() - np.array(ca)).max() < 2e-6, N
        assert np.abs(yh[0, 0].cpu().numpy() - np.array(cd)).max() < 2e-6, N
    for N in (5, 64, 257, 8192, 4098):  # 8192 -> 4098 -> 2051: the two pooling stages of the default MSD
        x = rng.standard_normal((3, N)).astype(np.float32)
        yl, (yh,) = m(torch.from_numpy(x)[:, None, :].to(dev))
        for b in range(3):
            assert np.abs(yl[b, 0].cpu().numpy() - np.convolve(x[b].astype(np.float64), TP.DB3_DEC_LO)[1::2]).max() < 5e-6
            assert np.abs(yh[b, 0].cpu().numpy() - np.convolve(x[b].astype(np.float64), TP.DB3_DEC_HI)[1::2]).max() < 5e-6


def test_product_dwt_follows_the_definition_emulated(emulated_cabi):
    _product_dwt_vs_definition("cpu")


@pytest.mark.gpu
def test_product_dwt_follows_the_definition_gpu():
    _product_dwt_vs_definition("cuda")
