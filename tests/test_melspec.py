"""mel-STFT feature extractor: host logic under the emulated ABI (CPU) and kernel parity on the GPU
against oracle/audio_oracle.py and the golden fixture recorded from the reference."""
import os

import pytest
import torch

import audio_oracle as A
from util import GOLDEN, assert_close, emulation


def _cases():
    g = torch.Generator().manual_seed(11)
    return [
        ("v1", dict(), torch.randn(3, 8192, generator=g) * 0.1),
        ("16k", dict(fs=16000, fft_size=2048, hop_size=200, win_length=1000, fmin=0, fmax=8000),
         torch.randn(2, 9600, generator=g) * 0.1),
        ("ragged", dict(fs=22050, fft_size=1024, hop_size=256), torch.randn(1, 5000, generator=g) * 0.3),
    ]


def _check(device):
    from kantts.utils.audio_torch import MelSpectrogram, stft

    for name, kw, x in _cases():
        ms = MelSpectrogram(**kw).to(device)
        got = ms(x[:, None, :].to(device)).cpu()
        ref = A.mel_spectrogram(x, **{k: v for k, v in kw.items()})
        assert_close(got, ref, 1e-4, what="mel " + name)  # <= 1e-4 abs in normalised units
    x = _cases()[0][2]
    got = stft(x.to(device), 1024, 120, 600, "hann").cpu()
    assert_close(got, A.stft_magnitude(x, 1024, 120, 600), 2e-5, what="stft magnitude")
    fix = torch.load(os.path.join(GOLDEN, "melspec.pt"), weights_only=False)
    got = MelSpectrogram().to(device)(fix["wav"].to(device)).cpu()
    assert_close(got, fix["mel_v1"], 1e-4, what="golden mel V1")
    got = MelSpectrogram(fs=16000, fft_size=2048, hop_size=200, win_length=1000, fmin=0, fmax=8000).to(device)(
        fix["wav"].to(device)).cpu()
    assert_close(got, fix["mel_16k"], 1e-4, what="golden mel 16k")
    assert_close(stft(fix["wav"].to(device), 1024, 120, 600, torch.hann_window(600)).cpu(), fix["stft_1024_120_600"],
                 2e-5, what="golden stft")


def _register_form_cases(device):
    """n_fft 1024 / 2048 take the register-resident FFT (csrc/melspec.hip: melspec_reg_kernel), KANTTS_MEL_GENERIC=1 the
    radix-2 LDS kernel every other size uses: same mel / magnitudes to rounding, against the oracle as well -- odd lengths,
    hops that leave a partial group of frames, both paddings, 128 mel channels (both channel slots of a lane), the
    magnitude output, frames that reach over both ends of the waveform."""
    from kantts.utils.audio_torch import MelSpectrogram, stft

    g = torch.Generator().manual_seed(31)
    cases = [
        (dict(fft_size=1024, hop_size=256), (5, 4099)),
        (dict(fft_size=1024, hop_size=200, win_length=800, num_mels=128, fmin=0, fmax=11025, pad_mode="reflect"), (2, 2311)),
        (dict(fs=16000, fft_size=2048, hop_size=200, win_length=1000, fmin=0, fmax=8000), (3, 3001)),
        (dict(fs=16000, fft_size=2048, hop_size=512, num_mels=40, fmin=50, fmax=7000, pad_mode="reflect"), (1, 2600)),
        (dict(fft_size=1024, hop_size=256), (1, 700)),   # every frame reaches over an end of the waveform; odd frame count
        (dict(fft_size=1024, hop_size=300, num_mels=20, fmin=0, fmax=4000), (4, 5)),  # one frame per utterance, T < hop
    ]
    for kw, (B, T) in cases:
        x = torch.randn(B, T, generator=g) * 0.2
        ms = MelSpectrogram(**kw).to(device)
        out = {}
        for generic in (False, True):
            if generic:
                os.environ["KANTTS_MEL_GENERIC"] = "1"
            try:
                out[generic] = ms(x[:, None, :].to(device)).cpu()
            finally:
                os.environ.pop("KANTTS_MEL_GENERIC", None)
        if "pad_mode" not in kw:  # the oracle restates the reference's zero padding
            assert_close(out[False], A.mel_spectrogram(x, **kw), 1e-4, what="register-form mel vs oracle %r" % (kw,))
        assert_close(out[False], out[True], 5e-5, what="register-form mel vs generic kernel %r" % (kw,))
    for n_fft, hop, win, (B, T) in ((1024, 120, 600, (3, 1500)), (2048, 240, 1200, (2, 4100))):
        x = torch.randn(B, T, generator=g) * 0.2
        out = {}
        for generic in (False, True):
            if generic:
                os.environ["KANTTS_MEL_GENERIC"] = "1"
            try:
                out[generic] = stft(x.to(device), n_fft, hop, win, "hann").cpu()
            finally:
                os.environ.pop("KANTTS_MEL_GENERIC", None)
        assert_close(out[False], A.stft_magnitude(x, n_fft, hop, win), 2e-5, rtol=2e-5, what="register-form |STFT| vs oracle")
        assert_close(out[False], out[True], 2e-5, rtol=2e-5, what="register-form |STFT| vs generic kernel")


def _wide_filterbank_case(device):
    """A filterbank no MelSpectrogram builds (Slaney triangles of <= 128 channels over 513 bins stay under 242 chunks of 8
    bins) but the C ABI accepts: 128 channels, each ~44 bins wide -> ~830 chunks, more than the register kernel's chunk table
    holds (MELR_CHMAX 256), so its per-channel loop over the global weights runs instead (csrc/melspec.hip, the `else` of
    `mel_fast`).  Against the radix-2 kernel and against the dense contraction of the magnitudes."""
    from kantts.utils.audio_torch import MelSpectrogram, stft

    g = torch.Generator().manual_seed(77)
    ms = MelSpectrogram(fft_size=1024, hop_size=256, num_mels=128, fmin=0, fmax=11025)
    nb = 513
    W = torch.zeros(nb, 128)
    for c in range(128):
        lo = 3 + (c * 461) // 128 + (c % 5)  # unaligned starts, the last channel ends at bin 512
        width = 40 + (c % 9)
        hi = min(lo + width, nb)
        W[lo:hi, c] = torch.rand(hi - lo, generator=g) * 0.05 + 0.001
    nchunks = sum(int((W[:, c].nonzero()[-1] // 8) - (W[:, c].nonzero()[0] // 8) + 1) for c in range(128))
    assert nchunks > 256  # the point of the case
    ms.melmat.copy_(W)
    ms._support = None
    ms = ms.to(device)
    x = torch.randn(3, 2900, generator=g) * 0.2
    out = {}
    for generic in (False, True):
        if generic:
            os.environ["KANTTS_MEL_GENERIC"] = "1"
        try:
            out[generic] = ms(x[:, None, :].to(device)).cpu()
        finally:
            os.environ.pop("KANTTS_MEL_GENERIC", None)
    assert_close(out[False], out[True], 5e-5, what="wide filterbank: register-form kernel vs generic kernel")
    # dense restatement from the magnitudes (zero padding, clamp(power, eps) -> sqrt; eps = 1e-10 as MelSpectrogram)
    xp = torch.nn.functional.pad(x.double(), (512, 512))
    fr = xp.unfold(1, 1024, 256) * torch.hann_window(1024, dtype=torch.float64)
    spec = torch.fft.rfft(fr, dim=-1)
    mag = torch.sqrt(torch.clamp(spec.real ** 2 + spec.imag ** 2, min=1e-10))
    mel = torch.clamp(mag @ W.double(), min=1e-10)
    ref = torch.clamp(2 * 4.0 * ((20 * torch.log10(torch.clamp(mel, min=1e-5)) - 20.0 + 100.0) / 100.0) - 4.0, -4.0, 4.0)
    assert_close(out[False], ref.transpose(1, 2).float(), 1e-4, what="wide filterbank: register-form kernel vs dense restatement")


def _several_pairs_per_wave_case(device):
    """melspec_reg_kernel is a persistent grid: past 768 workgroups x 4 waves a wave walks SEVERAL pairs of frames (the next
    pair's samples are fetched beside the filterbank, the exchange cells are reused).  The shipped batch (32 x 8192 samples:
    528 pairs) never gets there; KANTTS_MEL_WGS caps the grid, so that 45 pairs on one or two workgroups do: bit-identical
    to the one-pair-per-wave launch, mel (both channel slots) and magnitudes."""
    from kantts.utils.audio_torch import MelSpectrogram, stft

    g = torch.Generator().manual_seed(3)
    for kw, (B, T) in ((dict(fft_size=1024, hop_size=256), (5, 4099)),
                       (dict(fft_size=1024, hop_size=200, win_length=800, num_mels=128, fmin=0, fmax=11025, pad_mode="reflect"), (3, 2311))):
        x = (torch.randn(B, T, generator=g) * 0.2).to(device)
        ms = MelSpectrogram(**kw).to(device)
        full, mag = ms(x[:, None, :]).cpu(), stft(x, 1024, 120, 600, "hann").cpu()
        for cap in ("1", "2"):
            os.environ["KANTTS_MEL_WGS"] = cap
            try:
                assert torch.equal(ms(x[:, None, :]).cpu(), full), (kw, cap)
                assert torch.equal(stft(x, 1024, 120, 600, "hann").cpu(), mag), (kw, cap)
            finally:
                os.environ.pop("KANTTS_MEL_WGS", None)


@pytest.mark.gpu
def test_register_resident_fft_form_gpu():
    _register_form_cases("cuda")


@pytest.mark.gpu
def test_register_form_walks_several_pairs_of_frames_per_wave_gpu():
    """Device twin of the kernel-source case: the persistent-grid loop of melspec_reg_kernel (several pairs of frames per
    wave) is bit-identical to the one-pair-per-wave launch."""
    _several_pairs_per_wave_case("cuda")


@pytest.mark.gpu
def test_register_form_with_a_filterbank_beyond_its_chunk_table_gpu():
    _wide_filterbank_case("cuda")


@pytest.mark.gpu
def test_saturating_launch_values_gpu():
    """The 67 584-frame launch bench.py quotes as `roofline_saturating` (2048 x 8192 samples: every wave of the persistent
    grid walks many pairs) compared by VALUE with the radix-2 kernel on the same input, and a sample of its rows with the
    float64 oracle."""
    import audio_oracle as A
    from kantts.utils.audio_torch import MelSpectrogram

    g = torch.Generator().manual_seed(11)
    ms = MelSpectrogram().cuda()
    x = (torch.randn(2048, 8192, generator=g) * 0.1).cuda()
    a = ms(x[:, None, :])
    assert a.shape == (2048, 80, 33)
    os.environ["KANTTS_MEL_GENERIC"] = "1"
    try:
        b = ms(x[:, None, :])
    finally:
        os.environ.pop("KANTTS_MEL_GENERIC", None)
    assert float((a - b).abs().max()) < 5e-5
    rows = [0, 1, 777, 1024, 2047]
    ref = A.mel_spectrogram(x[rows].cpu())
    assert_close(a[rows].cpu(), ref.float(), 1e-4, what="67 584-frame launch vs the float64 oracle (sampled rows)")


def _layouts_agree(device):
    """The C ABI in both mel layouts: kantts_melspec_fwd / _norm_fwd / _bwd (channel-major, the reference's
    (B, n_mels, frames)) against kantts_melspec_norm_fwd_fm / kantts_melspec_bwd_fm (frame-major, what the host layer uses
    and hands out as a transposed view) -- the same numbers, forward bit for bit, the waveform gradient to the order of its
    atomics; n_fft 1024 (register kernel) and 512 (radix-2 kernel)."""
    from kantts._hip import check, lib, ptr
    from kantts.utils.audio_torch import MelSpectrogram, _fft_consts

    g = torch.Generator().manual_seed(41)
    for n_fft, hop in ((1024, 256), (512, 128)):
        ms = MelSpectrogram(fft_size=n_fft, hop_size=hop).to(device)
        B, T = 3, 2500
        x = (torch.randn(B, T, generator=g) * 0.2).to(device)
        frames = 1 + T // hop
        wpad, tw = _fft_consts(n_fft, n_fft, "hann", x.device)
        st, ln, of, w = ms._mel_support(x.device)
        n_mels = st.numel()
        cm = torch.empty(B, n_mels, frames, device=device)
        cm2 = torch.empty_like(cm)
        fm = torch.empty(B, frames, n_mels, device=device)
        L = lib()
        check(L.kantts_melspec_fwd(ptr(x), B, T, n_fft, hop, frames, 0, ptr(wpad), ptr(tw), 1e-10, ptr(st), ptr(ln), ptr(of),
                                   ptr(w), n_mels, 1e-10, ptr(cm), None, None), "melspec_fwd")
        check(L.kantts_melspec_norm_fwd(ptr(x), B, T, n_fft, hop, frames, 0, ptr(wpad), ptr(tw), 1e-10, ptr(st), ptr(ln),
                                        ptr(of), ptr(w), n_mels, 1e-10, 20.0, -100.0, 4.0, 1, ptr(cm2), None, None), "norm_fwd")
        check(L.kantts_melspec_norm_fwd_fm(ptr(x), B, T, n_fft, hop, frames, 0, ptr(wpad), ptr(tw), 1e-10, ptr(st), ptr(ln),
                                           ptr(of), ptr(w), n_mels, 1e-10, 20.0, -100.0, 4.0, 1, 1, ptr(fm), None, None),
              "norm_fwd_fm")
        assert torch.equal(cm, cm2) and torch.equal(cm, fm.transpose(1, 2)), n_fft
        dm = torch.randn(B, n_mels, frames, generator=g).to(device)
        da, db = torch.zeros_like(x), torch.zeros_like(x)
        check(L.kantts_melspec_bwd(ptr(x), ptr(dm), B, T, n_fft, hop, frames, 0, ptr(wpad), ptr(tw), 1e-10, ptr(st), ptr(ln),
                                   ptr(of), ptr(w), n_mels, 1e-10, ptr(da), None), "melspec_bwd")
        dm_fm = dm.transpose(1, 2).contiguous()
        check(L.kantts_melspec_bwd_fm(ptr(x), ptr(dm_fm), B, T, n_fft, hop, frames, 0, ptr(wpad),
                                      ptr(tw), 1e-10, ptr(st), ptr(ln), ptr(of), ptr(w), n_mels, 1e-10, 1, ptr(db), None),
              "melspec_bwd_fm")
        assert float((da - db).abs().max()) <= 1e-5 * float(da.abs().max()), n_fft
        assert float(da.abs().max()) > 0


def test_mel_layouts_agree_emulated():
    with emulation():
        _layouts_agree("cpu")


@pytest.mark.gpu
def test_mel_layouts_agree_gpu():
    _layouts_agree("cuda")


def test_melspec_host_logic_emulated():
    with emulation():
        _check("cpu")


@pytest.mark.gpu
def test_melspec_gpu_matches_oracle_and_golden():
    _check("cuda")


@pytest.mark.gpu
def test_melspec_linearity_and_batch_independence_full_size():
    """Size-independent properties at the BASELINE shape (32 x 8192): scaling the waveform by 10 adds
    exactly 20 dB = 1.6 normalised units where nothing clips; rows do not interact."""
    from kantts.utils.audio_torch import MelSpectrogram

    g = torch.Generator().manual_seed(5)
    x = (torch.randn(32, 8192, generator=g) * 0.05).cuda()
    ms = MelSpectrogram().cuda()
    a, b = ms(x), ms(x * 10.0)
    inside = (a > -3.9) & (b < 3.9) & (a < 2.3)
    assert inside.float().mean() > 0.5
    assert float(((b - a) - 1.6)[inside].abs().max()) < 2e-4
    one = ms(x[5:6])
    assert torch.equal(one, a[5:6])


def _check_backward(device):
    """d(mel L1 loss)/d(wav): kernel (or emulated ABI) vs autograd through oracle/audio_oracle.py."""
    from kantts.utils.audio_torch import MelSpectrogram

    g = torch.Generator().manual_seed(21)
    x = torch.randn(3, 4096, generator=g) * 0.1
    tgt = A.mel_spectrogram(torch.randn(3, 4096, generator=g) * 0.1)
    xr = x.clone().requires_grad_(True)
    (A.mel_spectrogram(xr) - tgt).abs().mean().backward()
    ms = MelSpectrogram().to(device)
    xd = x.clone().to(device).requires_grad_(True)
    (ms(xd[:, None, :]) - tgt.to(device)).abs().mean().backward()
    err = (xd.grad.cpu() - xr.grad).norm() / xr.grad.norm()
    assert float(err) < 2e-3, float(err)


def test_melspec_backward_emulated():
    with emulation():
        _check_backward("cpu")


@pytest.mark.gpu
def test_melspec_backward_gpu():
    _check_backward("cuda")


def _dsp_case(device):
    """Offline extractor dsp.melspectrogram against the float64 numpy restatement."""
    import numpy as np

    import audio_oracle as A
    from kantts.preprocess.audio_processor.core import dsp

    g = np.random.default_rng(3)
    y = (g.standard_normal(5000) * 0.1).astype(np.float32)
    for kw in (dict(sample_rate=16000, n_fft=1024, hop_length=200, win_length=1000, fmin=0, fmax=8000),
               dict(sample_rate=16000, n_fft=2048, hop_length=200, win_length=1000, fmin=0, fmax=8000, max_norm=1.0),
               dict(sample_rate=22050, symmetric=True, max_norm=4.0, preemphasize=True)):
        ref = A.dsp_melspectrogram(y, **kw)
        wav = torch.from_numpy(y).to(device)[None, :]
        args = dict(kw)
        sr = args.pop("sample_rate")
        got = dsp.melspectrogram_batch(wav, sr, **args)[0].cpu().numpy()
        assert got.shape == ref.shape == (1 + len(y) // kw.get("hop_length", 256), 80)
        assert np.abs(got - ref).max() < 2e-4 * kw.get("max_norm", 1.0) + 1e-5, kw


def test_dsp_melspectrogram_emulated():
    from util import emulation

    with emulation():
        _dsp_case("cpu")


@pytest.mark.gpu
def test_dsp_melspectrogram_gpu():
    _dsp_case("cuda")
