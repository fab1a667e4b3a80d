"""Edge cases of the hot path: single-utterance and very ragged batches, one-token sequences, one-frame vocoder
inputs, waveform lengths that are not multiples of the hop / the discriminator periods, and empty problems through
the C ABI.  CPU: emulated ABI vs the oracle; GPU: the kernels vs the oracle."""
import ctypes

import pytest
import torch

import audio_oracle as A
import hifigan_oracle as H
import torch_oracle as O
from util import assert_close, emulation


def _sambert_ragged(device):
    from kantts.models.sambert.kantts_sambert import KanTtsSAMBERT
    from kantts.train.loss import MelReconLoss, ProsodyReconLoss

    cfg = O.sambert_config(tiny=True)
    for B, T_in, min_len, dur_hi in ((1, 9, 8, 5), (4, 14, 1, 4)):
        torch.manual_seed(0)
        m = KanTtsSAMBERT(dict(cfg)).eval()
        P = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
        batch = O.synthetic_sambert_batch(B=B, T_in=T_in, seed=3, min_len=min_len, dur_hi=dur_hi)
        if B > 1:
            # one utterance with a single phoneme next to a full-length one
            lens = batch["input_lengths"]
            lens[1] = 1
            d = batch["duration_targets"]
            d[1] = 0
            d[1, 0] = 3
            batch["output_lengths"][1] = 3
            Tm = batch["mel_targets"].shape[1]
            d[1, 1] = Tm - 3  # r-padding frames parked on token len (Padder._pad_durations)
            batch["mel_targets"][1, 3:] = 0
        m = m.to(device)
        gb = {k: v.to(device) for k, v in batch.items()}
        res = m(**gb)
        out = O.sambert_forward(P, cfg, **batch)
        assert torch.equal(res["LR_length_rounded"].cpu(), out["LR_length_rounded"])
        for k in ("dec_outputs", "postnet_outputs", "log_duration_predictions", "pitch_predictions"):
            assert_close(res[k].detach().cpu(), out[k].detach(), 5e-5, what="%s B=%d" % (k, B))
        mel_, mel = MelReconLoss()(gb["output_lengths"], gb["mel_targets"], res["dec_outputs"], res["postnet_outputs"])
        dl, pl, el = ProsodyReconLoss()(gb["input_lengths"], res["duration_targets"], res["pitch_targets"],
                                        res["energy_targets"], res["log_duration_predictions"],
                                        res["pitch_predictions"], res["energy_predictions"])
        (mel_ + mel + dl + pl + el).backward()
        O.sambert_losses(out, batch["input_lengths"], batch["output_lengths"], batch["mel_targets"])["total"].backward()
        num = den = 0.0
        for n, p in m.named_parameters():
            if p.grad is not None:
                num += float((p.grad.cpu().double() - P[n].grad.double()).pow(2).sum())
                den += float(P[n].grad.double().pow(2).sum())
        assert (num / den) ** 0.5 < 2e-3, (B, (num / den) ** 0.5)


def test_sambert_single_and_one_token_utterances_emulated():
    with emulation():
        _sambert_ragged("cpu")


@pytest.mark.gpu
def test_sambert_single_and_one_token_utterances_gpu():
    import kantts._hip as hip

    hip.set_precision("fp32")
    _sambert_ragged("cuda")


def _vocoder_edges(device):
    from kantts.models.hifigan.hifigan import Generator, MultiPeriodDiscriminator, MultiScaleDiscriminator
    from kantts.utils.audio_torch import MelSpectrogram

    torch.manual_seed(0)
    G, D1, D2 = Generator(channels=32), MultiPeriodDiscriminator(), MultiScaleDiscriminator()
    PG = {k: v.detach().clone() for k, v in G.state_dict().items()}
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 80, 1, generator=g)  # ONE mel frame
    with torch.no_grad():
        y = G.to(device)(x.to(device))
        assert y.shape == (1, 1, 256)
        assert_close(y.cpu(), H.generator(PG, x), 2e-5, what="one-frame generator")
    # 1531 samples: prime-ish, not a multiple of any period (reflect padding inside every MPD branch) nor of 2 (DWT)
    w = torch.randn(2, 1, 1531, generator=g).clamp(-1, 1)
    for D, f in ((D1, H.mpd), (D2, H.msd)):
        P = {k: v.detach().clone() for k, v in D.state_dict().items()}
        with torch.no_grad():
            o, fm = D.to(device)(w.to(device))
            o_r, f_r = f(P, w)
        for a, b in zip(o, o_r):
            assert_close(a.cpu(), b, 5e-5, what="odd-length discriminator output")
        for fa, fb in zip(fm, f_r):
            for a, b in zip(fa, fb):
                assert a.shape == b.shape
    # mel of a waveform whose length is not a multiple of the hop
    xw = torch.randn(3, 1000, generator=g) * 0.1
    got = MelSpectrogram().to(device)(xw[:, None, :].to(device)).cpu()
    assert got.shape == (3, 80, 1 + 1000 // 256)
    assert_close(got, A.mel_spectrogram(xw), 1e-4, what="short mel")


def test_vocoder_minimal_and_odd_lengths_emulated():
    with emulation():
        _vocoder_edges("cpu")


@pytest.mark.gpu
def test_vocoder_minimal_and_odd_lengths_gpu():
    import kantts._hip as hip

    hip.set_precision("fp32")
    _vocoder_edges("cuda")


@pytest.mark.gpu
def test_empty_problems_are_accepted_by_the_abi():
    """M = 0 / B = 0 / T = 0 calls return KANTTS_OK without launching; NULL mandatory pointers are rejected."""
    import kantts._hip as hip

    L = hip.lib()
    x = torch.zeros(16, device="cuda")
    p = x.data_ptr()
    g = hip.GemmArgs()
    g.seg[0] = hip.make_seg(x, 4, 1, x, 4, 1, 4)
    g.nseg, g.M, g.N, g.c, g.c_is, g.c_js, g.alpha, g.splitk, g.groups = 1, 0, 4, p, 4, 1, 1.0, 1, 1
    assert L.kantts_gemm_seg_launch(ctypes.byref(g), None) == 0
    g.c = None
    g.M = 4
    assert L.kantts_gemm_seg_launch(ctypes.byref(g), None) < 0
    assert L.kantts_layernorm_fwd(p, p, p, p, p, p, 0, 4, 1e-6, None) == 0
    assert L.kantts_lstm_fwd(p, p, p, None, p, p, p, 0, 5, 128, 1, 0, 0, None) == 0
    assert L.kantts_lstm_fwd(p, p, p, None, p, p, p, 2, 5, 64, 1, 0, 0, None) == -2  # H != 128: unsupported
    c = hip.ConvArgs()
    c.in_, c.w, c.out = p, p, p
    c.B, c.Tsrc, c.Tdst, c.Cin_tot, c.Ntot, c.CR, c.NG, c.groups, c.K = 0, 8, 8, 4, 4, 4, 4, 1, 3
    c.in_mul, c.in_kstep, c.in_div, c.phases, c.inner = 1, 1, 1, 1, 1
    assert L.kantts_conv_win_launch(ctypes.byref(c), None) == 0
    c.in_div = 0
    assert L.kantts_conv_win_launch(ctypes.byref(c), None) < 0
    assert L.kantts_sinadd_fwd(p, p, 0, None) == 0
    torch.cuda.synchronize()
