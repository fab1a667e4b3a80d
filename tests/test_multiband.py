"""PQMF, MultiResolutionSTFTLoss (STFT-magnitude backward), SpecDiscriminator /
MultiSpecDiscriminator -- against tests/golden/multiband.pt, recorded from the untouched reference by
oracle/make_golden.py::multiband_case.  Host logic under the emulated ABI on the CPU, kernels on the GPU."""
import os

import pytest
import torch

from util import GOLDEN, assert_close, emulation, rel_l2


def _fix():
    return torch.load(os.path.join(GOLDEN, "multiband.pt"), weights_only=False)


def _pqmf(device):
    from kantts.models.pqmf import PQMF

    f = _fix()["pqmf"]
    pq = PQMF()
    for k, v in f["filters"].items():  # buffers: same names, shapes and values as the reference's
        assert_close(pq.state_dict()[k], v, 1e-7, what=k)
    pq = pq.to(device)
    x = f["x"].to(device)
    z = pq.analysis(x)
    assert_close(z.cpu(), f["analysis"], 2e-6, what="analysis")
    y = pq.synthesis(f["analysis"].to(device))
    assert_close(y.cpu(), f["synthesis"], 2e-6, what="synthesis")
    # near-perfect reconstruction (delay-free here: zero-phase padding on both sides)
    mid = slice(200, -200)
    assert float((y.cpu()[..., mid] - f["x"][..., mid]).abs().max()) < 2e-3
    # gradients flow through both directions
    xg = x.clone().requires_grad_(True)
    pq.synthesis(pq.analysis(xg)).pow(2).sum().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad).all()


def _mrstft(device, key, grad_tol=None):
    from kantts.train.loss import MultiResolutionSTFTLoss

    f = _fix()[key]
    if key == "mrstft":
        crit = MultiResolutionSTFTLoss()
    else:
        crit = MultiResolutionSTFTLoss(fft_sizes=[384, 683, 171], hop_sizes=[30, 60, 10], win_lengths=[150, 300, 60])
    crit = crit.to(device)
    yh = f["y_hat"].to(device).requires_grad_(True)
    sc, mag = crit(yh, f["y"].to(device))
    assert abs(float(sc) - f["sc"]) <= 2e-5 * max(1.0, f["sc"]) and abs(float(mag) - f["mag"]) <= 2e-5 * max(1.0, f["mag"])
    (sc + mag).backward()
    # d log|X| = d|X| / |X|: bins with tiny magnitudes amplify the fp32 rounding of the two FFT implementations
    assert rel_l2(yh.grad.cpu(), f["grad"]) <= (grad_tol or (5e-3 if device == "cuda" else 2e-4))


def _multispec(device):
    from kantts.models.hifigan.hifigan import MultiSpecDiscriminator

    f = _fix()["multispec"]
    torch.manual_seed(4)
    D = MultiSpecDiscriminator(fft_sizes=[256, 512], hop_sizes=[60, 120], win_lengths=[240, 400],
                               discriminator_params=f["params"])
    sd = D.state_dict()
    assert list(sd.keys()) == list(f["checksums"].keys())
    for k, (shape, s_, a_) in f["checksums"].items():
        assert tuple(sd[k].shape) == tuple(shape) and abs(float(sd[k].double().sum()) - s_) <= 1e-9 * max(1.0, a_), k
    D = D.to(device)
    y = f["y"].to(device).requires_grad_(True)
    outs, fmaps = D(y)
    for a, b in zip(outs, f["outs"]):
        assert_close(a.detach().cpu(), b, 2e-5, what="multispec out")
    for fm, fr in zip(fmaps, f["fmap_sums"]):
        assert len(fm) == len(fr)
        for a, (shape, s_, a_) in zip(fm, fr):
            assert tuple(a.shape) == tuple(shape)
            assert abs(float(a.double().sum()) - s_) <= 2e-4 * max(1.0, a_)
    loss = sum((o * o).mean() for o in outs) + sum(a.abs().mean() for fm in fmaps for a in fm)
    assert abs(float(loss) - f["loss"]) <= 2e-5 * max(1.0, f["loss"])
    loss.backward()
    assert (y.grad is None) == f["input_grad_is_none"]  # the reference takes the spectrogram under no_grad
    for n, p in D.named_parameters():
        ref = f["grad_norms"][n]
        assert abs(float(p.grad.double().norm()) - ref) <= 2e-3 * ref + 1e-7, n


def test_multispec_default_constructor_fails_like_the_reference():
    from kantts.models.hifigan.hifigan import MultiSpecDiscriminator

    assert _fix()["multispec_default_ctor"] == "TypeError"
    with pytest.raises(TypeError):
        MultiSpecDiscriminator()


def test_pqmf_emulated():
    with emulation():
        _pqmf("cpu")


@pytest.mark.parametrize("key", ["mrstft", "mrstft_subband"])
def test_mrstft_loss_emulated(key):
    with emulation():
        _mrstft("cpu", key)


def test_multispec_emulated():
    with emulation():
        _multispec("cpu")


@pytest.mark.gpu
def test_multiband_pieces_gpu():
    import kantts._hip as hip

    hip.set_precision("fp32")
    _pqmf("cuda")
    _mrstft("cuda", "mrstft")
    _mrstft("cuda", "mrstft_subband")
    _multispec("cuda")


def test_model_builder_multiband_and_criteria(emulated_cabi):
    """model_builder adds model["pqmf"] for multi-band generators (reference models/__init__.py:64-68) and
    criterion_builder knows stft_loss / subband_stft_loss."""
    from kantts.models import model_builder
    from kantts.train.loss import criterion_builder

    opt = {"type": "Adam", "params": {"lr": 2e-4, "betas": [0.5, 0.9], "weight_decay": 0.0}}
    sch = {"type": "MultiStepLR", "params": {"gamma": 0.5, "milestones": [10]}}
    config = {"model_type": "hifigan", "Model": {
        "Generator": {"params": {"channels": 32, "out_channels": 4, "upsample_scales": [8, 4, 2],
                                 "upsample_kernal_sizes": [16, 8, 4]}, "optimizer": opt, "scheduler": sch},
        "MultiPeriodDiscriminator": {"params": {"periods": [2, 3]}, "optimizer": opt, "scheduler": sch}},
        "Loss": {"stft_loss": {"enable": True, "params": {}, "weights": 1.0},
                 "subband_stft_loss": {"enable": True, "params": {"fft_sizes": [384, 683, 171], "hop_sizes": [30, 60, 10],
                                                                  "win_lengths": [150, 300, 60]}, "weights": 1.0}}}
    model, _, _ = model_builder(config, device="cpu")
    assert model["pqmf"].subbands == 4
    x = torch.randn(2, 80, 10)
    y_mb = model["generator"](x)
    assert y_mb.shape == (2, 4, 10 * 64)
    y = model["pqmf"].synthesis(y_mb)
    assert y.shape == (2, 1, 10 * 256)
    crit = criterion_builder(config, device="cpu")
    sc, mag = crit["stft_loss"](y, torch.randn(2, 1, 2560) * 0.1)
    sc2, mag2 = crit["subband_stft_loss"](y_mb, model["pqmf"].analysis(torch.randn(2, 1, 2560) * 0.1))
    (sc + mag + sc2 + mag2).backward()
    assert all(p.grad is not None for p in model["generator"].parameters())


def test_multiband_gan_step_emulated(emulated_cabi):
    """A multi-band GAN step (out_channels 4 + PQMF): generator_loss follows reference trainer.py:469-525 -- full-band
    losses on the synthesised signal, the sum halved before half the sub-band STFT loss is added, discriminators on the
    synthesised signal -- and gan_train_step runs both phases."""
    from kantts.models import model_builder
    from kantts.train.gan_step import gan_train_step, generator_loss
    from kantts.train.loss import criterion_builder

    opt = {"type": "Adam", "params": {"lr": 2e-4, "betas": [0.5, 0.9], "weight_decay": 0.0}}
    sch = {"type": "MultiStepLR", "params": {"gamma": 0.5, "milestones": [10]}}
    sub = {"fft_sizes": [384, 683, 171], "hop_sizes": [30, 60, 10], "win_lengths": [150, 300, 60]}
    config = {"model_type": "hifigan", "Model": {
        "Generator": {"params": {"channels": 32, "out_channels": 4, "upsample_scales": [8, 4, 2],
                                 "upsample_kernal_sizes": [16, 8, 4]}, "optimizer": opt, "scheduler": sch},
        "MultiPeriodDiscriminator": {"params": {"periods": [2, 3]}, "optimizer": opt, "scheduler": sch}},
        "Loss": {"stft_loss": {"enable": True, "params": {}, "weights": 1.5},
                 "subband_stft_loss": {"enable": True, "params": sub, "weights": 1.0},
                 "mel_loss": {"enable": True, "params": {}, "weights": 45.0},
                 "generator_adv_loss": {"enable": True, "params": {}, "weights": 1.0},
                 "discriminator_adv_loss": {"enable": True, "params": {}, "weights": 1.0},
                 "feat_match_loss": {"enable": True, "params": {}, "weights": 2.0}},
        "generator_grad_norm": -1, "discriminator_grad_norm": -1, "discriminator_train_start_steps": 0,
        "generator_train_start_steps": 0}
    torch.manual_seed(1)
    model, optimizer, scheduler = model_builder(config, device="cpu")
    crit = criterion_builder(config, device="cpu")
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 80, 10, generator=g)
    y = (torch.randn(2, 1, 2560, generator=g) * 0.3).clamp(-1, 1)
    gen_loss, losses, y_ = generator_loss(model, crit, x, y)
    assert y_.shape == y.shape  # the discriminators and the full-band losses see the synthesised signal
    with torch.no_grad():
        y_mb_ = model["generator"](x)
        full = model["pqmf"].synthesis(y_mb_)
        sc, mag = crit["stft_loss"](full, y)
        ssc, smag = crit["subband_stft_loss"](y_mb_, model["pqmf"].analysis(y))
        want = 0.5 * (sc + mag) * 1.5 + 0.5 * (ssc + smag) + 45.0 * losses["mel_loss"] + losses["adversarial_loss"] \
            + 2.0 * losses["feature_matching_loss"]
    assert abs(float(losses["spectral_convergence_loss"]) - float(sc)) < 1e-5
    assert abs(float(losses["sub_log_stft_magnitude_loss"]) - float(smag)) < 1e-5
    assert abs(float(gen_loss) - float(want)) < 1e-4 * max(1.0, abs(float(want)))
    w0 = [p.detach().clone() for p in model["generator"].parameters()]
    out = gan_train_step(model, optimizer, scheduler, crit, config, y, x, steps=1)
    assert {"generator_loss", "discriminator_loss", "sub_spectral_convergence_loss"} <= set(out)
    assert all(bool(torch.isfinite(v)) for v in out.values())
    assert any(float((p - q).abs().max()) > 0 for p, q in zip(model["generator"].parameters(), w0))
