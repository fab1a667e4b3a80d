"""NSF generator: SourceModule sine excitation + strided excitation down-convolutions added to every
upsampling stage (reference hifigan.py:119-176, layers.py:229-290).  Pinned to a forward / backward of the reference
(tests/golden/hifigan_nsf.pt, oracle/make_golden.py::nsf_generator_case), causal and non-causal."""
import os

import pytest
import torch

from util import GOLDEN


def _build(causal):
    from kantts.models.hifigan.hifigan import Generator

    torch.manual_seed(3)
    return Generator(in_channels=80, channels=32, upsample_scales=[4, 4, 2, 2], upsample_kernal_sizes=[8, 8, 4, 4],
                     causal=causal, nsf_params={"nb_harmonics": 7, "sampling_rate": 16000})


def _check(name, device, exact_rng):
    fix = torch.load(os.path.join(GOLDEN, "hifigan_nsf.pt"), weights_only=False)[name]
    G = _build(name == "causal")
    assert sorted(G.state_dict().keys()) == fix["state_keys"]
    for k, (shape, s, a) in fix["weight_checksums"].items():
        v = G.state_dict()[k]
        assert tuple(v.shape) == shape and abs(float(v.double().sum()) - s) <= 1e-6 * max(1.0, a), k
    G = G.to(device)
    x = fix["x"].to(device)
    torch.manual_seed(1234)
    y = G(x)
    assert tuple(y.shape) == tuple(fix["y"].shape)
    (y * fix["cot"].to(device)).sum().backward()
    if exact_rng:
        # CPU: the excitation draws are the reference's own (same torch.distributions calls on the same generator)
        d = (y.detach().cpu() - fix["y"]).abs()
        assert float(d.max()) <= 5e-5, float(d.max())
        for n, p in G.named_parameters():
            ref = fix["grad_norms"][n]
            assert abs(float(p.grad.double().norm()) - ref) <= 2e-3 * ref + 1e-6, n
    else:
        # GPU: phase / noise come from the device generator -> statistical agreement only
        assert torch.isfinite(y).all() and float(y.abs().max()) <= 1.0
        assert abs(float(y.abs().mean()) - float(fix["y"].abs().mean())) < 0.05
        for n, p in G.named_parameters():
            assert p.grad is not None and torch.isfinite(p.grad).all(), n


@pytest.mark.parametrize("name", ["causal", "noncausal"])
def test_nsf_generator_host_logic_matches_reference_fixture(emulated_cabi, name):
    _check(name, "cpu", exact_rng=True)


def test_source_module_excitation_statistics():
    """Unvoiced frames carry pure noise of std alpha/3, voiced frames a sine of amplitude alpha per harmonic."""
    from kantts.models.hifigan.layers import SourceModule

    sm = SourceModule(nb_harmonics=7, upsample_ratio=64, sampling_rate=16000)
    pitch = torch.full((2, 1, 50), 200.0)
    uv = torch.zeros(2, 1, 50)
    uv[0] = 1.0
    pitch = pitch * uv
    torch.manual_seed(0)
    e = sm.excitation(pitch, uv)
    assert tuple(e.shape) == (2, 8, 3200)
    assert abs(float(e[1].std()) - 0.1 / 3) < 2e-3                      # unvoiced: alpha / 3 / sigma * N(0, sigma)
    assert abs(float(e[0, 0].pow(2).mean().sqrt()) - 0.1 / 2 ** 0.5) < 3e-3   # voiced fundamental: alpha * sin(...)
    # fundamental: 200 Hz at 16 kHz -> period of 80 samples
    z = e[0, 0, :1600] - e[0, 0, 80:1680]
    assert float(z.abs().mean()) < 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["causal", "noncausal"])
def test_nsf_generator_gpu(name):
    import kantts._hip as hip

    hip.set_precision("fp32")
    _check(name, "cuda", exact_rng=False)


@pytest.mark.gpu
def test_nsf_generator_gpu_matches_cpu_oracle_with_shared_excitation():
    """Same excitation tensor on both sides: GPU kernels == emulated-ABI oracle for the whole NSF generator."""
    import kantts._hip as hip
    from kantts.models.hifigan.layers import SourceModule
    from util import emulation

    hip.set_precision("fp32")
    fix = torch.load(os.path.join(GOLDEN, "hifigan_nsf.pt"), weights_only=False)["causal"]
    torch.manual_seed(5)
    e_fixed = SourceModule(7, 64, 16000).excitation(fix["x"][:, -2:-1], fix["x"][:, -1:])
    outs = []
    orig = SourceModule.excitation
    SourceModule.excitation = lambda self, pitch, uv: e_fixed.to(pitch.device)
    try:
        for dev in ("cuda", "cpu"):
            G = _build(True).to(dev)
            if dev == "cpu":
                with emulation():
                    outs.append(G(fix["x"]).detach())
            else:
                outs.append(G(fix["x"].cuda()).detach().cpu())
    finally:
        SourceModule.excitation = orig
    assert float((outs[0] - outs[1]).abs().max()) <= 5e-5


def test_nsf_training_cli_emulated(tmp_path):
    """kantts.bin.train_hifigan on synthetic NSF batches (mel + f0 + voiced flag): generator with source module,
    discriminator, both updates, checkpoint."""
    from util import emulation

    from kantts.bin.train_hifigan import train as train_voc

    opt = {"type": "Adam", "params": {"lr": 2e-4, "betas": [0.5, 0.9], "weight_decay": 0.0}}
    sch = {"type": "MultiStepLR", "params": {"gamma": 0.5, "milestones": [200000]}}
    voc = {"model_type": "hifigan", "Model": {
        "Generator": {"params": {"channels": 32, "nsf_params": {"nb_harmonics": 7, "sampling_rate": 16000}},
                      "optimizer": opt, "scheduler": sch},
        "MultiPeriodDiscriminator": {"params": {"periods": [2, 3]}, "optimizer": opt, "scheduler": sch}},
        "Loss": {"generator_adv_loss": {"enable": True, "params": {}, "weights": 1.0},
                 "discriminator_adv_loss": {"enable": True, "params": {}, "weights": 1.0},
                 "mel_loss": {"enable": True, "params": {}, "weights": 45.0},
                 "feat_match_loss": {"enable": True, "params": {}, "weights": 2.0}},
        "generator_grad_norm": -1, "discriminator_grad_norm": -1, "discriminator_train_start_steps": 0,
        "generator_train_start_steps": 0, "batch_size": 2, "batch_max_steps": 1024, "log_interval_steps": 1,
        "save_interval_steps": 2, "audio_config": {"hop_length": 256, "sampling_rate": 16000}}
    with emulation():
        tr = train_voc(voc, [], str(tmp_path / "voc_nsf"), synthetic=2)
    assert tr.steps == 3 and os.path.exists(tmp_path / "voc_nsf" / "ckpt" / "checkpoint_2.pth")
    sd = torch.load(tmp_path / "voc_nsf" / "ckpt" / "checkpoint_2.pth", map_location="cpu")["model"]["generator"]
    assert "source_module.ffn.0.weight_v" in sd and "source_downs.0.conv1d.weight_g" in sd
