"""HiFi-GAN generator / MPD / MSD: host logic under the emulated ABI (CPU) and kernel parity on the GPU
against oracle/hifigan_oracle.py.  The oracle itself is pinned to the untouched reference twice: live in the build
container by oracle/check_vs_reference.py (its HiFi-GAN section: G / MPD / MSD forward + gradients at 512 and 64
channels) and, travelling with the repo, by tests/golden/hifigan_v1.pt (the reference's V1 class defaults;
test_oracle_matches_reference_v1_fixture below).
Tolerances: fp32 path wav mean-abs <= 1e-4 (asserted at 1e-5), parameter gradients rel-L2 <= 2e-3."""
import os

import pytest
import torch
import torch.nn.functional as F

import hifigan_oracle as H
from util import assert_close, assert_grads_close, emulation, rel_l2, run_both


def _models(channels, seed=0):
    from kantts.models.hifigan.hifigan import Generator, MultiPeriodDiscriminator, MultiScaleDiscriminator

    torch.manual_seed(seed)
    return Generator(channels=channels), MultiPeriodDiscriminator(), MultiScaleDiscriminator()


def _check_models(device, channels, B, frames, T_wav, gtol, wtol):
    G, D1, D2 = _models(channels)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 80, frames, generator=g)
    y = torch.randn(B, 1, T_wav, generator=g).clamp(-1, 1)
    PG = {k: v.detach().clone().requires_grad_(True) for k, v in G.state_dict().items()}
    G = G.to(device)
    yo = G(x.to(device))
    yr = H.generator(PG, x)
    assert yo.shape == yr.shape == (B, 1, frames * 256)
    assert float((yo.detach().cpu() - yr.detach()).abs().mean()) <= wtol
    cot = torch.randn(yr.shape, generator=g)
    (yo * cot.to(device)).sum().backward()
    (yr * cot).sum().backward()
    assert_grads_close([(n, p.grad, PG[n].grad) for n, p in G.named_parameters()], gtol, "G")
    for D, f, nm in ((D1, H.mpd, "mpd"), (D2, H.msd, "msd")):
        P = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in D.state_dict().items()}
        D = D.to(device)
        yy = y.clone().to(device).requires_grad_(True)
        y2 = y.clone().requires_grad_(True)
        o, fm = D(yy)
        o_r, f_r = f(P, y2)
        for a, b in zip(o, o_r):
            assert_close(a.detach().cpu(), b.detach(), 2e-5, what=nm + " out")
        for fa, fb in zip(fm, f_r):
            assert len(fa) == len(fb)
            for a, b in zip(fa, fb):
                assert a.shape == b.shape
                assert_close(a.detach().cpu(), b.detach(), 5e-5, what=nm + " fmap")
        l1 = sum((a * a).sum() for a in o) + sum(a.abs().mean() for fa in fm for a in fa)
        l2 = sum((a * a).sum() for a in o_r) + sum(a.abs().mean() for fa in f_r for a in fa)
        l1.backward()
        l2.backward()
        assert_grads_close([(n, p.grad, P[n].grad) for n, p in D.named_parameters()], gtol, nm)
        assert rel_l2(yy.grad.cpu(), y2.grad) <= gtol, nm + " d(input)"


def test_hifigan_state_dict_keys_match_reference_layout():
    G, D1, D2 = _models(32)
    ks = list(G.state_dict().keys())
    assert "transpose_upsamples.0.1.deconv.weight_g" in ks and "conv_blocks.11.convs2.2.conv1d.weight_v" in ks
    assert len(ks) == 246  # reference Generator key count
    assert "discriminators.4.convs.3.0.weight_g" in D1.state_dict() and "discriminators.0.conv_post.weight" in D1.state_dict()
    kd = D2.state_dict()
    assert "aux_convs.1.weight_v" in kd and "meanpools.0.h0" in kd and "discriminators.2.conv_post.weight_g" in kd


def test_hifigan_host_logic_emulated():
    with emulation():
        _check_models("cpu", channels=32, B=2, frames=4, T_wav=640, gtol=2e-3, wtol=1e-6)


def test_conv_variants_emulated_match_torch():
    from kantts._hip import ops

    with emulation():
        B, T = 2, 19
        x = torch.randn(B, T, 16, requires_grad=True)
        w = torch.randn(32, 4, 7, requires_grad=True)
        b = torch.randn(32, requires_grad=True)
        y = ops.conv_cl(x, w, b, stride=2, pad=3, groups=4, out_leaky=0.1)
        ref = F.leaky_relu(F.conv1d(x.transpose(1, 2), w, b, stride=2, padding=3, groups=4), 0.1).transpose(1, 2)
        assert_close(y.detach(), ref.detach(), 1e-5, what="grouped strided")
        gy = torch.autograd.grad(y.sum(), (x, w, b))
        gr = torch.autograd.grad(ref.sum(), (x, w, b))
        for a, c in zip(gy, gr):
            assert rel_l2(a, c) < 1e-5


_WIN_CASES = [
    # (B, Tin, Cin, Cout, K, stride, dil, pad, groups, in_leaky, out_leaky, use_res)
    (2, 300, 64, 128, 41, 4, 1, 20, 4, None, 0.1, False),      # MSD strided grouped (tall tile)
    (3, 150, 256, 256, 41, 1, 1, 20, 16, None, 0.1, False),    # MSD stride-1 grouped, CR = NG = 16
    (2, 517, 32, 32, 11, 1, 5, 50, 1, 0.1, None, True),        # residual block: causal dilated + res
    (2, 260, 128, 256, 7, 1, 1, 6, 1, 0.1, None, False),       # wide tile (NG >= 128)
    (2, 200, 32, 1, 7, 1, 1, 6, 1, 0.01, None, False),         # conv_post: one output channel
    (1, 40, 8, 12, 3, 2, 3, 3, 1, None, None, False),          # short, stride 2, Cout % 4 == 0 only
    (2, 90, 24, 20, 5, 3, 2, 4, 2, 0.2, 0.3, False),           # stride 3 dil 2: dgrad phases with uneven taps
    (3, 700, 1, 128, 15, 1, 1, 7, 1, None, 0.1, False),        # MSD first layer: one input channel (conv_c1.hip)
    (2, 333, 1, 32, 5, 3, 1, 2, 1, None, 0.1, False),          # MPD first layer shape, strided
    (2, 96, 64, 128, 5, 2, 1, 2, 8, None, 0.1, False),         # 8 -> 16 channels per group: 4 groups packed block-diagonally
    (2, 70, 128, 256, 7, 4, 1, 3, 8, 0.1, None, False),        # 16 -> 32 per group: 2 groups packed; strided dgrad phases
    (1, 64, 48, 24, 3, 1, 2, 2, 12, None, None, False),        # 4 -> 2 per group (direct kernels, packing declined: CR % 4)
    (2, 128, 128, 256, 41, 2, 1, 20, 16, None, 0.1, False),    # the MSD layer the packing is for (k = 41, 8 -> 16 per group)
]


def test_group_packing_helpers_roundtrip():
    from kantts._hip import ops

    assert ops._group_pack(16, 8, 16) == 4 and ops._group_pack(16, 16, 32) == 2 and ops._group_pack(16, 16, 8) == 4
    assert ops._group_pack(1, 8, 8) == 1 and ops._group_pack(4, 32, 32) == 1 and ops._group_pack(16, 64, 64) == 1
    assert ops._group_pack(3, 8, 8) == 1 and ops._group_pack(6, 8, 8) == 2  # P must divide the group count
    g = torch.Generator().manual_seed(0)
    K, groups, NG, CR, P = 3, 8, 16, 8, 4
    w = torch.randn(K, groups * NG, CR, generator=g)
    wp = ops._blockdiag_pack(w, groups, P)
    assert tuple(wp.shape) == (K, groups * NG, P * CR)
    assert torch.equal(ops._blockdiag_unpack(wp, groups, P), w)
    # dense contraction with the packed weight == grouped contraction with the original one
    x = torch.randn(5, groups * CR, generator=g)
    ref = torch.cat([x[:, gi * CR:(gi + 1) * CR] @ w[1, gi * NG:(gi + 1) * NG].T for gi in range(groups)], dim=1)
    got = torch.cat([x[:, m * P * CR:(m + 1) * P * CR] @ wp[1, m * P * NG:(m + 1) * P * NG].T for m in range(groups // P)], dim=1)
    assert torch.allclose(got, ref, atol=1e-5)
    # everything off the diagonal blocks is exactly zero
    assert int((wp != 0).sum()) == int((w != 0).sum())


def _win_case(case, device):
    from kantts._hip import ops

    B, T, Cin, Cout, K, stride, dil, pad, groups, il, ol, use_res = case
    g = torch.Generator().manual_seed(K * 1000 + Cin)
    x = torch.randn(B, T, Cin, generator=g).to(device).requires_grad_(True)
    w = (torch.randn(Cout, Cin // groups, K, generator=g) * (1.0 / (Cin // groups * K) ** 0.5)).to(device).requires_grad_(True)
    b = torch.randn(Cout, generator=g).to(device).requires_grad_(True)
    if stride == 1:
        Tout = T
    else:
        Tout = (T + 2 * pad - dil * (K - 1) - 1) // stride + 1
    res = torch.randn(B, Tout, Cout, generator=g).to(device).requires_grad_(True) if use_res else None
    y = ops.conv_cl(x, w, b, stride=stride, dilation=dil, pad=pad, Tout=Tout, groups=groups, in_leaky=il,
                    out_leaky=ol, res=res)
    xin = F.leaky_relu(x, il) if il is not None else x
    # the op pads ``pad`` on the left and whatever Tout requires on the right
    need = (Tout - 1) * stride + dil * (K - 1) + 1
    xp = F.pad(xin.transpose(1, 2), (pad, max(0, need - T - pad)))
    ref = F.conv1d(xp, w, b, stride=stride, dilation=dil, groups=groups)[..., :Tout]
    if ol is not None:
        ref = F.leaky_relu(ref, ol)
    ref = ref.transpose(1, 2)
    if res is not None:
        ref = ref + res
    cot = torch.randn(ref.shape, generator=g).to(device)
    ins = (x, w, b) + ((res,) if use_res else ())
    gy = torch.autograd.grad((y * cot).sum(), ins)
    gr = torch.autograd.grad((ref * cot).sum(), ins)
    return y.detach(), ref.detach(), gy, gr


def test_conv_win_emulated_matches_torch():
    with emulation():
        for case in _WIN_CASES[2:-1]:
            y, ref, gy, gr = _win_case(case, "cpu")
            assert_close(y, ref, 2e-5, what=str(case))
            for a, c in zip(gy, gr):
                assert rel_l2(a, c) < 1e-5, case


@pytest.mark.gpu
@pytest.mark.parametrize("prec,otol,gtol", [("fp32", 5e-5, 2e-4), ("bf16", 4e-2, 6e-2), ("ref", 5e-5, 2e-4)])
def test_conv_win_gpu_matches_torch(prec, otol, gtol):
    """LDS-window convolution kernel (csrc/conv_win.hip) forward + input gradient, and the GEMM weight gradient,
    against ATen's conv1d on the same device; 'ref' routes through the scalar reference GEMM instead."""
    import kantts._hip as hip

    hip.set_precision(prec)
    try:
        for case in _WIN_CASES:
            y, ref, gy, gr = _win_case(case, "cuda")
            assert float((y - ref).abs().max()) <= otol * max(1.0, float(ref.abs().max())), (prec, case)
            for a, c in zip(gy, gr):
                assert rel_l2(a, c) < gtol, (prec, case)
    finally:
        hip.set_precision("fp32")


@pytest.mark.gpu
def test_hifigan_gpu_matches_oracle():
    import kantts._hip as hip

    hip.set_precision("fp32")
    _check_models("cuda", channels=64, B=2, frames=8, T_wav=2048, gtol=2e-3, wtol=1e-5)


@pytest.mark.gpu
def test_conv_ops_gpu_vs_emulated():
    from kantts._hip import ops

    B, T = 3, 37

    def r(*s, seed, scale=1.0, grad=True):
        g = torch.Generator().manual_seed(seed)
        return (torch.randn(*s, generator=g) * scale).requires_grad_(grad)

    cases = [
        ("causal dilated + res", lambda x, w, b, res: ops.conv_cl(x, w, b, dilation=3, pad=12, in_leaky=0.1, res=res),
         (r(B, T, 24, seed=1), r(40, 24, 5, seed=2, scale=0.2), r(40, seed=3), r(B, T, 40, seed=4))),
        ("strided grouped", lambda x, w, b: ops.conv_cl(x, w, b, stride=4, pad=20, groups=4, out_leaky=0.1),
         (r(B, 64, 32, seed=5), r(64, 8, 41, seed=6, scale=0.1), r(64, seed=7))),
        ("period fold", lambda x, w, b: ops.conv_cl(x, w, b, stride=3, pad=2, inner=5, out_leaky=0.1),
         (r(B, 20, 5, 8, seed=8), r(16, 8, 5, seed=9, scale=0.2), r(16, seed=10))),
        ("period fold, one input channel", lambda x, w, b: ops.conv_cl(x, w, b, stride=3, pad=2, inner=11, out_leaky=0.1),
         (r(2, 100, 11, 1, seed=24), r(32, 1, 5, seed=25, scale=0.3), r(32, seed=26))),
        ("period fold, long", lambda x, w, b: ops.conv_cl(x, w, b, stride=3, pad=2, inner=7, in_leaky=0.1, out_leaky=0.1),
         (r(2, 301, 7, 32, seed=21), r(128, 32, 5, seed=22, scale=0.1), r(128, seed=23))),
        ("nearest upsample", lambda x, w, b: ops.conv_cl(x, w, b, pad=6, up=8, in_leaky=0.1),
         (r(B, T, 12, seed=11), r(20, 12, 7, seed=12, scale=0.2), r(20, seed=13))),
        ("polyphase transposed", lambda x, w, b, res: ops.conv_transpose_cl(x, w, b, 8, in_leaky=0.1, res=res),
         (r(B, T, 16, seed=14), r(16, 12, 16, seed=15, scale=0.2), r(12, seed=16), r(B, T * 8, 12, seed=17))),
        ("weight norm", lambda v, g: ops.weight_norm(v, g), (r(33, 7, 5, seed=18), r(33, 1, 1, seed=19))),
        ("sin add", lambda x: ops.sin_add(x), (r(5, 999, seed=20),)),
    ]
    for name, fn, args in cases:
        go, gg, co, cg = run_both(fn, *args)
        assert_close(go[0], co[0], 5e-5, what=name)
        for a, c in zip(gg, cg):
            assert rel_l2(a, c) < 2e-4, name


def _gan_losses_check(device, channels, B, frames):
    """Generator / discriminator loss values and gradients of one GAN step vs the oracle
    (hifigan_oracle + audio_oracle), reference weights 45 / 1 / 2 (hifigan_v1_16k.yaml:115-158)."""
    import audio_oracle as A
    from kantts.train.gan_step import discriminator_loss, generator_loss
    from kantts.train.loss import (DiscriminatorAdversarialLoss, FeatureMatchLoss, GeneratorAdversarialLoss,
                                   MelSpectrogramLoss)

    G, D1, D2 = _models(channels)
    PG = {k: v.detach().clone().requires_grad_(True) for k, v in G.state_dict().items()}
    P1 = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in D1.state_dict().items()}
    P2 = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in D2.state_dict().items()}
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, 80, frames, generator=g)
    y = torch.randn(B, 1, frames * 256, generator=g).clamp(-1, 1)
    model = {"generator": G.to(device), "discriminator": {"MultiScaleDiscriminator": D2.to(device),
                                                           "MultiPeriodDiscriminator": D1.to(device)}}
    crit = {"mel_loss": MelSpectrogramLoss().to(device), "generator_adv_loss": GeneratorAdversarialLoss(),
            "discriminator_adv_loss": DiscriminatorAdversarialLoss(), "feat_match_loss": FeatureMatchLoss()}
    for k, w in (("mel_loss", 45.0), ("generator_adv_loss", 1.0), ("discriminator_adv_loss", 1.0),
                 ("feat_match_loss", 2.0)):
        crit[k].weights = w
    gen_loss, losses, _ = generator_loss(model, crit, x.to(device), y.to(device))
    gen_loss.backward()
    # oracle
    y_ = H.generator(PG, x)
    mel_l = (A.mel_spectrogram(y_[:, 0]) - A.mel_spectrogram(y[:, 0])).abs().mean()
    adv, fm = 0.0, 0.0
    for P, f in ((P2, H.msd), (P1, H.mpd)):
        o_h, f_h = f(P, y_)
        adv = adv + H.gen_adv_loss(o_h)
        with torch.no_grad():
            _, f_r = f(P, y)
        fm = fm + H.feat_match_loss(f_r, [[t.detach() for t in fl] for fl in f_h])
    ref = 45.0 * mel_l + adv + 2.0 * fm
    ref.backward()
    assert abs(float(losses["mel_loss"]) - float(mel_l)) < 1e-4
    assert abs(float(losses["adversarial_loss"]) - float(adv)) < 1e-4 * max(1.0, float(adv))
    assert abs(float(gen_loss) - float(ref)) < 2e-4 * max(1.0, abs(float(ref)))
    assert_grads_close([(n, p.grad, PG[n].grad) for n, p in model["generator"].named_parameters()], 5e-3, "G")
    for d in model["discriminator"].values():
        d.zero_grad()
    dis_loss, _ = discriminator_loss(model, crit, x.to(device), y.to(device))
    dis_loss.backward()
    for P in (P1, P2):
        for v in P.values():
            v.grad = None
    ref_d = 0.0
    with torch.no_grad():
        y2 = H.generator(PG, x)
    for P, f in ((P2, H.msd), (P1, H.mpd)):
        real, fake = H.dis_adv_loss(f(P, y2)[0], f(P, y)[0])
        ref_d = ref_d + real + fake
    ref_d.backward()
    assert abs(float(dis_loss) - float(ref_d)) < 1e-4 * max(1.0, float(ref_d))
    for P, key in ((P1, "MultiPeriodDiscriminator"), (P2, "MultiScaleDiscriminator")):
        assert_grads_close([(n, p.grad, P[n].grad) for n, p in model["discriminator"][key].named_parameters()],
                           5e-3, key)


def test_gan_step_losses_emulated():
    with emulation():
        _gan_losses_check("cpu", channels=16, B=1, frames=8)


@pytest.mark.gpu
def test_gan_step_losses_gpu():
    import kantts._hip as hip

    hip.set_precision("fp32")
    _gan_losses_check("cuda", channels=64, B=2, frames=16)


@pytest.mark.gpu
def test_gan_train_step_runs_and_updates():
    """Full step through model_builder / ArenaAdam / MultiStepLR at a reduced width: parameters move, losses finite."""
    import kantts._hip as hip
    from kantts.models import model_builder
    from kantts.train.gan_step import gan_train_step
    from kantts.train.loss import criterion_builder

    hip.set_precision("fp32")
    opt = {"type": "Adam", "params": {"lr": 2e-4, "betas": [0.5, 0.9], "weight_decay": 0.0}}
    sch = {"type": "MultiStepLR", "params": {"gamma": 0.5, "milestones": [200000]}}
    config = {"model_type": "hifigan", "Model": {
        "Generator": {"params": {"channels": 64}, "optimizer": opt, "scheduler": sch},
        "MultiScaleDiscriminator": {"params": {}, "optimizer": opt, "scheduler": sch},
        "MultiPeriodDiscriminator": {"params": {}, "optimizer": opt, "scheduler": sch}},
        "Loss": {"generator_adv_loss": {"enable": True, "params": {}, "weights": 1.0},
                 "discriminator_adv_loss": {"enable": True, "params": {}, "weights": 1.0},
                 "stft_loss": {"enable": False},
                 "mel_loss": {"enable": True, "params": {}, "weights": 45.0},
                 "feat_match_loss": {"enable": True, "params": {}, "weights": 2.0}},
        "generator_grad_norm": -1, "discriminator_grad_norm": -1, "discriminator_train_start_steps": 0,
        "generator_train_start_steps": 0}
    torch.manual_seed(0)
    model, optimizer, scheduler = model_builder(config, device="cuda")
    crit = criterion_builder(config, device="cuda")
    x = torch.randn(2, 80, 16, device="cuda")
    y = torch.randn(2, 1, 4096, device="cuda").clamp(-1, 1)
    w0 = optimizer["generator"].arena.flat.clone()
    for it in range(2):
        out = gan_train_step(model, optimizer, scheduler, crit, config, y, x, steps=1)
    assert all(torch.isfinite(torch.as_tensor(float(v))) for v in out.values())
    assert float((optimizer["generator"].arena.flat - w0).abs().max()) > 0


def _small_gan_setup(seed=0):
    from kantts.models import model_builder
    from kantts.train.loss import criterion_builder

    opt = {"type": "Adam", "params": {"lr": 2e-4, "betas": [0.5, 0.9], "weight_decay": 0.0}}
    sch = {"type": "MultiStepLR", "params": {"gamma": 0.5, "milestones": [2]}}  # the lr halves after the second step
    config = {"model_type": "hifigan", "Model": {
        "Generator": {"params": {"channels": 64}, "optimizer": opt, "scheduler": sch},
        "MultiScaleDiscriminator": {"params": {}, "optimizer": opt, "scheduler": sch},
        "MultiPeriodDiscriminator": {"params": {}, "optimizer": opt, "scheduler": sch}},
        "Loss": {"generator_adv_loss": {"enable": True, "params": {}, "weights": 1.0},
                 "discriminator_adv_loss": {"enable": True, "params": {}, "weights": 1.0},
                 "stft_loss": {"enable": False},
                 "mel_loss": {"enable": True, "params": {}, "weights": 45.0},
                 "feat_match_loss": {"enable": True, "params": {}, "weights": 2.0}},
        "generator_grad_norm": -1, "discriminator_grad_norm": -1, "discriminator_train_start_steps": 0,
        "generator_train_start_steps": 0}
    torch.manual_seed(seed)
    model, optimizer, scheduler = model_builder(config, device="cuda")
    crit = criterion_builder(config, device="cuda")
    return config, model, optimizer, scheduler, crit


@pytest.mark.gpu
@pytest.mark.parametrize("prec,tol", [("fp32", 2e-4), ("bf16", 2e-3)])
def test_graphed_gan_step_matches_eager_gpu(prec, tol):
    """The captured GAN step (kantts/train/gan_graph_step.py: both phases, eight branch streams, three Adam updates in
    one hipGraph) against the eager step from the same initial state over four steps with a learning-rate milestone in
    between: same losses, same weights.  Differences come from the fp32 atomics of the split weight gradients (summation
    order) only; in bf16 mode the cconv kernels (forced on at this small size) are inside the capture."""
    import kantts._hip as hip
    from kantts._hip import ops
    from kantts.train.gan_graph_step import GraphedGanStep
    from kantts.train.gan_step import gan_train_step

    hip.set_precision(prec)
    old_thr = ops.CCONV_MIN_FLOPS
    ops.CCONV_MIN_FLOPS = 0.0
    try:
        g = torch.Generator().manual_seed(5)
        xs = [torch.randn(2, 80, 16, generator=g).cuda() for _ in range(4)]
        ys = [torch.randn(2, 1, 4096, generator=g).clamp(-1, 1).cuda() for _ in range(4)]
        config, model, optimizer, scheduler, crit = _small_gan_setup()
        eager_losses = []
        for x, y in zip(xs, ys):
            out = gan_train_step(model, optimizer, scheduler, crit, config, y, x, steps=1)
            eager_losses.append({k: float(v.detach()) for k, v in out.items()})
        eager_w = [optimizer["generator"].arena.flat.clone()] + [o.arena.flat.clone() for o in optimizer["discriminator"].values()]
        eager_lr = optimizer["generator"].param_groups[0]["lr"]

        config, model, optimizer, scheduler, crit = _small_gan_setup()
        step = GraphedGanStep(model, optimizer, scheduler, crit, config, ys[0], xs[0], steps=1)
        for i, (x, y) in enumerate(zip(xs, ys)):
            step.load_batch(y, x)
            out = step()
            got = {k: float(v.detach()) for k, v in out.items()}
            for k, v in eager_losses[i].items():
                assert abs(got[k] - v) <= tol * max(1.0, abs(v)), (prec, i, k, got[k], v)
        graph_w = [optimizer["generator"].arena.flat] + [o.arena.flat for o in optimizer["discriminator"].values()]
        assert optimizer["generator"].param_groups[0]["lr"] == eager_lr
        assert optimizer["generator"]._step == 4
        for a, b in zip(graph_w, eager_w):
            assert rel_l2(a, b) <= tol, prec
    finally:
        ops.CCONV_MIN_FLOPS = old_thr
        hip.set_precision("fp32")


def _noncausal_and_spectral(device):
    """causal=False generator (symmetric conv padding, ConvTranspose1d with padding (k-s)/2) vs the oracle, and the
    follow_official_norm discriminators (torch spectral_norm holders) vs the oracle fed with weight / sigma."""
    from kantts.models.hifigan.hifigan import Generator, MultiScaleDiscriminator

    torch.manual_seed(3)
    G = Generator(channels=32, causal=False)
    PG = {k: v.detach().clone().requires_grad_(True) for k, v in G.state_dict().items()}
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 80, 5, generator=g)
    G = G.to(device)
    yo = G(x.to(device))
    yr = H.generator(PG, x, causal=False)
    assert yo.shape == yr.shape == (2, 1, 5 * 256)
    assert_close(yo.detach().cpu(), yr.detach(), 2e-5, what="non-causal generator")
    cot = torch.randn(yr.shape, generator=g)
    (yo * cot.to(device)).sum().backward()
    (yr * cot).sum().backward()
    assert_grads_close([(n, p.grad, PG[n].grad) for n, p in G.named_parameters()], 2e-3, "non-causal G")
    # spectral norm: eval mode (no power iteration) -> weight = weight_orig / (u^T W v)
    D = MultiScaleDiscriminator(follow_official_norm=True).eval()
    sd = D.state_dict()
    assert "discriminators.0.convs.0.0.weight_orig" in sd and "discriminators.1.convs.0.0.weight_g" in sd
    P = {}
    for k, v in sd.items():
        if k.endswith("weight_orig"):
            pre = k[: -len("weight_orig")]
            W = v.flatten(1)
            sigma = torch.dot(sd[pre + "weight_u"], W @ sd[pre + "weight_v"])
            P[pre + "weight"] = v / sigma
        elif not (k.endswith("weight_u") or (k.endswith("weight_v") and k[: -len("weight_v")] + "weight_orig" in sd)):
            P[k] = v
    y = torch.randn(2, 1, 1024, generator=g).clamp(-1, 1)
    with torch.no_grad():
        o, fm = D.to(device)(y.to(device))
        o_r, f_r = H.msd(P, y)
    for a, b in zip(o, o_r):
        assert rel_l2(a.cpu(), b) < 1e-4, "spectral-norm MSD output"  # (1/sigma makes the untrained outputs huge)


def test_noncausal_generator_and_spectral_norm_emulated():
    with emulation():
        _noncausal_and_spectral("cpu")


@pytest.mark.gpu
def test_noncausal_generator_and_spectral_norm_gpu():
    import kantts._hip as hip

    hip.set_precision("fp32")
    _noncausal_and_spectral("cuda")


def _v1_fixture():
    import os

    from util import GOLDEN

    return torch.load(os.path.join(GOLDEN, "hifigan_v1.pt"), weights_only=False)


def _v1_models(fix):
    """V1 class defaults from the fixture's seed; the seeded weights must be the reference's (checksums)."""
    G, D1, D2 = _models(512, seed=0)
    for m, key in ((G, "G_checksums"), (D1, "mpd_checksums"), (D2, "msd_checksums")):
        sd = m.state_dict()
        assert list(sd.keys()) == list(fix[key].keys())
        for k, (shape, s_, a_) in fix[key].items():
            assert tuple(sd[k].shape) == tuple(shape), k
            assert abs(float(sd[k].double().sum()) - s_) <= 1e-9 * max(1.0, a_), k
    return G, D1, D2


def test_oracle_matches_reference_v1_fixture():
    """oracle/hifigan_oracle.py == the untouched reference at HiFi-GAN V1 (512 ch, MPD x5, MSD x3): outputs recorded by
    oracle/make_golden.py::hifigan_v1_case."""
    fix = _v1_fixture()
    G, D1, D2 = _v1_models(fix)
    PG = {k: v.detach().clone().requires_grad_(True) for k, v in G.state_dict().items()}
    wav = H.generator(PG, fix["x"])
    assert_close(wav.detach(), fix["wav"], 2e-6, what="oracle wav")
    (wav * fix["cot"]).sum().backward()
    for n, ref in fix["G_grad_norms"].items():
        assert abs(float(PG[n].grad.double().norm()) - ref) <= 1e-3 * ref + 1e-7, n  # a LeakyReLU flip moves a small bias gradient by ~1e-4
    for D, f, nm in ((D1, H.mpd, "mpd"), (D2, H.msd, "msd")):
        o, fm = f({k: v for k, v in D.state_dict().items()}, fix["y"])
        for a, b in zip(o, fix[nm + "_out"]):
            assert_close(a, b, 2e-6, what=nm)
        for fa, fb in zip(fm, fix[nm + "_fmap_sums"]):
            for a, (shape, s_, a_) in zip(fa, fb):
                assert tuple(a.shape) == tuple(shape)
                assert abs(float(a.double().sum()) - s_) <= 1e-5 * max(1.0, a_)


@pytest.mark.gpu
def test_hifigan_v1_gpu_matches_reference_fixture():
    """HIP path (fp32 mode) at the V1 class defaults against outputs recorded from the reference itself."""
    import kantts._hip as hip

    hip.set_precision("fp32")
    fix = _v1_fixture()
    G, D1, D2 = _v1_models(fix)
    G = G.cuda()
    wav = G(fix["x"].cuda())
    d = (wav.detach().cpu() - fix["wav"]).abs()
    assert float(d.mean()) <= 1e-5 and float(d.max()) <= 2e-4, (float(d.mean()), float(d.max()))
    (wav * fix["cot"].cuda()).sum().backward()
    for n, p in G.named_parameters():
        ref = fix["G_grad_norms"][n]
        assert abs(float(p.grad.double().norm()) - ref) <= 2e-3 * ref + 1e-6, n
    for D, nm in ((D1, "mpd"), (D2, "msd")):
        D = D.cuda()
        o, fm = D(fix["y"].cuda())
        for a, b in zip(o, fix[nm + "_out"]):
            assert_close(a.detach().cpu(), b, 2e-5, what=nm)
        for fa, fb in zip(fm, fix[nm + "_fmap_sums"]):
            for a, (shape, s_, a_) in zip(fa, fb):
                assert tuple(a.shape) == tuple(shape)
                assert abs(float(a.double().sum()) - s_) <= 2e-4 * max(1.0, a_)


@pytest.mark.gpu
@pytest.mark.parametrize("Cin,Cout,s,T", [(64, 32, 2, 4096), (128, 64, 2, 2048), (128, 64, 2, 37), (256, 128, 8, 256),
                                            (512, 256, 8, 32)])
def test_upsample_streaming_kernels_gpu(Cin, Cout, s, T):
    """csrc/upsample.hip + the two-segment bf16 contraction against torch's conv_transpose1d on the same bf16-rounded
    operands (causal trim), with bias and residual; bf16 and fp32 outputs; sequence starts inside a 16-token tile."""
    import kantts._hip as hip
    from kantts._hip import ops

    hip.set_precision("bf16")
    try:
        g = torch.Generator().manual_seed(Cin + T)
        B = 3
        x = torch.randn(B, T, Cin, generator=g)
        w = torch.randn(Cin, Cout, 2 * s, generator=g) * (1.0 / (2 * Cin) ** 0.5)
        b = torch.randn(Cout, generator=g)
        res = torch.randn(B, T * s, Cout, generator=g)
        h, act = ops.sin_add(x.cuda(), act_slope=0.1)
        assert_close(h.cpu(), torch.sin(x) + x, 1e-6, what="sin_add")
        a = F.leaky_relu(torch.sin(x) + x, 0.1).to(torch.bfloat16)
        assert float((act.cpu().float() - a.float()).abs().max()) <= 2e-2  # one bf16 ulp where sinf differs in the last bit
        a = act.cpu()
        wq = w.to(torch.bfloat16).float()
        ref = F.conv_transpose1d(a.float().transpose(1, 2), wq, b, stride=s)[:, :, :T * s].transpose(1, 2)
        y = ops.upsample_forward(act, w.cuda(), b.cuda(), s, res=res.cuda())
        assert y is not None and y.dtype == torch.float32
        assert rel_l2(y.cpu(), ref + res) <= 2e-5
        if Cin <= 128:
            y2 = ops.upsample_forward(act, w.cuda(), b.cuda(), s, out_bf16=True)
            assert y2.dtype == torch.bfloat16 and rel_l2(y2.cpu().float(), ref) <= 4e-3
            hb = h.to(torch.bfloat16)
            a3 = F.leaky_relu(hb.cpu().float(), 0.1).to(torch.bfloat16).float()
            ref3 = F.conv_transpose1d(a3.transpose(1, 2), wq, b, stride=s)[:, :, :T * s].transpose(1, 2)
            y3 = ops.upsample_forward(hb, w.cuda(), b.cuda(), s, out_bf16=True, in_slope=0.1)
            assert rel_l2(y3.cpu().float(), ref3) <= 4e-3
    finally:
        hip.set_precision("fp32")


def test_branch_exit_is_an_identity_with_a_private_gradient():
    """ops._BranchExit (end of a branch of ops.parallel_branches(private_grads=True)): identity forward; backward returns
    a COPY of the incoming gradient, so in-place accumulation further down a branch never touches the tensor its sibling
    branches still read (the cross-stream race of DESIGN section 5)."""
    from kantts._hip import ops

    x = torch.randn(3, 5, requires_grad=True)
    y = ops._BranchExit.apply(x)
    assert torch.equal(y, x) and y.data_ptr() == x.data_ptr()
    seen = {}

    class Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.view_as(t)

        @staticmethod
        def backward(ctx, g):
            seen["in"] = g
            return g

    g0 = torch.randn(3, 5)
    out = ops._private_grad([Probe.apply(x), "not a tensor"])
    assert isinstance(out, list) and out[1] == "not a tensor"
    y2 = ops._BranchExit.apply(x)
    hooked = {}
    y2.register_hook(lambda g: hooked.setdefault("g", g))
    x.grad = None
    y2.backward(g0)
    assert torch.equal(x.grad, g0) and x.grad.data_ptr() != hooked["g"].data_ptr()


def test_multiscale_discriminator_with_average_pooling_matches_the_reference_fixture():
    """downsample_pooling other than "DWT" (no shipped yaml): the reference's AvgPool1d(4, 2, padding=2) pair without
    auxiliary convolutions (hifigan.py:456-471) -- same state_dict keys, seeded weights, outputs, feature-map sums and
    input gradient as recorded from the reference (tests/golden/msd_avgpool.pt, make_golden.py::msd_avgpool_case)."""
    from kantts.models.hifigan.hifigan import MultiScaleDiscriminator

    import os

    from util import GOLDEN

    fix = torch.load(os.path.join(GOLDEN, "msd_avgpool.pt"), weights_only=False)
    torch.manual_seed(21)
    msd = MultiScaleDiscriminator(scales=3, downsample_pooling="AvgPool1d", discriminator_params=fix["discriminator_params"])
    sd = msd.state_dict()
    assert sorted(sd.keys()) == fix["keys"]
    for k, (shape, s_, a_) in fix["checksums"].items():
        assert tuple(sd[k].shape) == shape and abs(float(sd[k].double().sum()) - s_) <= 1e-6 * max(1.0, a_), k
    x = fix["x"].clone().requires_grad_(True)
    with emulation():
        outs, fmaps = msd(x)
        sum(o.pow(2).mean() for o in outs).backward()
    for o, w in zip(outs, fix["outs"]):
        assert_close(o.detach(), w, 2e-6, what="MSD output")
    for fm, ws in zip(fmaps, fix["fmap_sums"]):
        for f, (shape, s_, a_) in zip(fm, ws):
            assert tuple(f.shape) == shape and abs(float(f.double().sum()) - s_) <= 2e-5 * max(1.0, a_)
    assert rel_l2(x.grad, fix["dx"]) <= 1e-5


@pytest.mark.parametrize("name", ["causal", "noncausal"])
def test_generator_with_relu_activation_matches_the_reference_fixture(name):
    """nonlinear_activation="ReLU" (no shipped yaml): every convolution kernel applies its activation itself, parametrised
    by one slope, and ReLU is slope 0 -- output and per-parameter gradient norms as recorded from the reference
    (tests/golden/hifigan_relu.pt, make_golden.py::relu_generator_case); other activations are refused."""
    import os

    from kantts.models.hifigan.hifigan import Generator
    from util import GOLDEN

    fix = torch.load(os.path.join(GOLDEN, "hifigan_relu.pt"), weights_only=False)[name]
    torch.manual_seed(5)
    G = Generator(in_channels=80, channels=32, upsample_scales=[4, 4, 2, 2], upsample_kernal_sizes=[8, 8, 4, 4],
                  causal=(name == "causal"), nonlinear_activation="ReLU", nonlinear_activation_params={})
    sd = G.state_dict()
    for k, (shape, s_, a_) in fix["weight_checksums"].items():
        assert tuple(sd[k].shape) == shape and abs(float(sd[k].double().sum()) - s_) <= 1e-6 * max(1.0, a_), k
    with emulation():
        y = G(fix["x"])
        (y * fix["cot"]).sum().backward()
    assert_close(y.detach(), fix["y"], 2e-6, what="waveform")
    for n, p in G.named_parameters():
        if n in fix["grad_norms"]:
            w = fix["grad_norms"][n]
            assert abs(float(p.grad.double().norm()) - w) <= 2e-4 * max(w, 1e-3), n
    with pytest.raises(NotImplementedError):
        Generator(in_channels=80, channels=32, nonlinear_activation="ELU", nonlinear_activation_params={})


# ---- weight-norm images of a whole network from one table-driven launch (ParamArena.build_weight_norm_images) ----------
def _wn_table_check(device, count_calls=None, train=True):
    """Every layer's slice of the arena's image buffers equals what the per-layer entry point
    (kantts_weight_norm_tap_images) writes for that layer -- fp32 weight and both bf16 images, bit for bit (same
    arithmetic, same summation order per row); two training steps with the table equal two steps without it; and the
    table is launched once per network and optimizer step, not once per layer and forward pass."""
    import kantts._hip as hip
    from kantts._hip import ops
    from kantts.models import hifigan_model_builder
    from kantts.train.gan_step import gan_train_step
    from kantts.train.loss import criterion_builder

    opt = {"type": "Adam", "params": {"lr": 2e-3, "betas": [0.5, 0.9], "weight_decay": 0.0}}
    sch = {"type": "MultiStepLR", "params": {"gamma": 0.5, "milestones": [200000]}}
    config = {"model_type": "hifigan", "Model": {
        "Generator": {"params": {"channels": 32, "upsample_scales": [4, 4], "upsample_kernal_sizes": [8, 8],
                                 "resblock_kernel_sizes": [3, 7], "resblock_dilations": [[1, 3, 5], [1, 3, 5]]},
                       "optimizer": opt, "scheduler": sch},
        "MultiScaleDiscriminator": {"params": {"scales": 2}, "optimizer": opt, "scheduler": sch},
        "MultiPeriodDiscriminator": {"params": {"periods": [2, 3]}, "optimizer": opt, "scheduler": sch}},
        "Loss": {"generator_adv_loss": {"enable": True, "params": {}, "weights": 1.0},
                 "discriminator_adv_loss": {"enable": True, "params": {}, "weights": 1.0},
                 "mel_loss": {"enable": True, "params": {"fft_size": 256, "hop_size": 64, "win_length": 256}, "weights": 45.0},
                 "feat_match_loss": {"enable": True, "params": {}, "weights": 2.0}},
        "generator_grad_norm": -1, "discriminator_grad_norm": -1, "discriminator_train_start_steps": 0,
        "generator_train_start_steps": 0}

    def build():
        torch.manual_seed(3)
        model, optimizer, scheduler = hifigan_model_builder(config, device, 0, False, use_arena=True)
        return model, optimizer, scheduler, criterion_builder(config, device=device)

    hip.set_precision("bf16")
    try:
        model, optimizer, scheduler, crit = build()
        n_layers = 0
        for o in [optimizer["generator"], *optimizer["discriminator"].values()]:
            arena = o.arena
            assert arena._wn_table is not None
            assert arena.refresh_weight_norm_images(force=True)
            for m in arena.module.modules():
                wn = getattr(m, "_kantts_wn", None)
                if wn is None:
                    continue
                n_layers += 1
                v = m.weight_v.detach()
                v = v.squeeze(-1) if v.dim() == 4 else v
                ref = ops.weight_norm_tap(v, m.weight_g.detach(), groups=wn[4])
                rf, rd = ops.weight_images(ref, wn[4])
                if wn[2] is not None:  # same kernel body per row: bit for bit
                    assert torch.equal(ref, wn[1]), type(m)
                    assert rf is not None and torch.equal(rf, wn[2]) and torch.equal(rd, wn[3])
                else:  # layers without bf16 images take kantts_weight_norm_strided_fwd per layer (another summation order)
                    assert rf is None and float((ref - wn[1]).abs().max()) <= 1e-6 * float(ref.abs().max())
        assert n_layers >= 30
        if not train:
            return None
        g = torch.Generator().manual_seed(1)
        x = torch.randn(2, 80, 8, generator=g).to(device)
        y = torch.randn(2, 1, 128, generator=g).clamp(-1, 1).to(device)
        res = {}
        # third run (device only): the table arm again -- the weight gradients' fp32 atomics order differently from run to
        # run and two Adam steps amplify that; what the two arms may differ by is measured, not assumed
        for table in (True, False) + (("again",) if device != "cpu" else ()):
            if not table:
                os.environ["KANTTS_NO_WEIGHT_NORM_TABLE"] = "1"
            try:
                model, optimizer, scheduler, crit = build()
                if count_calls is not None:
                    count_calls.clear()
                losses = []
                for it in range(2):
                    out = gan_train_step(model, optimizer, scheduler, crit, config, y, x, steps=1)
                    losses.append({k: float(v) for k, v in out.items()})
                res[table] = (losses, [o.arena.flat.clone() for o in [optimizer["generator"], *optimizer["discriminator"].values()]],
                              None if count_calls is None else dict(count_calls))
            finally:
                os.environ.pop("KANTTS_NO_WEIGHT_NORM_TABLE", None)
        floor_l = floor_w = 0.0
        if "again" in res:
            floor_l = max(abs(a[k] - b[k]) / max(1.0, abs(b[k])) for a, b in zip(res[True][0], res["again"][0]) for k in a)
            floor_w = max(rel_l2(a, b) for a, b in zip(res[True][1], res["again"][1]))
        for a, b in zip(res[True][0], res[False][0]):
            for k in a:
                assert abs(a[k] - b[k]) <= max(1e-5, 3 * floor_l) * max(1.0, abs(b[k])), (k, a[k], b[k], floor_l)
        # weights after two Adam steps: the two arms' weights of the layers WITHOUT bf16 images differ in the last bit
        # (another summation order per row, asserted above at 1e-6), which now and then flips one LeakyReLU gate or one
        # sign of an L1 gradient somewhere -- a discrete event that Adam (lr 2e-3, |update| = lr whatever the gradient's
        # size) turns into 7.8e-5 of one arena's norm, always the same value, in about one run in five on the device
        for a, b in zip(res[True][1], res[False][1]):
            assert rel_l2(a, b) <= max(3e-4, 3 * floor_w), (rel_l2(a, b), floor_w)
        return res
    finally:
        hip.set_precision("fp32")


def test_weight_norm_table_equals_per_layer_launches_emulated(emulated_cabi, monkeypatch):
    import os  # noqa: F401

    counts = {}
    for name in ("kantts_weight_norm_table", "kantts_weight_norm_tap_images", "kantts_weight_norm_table_bwd"):
        orig = getattr(emulated_cabi, name)
        monkeypatch.setattr(emulated_cabi, name, (lambda o, n: lambda *a: (counts.__setitem__(n, counts.get(n, 0) + 1), o(*a))[1])(orig, name),
                            raising=False)
    res = _wn_table_check("cpu", counts)
    with_table, without = res[True][2], res[False][2]
    # two steps, three networks: step 1 refreshes G, both discriminators, G again after its update; step 2 the
    # discriminators (updated at the end of step 1) and G again -> 7 launches, none per layer
    assert with_table.get("kantts_weight_norm_table", 0) == 7, with_table
    assert with_table.get("kantts_weight_norm_tap_images", 0) == 0, with_table
    # backward of the reparametrisation: one launch per network and backward pass (generator phase: the generator; the
    # discriminator phase: each discriminator) = 3 per step
    assert with_table.get("kantts_weight_norm_table_bwd", 0) == 6, with_table
    assert without.get("kantts_weight_norm_table_bwd", 0) == 0, without
    assert without.get("kantts_weight_norm_table", 0) == 0 and without.get("kantts_weight_norm_tap_images", 0) > 100, without


def test_weight_norm_table_images_equal_the_per_layer_kernel_emulated(emulated_cabi):
    """The image comparison alone (the variant tests/test_kernel_source_on_cpu.py runs on the kernel source)."""
    _wn_table_check("cpu", train=False)


def _wn_table_bwd_check(device):
    """kantts_weight_norm_table_bwd (every layer's dv / dg from one launch, written into the gradient arena) against the
    per-layer kantts_weight_norm_strided_bwd on the same weight gradients."""
    import kantts._hip as hip
    from kantts._hip import ops
    from kantts.models.hifigan.hifigan import MultiPeriodDiscriminator
    from kantts.train.optim import ArenaAdam, ParamArena

    torch.manual_seed(5)
    net = MultiPeriodDiscriminator(periods=[2, 3]).to(device)
    arena = ParamArena(net)
    assert arena.build_weight_norm_images() >= 8
    opt = ArenaAdam(arena, lr=1e-3)
    opt.zero_grad()  # arms the deferral
    g = torch.Generator().manual_seed(6)
    want = {}
    for m in net.modules():
        wn = getattr(m, "_kantts_wn", None)
        if wn is None:
            continue
        v = m.weight_v.detach()
        v3 = v.squeeze(-1) if v.dim() == 4 else v
        cout, cin, k = v3.shape
        dw = torch.randn(k, cout, cin, generator=g).to(device)
        dv, dg = torch.empty_like(v3), torch.empty_like(m.weight_g.detach())
        hip.check(hip.lib().kantts_weight_norm_strided_bwd(hip.ptr(dw), hip.ptr(v3.contiguous()), hip.ptr(m.weight_g.detach()),
                                                           hip.ptr(dv), hip.ptr(dg), cout, cin, k, cin, 1, cout * cin,
                                                           hip.stream()), "strided_bwd")
        want[id(m)] = (dv.reshape(v.shape).clone(), dg.clone())
        assert arena.defer_weight_norm_backward(wn[5], dw)
    ops.flush_weight_norm_backward()
    n = 0
    for m in net.modules():
        if id(m) in want:
            dv, dg = want[id(m)]
            assert rel_l2(arena.grad_slot(m.weight_v), dv) <= 1e-5 and rel_l2(arena.grad_slot(m.weight_g), dg) <= 1e-5
            n += 1
    assert n >= 8 and not arena._wn_pending


def test_weight_norm_table_backward_equals_per_layer_emulated(emulated_cabi):
    _wn_table_bwd_check("cpu")


def _wn_table_two_applications(device):
    """A weight-normed network applied TWICE between zero_grad and step (the reference's two-pass discriminator loss,
    kantts/train/trainer.py:540-560; gradient accumulation over two backward passes likewise): the table launch assigns a
    layer's gradient slot, so the second application's reparametrisation backward is added to it at the flush.  Against
    the same two passes with the deferral off (per-layer kernels, autograd's own accumulation)."""
    from kantts._hip import ops
    from kantts.models.hifigan.hifigan import MultiPeriodDiscriminator
    from kantts.train.optim import ArenaAdam, ParamArena

    g = torch.Generator().manual_seed(9)
    ya = (torch.randn(2, 1, 600, generator=g) * 0.3).to(device)
    yb = (torch.randn(2, 1, 600, generator=g) * 0.3).to(device)

    def run(defer, two_backwards):
        torch.manual_seed(5)
        net = MultiPeriodDiscriminator(periods=[2, 3]).to(device)
        arena = ParamArena(net)
        assert arena.build_weight_norm_images() >= 8
        opt = ArenaAdam(arena, lr=1e-3)
        opt.zero_grad()
        arena._wn_defer = defer
        la = sum((o ** 2).mean() for o in net(ya)[0])
        if two_backwards:
            la.backward()
            ops.wgrad_overlap.join()
            sum(((o - 1) ** 2).mean() for o in net(yb)[0]).backward()
        else:
            (la + sum(((o - 1) ** 2).mean() for o in net(yb)[0])).backward()
        ops.wgrad_overlap.join()  # flushes the deferred reparametrisation backward
        assert not arena._wn_pending and not arena._wn_extra
        return arena.pack_grads().clone(), (len(arena._wn_seen) if defer else 0)

    for two in (False, True):
        ref, _ = run(False, two)
        got, seen = run(True, two)
        assert seen >= 8
        assert float(ref.abs().max()) > 0
        assert rel_l2(got, ref) <= 1e-5, two


def test_weight_norm_table_backward_with_a_layer_applied_twice_emulated(emulated_cabi):
    _wn_table_two_applications("cpu")


@pytest.mark.gpu
def test_weight_norm_table_backward_with_a_layer_applied_twice_gpu():
    _wn_table_two_applications("cuda")


@pytest.mark.gpu
def test_weight_norm_table_backward_equals_per_layer_gpu():
    _wn_table_bwd_check("cuda")


@pytest.mark.gpu
def test_weight_norm_table_equals_per_layer_launches_gpu():
    _wn_table_check("cuda")


def _c1_persistent(device, monkeypatch):
    """conv_c1_wgrad_mfma_kernel walks several runs per workgroup (at most 512 workgroups per launch): with the cap forced
    down to 2 / 3 workgroups the 9 (MSD shape) and 4 (MPD shape) runs of these cases are spread unevenly -- same gradients
    as torch's."""
    for cap in ("2", "3", "512"):
        monkeypatch.setenv("KANTTS_C1_WGRAD_WGS", cap)
        for case in ((3, 700, 1, 128, 15, 1, 1, 7, 1, None, 0.1, False), (2, 333, 1, 32, 5, 3, 1, 2, 1, None, 0.1, False),
                     (5, 1100, 1, 64, 15, 1, 1, 7, 1, None, 0.1, False)):
            y, ref, gy, gr = _win_case(case, device)
            assert_close(y, ref, 5e-5, what=str(case))
            for a, c in zip(gy, gr):
                assert rel_l2(a, c) < 2e-4, (cap, case, rel_l2(a, c))


def _one_output_channel(device):
    """csrc/conv_n1.hip (conv_post of the generator / the sub-discriminators: one output channel = a dot product per output
    position) against torch: forward, input gradient, weight and bias gradient -- 1024 -> 1 with the period axis folded,
    32 -> 1 with k = 7 behind a LeakyReLU, a strided / dilated odd case, both weight layouts."""
    import torch.nn.functional as F

    from kantts._hip import ops

    g = torch.Generator().manual_seed(23)
    cases = [  # B, T, inner, Cin, K, stride, dil, pad, in_leaky, tap_major
        (2, 23, 3, 1024, 3, 1, 1, 1, None, False),
        (3, 301, 1, 32, 7, 1, 1, 6, 0.01, False),
        (2, 40, 2, 64, 5, 2, 2, 3, 0.1, True),
        (1, 9, 1, 1024, 3, 1, 1, 1, None, True),
        (2, 11, 1, 4, 8, 1, 1, 0, None, False),    # the narrowest input (one float4 per position), the longest filter
        (1, 5, 2, 128, 1, 1, 1, 0, 0.2, True),     # a 1-tap layer
        (1, 2, 1, 8, 3, 1, 1, 2, None, False),     # more padding than signal
    ]
    for (B, T, inner, Cin, K, stride, dil, pad, slope, tap) in cases:
        shape = (B, T, inner, Cin) if inner > 1 else (B, T, Cin)
        x = torch.randn(shape, generator=g).to(device).requires_grad_(True)
        wp = (torch.randn(1, Cin, K, generator=g) / (Cin * K) ** 0.5).to(device)
        w = (wp.permute(2, 0, 1).contiguous() if tap else wp).requires_grad_(True)
        b = torch.randn(1, generator=g).to(device).requires_grad_(True)
        Tout = (T + 2 * pad - dil * (K - 1) - 1) // stride + 1
        y = ops.conv_cl(x, w, b, stride=stride, dilation=dil, pad=pad, inner=inner, in_leaky=slope, tap_major=tap, Tout=Tout)
        xr = x.detach().clone().requires_grad_(True)
        wr = wp.clone().requires_grad_(True)
        br = b.detach().clone().requires_grad_(True)
        xa = F.leaky_relu(xr, slope) if slope is not None else xr
        xt = xa.reshape(B, T, inner, Cin).permute(0, 2, 3, 1).reshape(B * inner, Cin, T)
        ref = F.conv1d(xt, wr, br, stride=stride, padding=pad, dilation=dil)  # (B * inner, 1, Tout)
        ref = ref.reshape(B, inner, Tout).permute(0, 2, 1).reshape(y.shape)
        assert float((y - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max())), (Cin, K)
        cot = torch.randn(y.shape, generator=g).to(device)
        gy = torch.autograd.grad(y, [x, w, b], cot)
        gr = torch.autograd.grad(ref, [xr, wr, br], cot)
        assert rel_l2(gy[0], gr[0]) < 2e-5, ("dx", Cin, K, rel_l2(gy[0], gr[0]))
        gw = gy[1].permute(1, 2, 0) if tap else gy[1]
        assert rel_l2(gw, gr[1]) < 5e-5, ("dw", Cin, K, rel_l2(gw, gr[1]))
        assert abs(float(gy[2]) - float(gr[2])) <= 1e-4 * max(1.0, abs(float(gr[2]))), ("db", Cin, K)


def test_one_output_channel_convolution_emulated(emulated_cabi):
    _one_output_channel("cpu")


@pytest.mark.gpu
def test_one_output_channel_convolution_gpu():
    import kantts._hip as hip

    hip.set_precision("fp32")
    _one_output_channel("cuda")


def _c1_image(device):
    """bf16 mode: the 1-channel first layer of a sub-discriminator hands its consumer the bf16 image of its (activated)
    output from the same launch (kantts_conv_c1_args.y_bf16) -- bit for bit what the cast pass it replaces would write."""
    import kantts._hip as hip
    from kantts._hip import ops

    hip.set_precision("bf16")
    try:
        g = torch.Generator().manual_seed(12)
        for (B, T, inner, Cout, K, stride, pad) in ((2, 301, 1, 128, 15, 1, 7), (3, 90, 2, 32, 5, 3, 2)):
            x = torch.randn(B, T, inner, 1, generator=g).to(device) if inner > 1 else torch.randn(B, T, 1, generator=g).to(device)
            w = (torch.randn(Cout, 1, K, generator=g) * 0.3).to(device)
            b = torch.randn(Cout, generator=g).to(device)
            y = ops.conv_cl(x, w, b, stride=stride, pad=pad, inner=inner, out_leaky=0.1, image=None)
            img = ops.get_image(y, None)
            assert img is not None and img.dtype == torch.bfloat16 and img.shape == y.shape
            assert torch.equal(img, y.to(torch.bfloat16))
    finally:
        hip.set_precision("fp32")


def test_one_channel_layer_hands_over_its_bf16_image_emulated(emulated_cabi):
    _c1_image("cpu")


@pytest.mark.gpu
def test_one_channel_layer_hands_over_its_bf16_image_gpu():
    _c1_image("cuda")


def test_one_channel_weight_gradient_with_a_persistent_grid_emulated(emulated_cabi, monkeypatch):
    _c1_persistent("cpu", monkeypatch)


@pytest.mark.gpu
def test_one_channel_weight_gradient_with_a_persistent_grid_gpu(monkeypatch):
    import kantts._hip as hip

    hip.set_precision("fp32")
    _c1_persistent("cuda", monkeypatch)
