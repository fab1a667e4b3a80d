"""Pins oracle/torch_oracle.py (and the product's parameter construction) against fixtures dumped
from the UNTOUCHED reference by oracle/make_golden.py.  CPU only; travels to the GPU box."""
import os

import pytest
import torch

import torch_oracle as O
from util import GOLDEN, assert_close


def _load(name):
    return torch.load(os.path.join(GOLDEN, name + ".pt"), weights_only=False)


def _product_weights(fix):
    from kantts.models.sambert.kantts_sambert import KanTtsSAMBERT

    torch.manual_seed(fix["seed_w"])
    m = KanTtsSAMBERT(dict(fix["cfg"]))
    return m.state_dict()


@pytest.mark.parametrize("name", ["sambert_tiny", "sambert_tiny16"])
def test_state_dict_matches_reference(name):
    """Same keys, shapes and (seeded) values as the reference's KanTtsSAMBERT.state_dict()."""
    fix = _load(name)
    sd = _product_weights(fix)
    ref = fix["weight_checksums"]
    assert list(sd.keys()) == list(ref.keys())
    for k, v in sd.items():
        shape, s, a = ref[k]
        assert tuple(v.shape) == tuple(shape), k
        assert abs(float(v.double().sum()) - s) <= 1e-9 * max(1.0, abs(a)), k
        assert abs(float(v.double().abs().sum()) - a) <= 1e-9 * max(1.0, abs(a)), k


@pytest.mark.parametrize("name", ["sambert_tiny", "sambert_tiny16"])
def test_oracle_reproduces_reference_outputs(name):
    fix = _load(name)
    sd = _product_weights(fix)
    P = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    batch = O.synthetic_sambert_batch(**fix["batch_args"])
    out = O.sambert_forward(P, fix["cfg"], **batch)
    assert out["x_band_width"] == fix["x_band_width"] and out["h_band_width"] == fix["h_band_width"]
    assert torch.equal(out["LR_length_rounded"], fix["outputs"]["LR_length_rounded"])  # bit-exact
    for k, ref in fix["outputs"].items():
        if ref.is_floating_point():
            assert_close(out[k].detach(), ref, atol=2e-5, what=k)
    nb = 2
    assert_close(out["enc_slf_attn_lst"][0][:nb].detach(), fix["attn_checks"]["enc0"], 1e-6, what="enc attn")
    assert_close(out["pnca_x_attn_lst"][-1][:nb].detach(), fix["attn_checks"]["pnca_x_last"], 1e-6, what="x attn")
    assert_close(out["pnca_h_attn_lst"][-1][:nb].detach(), fix["attn_checks"]["pnca_h_last"], 1e-6, what="h attn")
    L = O.sambert_losses(out, batch["input_lengths"], batch["output_lengths"], batch["mel_targets"])
    for k, v in fix["losses"].items():
        assert abs(float(L[k]) - v) <= 2e-5 * max(1.0, abs(v)), k
    L["total"].backward()
    for k, g in fix["grads"].items():
        assert_close(P[k].grad, g, atol=1e-6 + 1e-4 * float(g.abs().max()), what="grad " + k)
    for k, (s, nrm) in fix["grad_summaries"].items():
        assert abs(float(P[k].grad.double().norm()) - nrm) <= 1e-3 * nrm + 1e-7, k


def test_audio_oracle_reproduces_reference():
    import audio_oracle as A

    fix = _load("melspec")
    x = fix["wav"]
    assert_close(A.mel_spectrogram(x), fix["mel_v1"], 2e-5, what="mel V1")
    assert_close(A.mel_spectrogram(x, 16000, 2048, 200, 1000, 80, 0, 8000), fix["mel_16k"], 2e-5, what="mel 16k")
    assert_close(A.stft_magnitude(x, 1024, 120, 600), fix["stft_1024_120_600"], 2e-5, what="stft")


def _infer_inputs(fix, B=None, seed=None):
    args = dict(fix["batch_args"])
    if B is not None:
        args["B"] = B
    if seed is not None:
        args["seed"] = seed
    batch = O.synthetic_sambert_batch(**args)
    return {k: batch[k] for k in ("inputs_ling", "inputs_emotion", "inputs_speaker", "input_lengths")}


def test_oracle_reproduces_reference_free_running_inference():
    """Free-running path (AR duration predictor, predicted-duration length regulation, AR decoder loop) of the
    oracle against the reference's own inference run (batch 1 -- the only size the reference can infer at)."""
    fix = _load("sambert_tiny_infer")
    sd = _product_weights(fix)
    sd["variance_adaptor.duration_predictor.fc.bias"] = torch.full_like(
        sd["variance_adaptor.duration_predictor.fc.bias"], fix["dur_bias"])
    with torch.no_grad():
        out = O.sambert_forward(dict(sd), fix["cfg"], **_infer_inputs(fix))
    assert out["x_band_width"] == fix["x_band_width"]
    assert torch.equal(out["LR_length_rounded"], fix["outputs"]["LR_length_rounded"])  # bit-exact frame counts
    for k, ref in fix["outputs"].items():
        if ref.is_floating_point():
            assert_close(out[k], ref, atol=2e-5, what=k)


def test_get_mask_from_lengths_bit_exact():
    """kantts.models.utils.get_mask_from_lengths and the oracle's pad_mask against masks recorded from the reference's
    own function (kantts/models/utils.py:13-23), with and without max_len; SeqInfo round-trips the lengths."""
    from kantts.models.utils import SeqInfo, get_mask_from_lengths

    for c in _load("masks"):
        m = get_mask_from_lengths(c["lengths"])
        assert m.dtype == torch.bool and torch.equal(m, c["mask"])
        assert torch.equal(get_mask_from_lengths(c["lengths"], max_len=c["max_len"]), c["mask_maxlen"])
        assert torch.equal(O.pad_mask(c["lengths"], c["max_len"]), c["mask_maxlen"])
        info = SeqInfo.of(c["mask_maxlen"])
        assert torch.equal(info.lens64, c["lengths"]) and torch.equal(info.mask, c["mask_maxlen"])


def test_mse_and_hinge_criteria_match_the_reference_classes():
    """The criteria variants no shipped yaml selects (loss_type="mse" of the reconstruction losses, loss_type="hinge" of the
    adversarial losses) against values and input gradients recorded from the reference classes
    (tests/golden/loss_variants.pt, oracle/make_golden.py::loss_variants_case).  They are plain elementwise arithmetic on
    whatever device the tensors live on, so the host tensors of the fixture exercise the shipped code."""
    from kantts.train.loss import (DiscriminatorAdversarialLoss, GeneratorAdversarialLoss, MelReconLoss,
                                   ProsodyReconLoss)

    fix = torch.load(os.path.join(GOLDEN, "loss_variants.pt"), weights_only=False)

    def close(a, b, tol=1e-6):
        assert torch.allclose(torch.as_tensor(a), torch.as_tensor(b), rtol=1e-5, atol=tol), (a, b)

    dec, post = fix["dec"].clone().requires_grad_(True), fix["post"].clone().requires_grad_(True)
    a, b = MelReconLoss("mse")(fix["output_lengths"], fix["mel_targets"], dec, post)
    (a + 2 * b).backward()
    for got, want in zip((a, b, dec.grad, post.grad), fix["mel_mse"]):
        close(got.detach(), want)
    only, zero = MelReconLoss("mse")(fix["output_lengths"], fix["mel_targets"], fix["dec"])
    close(only, fix["mel_mse_no_postnet"])
    assert zero == 0.0
    lp, pp, ep = (fix[k].clone().requires_grad_(True) for k in ("logdur_p", "pitch_p", "energy_p"))
    d, p, e = ProsodyReconLoss("mse")(fix["input_lengths"], fix["dur"], fix["pitch"], fix["energy"], lp, pp, ep)
    (d + 2 * p + 3 * e).backward()
    for got, want in zip((d, p, e, lp.grad, pp.grad, ep.grad), fix["prosody_mse"]):
        close(got.detach(), want)
    with pytest.raises(ValueError):
        MelReconLoss("huber")
    with pytest.raises(ValueError):
        ProsodyReconLoss("huber")

    fake, real, hinge = fix["d_fake"], fix["d_real"], fix["hinge"]
    for avg in (True, False):
        scores = [f[-1].clone().requires_grad_(True) for f in fake]
        v = GeneratorAdversarialLoss(average_by_discriminators=avg, loss_type="hinge")(scores)
        v.backward()
        want_v, want_g = hinge[("g_list", avg)]
        close(v.detach(), want_v)
        for s_, w in zip(scores, want_g):
            close(s_.grad, w)
        fk = [[t.clone().requires_grad_(True) for t in f] for f in fake]
        rl = [[t.clone().requires_grad_(True) for t in f] for f in real]
        r_, f_ = DiscriminatorAdversarialLoss(average_by_discriminators=avg, loss_type="hinge")(fk, rl)
        (r_ + 2 * f_).backward()
        want_r, want_f, gf, gr = hinge[("d_nested", avg)]
        close(r_.detach(), want_r)
        close(f_.detach(), want_f)
        for t, w in zip(fk, gf):
            close(t[-1].grad, w)
        for t, w in zip(rl, gr):
            close(t[-1].grad, w)
    one = fake[0][-1].clone().requires_grad_(True)
    v = GeneratorAdversarialLoss(loss_type="hinge")(one)
    v.backward()
    close(v.detach(), hinge["g_tensor"][0])
    close(one.grad, hinge["g_tensor"][1])
    r_, f_ = DiscriminatorAdversarialLoss(loss_type="hinge")(fake[1][-1], real[1][-1])
    close(r_, hinge["d_tensor"][0])
    close(f_, hinge["d_tensor"][1])
    with pytest.raises(AssertionError):
        GeneratorAdversarialLoss(loss_type="wasserstein")


def test_lstm_aten_equals_the_time_loop():
    """oracle/torch_oracle.py::lstm_aten (the ATen kernel the reference's nn.LSTM modules call, packed where the reference
    packs) against the readable time loop ``lstm_layer`` it replaced: values and gradients, uni- and bidirectional, one and
    two layers, ragged lengths (padded outputs zero, the reverse direction starting at each row's last valid token)."""
    import torch_oracle as O

    g = torch.Generator().manual_seed(3)
    B, T, C, H = 5, 11, 12, 8

    def weights(pre, layers, bidir):
        P = {}
        for l in range(layers):
            for sfx in ("", "_reverse") if bidir else ("",):
                cin = C if l == 0 else H * (2 if bidir else 1)
                for n, shape in (("weight_ih", (4 * H, cin)), ("weight_hh", (4 * H, H)), ("bias_ih", (4 * H,)), ("bias_hh", (4 * H,))):
                    P["%s.%s_l%d%s" % (pre, n, l, sfx)] = (0.4 * torch.randn(*shape, generator=g)).requires_grad_(True)
        return P

    for layers, bidir, lens in ((1, False, None), (2, False, None), (1, True, [11, 3, 7, 1, 11]), (1, True, None)):
        P = weights("m", layers, bidir)
        x = torch.randn(B, T, C, generator=g).requires_grad_(True)
        lengths = None if lens is None else torch.tensor(lens)
        got = O.lstm_aten(P, "m", x, layers, bidir, lengths)
        h = x
        for l in range(layers):
            outs = [O._lstm_named(P, "m", h, l, lengths, rev) for rev in ((False, True) if bidir else (False,))]
            h = torch.cat(outs, -1)
        assert got.shape == h.shape
        assert float((got - h).abs().max()) < 2e-6
        if lens is not None:
            for b, n in enumerate(lens):
                assert float(got[b, n:].abs().max() if n < T else 0.0) == 0.0
        w = torch.randn(got.shape, generator=g)
        ga = torch.autograd.grad((got * w).sum(), [x] + list(P.values()))
        gb = torch.autograd.grad((h * w).sum(), [x] + list(P.values()))
        for a, b in zip(ga, gb):
            assert float((a - b).abs().max()) < 1e-5 * max(1.0, float(b.abs().max()))
