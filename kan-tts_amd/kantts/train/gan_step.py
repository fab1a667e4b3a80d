"""HiFi-GAN training step on the HIP path -- the arithmetic of GAN_Trainer.train_step
(reference kantts/train/trainer.py:469-589) without its logging shell.

Per step: y_ = G(x); generator loss = w_mel * L1(mel(y_), mel(y)) + w_adv * sum_D mse(D(y_), 1) + w_fm * FM;
generator update; y_ = G(x) again without grad (the reference re-computes it, :556-559); discriminator loss =
sum_D [mse(D(y), 1) + mse(D(y_), 0)]; discriminator updates.

Reference quirk kept on purpose: the feature-matching call swaps its arguments (:527-531 builds
``zip(fmap_lst, fmap_lst_)`` with the names exchanged), so the real feature maps (computed under no_grad) sit
in the differentiable slot and the generated ones are detached -- the term changes the logged loss value but
contributes no gradient.  ``feature_match_gradient=True`` gives the textbook behaviour instead.
"""
import torch


def _clip(optimizer, params, max_norm):
    if max_norm and max_norm > 0:
        if hasattr(optimizer, "set_grad_clip"):
            optimizer.set_grad_clip(max_norm)
        else:
            torch.nn.utils.clip_grad_norm_(params, max_norm)


def generator_loss(model, criterion, x, y, adversarial=True, feature_match_gradient=False):
    y_ = model["generator"](x)
    losses = {}
    gen_loss = 0.0
    # multi-band generators (out_channels = subbands): the full-band signal comes out of the PQMF synthesis bank and
    # carries every full-band loss and the discriminators (reference trainer.py:476-478)
    pqmf = model.get("pqmf", None)
    y_mb_ = None
    if pqmf is not None:
        y_mb_ = y_
        y_ = pqmf.synthesis(y_mb_)
    if criterion.get("stft_loss", None) is not None:  # multi-resolution STFT loss (reference :484-493)
        sc, mag = criterion["stft_loss"](y_, y)
        losses["spectral_convergence_loss"], losses["log_stft_magnitude_loss"] = sc, mag
        gen_loss = gen_loss + (sc + mag) * criterion["stft_loss"].weights
    if criterion.get("subband_stft_loss", None) is not None:
        # reference :496-506 halves what has been summed so far and adds half the sub-band loss.  (It looks the
        # criterion up under the key "sub_stft", which criterion_builder never creates -- upstream raises KeyError as
        # soon as the loss is enabled; the key it tests for, "subband_stft_loss", is used here.)
        if pqmf is None:
            raise ValueError("subband_stft_loss needs a multi-band generator (model['pqmf'])")
        gen_loss = gen_loss * 0.5
        ssc, smag = criterion["subband_stft_loss"](y_mb_, pqmf.analysis(y))
        losses["sub_spectral_convergence_loss"], losses["sub_log_stft_magnitude_loss"] = ssc, smag
        gen_loss = gen_loss + 0.5 * (ssc + smag)
    if criterion.get("mel_loss", None) is not None:
        losses["mel_loss"] = criterion["mel_loss"](y_, y)
        gen_loss = gen_loss + losses["mel_loss"] * criterion["mel_loss"].weights
    if adversarial:
        adv, fmaps_hat = 0.0, []
        for name, d in model["discriminator"].items():
            p_, fmap_ = d(y_)
            fmaps_hat.append(fmap_)
            adv = adv + criterion["generator_adv_loss"](p_)
        losses["adversarial_loss"] = adv
        gen_loss = gen_loss + adv * criterion["generator_adv_loss"].weights
        if criterion.get("feat_match_loss", None) is not None:
            fmaps = []
            with torch.no_grad():
                for name, d in model["discriminator"].items():
                    fmaps.append(d(y)[1])
            fm = 0.0
            for real, fake in zip(fmaps, fmaps_hat):
                fm = fm + (criterion["feat_match_loss"](fake, real) if feature_match_gradient
                           else criterion["feat_match_loss"](real, fake))
            losses["feature_matching_loss"] = fm
            gen_loss = gen_loss + fm * criterion["feat_match_loss"].weights
    losses["generator_loss"] = gen_loss
    return gen_loss, losses, y_


def discriminator_loss(model, criterion, x, y, batched=True):
    """``batched``: every discriminator sees real and generated audio as ONE batch of 2B (same sums, same
    gradients -- the discriminators have no cross-sample statistics) instead of two passes of B as in the
    reference (:556-577): half the launches, weight-norm / re-layout work and twice the rows per tile."""
    with torch.no_grad():
        y_ = model["generator"](x)
        if model.get("pqmf", None) is not None:  # reference :561-562
            y_ = model["pqmf"].synthesis(y_)
    dis_loss, losses = 0.0, {"real_loss": 0.0, "fake_loss": 0.0}
    B = y.size(0)
    both = torch.cat([y, y_.detach()], dim=0) if batched else None
    for name, d in model["discriminator"].items():
        if batched:
            outs, _ = d(both)
            p = [o[:B] for o in outs] if isinstance(outs, (list, tuple)) else outs[:B]
            p_ = [o[B:] for o in outs] if isinstance(outs, (list, tuple)) else outs[B:]
        else:
            p, _ = d(y)
            p_, _ = d(y_.detach())
        real, fake = criterion["discriminator_adv_loss"](p_, p)
        dis_loss = dis_loss + real + fake
        losses["real_loss"] = losses["real_loss"] + real
        losses["fake_loss"] = losses["fake_loss"] + fake
    losses["discriminator_loss"] = dis_loss
    return dis_loss, losses


def gan_train_step(model, optimizer, scheduler, criterion, config, y, x, steps=10 ** 9):
    """One GAN step; returns a dict of (device) loss tensors.  ``steps`` gates the phases like the reference."""
    out = {}
    train_d = steps > config.get("discriminator_train_start_steps", 0)
    if steps >= config.get("generator_train_start_steps", 0):
        # The generator phase back-propagates THROUGH the discriminators; the reference (trainer.py:600-680) leaves their
        # parameters trainable there, so autograd also produces discriminator weight gradients that nothing ever reads
        # (they are zeroed before the discriminator phase).  Freezing the parameters for this phase removes a third of
        # all discriminator weight-gradient launches; every update and every loss stays bit-identical
        # (tests/test_trainer.py retraces the reference's GAN loss curve).
        frozen = [p for d in model["discriminator"].values() for p in d.parameters() if p.requires_grad] if train_d else []
        for p in frozen:
            p.requires_grad_(False)
        try:
            gen_loss, losses, _ = generator_loss(model, criterion, x, y, adversarial=train_d)
            out.update(losses)
            optimizer["generator"].zero_grad()
            gen_loss.backward()
        finally:
            for p in frozen:
                p.requires_grad_(True)
        _clip(optimizer["generator"], model["generator"].parameters(), config.get("generator_grad_norm", -1))
        optimizer["generator"].step()
        scheduler["generator"].step()
    if train_d:
        dis_loss, losses = discriminator_loss(model, criterion, x, y)
        out.update(losses)
        for key in optimizer["discriminator"]:
            optimizer["discriminator"][key].zero_grad()
        dis_loss.backward()
        for key in optimizer["discriminator"]:
            _clip(optimizer["discriminator"][key], model["discriminator"][key].parameters(),
                  config.get("discriminator_grad_norm", -1))
            optimizer["discriminator"][key].step()
        for key in scheduler["discriminator"]:
            scheduler["discriminator"][key].step()
    # detached: a caller that keeps the loss values must not keep both autograd graphs (and their AccumulateGrad nodes,
    # which pin the stream they first ran on -- a later hipGraph capture of the step would trip over them) alive
    return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}
