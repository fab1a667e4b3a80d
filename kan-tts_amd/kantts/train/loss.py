"""Training criteria (reference kantts/train/loss.py).  The masked-L1 reductions of the SAM-BERT
step run in one kernel each (kantts_masked_l1: loss scalar + d loss/d pred in the same pass)."""
import torch
import torch.nn.functional as F

from kantts._hip import ops


def _masked_mse(pred, target, lens):
    """sum over valid rows of (target - pred)^2 / (valid rows * row width) -- loss_type="mse" of the two criteria below
    (reference :13-14, :46-47).  No shipped yaml selects it, so it is a handful of elementwise launches rather than a
    variant of kantts_masked_l1."""
    T = pred.size(1)
    valid = torch.arange(T, device=pred.device)[None, :] < lens.to(pred.device)[:, None]
    width = pred.numel() // (pred.size(0) * T)
    sq = (target.to(pred.dtype) - pred) ** 2
    if sq.dim() == 3:
        sq = sq * valid.unsqueeze(-1)
    else:
        sq = sq * valid
    return sq.sum() / (valid.sum() * width)


def _check_loss_type(loss_type):
    if loss_type not in ("mae", "mse"):
        raise ValueError("Unknown loss type: {}".format(loss_type))
    return ops.masked_l1 if loss_type == "mae" else _masked_mse


class MelReconLoss(torch.nn.Module):
    """Masked mean |target - output| (``"mae"``) or squared error (``"mse"``) for decoder and postnet mels
    (reference :7-37)."""

    def __init__(self, loss_type="mae"):
        super(MelReconLoss, self).__init__()
        self.loss_type = loss_type
        _check_loss_type(loss_type)

    def forward(self, output_lengths, mel_targets, dec_outputs, postnet_outputs=None):
        lens = output_lengths.to(torch.int64)
        criterion = _check_loss_type(self.loss_type)
        mel_loss_ = criterion(dec_outputs, mel_targets, lens)
        mel_loss = criterion(postnet_outputs, mel_targets, lens) if postnet_outputs is not None else 0.0
        return mel_loss_, mel_loss


class ProsodyReconLoss(torch.nn.Module):
    """Masked L1 (``"mae"``) or squared error (``"mse"``) on log-duration, pitch, energy (reference :40-85)."""

    def __init__(self, loss_type="mae"):
        super(ProsodyReconLoss, self).__init__()
        self.loss_type = loss_type
        _check_loss_type(loss_type)

    def forward(self, input_lengths, duration_targets, pitch_targets, energy_targets, log_duration_predictions,
                pitch_predictions, energy_predictions):
        lens = input_lengths.to(torch.int64)
        criterion = _check_loss_type(self.loss_type)
        dur_loss = criterion(log_duration_predictions, torch.log(duration_targets.float() + 1), lens)
        pitch_loss = criterion(pitch_predictions, pitch_targets, lens)
        energy_loss = criterion(energy_predictions, energy_targets, lens)
        return dur_loss, pitch_loss, energy_loss


def sambert_loss_sum(mel_criterion, prosody_criterion, batch, res, prosody_lengths=None):
    """The training objective of a SAM-BERT step as the reference's trainer forms it (kantts/train/trainer.py:940-975:
    mel_loss_ + mel_loss + dur_loss + pitch_loss + energy_loss) -> (total, {name: detached component}).
    Both criteria in their default "mae" form on device tensors: ONE launch for the five terms, their sum and all five
    gradients, one launch in backward (ops.masked_l1_many) instead of five reductions, the duration-target logarithm and
    the scalar sums / products around them (~30 graph nodes between the end of forward and the start of backward).
    Anything else: the two criteria are called as modules."""
    names = ("mel_loss_", "mel_loss", "dur_loss", "pitch_loss", "energy_loss")
    dec, post = res["dec_outputs"], res["postnet_outputs"]
    if (getattr(mel_criterion, "loss_type", None) == "mae" and getattr(prosody_criterion, "loss_type", None) == "mae"
            and type(mel_criterion) is MelReconLoss and type(prosody_criterion) is ProsodyReconLoss and post is not None
            and dec.dtype == torch.float32 and not __import__("os").environ.get("KANTTS_NO_FUSED_LOSS")):
        il = (batch["input_lengths"] if prosody_lengths is None else prosody_lengths).to(torch.int64)
        ol = batch["output_lengths"].to(torch.int64)
        mel_t = batch["mel_targets"]
        dur_t = res["duration_targets"]
        if dur_t.dtype == torch.int64 and mel_t.dtype == torch.float32:
            total, comps = ops.masked_l1_many([
                (dec, mel_t, ol, False), (post, mel_t, ol, False),
                (res["log_duration_predictions"], dur_t, il, True),
                (res["pitch_predictions"], res["pitch_targets"].float(), il, False),
                (res["energy_predictions"], res["energy_targets"].float(), il, False)])
            return total, {n: comps[k] for k, n in enumerate(names)}
    mel_, mel = mel_criterion(batch["output_lengths"], batch["mel_targets"], dec, post)
    d, p, e = prosody_criterion(batch["input_lengths"] if prosody_lengths is None else prosody_lengths,
                                res["duration_targets"], res["pitch_targets"],
                                res["energy_targets"], res["log_duration_predictions"], res["pitch_predictions"],
                                res["energy_predictions"])
    comps = dict(zip(names, (mel_, mel, d, p, e)))
    return mel_ + mel + d + p + e, {k: (v.detach() if torch.is_tensor(v) else v) for k, v in comps.items()}


def _fused_gan_losses(tensors):
    """The one-launch criteria (ops.elem_loss_many) apply to fp32 tensors; KANTTS_NO_FUSED_GAN_LOSS=1: the per-term path."""
    return (len(tensors) > 0 and all(torch.is_tensor(t) and t.dtype == torch.float32 and t.numel() > 0 for t in tensors)
            and not __import__("os").environ.get("KANTTS_NO_FUSED_GAN_LOSS"))


class GeneratorAdversarialLoss(torch.nn.Module):
    """Generator adversarial loss, mean (or sum) over discriminators: LSGAN ``mse(D(G(x)), 1)`` (``"mse"``, every shipped
    yaml: one reduction kernel per score) or ``-mean(D(G(x)))`` (``"hinge"``) (reference :108-151)."""

    def __init__(self, average_by_discriminators=True, loss_type="mse"):
        super().__init__()
        assert loss_type in ["mse", "hinge"], f"{loss_type} is not supported."
        self.average_by_discriminators = average_by_discriminators
        self.loss_type = loss_type

    def criterion(self, x):
        return ops.mse_to_const(x, 1.0) if self.loss_type == "mse" else -x.mean()

    def forward(self, outputs):
        if not isinstance(outputs, (tuple, list)):
            return self.criterion(outputs)
        scores = [o[-1] if isinstance(o, (tuple, list)) else o for o in outputs]
        if self.loss_type == "mse" and _fused_gan_losses(scores):
            # every discriminator's score in ONE launch (was: a zero-fill + a reduction + an addition per discriminator)
            w = 1.0 / len(scores) if self.average_by_discriminators else 1.0
            return ops.elem_loss_many([(s, None, 1.0, 1, w / s.numel(), 0) for s in scores])[0]
        adv = 0.0
        for o in scores:
            adv = adv + self.criterion(o)
        return adv / len(outputs) if self.average_by_discriminators else adv


class DiscriminatorAdversarialLoss(torch.nn.Module):
    """Discriminator adversarial loss -> (real, fake): LSGAN (``"mse"``, every shipped yaml) or hinge
    ``mean(relu(1 - D(x)))``, ``mean(relu(1 + D(G(x))))`` (reference :154-216)."""

    def __init__(self, average_by_discriminators=True, loss_type="mse"):
        super().__init__()
        assert loss_type in ["mse", "hinge"], f"{loss_type} is not supported."
        self.average_by_discriminators = average_by_discriminators
        self.loss_type = loss_type

    def real_criterion(self, x):
        if self.loss_type == "mse":
            return ops.mse_to_const(x, 1.0)
        return torch.clamp(1.0 - x, min=0.0).mean()  # = -mean(min(x - 1, 0)) of the reference

    def fake_criterion(self, x):
        if self.loss_type == "mse":
            return ops.mse_to_const(x, 0.0)
        return torch.clamp(1.0 + x, min=0.0).mean()  # = -mean(min(-x - 1, 0))

    def forward(self, outputs_hat, outputs):
        if not isinstance(outputs, (tuple, list)):
            return self.real_criterion(outputs), self.fake_criterion(outputs_hat)
        pairs = [(oh[-1], o[-1]) if isinstance(oh, (tuple, list)) else (oh, o) for oh, o in zip(outputs_hat, outputs)]
        if self.loss_type == "mse" and _fused_gan_losses([t for p in pairs for t in p]):
            w = 1.0 / len(pairs) if self.average_by_discriminators else 1.0
            both = ops.elem_loss_many([(o, None, 1.0, 1, w / o.numel(), 0) for _, o in pairs]
                                      + [(oh, None, 0.0, 1, w / oh.numel(), 1) for oh, _ in pairs], n_out=2)
            return both[0], both[1]
        real, fake = 0.0, 0.0
        for oh, o in pairs:
            real = real + self.real_criterion(o)
            fake = fake + self.fake_criterion(oh)
        if self.average_by_discriminators:
            real, fake = real / len(outputs), fake / len(outputs)
        return real, fake


class FeatureMatchLoss(torch.nn.Module):
    """Sum over discriminators of the mean L1 between feature maps (reference :219-256)."""

    def __init__(self, average_by_layers=True, average_by_discriminators=True):
        super().__init__()
        self.average_by_layers = average_by_layers
        self.average_by_discriminators = average_by_discriminators

    def forward(self, feats_hat, feats):
        flat = [a for fh in feats_hat for a in fh]
        if _fused_gan_losses(flat):
            # ~48 feature maps (one per discriminator layer): ONE launch instead of 48 x (zero-fill, reduction, addition)
            terms = []
            for fh, fr in zip(feats_hat, feats):
                w = (1.0 / len(fh) if self.average_by_layers else 1.0) * (1.0 / len(feats) if self.average_by_discriminators else 1.0)
                terms += [(a, b, 0.0, 0, w / a.numel(), 0) for a, b in zip(fh, fr)]
            return ops.elem_loss_many(terms)[0]
        total = 0.0
        for fh, fr in zip(feats_hat, feats):
            part = 0.0
            for a, b in zip(fh, fr):
                part = part + ops.l1_mean(a, b)
            total = total + (part / len(fh) if self.average_by_layers else part)
        return total / len(feats) if self.average_by_discriminators else total


class MelSpectrogramLoss(torch.nn.Module):
    """L1 between normalised log-mels of generated and real audio (reference :259-311); forward and backward
    of the generated branch run in the fused mel-STFT kernels."""

    def __init__(self, fs=22050, fft_size=1024, hop_size=256, win_length=None, window="hann", num_mels=80, fmin=80,
                 fmax=7600, center=True, normalized=False, onesided=True, eps=1e-10, log_base=10.0):
        super().__init__()
        from kantts.utils.audio_torch import MelSpectrogram

        self.mel_spectrogram = MelSpectrogram(fs=fs, fft_size=fft_size, hop_size=hop_size, win_length=win_length,
                                              window=window, num_mels=num_mels, fmin=fmin, fmax=fmax, center=center,
                                              normalized=normalized, onesided=onesided, eps=eps, log_base=log_base)

    def forward(self, y_hat, y):
        mel_hat = self.mel_spectrogram(y_hat)
        with torch.no_grad():
            mel = self.mel_spectrogram(y)
        return ops.l1_mean(mel_hat, mel)


class SpectralConvergenceLoss(torch.nn.Module):
    """|| |Y| - |X| ||_F / || |Y| ||_F (reference :312-332)."""

    def forward(self, x_mag, y_mag):
        return torch.norm(y_mag - x_mag, p="fro") / torch.norm(y_mag, p="fro")


class LogSTFTMagnitudeLoss(torch.nn.Module):
    """L1(log |Y|, log |X|) (reference :335-354)."""

    def forward(self, x_mag, y_mag):
        return ops.l1_mean(torch.log(x_mag), torch.log(y_mag))


class STFTLoss(torch.nn.Module):
    """One resolution (reference :356-396): both magnitudes come from the fused framed-STFT kernel; the generated
    branch back-propagates through kantts_stft_mag_bwd."""

    def __init__(self, fft_size=1024, shift_size=120, win_length=600, window="hann_window"):
        super(STFTLoss, self).__init__()
        self.fft_size, self.shift_size, self.win_length = fft_size, shift_size, win_length
        self.window_name = window
        self.spectral_convergence_loss = SpectralConvergenceLoss()
        self.log_stft_magnitude_loss = LogSTFTMagnitudeLoss()
        self.register_buffer("window", getattr(torch, window)(win_length))

    def forward(self, x, y):
        from kantts.utils.audio_torch import stft

        x_mag = stft(x, self.fft_size, self.shift_size, self.win_length, self.window_name)
        with torch.no_grad():
            y_mag = stft(y, self.fft_size, self.shift_size, self.win_length, self.window_name)
        return self.spectral_convergence_loss(x_mag, y_mag), self.log_stft_magnitude_loss(x_mag, y_mag)


class MultiResolutionSTFTLoss(torch.nn.Module):
    """Mean over resolutions of (spectral convergence, log-magnitude L1) -- reference :399-441; (B, T) or
    (B, #subband, T) inputs."""

    def __init__(self, fft_sizes=[1024, 2048, 512], hop_sizes=[120, 240, 50], win_lengths=[600, 1200, 240],
                 window="hann_window"):
        super(MultiResolutionSTFTLoss, self).__init__()
        assert len(fft_sizes) == len(hop_sizes) == len(win_lengths)
        self.stft_losses = torch.nn.ModuleList()
        for fs, ss, wl in zip(fft_sizes, hop_sizes, win_lengths):
            self.stft_losses += [STFTLoss(fs, ss, wl, window)]

    def forward(self, x, y):
        if len(x.shape) == 3:
            x = x.reshape(-1, x.size(2))
            y = y.reshape(-1, y.size(2))
        sc_loss, mag_loss = 0.0, 0.0
        for f in self.stft_losses:
            sc_l, mag_l = f(x, y)
            sc_loss = sc_loss + sc_l
            mag_loss = mag_loss + mag_l
        return sc_loss / len(self.stft_losses), mag_loss / len(self.stft_losses)


class AttentionBinarizationLoss(torch.nn.Module):
    """-mean log soft attention on the hard (MAS) path, warmed up over epochs (reference :463-479).
    ``soft[hard == 1]`` of the reference is a boolean gather with a host sync; hard is 0/1, so the same sum is
    ``sum(hard * log(clamp(soft)))`` with static shapes."""

    def __init__(self, start_epoch=0, warmup_epoch=100):
        super(AttentionBinarizationLoss, self).__init__()
        self.start_epoch = start_epoch
        self.warmup_epoch = warmup_epoch

    def forward(self, epoch, hard_attention, soft_attention, eps=1e-12):
        log_sum = (hard_attention * torch.log(torch.clamp(soft_attention, min=eps))).sum()
        kl_loss = -log_sum / hard_attention.sum()
        if epoch < self.start_epoch:
            warmup_ratio = 0
        else:
            warmup_ratio = min(1.0, (epoch - self.start_epoch) / self.warmup_epoch)
        return kl_loss * warmup_ratio


class _CtcAttn(torch.autograd.Function):
    """loss (B,) = CTC(log_softmax([blank | logits]), target 1..S) / S per utterance; the launch (csrc/ctc.hip) computes the
    gradient with the loss, backward only scales it."""

    @staticmethod
    def forward(ctx, logits, in_lens32, out_lens32, blank):
        from kantts._hip import ctc_attn

        loss, grad = ctc_attn(logits, in_lens32, out_lens32, blank, 1.0)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (grad,) = ctx.saved_tensors
        return grad * dloss.view(-1, 1, 1), None, None, None


class AttentionCTCLoss(torch.nn.Module):
    """CTC over the alignment log-probabilities with the phoneme sequence 1..N as target and a constant blank score
    (reference :482-508).  The reference loops over utterances (slice, log_softmax, one torch.nn.CTCLoss call each, whose
    ATen implementation copies the lengths to the host); here the batch is ONE launch of kantts_ctc_attn (csrc/ctc.hip:
    a workgroup per utterance walks the alpha and beta recurrences and writes the gradient), with the lengths read on the
    device -- so the MAS training step can be captured in a hipGraph.  Result = mean_b(nll_b / S_b), which is what the
    reference's sum of per-utterance "mean" losses / B computes; zero_infinity semantics."""

    def __init__(self, blank_logprob=-1):
        super(AttentionCTCLoss, self).__init__()
        self.blank_logprob = blank_logprob

    def forward(self, attn_logprob, in_lens, out_lens):
        lg = attn_logprob[:, 0]
        if not lg.is_contiguous():
            lg = lg.contiguous()
        loss = _CtcAttn.apply(lg.float(), in_lens.to(torch.int32), out_lens.to(torch.int32), float(self.blank_logprob))
        return loss.mean()


loss_dict = {
    "AttentionBinarizationLoss": AttentionBinarizationLoss,
    "AttentionCTCLoss": AttentionCTCLoss,
    "MelReconLoss": MelReconLoss,
    "ProsodyReconLoss": ProsodyReconLoss,
    "generator_adv_loss": GeneratorAdversarialLoss,
    "discriminator_adv_loss": DiscriminatorAdversarialLoss,
    "feat_match_loss": FeatureMatchLoss,
    "mel_loss": MelSpectrogramLoss,
    "stft_loss": MultiResolutionSTFTLoss,
    "subband_stft_loss": MultiResolutionSTFTLoss,
}


def criterion_builder(config, device="cpu"):
    """Same contract as reference :528-544: dict of enabled criteria + ``.weights`` entries."""
    criterion = {}
    for key, value in config["Loss"].items():
        if key in loss_dict:
            if value["enable"]:
                criterion[key] = loss_dict[key](**value.get("params", {})).to(device)
                setattr(criterion[key], "weights", value.get("weights", 1.0))
        elif value.get("enable", False):
            raise NotImplementedError("{} is not implemented".format(key))
    return criterion

