"""TEST INFRASTRUCTURE -- CPU restatement (functional PyTorch fp32) of the reference's HiFi-GAN hot
path: Generator, MultiPeriodDiscriminator, MultiScaleDiscriminator (DWT pooling) and the GAN losses.
Cites kantts/models/hifigan/{hifigan,layers}.py and kantts/train/loss.py of /root/reference.
Only tests/, smoke() and bench.py's cpu_baseline leg may import it.  ``P`` is a reference
``state_dict`` (weight-normalised layers carry ``weight_g`` / ``weight_v``).
"""
import torch
import torch.nn.functional as F

from thirdparty import dwt_db3_zero


def wn(P, pre):
    """torch.nn.utils.weight_norm (dim=0): w = g * v / ||v|| over all dims but 0
    (kantts/models/hifigan/layers.py:29,67,105,139)."""
    if pre + ".weight" in P:
        return P[pre + ".weight"]
    v, g = P[pre + ".weight_v"], P[pre + ".weight_g"]
    return v * (g / v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1))))


def _bias(P, pre):
    return P.get(pre + ".bias")


def conv(P, pre, x, causal, dilation=1):
    """Conv1d / CausalConv1d (kantts/models/hifigan/layers.py:15-91): causal = left pad (k-1)*d."""
    w = wn(P, pre + ".conv1d")
    k = w.shape[-1]
    if causal:
        return F.conv1d(F.pad(x, ((k - 1) * dilation, 0)), w, _bias(P, pre + ".conv1d"), dilation=dilation)
    return F.conv1d(x, w, _bias(P, pre + ".conv1d"), dilation=dilation, padding=(k * dilation - dilation) // 2)


def conv_transpose(P, pre, x, stride, causal):
    """ConvTranspose1d / CausalConvTranspose1d (layers.py:94-165): causal drops the last k - stride."""
    w = wn(P, pre + ".deconv")
    k = w.shape[-1]
    if causal:
        return F.conv_transpose1d(x, w, _bias(P, pre + ".deconv"), stride=stride)[:, :, : -(k - stride)]
    return F.conv_transpose1d(x, w, _bias(P, pre + ".deconv"), stride=stride, padding=(k - stride) // 2)


def resblock(P, pre, x, dilations, causal):
    """ResidualBlock.forward (layers.py:213-220)."""
    for i, d in enumerate(dilations):
        xt = conv(P, "%s.convs1.%d" % (pre, i), F.leaky_relu(x, 0.1), causal, d)
        xt = conv(P, "%s.convs2.%d" % (pre, i), F.leaky_relu(xt, 0.1), causal, 1)
        x = xt + x
    return x


def generator(P, x, scales=(8, 8, 2, 2), n_kernels=3, dilations=((1, 3, 5),) * 3, causal=True):
    """Generator.forward (hifigan.py:145-182), repeat_upsample=True, no NSF."""
    x = conv(P, "conv_pre", x, causal)
    for i, s in enumerate(scales):
        x = torch.sin(x) + x
        rep = F.interpolate(x, scale_factor=float(s), mode="nearest")
        rep = conv(P, "repeat_upsamples.%d.2" % i, F.leaky_relu(rep, 0.1), causal)
        up = conv_transpose(P, "transpose_upsamples.%d.1" % i, F.leaky_relu(x, 0.1), s, causal)
        x = rep + up[:, :, : rep.shape[-1]]
        xs = None
        for j in range(n_kernels):
            y = resblock(P, "conv_blocks.%d" % (i * n_kernels + j), x, dilations[j], causal)
            xs = y if xs is None else xs + y
        x = xs / n_kernels
    x = F.leaky_relu(x)  # default slope 0.01 (hifigan.py:178)
    return torch.tanh(conv(P, "conv_post", x, causal))


def period_discriminator(P, pre, x, period, n_layers=5, strides=(3, 3, 3, 3, 1)):
    """PeriodDiscriminator.forward (hifigan.py:249-267); conv_post kernel (2,1) pad (1,0) (:241-247)."""
    b, c, t = x.shape
    if t % period:
        x = F.pad(x, (0, period - t % period), "reflect")
        t = x.shape[-1]
    x = x.view(b, c, t // period, period)
    fmap = []
    for l in range(n_layers):
        w = wn(P, "%s.convs.%d.0" % (pre, l))
        x = F.leaky_relu(F.conv2d(x, w, P["%s.convs.%d.0.bias" % (pre, l)], stride=(strides[l], 1),
                                  padding=((w.shape[2] - 1) // 2, 0)), 0.1)
        fmap.append(x)
    x = F.conv2d(x, P[pre + ".conv_post.weight"], P[pre + ".conv_post.bias"], padding=(1, 0))
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


def mpd(P, y, periods=(2, 3, 5, 7, 11)):
    outs, fmaps = [], []
    for i, p in enumerate(periods):
        o, f = period_discriminator(P, "discriminators.%d" % i, y, p)
        outs.append(o)
        fmaps.append(f)
    return outs, fmaps


def scale_discriminator(P, pre, x, strides=(2, 2, 4, 4, 1), groups=(4, 16, 16, 16, 16)):
    """ScaleDiscriminator.forward (hifigan.py:398-407)."""
    fmap = []
    w = wn(P, pre + ".convs.0.0")
    x = F.leaky_relu(F.conv1d(x, w, P[pre + ".convs.0.0.bias"], padding=(w.shape[-1] - 1) // 2), 0.1)
    fmap.append(x)
    for l in range(len(strides)):
        w = wn(P, "%s.convs.%d.0" % (pre, l + 1))
        x = F.leaky_relu(F.conv1d(x, w, P["%s.convs.%d.0.bias" % (pre, l + 1)], stride=strides[l],
                                  padding=(w.shape[-1] - 1) // 2, groups=groups[l]), 0.1)
        fmap.append(x)
    n = len(strides) + 1
    w = wn(P, "%s.convs.%d.0" % (pre, n))
    x = F.leaky_relu(F.conv1d(x, w, P["%s.convs.%d.0.bias" % (pre, n)], padding=(w.shape[-1] - 1) // 2), 0.1)
    fmap.append(x)
    w = wn(P, pre + ".conv_post")
    x = F.conv1d(x, w, P[pre + ".conv_post.bias"], padding=(w.shape[-1] - 1) // 2)
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


def msd(P, y, scales=3):
    """MultiScaleDiscriminator.forward with DWT pooling (hifigan.py:461-478)."""
    outs, fmaps = [], []
    for i in range(scales):
        if i:
            lo, hi = dwt_db3_zero(y)
            y = torch.cat([lo, hi], dim=1)
            w = wn(P, "aux_convs.%d" % (i - 1))
            y = F.leaky_relu(F.conv1d(y, w, P["aux_convs.%d.bias" % (i - 1)], padding=7), 0.1)
        o, f = scale_discriminator(P, "discriminators.%d" % i, y)
        outs.append(o)
        fmaps.append(f)
    return outs, fmaps


# ----------------------------------------------------------------------------- losses (loss.py:108-311)
def gen_adv_loss(outs):
    """GeneratorAdversarialLoss (mse, averaged over discriminators)."""
    return sum(F.mse_loss(o, torch.ones_like(o)) for o in outs) / len(outs)


def dis_adv_loss(outs_hat, outs):
    """DiscriminatorAdversarialLoss -> (real, fake)."""
    real = sum(F.mse_loss(o, torch.ones_like(o)) for o in outs) / len(outs)
    fake = sum(F.mse_loss(o, torch.zeros_like(o)) for o in outs_hat) / len(outs_hat)
    return real, fake


def feat_match_loss(fmaps_hat, fmaps):
    """FeatureMatchLoss (average_by_layers, average_by_discriminators)."""
    tot = 0.0
    for fh, fr in zip(fmaps_hat, fmaps):
        tot = tot + sum(F.l1_loss(a, b.detach()) for a, b in zip(fh, fr)) / len(fh)
    return tot / len(fmaps)
