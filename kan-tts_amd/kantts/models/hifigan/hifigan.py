"""HiFi-GAN generator and MPD / MSD discriminators on HIP kernels
(reference kantts/models/hifigan/hifigan.py:22-478).  Same constructors, ``forward`` contracts
(``(B, C, T)`` in, ``(B, out_ch, T*prod(scales))`` out; discriminators return ``(outputs, feature maps)``)
and ``state_dict`` keys.  Internally everything is channels-last; feature maps are handed back as
zero-copy ``(B, C, T[, p])`` views of the channels-last buffers.
"""
import copy
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils import spectral_norm, weight_norm

from kantts._hip import get_precision, ops
from kantts.models.hifigan.layers import (activation_slope, CausalConv1d, CausalConvTranspose1d, Conv1d, ConvTranspose1d,
                                          ResidualBlock, SourceModule, conv_weight, effective_weight)

DB3_DEC_LO = [0.035226291882100656, -0.08544127388224149, -0.13501102001039084, 0.4598775021193313,
              0.8068915093133388, 0.3326705529509569]
DB3_DEC_HI = [((-1.0) ** (k + 1)) * DB3_DEC_LO[5 - k] for k in range(6)]


_DUAL_SEL = {}


def _dual_path_selector(k7, s, device):
    """Sel[k, r + j*s] = 1 where tap k of the causal k7-wide convolution over the nearest-repeated signal, at output phase
    r (output sample q*s + r), reads INPUT token q - j:  floor((r + k - (k7 - 1)) / s) == -j.  (k7, J*s) with
    J = 1 + ceil((k7 - 1) / s) ... trimmed to the taps that occur."""
    key = (k7, s, str(device))
    hit = _DUAL_SEL.get(key)
    if hit is None:
        offs = [[-((r + k - (k7 - 1)) // s) for r in range(s)] for k in range(k7)]  # j >= 0
        J = 1 + max(max(row) for row in offs)
        sel = torch.zeros(k7, J * s)
        for k in range(k7):
            for r in range(s):
                sel[k, r + offs[k][r] * s] = 1.0
        hit = _DUAL_SEL[key] = (sel.to(device), J)
    return hit


class Generator(torch.nn.Module):
    """Dual-path upsampling generator (reference :22-197): per stage
    x = sin(x)+x;  x = ConvT(LReLU(x)) + Conv_k7(LReLU(nearest_x_s(x)));  x = mean_j ResBlock_j(x).
    [round 4] Both paths of a causal stage are ONE polyphase contraction, one launch, one output write: the k7-wide
    convolution over the s-times repeated signal reads, at output phase r, only the input tokens q, q - 1, .. (tap k
    reads token q + floor((r + k - 6) / s)), i.e. it IS a causal transposed convolution of stride s whose polyphase weight
    is the sum of the taps that fall on the same token -- and it is applied to the same LeakyReLU(x) as the transposed
    convolution proper.  The two weights are added (a few elementwise / small-matmul operations on the weight tensors,
    differentiable, so both parameters receive their gradients from the ONE weight-gradient contraction) and the stage is
    ``conv_transpose_cl`` with a kernel of J*s taps (J = 2 for the x8 stages: exactly the cost of the transposed
    convolution alone -- the 67.6 GFLOP of the four repeat convolutions at batch 32 and their backward passes are gone;
    J = 4 for the x2 stages).  ``rep`` is never formed.  KANTTS_NO_DUAL_FUSE=1 keeps the two-launch form (the k = 7
    convolution reading x through a //s token map, added in the transposed convolution's epilogue)."""

    def __init__(self, in_channels=80, out_channels=1, channels=512, kernel_size=7, upsample_scales=(8, 8, 2, 2),
                 upsample_kernal_sizes=(16, 16, 4, 4), resblock_kernel_sizes=(3, 7, 11),
                 resblock_dilations=[(1, 3, 5), (1, 3, 5), (1, 3, 5)], repeat_upsample=True, bias=True, causal=True,
                 nonlinear_activation="LeakyReLU", nonlinear_activation_params={"negative_slope": 0.1},
                 use_weight_norm=True, nsf_params=None):
        super(Generator, self).__init__()
        assert kernel_size % 2 == 1, "Kernal size must be odd number."
        assert len(upsample_scales) == len(upsample_kernal_sizes)
        assert len(resblock_dilations) == len(resblock_kernel_sizes)
        self.upsample_scales = upsample_scales
        self.repeat_upsample = repeat_upsample
        self.num_upsamples = len(upsample_kernal_sizes)
        self.num_kernels = len(resblock_kernel_sizes)
        self.out_channels = out_channels
        self.nsf_enable = nsf_params is not None
        self.causal = causal
        self.slope = activation_slope(nonlinear_activation, nonlinear_activation_params)
        self.transpose_upsamples = torch.nn.ModuleList()
        self.repeat_upsamples = torch.nn.ModuleList()
        self.conv_blocks = torch.nn.ModuleList()
        conv_cls = CausalConv1d if causal else Conv1d
        conv_transposed_cls = CausalConvTranspose1d if causal else ConvTranspose1d
        self.conv_pre = conv_cls(in_channels, channels, kernel_size, 1, padding=(kernel_size - 1) // 2)
        for i in range(len(upsample_kernal_sizes)):
            self.transpose_upsamples.append(torch.nn.Sequential(
                getattr(torch.nn, nonlinear_activation)(**nonlinear_activation_params),
                conv_transposed_cls(channels // (2 ** i), channels // (2 ** (i + 1)), upsample_kernal_sizes[i],
                                    upsample_scales[i], padding=(upsample_kernal_sizes[i] - upsample_scales[i]) // 2)))
            if repeat_upsample:
                self.repeat_upsamples.append(nn.Sequential(
                    nn.Upsample(mode="nearest", scale_factor=upsample_scales[i]),
                    getattr(torch.nn, nonlinear_activation)(**nonlinear_activation_params),
                    conv_cls(channels // (2 ** i), channels // (2 ** (i + 1)), kernel_size=kernel_size, stride=1,
                             padding=(kernel_size - 1) // 2)))
            for j in range(len(resblock_kernel_sizes)):
                self.conv_blocks.append(ResidualBlock(
                    channels=channels // (2 ** (i + 1)), kernel_size=resblock_kernel_sizes[j],
                    dilation=resblock_dilations[j], nonlinear_activation=nonlinear_activation,
                    nonlinear_activation_params=nonlinear_activation_params, causal=causal))
        self.conv_post = conv_cls(channels // (2 ** (i + 1)), out_channels, kernel_size, 1,
                                  padding=(kernel_size - 1) // 2)
        if self.nsf_enable:
            # neural source filter (reference :119-143): a sine excitation at the sample rate, brought down to every
            # stage's rate by a strided convolution (kernel 2u, stride u) and added to the stage input
            self.source_module = SourceModule(nb_harmonics=nsf_params["nb_harmonics"],
                                              upsample_ratio=np.cumprod(self.upsample_scales)[-1],
                                              sampling_rate=nsf_params["sampling_rate"])
            self.source_downs = nn.ModuleList()
            self.downsample_rates = [1] + list(self.upsample_scales[::-1][:-1])
            self.downsample_cum_rates = np.cumprod(self.downsample_rates)
            for i, u in enumerate(self.downsample_cum_rates[::-1]):
                u = int(u)
                if u == 1:
                    self.source_downs.append(Conv1d(1, channels // (2 ** (i + 1)), 1, 1))
                else:
                    self.source_downs.append(conv_cls(1, channels // (2 ** (i + 1)), u * 2, u, padding=u // 2))

    def forward(self, x):
        excitation = None
        if self.nsf_enable:  # the last two input channels carry f0 (Hz) and the voiced flag
            excitation = self.source_module.forward_cl(x[:, -2:-1, :], x[:, -1:, :])
            x = x[:, :-2, :]
        h = self.conv_pre.forward_cl(x.transpose(1, 2).contiguous())
        for i in range(self.num_upsamples):
            s = self.upsample_scales[i]
            act = None
            up_layer = self.transpose_upsamples[i][1]
            if (get_precision() == "bf16" and h.is_cuda and isinstance(up_layer, CausalConvTranspose1d)
                    and not os.environ.get("KANTTS_NO_UPSTREAM")):  # (A/B switch of scripts/gpu_*.sh)
                # sin(h) + h and the bf16 LeakyReLU image the streaming transposed convolution consumes, in one pass
                h, act = ops.sin_add(h, act_slope=self.slope)
            else:
                h = ops.sin_add(h)
            if (self.repeat_upsample and isinstance(up_layer, CausalConvTranspose1d)
                    and isinstance(self.repeat_upsamples[i][2], CausalConv1d) and not os.environ.get("KANTTS_NO_DUAL_FUSE")):
                w_dual, b_dual = self._dual_path_weight(i, s)
                exc = self.source_downs[i].forward_cl(excitation) if excitation is not None else None
                h = ops.conv_transpose_cl(h, w_dual, b_dual, s, in_leaky=self.slope, res=exc, act=act)
                h = self._residual_stacks(h, i)
                continue
            if act is not None:
                ops.set_image(h, self.slope, act)  # the repeat convolution below reads the same activated image
            if self.repeat_upsample:
                conv = self.repeat_upsamples[i][2]
                c = conv.conv1d
                w, tap = conv_weight(c)
                rep = ops.conv_cl(h, w, c.bias, pad=conv.pad, up=s, Tout=h.shape[1] * s, in_leaky=self.slope,
                                  tap_major=tap)
            else:
                rep = None
            if excitation is not None:  # rep + e, accumulated in the strided convolution's epilogue
                rep = self.source_downs[i].forward_cl(excitation, res=rep)
            if act is not None:
                h = up_layer.forward_cl(h, in_leaky=self.slope, res=rep, act=act)
            else:
                h = up_layer.forward_cl(h, in_leaky=self.slope, res=rep)
            h = self._residual_stacks(h, i)
        # F.leaky_relu default slope 0.01 (reference :178), fused into conv_post's loader
        h = self.conv_post.forward_cl(h, in_leaky=0.01)
        return torch.tanh(h).transpose(1, 2)

    def _residual_stacks(self, h, i, image_slope=None):
        # the num_kernels residual stacks of a stage read the same h and are summed: independent branches.  One stream
        # each when a backward pass will follow (their weight gradients are the low-occupancy launches that gain:
        # GAN step 66.9 -> 60.3 ms); a forward-only pass is a chain of chip-filling launches and is 9 % faster
        # sequentially (4.09 vs 4.47 ms at batch 32 x 8192, profiles/r02_runAB_*)
        blocks = self.conv_blocks[i * self.num_kernels:(i + 1) * self.num_kernels]
        ops.act_image(h, self.slope)  # one activated bf16 image for the first convolution of every stack (bf16 mode)
        thunks = [(lambda b=b, h=h: b.forward_cl(h)) for b in blocks]
        # [round 4] the mean over the stacks (reference :160-176) and the activated bf16 image its consumer reads are ONE
        # launch (ops.mean_many); its backward hands every branch its own gradient buffer, which is what private_grads
        # (one clone per branch) was for
        fused = ops.mean_many_applies(len(blocks), h)
        ys = (ops.parallel_branches(thunks, inputs=(h,), private_grads=not fused) if torch.is_grad_enabled()
              else [t() for t in thunks])
        if fused:
            return ops.mean_many(ys, image_slope=image_slope)
        xs = ys[0]
        for y in ys[1:]:
            xs = xs + y
        return xs / self.num_kernels

    def _dual_path_weight(self, i, s):
        """(Cin, Cout, J*s) weight and (Cout) bias of the stage's two paths as one causal transposed convolution:
        w[ci, co, r + j*s] = w_T[ci, co, r + j*s] (the transposed convolution's own taps, zero beyond its kernel)
                             + sum_{k : tap k at phase r reads token q - j} w_7[co, ci, k]."""
        up, conv = self.transpose_upsamples[i][1], self.repeat_upsamples[i][2]
        w_t = effective_weight(up.deconv)                      # (Cin, Cout, K_T)
        c = conv.conv1d
        w7, tap = conv_weight(c)                               # (k7, Cout, Cin) tap-major, or (Cout, Cin, k7)
        k7 = c.kernel_size[0]
        sel, J = _dual_path_selector(k7, s, w_t.device)
        J = max(J, w_t.shape[2] // s)
        if sel.shape[1] < J * s:
            sel = F.pad(sel, (0, J * s - sel.shape[1]))
        if tap:
            rep = torch.matmul(w7.permute(2, 1, 0), sel)       # (Cin, Cout, k7) @ (k7, J*s)
        else:
            rep = torch.matmul(w7.permute(1, 0, 2), sel)
        w = rep + F.pad(w_t, (0, J * s - w_t.shape[2])) if w_t.shape[2] < J * s else rep + w_t
        bias = None
        if up.deconv.bias is not None or c.bias is not None:
            bias = (up.deconv.bias if c.bias is None else c.bias if up.deconv.bias is None else up.deconv.bias + c.bias)
        return w, bias

    def remove_weight_norm(self):
        print("Removing weight norm...")
        for layer in self.transpose_upsamples:
            layer[-1].remove_weight_norm()
        for layer in self.repeat_upsamples:
            layer[-1].remove_weight_norm()
        for layer in self.conv_blocks:
            layer.remove_weight_norm()
        self.conv_pre.remove_weight_norm()
        self.conv_post.remove_weight_norm()
        if self.nsf_enable:
            self.source_module.remove_weight_norm()
            for layer in self.source_downs:
                layer.remove_weight_norm()


def _norm_f(use_spectral_norm):
    """weight_norm, or torch's spectral_norm for the ``follow_official_norm`` discriminators (reference :217,321)."""
    return spectral_norm if use_spectral_norm else weight_norm


class PeriodDiscriminator(torch.nn.Module):
    """(k,1)-Conv2d stack over the period-folded waveform (reference :200-267).  In channels-last the
    fold is a pure view: (B, T) -> (B, T/p, p, 1); each conv strides along T/p with p independent columns."""

    def __init__(self, in_channels=1, out_channels=1, period=3, kernel_sizes=[5, 3], channels=32,
                 downsample_scales=[3, 3, 3, 3, 1], max_downsample_channels=1024, bias=True,
                 nonlinear_activation="LeakyReLU", nonlinear_activation_params={"negative_slope": 0.1},
                 use_spectral_norm=False):
        super(PeriodDiscriminator, self).__init__()
        weight_norm = _norm_f(use_spectral_norm)
        self.period = period
        self.slope = activation_slope(nonlinear_activation, nonlinear_activation_params)
        self.convs = nn.ModuleList()
        in_chs, out_chs = in_channels, channels
        for downsample_scale in downsample_scales:
            self.convs.append(torch.nn.Sequential(
                weight_norm(nn.Conv2d(in_chs, out_chs, (kernel_sizes[0], 1), (downsample_scale, 1),
                                      padding=((kernel_sizes[0] - 1) // 2, 0))),
                getattr(torch.nn, nonlinear_activation)(**nonlinear_activation_params)))
            in_chs = out_chs
            out_chs = min(out_chs * 4, max_downsample_channels)
        self.conv_post = nn.Conv2d(out_chs, out_channels, (kernel_sizes[1] - 1, 1), 1,
                                   padding=((kernel_sizes[1] - 1) // 2, 0))

    def forward(self, x):
        b, c, t = x.shape
        p = self.period
        if t % p != 0:
            x = F.pad(x, (0, p - (t % p)), "reflect")
            t = x.shape[-1]
        h = x.reshape(b, t // p, p, c) if c == 1 else x.transpose(1, 2).reshape(b, t // p, p, c)
        fmap = []
        for layer in self.convs:
            conv = layer[0]
            w, tap = conv_weight(conv)
            h = ops.conv_cl(h, w, conv.bias, stride=conv.stride[0], pad=conv.padding[0], inner=p,
                            out_leaky=self.slope, tap_major=tap, image=None)
            fmap.append(h.permute(0, 3, 1, 2))
        cp = self.conv_post
        h = ops.conv_cl(h, cp.weight.squeeze(-1), cp.bias, stride=1, pad=cp.padding[0], inner=p,
                        Tout=h.shape[1] + 2 * cp.padding[0] - cp.kernel_size[0] + 1)
        out = h.permute(0, 3, 1, 2)
        fmap.append(out)
        return torch.flatten(out, 1, -1), fmap


class MultiPeriodDiscriminator(torch.nn.Module):
    def __init__(self, periods=[2, 3, 5, 7, 11], discriminator_params={
            "in_channels": 1, "out_channels": 1, "kernel_sizes": [5, 3], "channels": 32,
            "downsample_scales": [3, 3, 3, 3, 1], "max_downsample_channels": 1024, "bias": True,
            "nonlinear_activation": "LeakyReLU", "nonlinear_activation_params": {"negative_slope": 0.1},
            "use_spectral_norm": False}):
        super(MultiPeriodDiscriminator, self).__init__()
        self.discriminators = nn.ModuleList()
        for period in periods:
            params = copy.deepcopy(discriminator_params)
            params["period"] = period
            self.discriminators += [PeriodDiscriminator(**params)]

    def forward(self, y):
        # the period discriminators are independent: one HIP stream each (ops.parallel_branches), forward and backward
        outs = ops.parallel_branches([(lambda d=d: d(y)) for d in self.discriminators], inputs=(y,))
        return [o[0] for o in outs], [o[1] for o in outs]


class ScaleDiscriminator(torch.nn.Module):
    """Grouped strided Conv1d stack (reference :305-407); groups are batched over gridDim.z."""

    def __init__(self, in_channels=1, out_channels=1, kernel_sizes=[15, 41, 5, 3], channels=128,
                 max_downsample_channels=1024, max_groups=16, bias=True, downsample_scales=[2, 2, 4, 4, 1],
                 nonlinear_activation="LeakyReLU", nonlinear_activation_params={"negative_slope": 0.1},
                 use_spectral_norm=False):
        super(ScaleDiscriminator, self).__init__()
        weight_norm = _norm_f(use_spectral_norm)
        assert len(kernel_sizes) == 4
        for ks in kernel_sizes:
            assert ks % 2 == 1
        self.slope = activation_slope(nonlinear_activation, nonlinear_activation_params)
        act = lambda: getattr(torch.nn, nonlinear_activation)(**nonlinear_activation_params)  # noqa: E731
        self.convs = nn.ModuleList()
        self.convs.append(torch.nn.Sequential(
            weight_norm(nn.Conv1d(in_channels, channels, kernel_sizes[0], bias=bias, padding=(kernel_sizes[0] - 1) // 2)),
            act()))
        in_chs, out_chs, groups = channels, channels, 4
        for downsample_scale in downsample_scales:
            self.convs.append(torch.nn.Sequential(
                weight_norm(nn.Conv1d(in_chs, out_chs, kernel_size=kernel_sizes[1], stride=downsample_scale,
                                      padding=(kernel_sizes[1] - 1) // 2, groups=groups, bias=bias)), act()))
            in_chs = out_chs
            out_chs = min(in_chs * 2, max_downsample_channels)
            groups = min(groups * 4, max_groups)
        out_chs = min(in_chs * 2, max_downsample_channels)
        self.convs.append(torch.nn.Sequential(
            weight_norm(nn.Conv1d(in_chs, out_chs, kernel_size=kernel_sizes[2], stride=1,
                                  padding=(kernel_sizes[2] - 1) // 2, bias=bias)), act()))
        self.conv_post = weight_norm(nn.Conv1d(out_chs, out_channels, kernel_size=kernel_sizes[3], stride=1,
                                               padding=(kernel_sizes[3] - 1) // 2, bias=bias))

    def forward_cl(self, h):
        fmap = []
        for layer in self.convs:
            c = layer[0]
            w, tap = conv_weight(c)
            h = ops.conv_cl(h, w, c.bias, stride=c.stride[0], pad=c.padding[0], groups=c.groups,
                            out_leaky=self.slope, tap_major=tap, image=None)
            fmap.append(h.transpose(1, 2))
        c = self.conv_post
        w, tap = conv_weight(c)
        h = ops.conv_cl(h, w, c.bias, stride=1, pad=c.padding[0], tap_major=tap)
        out = h.transpose(1, 2)
        fmap.append(out)
        return torch.flatten(out, 1, -1), fmap

    def forward(self, x):
        return self.forward_cl(x.transpose(1, 2).contiguous())


class DWT1DForward(nn.Module):
    """Single-level db3 analysis, zero padding (pytorch_wavelets.DWT1DForward(wave="db3", J=1) in the reference,
    hifigan.py:445-448 -- an un-vendored dependency, restated: taps pinned by their closed form, the zero-padding /
    decimation-phase convention PARITY UNPINNED).  One stride-2
    two-channel FIR launch; output (B, out, 2) channels-last == cat([yl, yh], dim=1) of the reference."""

    def __init__(self, J=1, wave="db3", mode="zero"):
        super().__init__()
        assert J == 1 and wave == "db3" and mode == "zero"
        self.register_buffer("h0", torch.tensor(DB3_DEC_LO[::-1]).view(1, 1, 6))
        self.register_buffer("h1", torch.tensor(DB3_DEC_HI[::-1]).view(1, 1, 6))

    def forward_cl(self, x):
        N = x.shape[1]
        out = (N + 5) // 2
        p = 2 * (out - 1) - N + 6
        w = torch.cat([self.h0, self.h1], dim=0)  # (2, 1, 6) correlation filters
        return ops.conv_cl(x, w, None, stride=2, pad=p // 2, Tout=out)

    def forward(self, x):
        y = self.forward_cl(x.transpose(1, 2).contiguous())
        return y[:, :, 0:1].transpose(1, 2), [y[:, :, 1:2].transpose(1, 2)]


class MultiScaleDiscriminator(torch.nn.Module):
    def __init__(self, scales=3, downsample_pooling="DWT", downsample_pooling_params={
            "kernel_size": 4, "stride": 2, "padding": 2}, discriminator_params={
            "in_channels": 1, "out_channels": 1, "kernel_sizes": [15, 41, 5, 3], "channels": 128,
            "max_downsample_channels": 1024, "max_groups": 16, "bias": True, "downsample_scales": [2, 2, 4, 4, 1],
            "nonlinear_activation": "LeakyReLU", "nonlinear_activation_params": {"negative_slope": 0.1}},
            follow_official_norm=False):
        super(MultiScaleDiscriminator, self).__init__()
        self.discriminators = torch.nn.ModuleList()
        for i in range(scales):
            params = copy.deepcopy(discriminator_params)
            if follow_official_norm:
                params["use_spectral_norm"] = True if i == 0 else False
            self.discriminators += [ScaleDiscriminator(**params)]
        if downsample_pooling == "DWT":
            self.meanpools = nn.ModuleList([DWT1DForward(wave="db3", J=1), DWT1DForward(wave="db3", J=1)])
            self.aux_convs = nn.ModuleList([weight_norm(nn.Conv1d(2, 1, 15, 1, padding=7)),
                                            weight_norm(nn.Conv1d(2, 1, 15, 1, padding=7))])
        else:  # any other value: the reference's fixed AvgPool1d(4, 2, padding=2) pair, no auxiliary convolutions (:456-459)
            self.meanpools = nn.ModuleList([nn.AvgPool1d(4, 2, padding=2), nn.AvgPool1d(4, 2, padding=2)])
            self.aux_convs = None

    def forward(self, y):
        # the pooling chain is sequential (and cheap); the scale discriminators on its outputs are independent
        hs = [y.transpose(1, 2).contiguous()]
        for i in range(1, len(self.discriminators)):
            if self.aux_convs is None:  # one-channel average pooling (no shipped yaml selects it): two small launches
                hs.append(self.meanpools[i - 1](hs[-1].transpose(1, 2)).transpose(1, 2).contiguous())
                continue
            h = self.meanpools[i - 1].forward_cl(hs[-1])
            c = self.aux_convs[i - 1]
            hs.append(ops.conv_cl(h, effective_weight(c), c.bias, pad=7, out_leaky=0.1))
        outs = ops.parallel_branches([(lambda d=d, h=h: d.forward_cl(h)) for d, h in zip(self.discriminators, hs)],
                                     inputs=hs)
        return [o[0] for o in outs], [o[1] for o in outs]


class SpecDiscriminator(torch.nn.Module):
    """(k,1)-Conv2d stack over the STFT magnitude (reference :481-581).  The spectrogram is computed under no_grad, as
    in the reference (the generator receives no gradient through this discriminator's input transform), with the
    magnitude kernel of kantts.utils.audio_torch.stft.  Layout: (B, frames, W, bins) channels-last; W is the reference's
    trailing singleton axis, which its square ``padding=(k-1)//2`` widens by 2*pad of zero columns at every layer
    (those columns carry bias-only activations into the feature maps and the output, and are reproduced here)."""

    def __init__(self, channels=32, init_kernel=15, kernel_size=11, stride=2, use_spectral_norm=False, fft_size=1024,
                 shift_size=120, win_length=600, window="hann_window", nonlinear_activation="LeakyReLU",
                 nonlinear_activation_params={"negative_slope": 0.1}):
        super(SpecDiscriminator, self).__init__()
        self.fft_size, self.shift_size, self.win_length = fft_size, shift_size, win_length
        norm_f = _norm_f(use_spectral_norm)
        self.slope = activation_slope(nonlinear_activation, nonlinear_activation_params)
        final_kernel, post_conv_kernel, blocks = 5, 3, 3
        act = lambda: getattr(torch.nn, nonlinear_activation)(**nonlinear_activation_params)  # noqa: E731
        self.convs = nn.ModuleList()
        self.convs.append(torch.nn.Sequential(
            norm_f(nn.Conv2d(fft_size // 2 + 1, channels, (init_kernel, 1), (1, 1), padding=(init_kernel - 1) // 2)), act()))
        for _ in range(blocks):
            self.convs.append(torch.nn.Sequential(
                norm_f(nn.Conv2d(channels, channels, (kernel_size, 1), (stride, 1), padding=(kernel_size - 1) // 2)), act()))
        self.convs.append(torch.nn.Sequential(
            norm_f(nn.Conv2d(channels, channels, (final_kernel, 1), (1, 1), padding=(final_kernel - 1) // 2)), act()))
        self.conv_post = norm_f(nn.Conv2d(channels, 1, (post_conv_kernel, 1), (1, 1),
                                          padding=((post_conv_kernel - 1) // 2, 0)))
        self.window_name = window
        self.register_buffer("window", getattr(torch, window)(win_length))

    @staticmethod
    def _widen(h, pw):
        if pw == 0:
            return h
        z = h.new_zeros(h.shape[0], h.shape[1], pw, h.shape[3])
        return torch.cat([z, h, z], dim=2)

    def forward(self, wav):
        from kantts.utils.audio_torch import stft

        with torch.no_grad():
            x_mag = stft(torch.squeeze(wav, 1).detach(), self.fft_size, self.shift_size, self.win_length, self.window_name)
        h = x_mag.unsqueeze(2)  # (B, frames, W=1, bins)
        fmap = []
        for layer in self.convs:
            conv = layer[0]
            w, tap = conv_weight(conv)
            ph, pw = conv.padding
            h = self._widen(h, pw)
            h = ops.conv_cl(h, w, conv.bias, stride=conv.stride[0], pad=ph, inner=h.shape[2], out_leaky=self.slope,
                            tap_major=tap, Tout=(h.shape[1] + 2 * ph - conv.kernel_size[0]) // conv.stride[0] + 1)
            fmap.append(h.permute(0, 3, 1, 2))
        cp = self.conv_post
        w, tap = conv_weight(cp)
        h = ops.conv_cl(h, w, cp.bias, stride=1, pad=cp.padding[0], inner=h.shape[2], tap_major=tap,
                        Tout=h.shape[1] + 2 * cp.padding[0] - cp.kernel_size[0] + 1)
        out = h.permute(0, 3, 1, 2)
        fmap.append(out)
        return out.squeeze(-1), fmap


class MultiSpecDiscriminator(torch.nn.Module):
    """Reference :584-617.  (Its default ``discriminator_params`` carry a ``kernel_sizes`` key that SpecDiscriminator
    does not accept -- constructing the reference with defaults raises TypeError, and so does this class; configs pass
    their own parameters.)"""

    def __init__(self, fft_sizes=[1024, 2048, 512], hop_sizes=[120, 240, 50], win_lengths=[600, 1200, 240],
                 discriminator_params={"channels": 15, "init_kernel": 1, "kernel_sizes": 11, "stride": 2,
                                       "use_spectral_norm": False, "window": "hann_window",
                                       "nonlinear_activation": "LeakyReLU",
                                       "nonlinear_activation_params": {"negative_slope": 0.1}}):
        super(MultiSpecDiscriminator, self).__init__()
        self.discriminators = nn.ModuleList()
        for fft_size, hop_size, win_length in zip(fft_sizes, hop_sizes, win_lengths):
            params = copy.deepcopy(discriminator_params)
            params["fft_size"], params["shift_size"], params["win_length"] = fft_size, hop_size, win_length
            self.discriminators += [SpecDiscriminator(**params)]

    def forward(self, y):
        y_d, fmap = [], []
        for d in self.discriminators:
            x, x_map = d(y)
            y_d.append(x)
            fmap.append(x_map)
        return y_d, fmap
