"""HiFi-GAN building blocks on HIP kernels (reference kantts/models/hifigan/layers.py:15-226).

Same classes / constructor arguments / ``state_dict`` keys (``conv1d.weight_g``, ``conv1d.weight_v``,
``deconv.weight_g`` ...).  The torch modules are parameter holders; the arithmetic is the segmented MFMA
GEMM with convolution token maps (kantts._hip.ops.conv_cl / conv_transpose_cl) on channels-last
activations ``(B, T, C)``.  ``forward`` keeps the reference's ``(B, C, T)`` contract (it transposes at the
boundary); models call ``forward_cl`` and stay channels-last end to end, with the LeakyReLU that
precedes a convolution fused into its operand loader and residual adds fused into its epilogue.
"""
import torch
import torch.nn as nn
from torch.nn.utils import remove_weight_norm, weight_norm

from kantts._hip import ops
from kantts.models.utils import init_weights


def get_padding(kernel_size, dilation=1):
    return int((kernel_size * dilation - dilation) / 2)


def effective_weight(m):
    """g * v / ||v|| through the fused weight-norm kernel, or the plain weight after remove_weight_norm()."""
    if hasattr(m, "weight_g"):
        return ops.weight_norm(m.weight_v, m.weight_g)
    return m.weight


def conv_weight(m):
    """(weight, tap_major) for ops.conv_cl: weight-normed convolutions with >= 4 input channels per group get their
    weight straight in the kernels' tap-major layout (K, Cout, Cin_g) from the weight-norm kernel; 1-3 channel
    layers (streaming kernels) and plain weights keep the parameter layout (Cout, Cin_g, K)."""
    if hasattr(m, "weight_g"):
        w = ops.weight_norm_image(m)  # one launch per network and optimizer step (ParamArena.build_weight_norm_images)
        if w is not None:
            return w, True
        v = m.weight_v
        if v.dim() == 4:  # Conv2d((k,1)) of the period discriminators
            v = v.squeeze(-1)
        if v.shape[1] % 4 == 0:
            return ops.weight_norm_tap(v, m.weight_g, groups=getattr(m, "groups", 1)), True
        return ops.weight_norm(v, m.weight_g), False
    if hasattr(m, "weight_orig"):
        # torch.nn.utils.spectral_norm (follow_official_norm discriminators, reference hifigan.py:217,321): the
        # weight / sigma reparametrisation lives in a forward pre-hook of the holder module, which is never called
        # here -- run the hook (one power iteration in training mode, on the small weight matrix) ourselves
        for hook in m._forward_pre_hooks.values():
            if type(hook).__name__ == "SpectralNorm":
                hook(m, None)
    w = m.weight
    return (w.squeeze(-1) if w.dim() == 4 else w), False


class Conv1d(torch.nn.Module):
    """Weight-normed Conv1d with symmetric padding (reference :15-49)."""

    causal = False

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 padding_mode="zeros"):
        super().__init__()
        self.conv1d = weight_norm(nn.Conv1d(in_channels, out_channels, kernel_size, stride,
                                            padding=0 if self.causal else padding, dilation=dilation, groups=groups,
                                            bias=bias, padding_mode=padding_mode))
        self.conv1d.apply(init_weights)
        self.pad = (kernel_size - 1) * dilation if self.causal else padding

    def forward_cl(self, x, in_leaky=None, out_leaky=None, res=None, image=False):
        c = self.conv1d
        k, d, s = c.kernel_size[0], c.dilation[0], c.stride[0]
        Tin = x.shape[1]
        # causal: (k-1)*d zeros on the left only (reference :86-91) -> ceil(Tin / s) outputs
        Tout = (Tin - 1) // s + 1 if self.causal else (Tin + 2 * self.pad - d * (k - 1) - 1) // s + 1
        w, tap = conv_weight(c)
        return ops.conv_cl(x, w, c.bias, stride=s, dilation=d, pad=self.pad, Tout=Tout, groups=c.groups,
                           in_leaky=in_leaky, out_leaky=out_leaky, res=res, tap_major=tap, image=image)

    def forward(self, x):
        return self.forward_cl(x.transpose(1, 2).contiguous()).transpose(1, 2)

    def remove_weight_norm(self):
        remove_weight_norm(self.conv1d)


class CausalConv1d(Conv1d):
    """Left-padded (k-1)*d causal Conv1d (reference :52-91); the pad is an index shift, never a copy."""

    causal = True


class ConvTranspose1d(torch.nn.Module):
    """Non-causal weight-normed ConvTranspose1d (reference :94-121)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding=0, output_padding=0):
        super().__init__()
        self.deconv = weight_norm(nn.ConvTranspose1d(in_channels, out_channels, kernel_size, stride, padding=padding,
                                                     output_padding=0))
        self.deconv.apply(init_weights)

        if kernel_size % stride != 0 or (kernel_size - stride) % 2 != 0 or padding != (kernel_size - stride) // 2:
            raise NotImplementedError("only kernel = m*stride with padding (kernel - stride) / 2 (every shipped yaml)")
        self.stride, self.padding = stride, padding

    def forward_cl(self, x, in_leaky=None, res=None):
        """Full transposed convolution = the polyphase form over one extra (zero) input token; the symmetric
        padding of the reference (:94-121) drops ``padding`` samples at both ends of it."""
        B, T, _ = x.shape
        s, pad = self.stride, self.padding
        taps = self.deconv.kernel_size[0] // s
        xz = torch.nn.functional.pad(x, (0, 0, 0, taps - 1))
        full = ops.conv_transpose_cl(xz, effective_weight(self.deconv), self.deconv.bias, s, in_leaky=in_leaky)
        y = full[:, pad:pad + T * s, :]
        return y if res is None else y + res

    def forward(self, x):
        return self.forward_cl(x.transpose(1, 2).contiguous()).transpose(1, 2)

    def remove_weight_norm(self):
        remove_weight_norm(self.deconv)


class CausalConvTranspose1d(torch.nn.Module):
    """Causal ConvTranspose1d: full transposed conv, last (k - stride) samples dropped (reference :125-165)
    == polyphase form with k/stride taps per output phase (ops.conv_transpose_cl)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding=0, output_padding=0):
        super().__init__()
        self.deconv = weight_norm(nn.ConvTranspose1d(in_channels, out_channels, kernel_size, stride, padding=0,
                                                     output_padding=0))
        self.stride = stride
        self.deconv.apply(init_weights)
        self.pad = kernel_size - stride

    def forward_cl(self, x, in_leaky=None, res=None, act=None):
        return ops.conv_transpose_cl(x, effective_weight(self.deconv), self.deconv.bias, self.stride, in_leaky=in_leaky,
                                     res=res, act=act)

    def forward(self, x):
        return self.forward_cl(x.transpose(1, 2).contiguous()).transpose(1, 2)

    def remove_weight_norm(self):
        remove_weight_norm(self.deconv)


def activation_slope(name, params):
    """The reference instantiates ``getattr(torch.nn, nonlinear_activation)(**nonlinear_activation_params)``
    (hifigan.py:69-71, layers.py:209-211).  Every convolution kernel here applies its input / output activation itself,
    parametrised by one slope: LeakyReLU(negative_slope) and ReLU (= slope 0) are expressible, anything else is not."""
    if name == "LeakyReLU":
        return float(params.get("negative_slope", 0.01))
    if name == "ReLU":
        return 0.0
    raise NotImplementedError("activation %s: only LeakyReLU / ReLU are fused into the convolution kernels" % name)


class ResidualBlock(torch.nn.Module):
    """3 x [LeakyReLU -> dilated conv -> LeakyReLU -> conv -> + x] (reference :168-226): 6 GEMM launches,
    activations and residual adds fused."""

    def __init__(self, channels, kernel_size=3, dilation=(1, 3, 5), nonlinear_activation="LeakyReLU",
                 nonlinear_activation_params={"negative_slope": 0.1}, causal=False):
        super().__init__()
        assert kernel_size % 2 == 1, "Kernal size must be odd number."
        conv_cls = CausalConv1d if causal else Conv1d
        self.convs1 = nn.ModuleList([
            conv_cls(channels, channels, kernel_size, 1, dilation=dilation[i],
                     padding=get_padding(kernel_size, dilation[i])) for i in range(len(dilation))])
        self.convs2 = nn.ModuleList([
            conv_cls(channels, channels, kernel_size, 1, dilation=1, padding=get_padding(kernel_size, 1))
            for i in range(len(dilation))])
        self.activation = getattr(torch.nn, nonlinear_activation)(**nonlinear_activation_params)
        self.slope = activation_slope(nonlinear_activation, nonlinear_activation_params)  # refuses what cannot be fused

    def forward_cl(self, x):
        # bf16 mode, wide enough for the MFMA kernels: the whole stack as one autograd node (ops._ResStackBF16)
        if (ops.res_stack_ok(x, self.convs1[0].conv1d.kernel_size[0])
                and all(hasattr(c.conv1d, "weight_g") and c.conv1d.groups == 1 and c.conv1d.stride[0] == 1
                        and c.conv1d.weight_v.shape[1] % 4 == 0 for c in list(self.convs1) + list(self.convs2))):
            spec = []
            for c1, c2 in zip(self.convs1, self.convs2):
                (w1, t1), (w2, t2) = conv_weight(c1.conv1d), conv_weight(c2.conv1d)
                if not (t1 and t2):
                    spec = None
                    break
                spec.append((w1, c1.conv1d.bias, c1.conv1d.kernel_size[0], c1.conv1d.dilation[0], c1.pad, w2, c2.conv1d.bias,
                             c2.pad))
            y = ops.res_stack(x, self.slope, spec) if spec else None
            if y is not None:
                return y
        # otherwise: every convolution also writes the activated bf16 image its successor reads (ops.set_image)
        n = len(self.convs1)
        for i, (c1, c2) in enumerate(zip(self.convs1, self.convs2)):
            xt = c1.forward_cl(x, in_leaky=self.slope, image=self.slope)
            x = c2.forward_cl(xt, in_leaky=self.slope, res=x, image=self.slope if i + 1 < n else False)
        return x

    def forward(self, x):
        return self.forward_cl(x.transpose(1, 2).contiguous()).transpose(1, 2)

    def remove_weight_norm(self):
        for layer in self.convs1:
            layer.remove_weight_norm()
        for layer in self.convs2:
            layer.remove_weight_norm()


class SourceModule(torch.nn.Module):
    """NSF sine-excitation source (reference :229-290): harmonics 1..H+1 of the frame-level f0, phase by a running sum
    of f0/sr, a random initial phase per harmonic, Gaussian noise, voiced / unvoiced mixing -- all without gradient --
    then a weight-normed 1x1 convolution (H+1 -> 1) and tanh.

    The excitation is a no-grad elementwise preamble of (B, H+1, T) floats and runs as device tensor ops; unlike the
    reference, which samples phase and noise on the host and copies them over, the draws come from the generator of the
    tensors' own device (on the CPU -- the emulated tests -- that is the same torch.distributions call sequence, hence
    the same numbers as the reference for a given seed).  The projection runs on the conv kernels, channels last."""

    def __init__(self, nb_harmonics, upsample_ratio, sampling_rate, alpha=0.1, sigma=0.003):
        super(SourceModule, self).__init__()
        self.nb_harmonics = nb_harmonics
        self.upsample_ratio = upsample_ratio
        self.sampling_rate = sampling_rate
        self.alpha = alpha
        self.sigma = sigma
        self.ffn = nn.Sequential(weight_norm(nn.Conv1d(self.nb_harmonics + 1, 1, kernel_size=1, stride=1)), nn.Tanh())

    @torch.no_grad()
    def excitation(self, pitch, uv):
        """pitch (Hz), uv: (B, 1, frames) -> e (B, H+1, frames * upsample_ratio)."""
        import numpy as np
        from torch.distributions.normal import Normal
        from torch.distributions.uniform import Uniform

        up = int(self.upsample_ratio)
        dev = pitch.device
        # the reference's own call (its source-index rounding for non-power-of-two ratios included)
        pitch_samples = torch.nn.functional.interpolate(pitch, scale_factor=up, mode="nearest")
        uv_samples = torch.nn.functional.interpolate(uv, scale_factor=up, mode="nearest")
        harm = torch.arange(1, self.nb_harmonics + 2, device=dev, dtype=pitch.dtype).view(1, -1, 1)
        F_mat = pitch_samples * harm / self.sampling_rate
        theta_mat = 2 * np.pi * (torch.cumsum(F_mat, dim=-1) % 1)
        one = torch.ones((), device=dev)
        phase_vec = Uniform(low=-np.pi * one, high=np.pi * one).sample(sample_shape=(pitch.size(0), self.nb_harmonics + 1, 1))
        phase_vec[:, 0, :] = 0
        noise = Normal(loc=0.0 * one, scale=self.sigma * one).sample(
            sample_shape=(pitch_samples.size(0), self.nb_harmonics + 1, pitch_samples.size(-1)))
        e_voice = self.alpha * torch.sin(theta_mat + phase_vec) + noise
        e_unvoice = self.alpha / 3 / self.sigma * noise
        return e_voice * uv_samples + e_unvoice * (1 - uv_samples)

    def forward_cl(self, pitch, uv):
        """-> (B, T, 1) channels-last excitation signal."""
        e = self.excitation(pitch, uv).transpose(1, 2).contiguous()
        c = self.ffn[0]
        w, tap = conv_weight(c)
        return torch.tanh(ops.conv_cl(e, w, c.bias, tap_major=tap))

    def forward(self, pitch, uv):
        return self.forward_cl(pitch, uv).transpose(1, 2)

    def remove_weight_norm(self):
        remove_weight_norm(self.ffn[0])
