"""Randomised sweep of the channels-last convolutions (ops.conv_cl / conv_transpose_cl; fixture ``dv``: the emulated C ABI on
the CPU suite, the HIP kernels on device tensors under ``-m gpu``) against ATen's conv1d / conv_transpose1d on the CPU: strides, dilations, paddings, groups (incl. the block-diagonal group
packing), nearest-neighbour read-through upsampling, the MPD fold, fused LeakyReLU on either side, residuals -- forward,
input gradient, weight and bias gradients.  Small shapes, fixed seed: ~40 configurations in a few seconds; the same
wrapper code then drives the HIP kernels (per-entry-point GPU parity is in test_hifigan.py)."""
import random

import torch
import torch.nn.functional as F

from util import rel_l2


def _ref_conv(x, w, b, stride, dil, pad, Tout, groups, up, il, ol, res):
    # x (B, T, [P,] C) channels-last -> the op's contract in plain torch
    folded = x.dim() == 4
    if folded:
        B, T, P, C = x.shape
        xx = x.permute(0, 2, 3, 1).reshape(B * P, C, T)
    else:
        xx = x.transpose(1, 2)
    if il is not None:
        xx = F.leaky_relu(xx, il)
    if up > 1:
        xx = torch.repeat_interleave(xx, up, dim=2)
    K = w.shape[-1]
    need = (Tout - 1) * stride + dil * (K - 1) + 1
    right = max(0, need - xx.shape[2] - pad)
    y = F.conv1d(F.pad(xx, (pad, right)), w, b, stride=stride, dilation=dil, groups=groups)[:, :, :Tout]
    if ol is not None:
        y = F.leaky_relu(y, ol)
    if folded:
        y = y.reshape(B, P, -1, Tout).permute(0, 3, 1, 2)
    else:
        y = y.transpose(1, 2)
    return y if res is None else y + res


def test_conv_cl_random_configurations(dv):
    from kantts._hip import ops

    rnd = random.Random(20240917)
    g = torch.Generator().manual_seed(7)
    done = 0
    while done < 40:
        groups = rnd.choice([1, 1, 1, 2, 4, 8])
        cr, ng = rnd.choice([1, 2, 4, 8, 12, 16, 32]), rnd.choice([1, 4, 8, 16, 20])
        Cin, Cout = groups * cr, groups * ng
        K, stride, dil = rnd.choice([1, 2, 3, 5, 7, 9]), rnd.choice([1, 1, 2, 3, 4]), rnd.choice([1, 1, 2, 3])
        up = rnd.choice([1, 1, 1, 2, 4]) if stride == 1 and groups == 1 else 1
        P = rnd.choice([1, 1, 1, 2, 3]) if up == 1 else 1
        B, T = rnd.choice([1, 2, 3]), rnd.randint(5, 41)
        pad = rnd.randint(0, dil * (K - 1))
        span = T * up + pad - dil * (K - 1) - 1
        if span < 0:
            continue
        Tout = rnd.randint(1, span // stride + 1 + (1 if rnd.random() < 0.3 else 0))  # sometimes one past: right zero pad
        il, ol = rnd.choice([None, None, 0.1]), rnd.choice([None, None, 0.2])
        shape = (B, T, P, Cin) if P > 1 else (B, T, Cin)
        x = torch.randn(shape, generator=g).requires_grad_(True)
        w = (torch.randn(Cout, cr, K, generator=g) / (cr * K) ** 0.5).requires_grad_(True)
        b = torch.randn(Cout, generator=g).requires_grad_(True) if rnd.random() < 0.7 else None
        rshape = (B, Tout, P, Cout) if P > 1 else (B, Tout, Cout)
        res = torch.randn(rshape, generator=g).requires_grad_(True) if rnd.random() < 0.3 else None
        cfg = dict(groups=groups, cr=cr, ng=ng, K=K, stride=stride, dil=dil, up=up, P=P, B=B, T=T, pad=pad, Tout=Tout,
                   il=il, ol=ol, bias=b is not None, res=res is not None)
        y = ops.conv_cl(dv(x), dv(w), dv(b), stride=stride, dilation=dil, pad=pad, Tout=Tout, up=up, groups=groups, inner=P,
                        in_leaky=il, out_leaky=ol, res=dv(res))
        ref = _ref_conv(x, w, b, stride, dil, pad, Tout, groups, up, il, ol, res)
        assert y.shape == ref.shape, cfg
        assert float((dv.back(y) - ref).detach().abs().max()) <= 2e-5 * max(1.0, float(ref.detach().abs().max())), cfg
        cot = torch.randn(ref.shape, generator=g)
        leaves = [t for t in (x, w, b, res) if t is not None]
        got = dv.back(torch.autograd.grad((y * dv(cot)).sum(), dv(leaves)))
        exp = torch.autograd.grad((ref * cot).sum(), leaves)
        for a, e in zip(got, exp):
            assert rel_l2(a, e) < 2e-5 or float((a - e).abs().max()) < 1e-6, cfg
        done += 1


def test_conv_transpose_cl_random_configurations(dv):
    """Causal polyphase transposed convolution (kernel = taps * stride, output trimmed to T * stride) vs ATen."""
    from kantts._hip import ops

    rnd = random.Random(99)
    g = torch.Generator().manual_seed(3)
    for _ in range(16):
        s, taps = rnd.choice([2, 3, 4, 5, 8]), rnd.choice([1, 2, 3])
        K = s * taps
        Cin, Cout = rnd.choice([4, 8, 16, 32]), rnd.choice([1, 4, 8, 12])
        B, T = rnd.choice([1, 2]), rnd.choice([3, 7, 20, 70])   # T >= 64 takes the window kernel, below the GEMM form
        il = rnd.choice([None, 0.1])
        x = torch.randn(B, T, Cin, generator=g).requires_grad_(True)
        w = (torch.randn(Cin, Cout, K, generator=g) / (Cin * taps) ** 0.5).requires_grad_(True)
        b = torch.randn(Cout, generator=g).requires_grad_(True)
        res = torch.randn(B, T * s, Cout, generator=g).requires_grad_(True) if rnd.random() < 0.5 else None
        cfg = dict(s=s, taps=taps, Cin=Cin, Cout=Cout, B=B, T=T, il=il, res=res is not None)
        y = ops.conv_transpose_cl(dv(x), dv(w), dv(b), s, in_leaky=il, res=dv(res))
        xx = x.transpose(1, 2)
        if il is not None:
            xx = F.leaky_relu(xx, il)
        ref = F.conv_transpose1d(xx, w, b, stride=s)[:, :, :T * s].transpose(1, 2)
        if res is not None:
            ref = ref + res
        assert float((dv.back(y) - ref).detach().abs().max()) <= 2e-5 * max(1.0, float(ref.detach().abs().max())), cfg
        cot = torch.randn(ref.shape, generator=g)
        leaves = [t for t in (x, w, b, res) if t is not None]
        for a, e in zip(dv.back(torch.autograd.grad((y * dv(cot)).sum(), dv(leaves))),
                        torch.autograd.grad((ref * cot).sum(), leaves)):
            assert rel_l2(a, e) < 2e-5, cfg
