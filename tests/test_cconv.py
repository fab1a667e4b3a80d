"""bf16 convolution contractions (csrc/cconv.hip): host logic of the bf16-mode convolution path under the emulated ABI
(CPU) and kernel parity on the GPU.

The contract has three layers and each gets its own check:
  * the host logic (ops._CConvCL / the bf16 branch of ops._ConvTransposeCL: operand images, weight re-layouts, group
    packing, token rules of the three contractions) against ATen's conv1d / conv_transpose1d on bf16-rounded operands:
    forward to fp32 summation order, gradients to the one extra bf16 rounding of the incoming gradient;
  * every kernel tile against the numpy model of the entry point (same bf16 operands on both sides: the difference is
    fp32 summation order only) -- GPU;
  * at the HiFi-GAN V1 batch-32 sizes, where no CPU oracle finishes, against the fp32-operand kernels of round 1 on the
    same device (conv_win / conv_wgrad, themselves pinned to the oracle) -- GPU.
The references of the first two layers see bf16-rounded operands (straight-through rounding), so a LeakyReLU gate
cannot flip between the two sides; the batch-32 comparison against fp32 operands uses input-side activations only
(an output-side gate flips for ~0.3 % of the elements under bf16 operand rounding and would dominate the error).
Reference semantics: kantts/models/hifigan/layers.py:15-165, hifigan.py:200-267,305-407."""
import random

import pytest
import torch
import torch.nn.functional as F

from util import emulation, rel_l2, run_both


def _ste_bf16(t):
    """bf16 rounding with a straight-through gradient: the reference then sees the operands the kernels see (no
    LeakyReLU gate can flip between the two sides), while its autograd formulas stay the fp32 ones."""
    return t + (t.detach().to(torch.bfloat16).float() - t.detach())


def _ref_conv(x, w, b, stride, dil, pad, Tout, groups, up, il, ol, res):
    # x (B, T, [P,] C) channels-last -> the op's contract in plain torch, operands rounded to bf16
    folded = x.dim() == 4
    if folded:
        B, T, P, C = x.shape
        xx = x.permute(0, 2, 3, 1).reshape(B * P, C, T)
    else:
        xx = x.transpose(1, 2)
    if il is not None:
        xx = F.leaky_relu(xx, il)
    xx = _ste_bf16(xx)
    if up > 1:
        xx = torch.repeat_interleave(xx, up, dim=2)
    K = w.shape[-1]
    need = (Tout - 1) * stride + dil * (K - 1) + 1
    right = max(0, need - xx.shape[2] - pad)
    y = F.conv1d(F.pad(xx, (pad, right)), _ste_bf16(w), b, stride=stride, dilation=dil, groups=groups)[:, :, :Tout]
    if ol is not None:
        y = F.leaky_relu(y, ol)
    if folded:
        y = y.reshape(B, P, -1, Tout).permute(0, 3, 1, 2)
    else:
        y = y.transpose(1, 2)
    return y if res is None else y + res


@pytest.fixture
def bf16_all_sizes():
    """bf16 mode with the size threshold of the cconv path removed, restored afterwards."""
    import kantts._hip as hip
    from kantts._hip import ops

    old_p, old_t = hip.get_precision(), ops.CCONV_MIN_FLOPS
    hip.set_precision("bf16")
    ops.CCONV_MIN_FLOPS = 0.0
    yield ops
    ops.CCONV_MIN_FLOPS = old_t
    hip.set_precision(old_p)


def _count_calls(monkeypatch, ops):
    calls = {"cconv": 0, "wgrad": 0}
    real_c, real_w = ops.cconv, ops.cconv_wgrad

    def c(*a, **k):
        calls["cconv"] += 1
        return real_c(*a, **k)

    def w(*a, **k):
        calls["wgrad"] += 1
        return real_w(*a, **k)

    monkeypatch.setattr(ops, "cconv", c)
    monkeypatch.setattr(ops, "cconv_wgrad", w)
    return calls


def test_cconv_path_random_configurations_emulated(emulated_cabi, bf16_all_sizes, monkeypatch):
    ops = bf16_all_sizes
    calls = _count_calls(monkeypatch, ops)
    rnd = random.Random(31)
    g = torch.Generator().manual_seed(5)
    done = 0
    while done < 30:
        groups = rnd.choice([1, 1, 1, 2, 4])
        cr, ng = rnd.choice([8, 16, 24, 32, 40, 64]), rnd.choice([8, 16, 24, 32, 72])
        Cin, Cout = groups * cr, groups * ng
        K, stride, dil = rnd.choice([1, 2, 3, 5, 7, 9]), rnd.choice([1, 1, 2, 3, 4]), rnd.choice([1, 1, 2, 3])
        up = rnd.choice([1, 1, 1, 2, 4]) if stride == 1 and groups == 1 else 1
        P = rnd.choice([1, 1, 1, 2, 3]) if up == 1 else 1
        B, T = rnd.choice([1, 2, 3]), rnd.randint(5, 41)
        pad = rnd.randint(0, dil * (K - 1))
        span = T * up + pad - dil * (K - 1) - 1
        if span < 0:
            continue
        Tout = rnd.randint(1, span // stride + 1 + (1 if rnd.random() < 0.3 else 0))
        il, ol = rnd.choice([None, None, 0.1]), rnd.choice([None, None, 0.2])
        shape = (B, T, P, Cin) if P > 1 else (B, T, Cin)
        x = torch.randn(shape, generator=g).requires_grad_(True)
        w = (torch.randn(Cout, cr, K, generator=g) / (cr * K) ** 0.5).requires_grad_(True)
        b = torch.randn(Cout, generator=g).requires_grad_(True) if rnd.random() < 0.7 else None
        rshape = (B, Tout, P, Cout) if P > 1 else (B, Tout, Cout)
        res = torch.randn(rshape, generator=g).requires_grad_(True) if rnd.random() < 0.3 else None
        tap_major = rnd.random() < 0.5
        cfg = dict(groups=groups, cr=cr, ng=ng, K=K, stride=stride, dil=dil, up=up, P=P, B=B, T=T, pad=pad, Tout=Tout,
                   il=il, ol=ol, bias=b is not None, res=res is not None, tap_major=tap_major)
        before = dict(calls)
        wk = w.permute(2, 0, 1).contiguous() if tap_major else w
        y = ops.conv_cl(x, wk, b, stride=stride, dilation=dil, pad=pad, Tout=Tout, up=up, groups=groups, inner=P,
                        in_leaky=il, out_leaky=ol, res=res, tap_major=tap_major)
        ref = _ref_conv(x, w, b, stride, dil, pad, Tout, groups, up, il, ol, res)
        assert y.shape == ref.shape, cfg
        assert float((y - ref).detach().abs().max()) <= 2e-5 * max(1.0, float(ref.detach().abs().max())), cfg
        cot = torch.randn(ref.shape, generator=g)
        leaves = [t for t in (x, w, b, res) if t is not None]
        got = torch.autograd.grad((y * cot).sum(), leaves)
        exp = torch.autograd.grad((ref * cot).sum(), leaves)
        for a, e in zip(got, exp):  # the incoming gradient is rounded to bf16 once more
            ok = rel_l2(a, e) < 6e-3 or float((a - e).abs().max()) < 1e-3
            assert ok, cfg
        assert calls["cconv"] == before["cconv"] + 2 and calls["wgrad"] == before["wgrad"] + 1, cfg  # fwd + dgrad, wgrad
        done += 1


def test_cconv_transposed_random_configurations_emulated(emulated_cabi, bf16_all_sizes, monkeypatch):
    ops = bf16_all_sizes
    calls = _count_calls(monkeypatch, ops)
    rnd = random.Random(77)
    g = torch.Generator().manual_seed(4)
    for _ in range(12):
        s, taps = rnd.choice([2, 4, 8]), rnd.choice([1, 2, 3])
        K = s * taps
        Cin, Cout = rnd.choice([8, 16, 32]), rnd.choice([4, 8, 12])
        if (s * Cout) % 8:
            continue
        B, T = rnd.choice([1, 2]), rnd.choice([16, 20, 70])
        il = rnd.choice([None, 0.1])
        x = torch.randn(B, T, Cin, generator=g).requires_grad_(True)
        w = (torch.randn(Cin, Cout, K, generator=g) / (Cin * taps) ** 0.5).requires_grad_(True)
        b = torch.randn(Cout, generator=g).requires_grad_(True)
        res = torch.randn(B, T * s, Cout, generator=g).requires_grad_(True) if rnd.random() < 0.5 else None
        cfg = dict(s=s, taps=taps, Cin=Cin, Cout=Cout, B=B, T=T, il=il, res=res is not None)
        y = ops.conv_transpose_cl(x, w, b, s, in_leaky=il, res=res)
        xx = x.transpose(1, 2)
        if il is not None:
            xx = F.leaky_relu(xx, il)
        ref = F.conv_transpose1d(_ste_bf16(xx), _ste_bf16(w), b, stride=s)[:, :, :T * s].transpose(1, 2)
        if res is not None:
            ref = ref + res
        assert float((y - ref).detach().abs().max()) <= 2e-5 * max(1.0, float(ref.detach().abs().max())), cfg
        cot = torch.randn(ref.shape, generator=g)
        leaves = [t for t in (x, w, b, res) if t is not None]
        for a, e in zip(torch.autograd.grad((y * cot).sum(), leaves), torch.autograd.grad((ref * cot).sum(), leaves)):
            ok = rel_l2(a, e) < 6e-3
            assert ok, cfg
    assert calls["cconv"] > 0 and calls["wgrad"] > 0


def test_cconv_threshold_keeps_small_layers_on_the_fp32_operand_kernels(emulated_cabi, monkeypatch):
    import kantts._hip as hip
    from kantts._hip import ops

    calls = _count_calls(monkeypatch, ops)
    hip.set_precision("bf16")
    try:
        x = torch.randn(2, 30, 16, requires_grad=True)
        w = torch.randn(16, 16, 3, requires_grad=True)
        ops.conv_cl(x, w, None, pad=1).sum().backward()
        assert calls == {"cconv": 0, "wgrad": 0}
    finally:
        hip.set_precision("fp32")


# ------------------------------------------------------------------------------------------------------------- GPU
def _r(*s, seed, scale=1.0, grad=True):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*s, generator=g) * scale).requires_grad_(grad)


@pytest.mark.gpu
def test_cconv_ops_gpu_vs_emulated(bf16_all_sizes):
    """The bf16-mode wrappers on the device vs the numpy model of the entry points: identical bf16 operands on both
    sides, so forward outputs agree to fp32 summation order; gradients pass through one more bf16 rounding of the
    (device- vs host-computed) incoming gradient, hence 2e-3."""
    ops = bf16_all_sizes
    B, T = 3, 37
    cases = [
        ("causal dilated + res", lambda x, w, b, res: ops.conv_cl(x, w, b, dilation=3, pad=12, in_leaky=0.1, res=res),
         (_r(B, T, 24, seed=1), _r(40, 24, 5, seed=2, scale=0.2), _r(40, seed=3), _r(B, T, 40, seed=4))),
        ("strided grouped", lambda x, w, b: ops.conv_cl(x, w, b, stride=4, pad=20, groups=4, out_leaky=0.1),
         (_r(B, 64, 32, seed=5), _r(64, 8, 41, seed=6, scale=0.1), _r(64, seed=7))),
        ("period fold", lambda x, w, b: ops.conv_cl(x, w, b, stride=3, pad=2, inner=5, out_leaky=0.1),
         (_r(B, 20, 5, 8, seed=8), _r(16, 8, 5, seed=9, scale=0.2), _r(16, seed=10))),
        ("period fold, long", lambda x, w, b: ops.conv_cl(x, w, b, stride=3, pad=2, inner=7, in_leaky=0.1, out_leaky=0.1),
         (_r(2, 301, 7, 32, seed=21), _r(128, 32, 5, seed=22, scale=0.1), _r(128, seed=23))),
        ("nearest upsample", lambda x, w, b: ops.conv_cl(x, w, b, pad=6, up=8, in_leaky=0.1),
         (_r(B, T, 16, seed=11), _r(24, 16, 7, seed=12, scale=0.2), _r(24, seed=13))),
        ("wide k5 fold 11", lambda x, w, b: ops.conv_cl(x, w, b, pad=2, inner=11, out_leaky=0.1),
         (_r(4, 10, 11, 256, seed=31), _r(136, 256, 5, seed=32, scale=0.03), _r(136, seed=33))),
        ("80 -> 72 channels k7", lambda x, w, b: ops.conv_cl(x, w, b, pad=6),
         (_r(2, 300, 80, seed=34), _r(72, 80, 7, seed=35, scale=0.05), _r(72, seed=36))),
        ("polyphase transposed", lambda x, w, b, res: ops.conv_transpose_cl(x, w, b, 8, in_leaky=0.1, res=res),
         (_r(B, T, 16, seed=14), _r(16, 12, 16, seed=15, scale=0.2), _r(12, seed=16), _r(B, T * 8, 12, seed=17))),
        ("polyphase transposed 3 taps", lambda x, w, b: ops.conv_transpose_cl(x, w, b, 4, in_leaky=0.1),
         (_r(2, 70, 32, seed=41), _r(32, 16, 12, seed=42, scale=0.2), _r(16, seed=43))),
    ]
    for name, fn, args in cases:
        go, gg, co, cg = run_both(fn, *args)
        err = float((go[0] - co[0]).abs().max())
        assert err <= 2e-5 * max(1.0, float(co[0].abs().max())), (name, err)
        for a, c in zip(gg, cg):
            assert rel_l2(a, c) < 2e-3, name


@pytest.mark.gpu
@pytest.mark.parametrize("tile", [0, 128128, 256064, 128064, 256032, 64128, 64064, 4128064, 3064064, 4064128])
def test_cconv_every_tile_gpu_vs_emulated(tile):
    """kantts_cconv_launch through the binding with a forced tile: epilogue variants (bias, LeakyReLU, residual, bf16 /
    fp32 gates, both outputs), ragged channel counts, phases, groups, fold and upsampling."""
    import kantts._hip as hip

    g = torch.Generator().manual_seed(tile + 1)
    # (B, Tsrc, Tdst, inner, Cin, Cout, groups, K, in_mul, in_add, in_kstep, in_div, phases, up)
    shapes = [
        (2, 50, 50, 1, 64, 64, 1, 3, 1, -1, 1, 1, 1, 1),
        (3, 37, 37, 1, 80, 136, 1, 7, 1, -6, 1, 1, 1, 1),
        (3, 61, 21, 5, 32, 128, 1, 5, 3, -2, 1, 1, 1, 1),
        (3, 21, 61, 5, 128, 32, 1, 5, 1, 2, -1, 3, 3, 1),
        (2, 200, 400, 1, 128, 128, 4, 41, 1, 20, -1, 2, 2, 1),
        (2, 30, 240, 1, 64, 32, 1, 7, 1, -6, 1, 1, 1, 8),
        (4, 10, 10, 11, 64, 72, 1, 5, 1, -2, 1, 1, 1, 1),
    ]
    for si, (B, Ts, Td, P, Cin, Cout, G, K, mul, add, kstep, div, phases, up) in enumerate(shapes):
        CR, NG = Cin // G, Cout // G
        x = torch.randn(B, Ts, P, Cin, generator=g).to(torch.bfloat16)
        w = (torch.randn(K, Cout, CR, generator=g) / (K * CR) ** 0.5).to(torch.bfloat16)
        bias = torch.randn(Cout, generator=g)
        res = torch.randn(B, Td, P, Cout, generator=g)
        gate = torch.randn(B, Td, P, Cout, generator=g)
        variants = [dict(bias=bias, out_leaky=0.1), dict(res=res, out_gate=gate, out_gate_slope=0.3),
                    dict(bias=bias, out_gate=gate.to(torch.bfloat16), out_gate_slope=0.1, bf_leaky=0.2)]
        kw = variants[si % 3]
        outs = []
        for dev in ("cuda", "cpu"):
            o32 = torch.full((B, Td, P, Cout), float("nan"), device=dev)
            obf = torch.zeros((B, Td, P, Cout), device=dev, dtype=torch.bfloat16)
            kd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in kw.items()}

            def run():
                assert hip.cconv(x.to(dev), w.to(dev), out=o32, out_bf=obf, B=B, Tsrc=Ts, Tdst=Td, groups=G, CR=CR, NG=NG, K=K,
                                 in_mul=mul, in_add=add, in_kstep=kstep, in_div=div, phases=phases, inner=P, up=up,
                                 tile=tile if dev == "cuda" else 0, **kd)

            if dev == "cpu":
                with emulation():
                    run()
            else:
                run()
                torch.cuda.synchronize()
            outs.append((o32.cpu(), obf.float().cpu()))
        (a32, abf), (c32, cbf) = outs
        assert not torch.isnan(a32).any(), (tile, si)
        assert float((a32 - c32).abs().max()) <= 2e-5 * max(1.0, float(c32.abs().max())), (tile, si)
        # bf16 image: one ulp where the fp32 values straddle a rounding boundary
        assert float((abf - cbf).abs().max()) <= 1e-2 * max(1.0, float(cbf.abs().max())), (tile, si)
        assert float(((abf - cbf).abs() > 1e-6).float().mean()) < 0.05, (tile, si)


@pytest.mark.gpu
@pytest.mark.parametrize("slices", [0, 1, 3])
def test_cconv_wgrad_gpu_vs_emulated(slices):
    import kantts._hip as hip

    g = torch.Generator().manual_seed(slices + 11)
    # (B, Tsrc, Tdst, inner, Cin, Cout, groups, K, stride, dil, pad, up)
    shapes = [
        (2, 70, 70, 1, 64, 64, 1, 3, 1, 1, 1, 1),
        (3, 200, 200, 1, 128, 128, 1, 7, 1, 3, 18, 1),
        (3, 61, 21, 5, 64, 256, 1, 5, 3, 1, 2, 1),
        (2, 90, 90, 1, 72, 136, 1, 3, 1, 1, 1, 1),
        (2, 120, 60, 1, 128, 256, 2, 9, 2, 1, 4, 1),
        (2, 50, 200, 1, 64, 64, 1, 7, 1, 1, 6, 4),
        (1, 9, 9, 1, 8, 8, 1, 1, 1, 1, 0, 1),
    ]
    for si, (B, Ts, Td, P, Cin, Cout, G, K, stride, dil, pad, up) in enumerate(shapes):
        CR, NG = Cin // G, Cout // G
        x = torch.randn(B, Ts, P, Cin, generator=g).to(torch.bfloat16)
        dy = torch.randn(B, Td, P, Cout, generator=g).to(torch.bfloat16)
        base = torch.randn(K, Cout, CR, generator=g)  # the call accumulates
        outs = []
        for dev in ("cuda", "cpu"):
            dw, db = base.clone().to(dev), torch.ones(Cout, device=dev)

            def run():
                assert hip.cconv_wgrad(x.to(dev), dy.to(dev), dw, db, B=B, Tsrc=Ts, Tdst=Td, groups=G, CR=CR, NG=NG, K=K,
                                       stride=stride, dil=dil, pad=pad, inner=P, up=up, slices=slices if dev == "cuda" else 0)

            if dev == "cpu":
                with emulation():
                    run()
            else:
                run()
                torch.cuda.synchronize()
            outs.append((dw.cpu(), db.cpu()))
        (adw, adb), (cdw, cdb) = outs
        scale = (B * Td * P) ** 0.5
        assert float((adw - cdw).abs().max()) <= 2e-5 * scale, (slices, si)
        assert float((adb - cdb).abs().max()) <= 2e-5 * scale, (slices, si)


@pytest.mark.gpu
def test_cconv_at_v1_batch32_sizes_vs_fp32_operand_kernels():
    """HiFi-GAN V1 layer shapes at batch 32 x 8192 (BASELINE config 3): the bf16 path against the fp32-operand kernels of
    the same wrapper (fp32 mode) on the same device.  Differences = bf16 operand rounding: outputs <= 2e-2 of the output
    scale, gradients <= 1.5e-2 relative L2."""
    import kantts._hip as hip
    from kantts._hip import ops

    layers = [
        ("mpd 1024->1024 k5 p11", dict(x=(64, 10, 11, 1024), w=(1024, 1024, 5), kw=dict(pad=2, inner=11, in_leaky=0.1))),
        ("mpd 128->512 s3 p2", dict(x=(32, 456, 2, 128), w=(512, 128, 5), kw=dict(stride=3, pad=2, inner=2))),
        ("gen res c128 k11 d5", dict(x=(32, 2048, 128), w=(128, 128, 11), kw=dict(dilation=5, pad=50, in_leaky=0.1))),
        ("gen res c32 k7", dict(x=(32, 8192, 32), w=(32, 32, 7), kw=dict(pad=6, in_leaky=0.1))),
        ("rep up8 256->128", dict(x=(32, 256, 256), w=(128, 256, 7), kw=dict(pad=6, up=8, in_leaky=0.1))),
        ("msd g16 1024 k41 s4", dict(x=(32, 520, 1024), w=(1024, 64, 41), kw=dict(stride=4, pad=20, groups=16))),
        ("msd g16 128->256 k41 s2 (packed)", dict(x=(32, 1024, 128), w=(256, 8, 41), kw=dict(stride=2, pad=20, groups=16))),
    ]
    for name, L in layers:
        g = torch.Generator().manual_seed(len(name))
        x0 = torch.randn(L["x"], generator=g).cuda()
        Cin_g, K = L["w"][1], L["w"][2]
        w0 = (torch.randn(L["w"], generator=g) / (Cin_g * K) ** 0.5).cuda()
        b0 = torch.randn(L["w"][0], generator=g).cuda()
        res = {}
        for prec in ("fp32", "bf16"):
            hip.set_precision(prec)
            try:
                x, w, b = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
                y = ops.conv_cl(x, w, b, **L["kw"])
                cot = torch.randn(y.shape, generator=torch.Generator().manual_seed(3)).cuda()
                gr = torch.autograd.grad((y * cot).sum(), (x, w, b))
                res[prec] = (y.detach(), gr)
            finally:
                hip.set_precision("fp32")
        (yf, gf), (yb, gb) = res["fp32"], res["bf16"]
        assert float((yf - yb).abs().max()) <= 2e-2 * max(1.0, float(yf.abs().max())), name
        assert rel_l2(yb, yf) < 6e-3, name
        for a, c, what in zip(gb, gf, ("dx", "dw", "db")):
            assert rel_l2(a, c) < 1.5e-2, (name, what)


# ------------------------------------------------------------------------------------------------- image hand-over
def _block_and_stack_run(ops, device, use_images):
    """A residual block (3 x [LReLU -> conv -> LReLU -> conv -> +x]) and a period-discriminator style stack in bf16 mode."""
    from kantts.models.hifigan.hifigan import PeriodDiscriminator
    from kantts.models.hifigan.layers import ResidualBlock

    real_get, real_ok = ops.get_image, ops.res_stack_ok
    ops.res_stack_ok = lambda x, K: False  # this test is about the per-convolution chain (the fused node has its own)
    if not use_images:
        ops.get_image = lambda t, slope: None
    casts = {"n": 0}
    real_cast = ops.act_cast_bf16

    def counting(*a, **k):
        casts["n"] += 1
        return real_cast(*a, **k)

    ops.act_cast_bf16 = counting
    try:
        torch.manual_seed(3)
        blk = ResidualBlock(32, kernel_size=3, dilation=(1, 3, 5), causal=True).to(device)
        D = PeriodDiscriminator(period=3, channels=8).to(device)
        g = torch.Generator().manual_seed(9)
        x = torch.randn(2, 50, 32, generator=g).to(device).requires_grad_(True)
        wav = torch.randn(2, 1, 300, generator=g).to(device).requires_grad_(True)
        y = blk.forward_cl(ops.act_image(x, blk.slope))
        out, fmap = D(wav)
        loss = (y * y).sum() + out.pow(2).sum() + sum(f.abs().sum() for f in fmap)
        loss.backward()
        grads = [x.grad, wav.grad] + [p.grad for p in list(blk.parameters()) + list(D.parameters())]
        return y.detach().cpu(), [t.detach().cpu() for t in grads], casts["n"]
    finally:
        ops.get_image, ops.act_cast_bf16, ops.res_stack_ok = real_get, real_cast, real_ok


def _check_image_handover(ops, device, exact):
    y1, g1, n1 = _block_and_stack_run(ops, device, True)
    y0, g0, n0 = _block_and_stack_run(ops, device, False)
    # the epilogue's image is the rounding of the very value the separate pass would round: the forward pass is
    # bit-identical; gradients too, up to the summation order of the kernels that accumulate with atomics on the device
    # (the one-input-channel first layer, weight-norm reductions)
    assert torch.equal(y1, y0)
    for a, b in zip(g1, g0):
        if exact:
            assert torch.equal(a, b)
        else:
            assert rel_l2(a, b) < 1e-5
    assert n1 < n0 - 5, (n1, n0)  # the forward casts of the chained convolutions are gone


def test_image_handover_is_bit_identical_emulated(emulated_cabi, bf16_all_sizes):
    _check_image_handover(bf16_all_sizes, "cpu", True)


@pytest.mark.gpu
def test_image_handover_changes_nothing_gpu(bf16_all_sizes):
    _check_image_handover(bf16_all_sizes, "cuda", False)


# ------------------------------------------------------------------------------------------------- fused residual stack
def _res_block_run(ops, device, fused, C=32, K=3, T=80):
    from kantts.models.hifigan.layers import ResidualBlock

    real_ok = ops.res_stack_ok
    if not fused:
        ops.res_stack_ok = lambda x, K: False
    try:
        torch.manual_seed(11)
        blk = ResidualBlock(C, kernel_size=K, dilation=(1, 3, 5), causal=True).to(device)
        g = torch.Generator().manual_seed(2)
        x = torch.randn(2, T, C, generator=g).to(device).requires_grad_(True)
        y = blk.forward_cl(x)
        cot = torch.randn(y.shape, generator=g).to(device)
        (y * cot).sum().backward()
        return y.detach().cpu(), [x.grad.detach().cpu()] + [p.grad.detach().cpu() for p in blk.parameters()]
    finally:
        ops.res_stack_ok = real_ok


def _check_res_stack(ops, device):
    """The one-node residual stack (ops._ResStackBF16) against the chain of per-convolution nodes: the same bf16 images,
    the same roundings of the gradients, the same kernels -- so the same numbers, not merely close ones."""
    for C, K, T in ((32, 3, 80), (64, 7, 50), (40, 11, 300)):
        y1, g1 = _res_block_run(ops, device, True, C, K, T)
        y0, g0 = _res_block_run(ops, device, False, C, K, T)
        assert torch.equal(y1, y0), (C, K)
        for a, b in zip(g1, g0):
            assert rel_l2(a, b) < 1e-5, (C, K)


def test_fused_residual_stack_equals_the_chain_emulated(emulated_cabi, bf16_all_sizes, monkeypatch):
    calls = _count_calls(monkeypatch, bf16_all_sizes)
    _check_res_stack(bf16_all_sizes, "cpu")
    assert calls["cconv"] > 0


@pytest.mark.gpu
def test_fused_residual_stack_equals_the_chain_gpu(bf16_all_sizes):
    _check_res_stack(bf16_all_sizes, "cuda")
