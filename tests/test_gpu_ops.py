"""GPU parity of every C-ABI entry point: the same host wrapper is run once with device tensors
through libkantts_hip.so and once with host tensors through the emulated ABI (oracle/cabi_numpy.py).
fp32 tolerances are written at each assert; integer outputs must be bit-exact."""
import pytest
import torch

from util import assert_close, emulation, rel_l2, run_both

pytestmark = pytest.mark.gpu


def _hip():
    import kantts._hip as hip

    return hip


def _rand(*shape, seed=0, scale=1.0, grad=False):
    g = torch.Generator().manual_seed(seed)
    t = torch.randn(*shape, generator=g) * scale
    return t.requires_grad_(grad)


# ------------------------------------------------------------------------------------------- GEMM
def test_mfma_fragment_maps_identity_and_asymmetric():
    """A = I against an asymmetric B catches row/col swaps in the MFMA C-write (guide section 3)."""
    hip = _hip()
    from kantts._hip import gemm, make_seg

    for prec in (hip.PREC_FP32, hip.PREC_BF16, hip.PREC_REF):
        n = 64
        a = torch.eye(n, device="cuda")
        b = (torch.arange(n * n, device="cuda", dtype=torch.float32).view(n, n) % 13) + \
            torch.arange(n, device="cuda", dtype=torch.float32)[:, None] * 0.5
        c = torch.empty(n, n, device="cuda")
        gemm([make_seg(a, n, 1, b, n, 1, n)], n, n, c, n, 1, precision=prec)  # C = A @ B^T = B^T
        torch.cuda.synchronize()
        assert torch.equal(c, b.t().contiguous()), "precision %d" % prec


@pytest.mark.parametrize("prec,tol", [("fp32", 2e-5), ("ref", 2e-5), ("bf16", 3e-2)])
@pytest.mark.parametrize("M,N,K", [(70, 50, 45), (256, 384, 128), (33, 1, 256), (130, 240, 80)])
def test_linear_fwd_bwd(prec, tol, M, N, K):
    hip = _hip()
    from kantts._hip import ops

    hip.set_precision(prec)
    try:
        x, w, b = _rand(M, K, seed=1, grad=True), _rand(N, K, seed=2, scale=0.2, grad=True), _rand(N, seed=3, grad=True)
        res = _rand(M, N, seed=4, grad=True)
        mask = (torch.arange(M) % 7 == 3)

        def f(x, w, b, res, mask):
            return ops.linear(x, w, b, res=res, rowmask=mask, relu=False, alpha=0.5)

        go, gg, co, cg = run_both(f, x, w, b, res, mask)
        scale = float(co[0].abs().max())
        assert_close(go[0], co[0], tol * max(1.0, scale), what="y")
        for a, c, nm in zip(gg, cg, "x w b res".split()):
            assert rel_l2(a, c) < (5e-2 if prec == "bf16" else 1e-4), nm

        def f2(x, w, b):
            return ops.linear(x, w, b, relu=True)

        go, gg, co, cg = run_both(f2, x, w, b)
        assert_close(go[0], co[0], tol * max(1.0, scale), what="relu y")
        for a, c, nm in zip(gg, cg, "x w b".split()):
            assert rel_l2(a, c) < (5e-2 if prec == "bf16" else 1e-4), "relu " + nm
    finally:
        hip.set_precision("fp32")


def test_linear_concat_sum_conv_and_dropout_exact():
    """Multi-segment (concat / sum) and conv-tap addressing; dropout masks are regenerated from the
    counter RNG, so the GPU and the emulated ABI must agree element-for-element."""
    from kantts._hip import ops

    B, T = 3, 17
    x1, x2 = _rand(B, T, 40, seed=1, grad=True), _rand(B, T, 24, seed=2, grad=True)
    w = _rand(48, 64, seed=3, scale=0.2, grad=True)
    b = _rand(48, seed=4, grad=True)
    go, gg, co, cg = run_both(lambda a, c, w, b: ops.linear([a, c], w, b, mode="concat", alpha=2.0), x1, x2, w, b)
    assert_close(go[0], co[0], 5e-5, what="concat")
    for a, c in zip(gg, cg):
        assert rel_l2(a, c) < 1e-4
    w1, w2 = _rand(32, 40, seed=5, scale=0.2, grad=True), _rand(32, 24, seed=6, scale=0.2, grad=True)
    b1, b2 = _rand(32, seed=7, grad=True), _rand(32, seed=8, grad=True)
    go, gg, co, cg = run_both(lambda a, c, w1, w2, b1, b2: ops.linear([a, c], [w1, w2], b1, bias2=b2, mode="sum"),
                              x1, x2, w1, w2, b1, b2)
    assert_close(go[0], co[0], 5e-5, what="sum")
    for a, c in zip(gg, cg):
        assert rel_l2(a, c) < 1e-4
    wc = _rand(72, 40, 3, seed=9, scale=0.2, grad=True)
    bc = _rand(72, seed=10, grad=True)
    mask = torch.arange(T)[None, :] >= torch.tensor([17, 9, 13])[:, None]
    go, gg, co, cg = run_both(lambda a, w, b, m: ops.linear(a, w, b, mode="conv", pad=1, relu=True, rowmask=m),
                              x1, wc, bc, mask)
    assert_close(go[0], co[0], 5e-5, what="conv3")
    for a, c, nm in zip(gg, cg, ("x", "w", "b")):
        assert rel_l2(a, c) < 1e-4, "conv3 " + nm
    w9 = _rand(32, 1, 9, seed=11, grad=True)
    xs = _rand(B, T, 1, seed=12, grad=True)
    go, gg, co, cg = run_both(lambda a, w, b, r: ops.linear(a, w, b, mode="conv", pad=4, res=r), xs, w9,
                              _rand(32, seed=13, grad=True), _rand(B, T, 32, seed=14, grad=True))
    assert_close(go[0], co[0], 5e-5, what="conv9")
    for a, c in zip(gg, cg):
        assert rel_l2(a, c) < 1e-4
    # dropout: identical masks on both sides (same seed stream) -> near bit-equal results
    import kantts._hip.ops as O_

    for relu in (False, True):
        O_._seed_counter = __import__("itertools").count(1000)
        torch.manual_seed(5)
        a = ops.linear(x1.detach().cuda(), w.detach()[:, :40].contiguous().cuda(), b.detach().cuda(), relu=relu,
                       drop_p=0.25).cpu()
        O_._seed_counter = __import__("itertools").count(1000)
        torch.manual_seed(5)
        with emulation():
            c = ops.linear(x1.detach(), w.detach()[:, :40].contiguous(), b.detach(), relu=relu, drop_p=0.25)
        assert_close(a, c, 5e-5, what="dropout relu=%s" % relu)
        frac = float((a == 0).float().mean())
        assert (0.15 < frac < 0.35) if not relu else (0.5 < frac < 0.75)


# ------------------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("M,C", [(37, 128), (2048, 512), (5, 80)])
def test_layernorm(M, C):
    from kantts._hip import ops

    x, g, b = _rand(M, C, seed=1, grad=True), _rand(C, seed=2, grad=True), _rand(C, seed=3, grad=True)
    go, gg, co, cg = run_both(lambda x, g, b: ops.layer_norm(x, g, b, 1e-6), x, g, b)
    assert_close(go[0], co[0], 2e-5, what="ln y")
    for a, c, nm in zip(gg, cg, ("dx", "dgamma", "dbeta")):
        assert rel_l2(a, c) < 1e-4, nm


# ------------------------------------------------------------------------------------------- attention
@pytest.mark.parametrize("drop", [0.0, 0.1])
def test_self_attention(drop):
    import itertools

    import kantts._hip.ops as O_
    from kantts._hip import ops

    B, L, H = 3, 41, 8
    qkv = _rand(B, L, 3 * H * 16, seed=1, grad=True)
    lens = torch.tensor([41, 17, 30], dtype=torch.int32)

    def f(qkv, lens):
        O_._seed_counter = itertools.count(77)
        return ops.self_attention(qkv, lens, H, drop_p=drop, want_probs=True)

    go, gg, co, cg = run_both(f, qkv, lens)
    assert_close(go[0], co[0], 2e-5, what="ctx")
    assert_close(go[1], co[1], 2e-6, what="probs")
    assert rel_l2(gg[0], cg[0]) < 1e-4


@pytest.mark.parametrize("bw", [0, 3, 50])
def test_pnca_attention(bw):
    from kantts._hip import ops

    B, L, H = 3, 37, 8
    qkv = _rand(B, L, 3 * H * 16, seed=1, grad=True)
    hkv = _rand(B, L, 2 * H * 16, seed=2, grad=True)
    lens = torch.tensor([37, 12, 25], dtype=torch.int32)
    go, gg, co, cg = run_both(lambda a, b, l: ops.pnca_attention(a, b, l, bw, bw, H, want_probs=True), qkv, hkv, lens)
    for k in range(4):
        assert_close(go[k], co[k], 2e-5, what="pnca out %d" % k)
    assert rel_l2(gg[0], cg[0]) < 1e-4 and rel_l2(gg[1], cg[1]) < 1e-4
    go, gg, co, cg = run_both(lambda a, b: ops.pnca_attention(a, b, None, bw, bw, H), qkv, hkv)
    assert_close(go[0], co[0], 2e-5, what="pnca nolens")


@pytest.mark.parametrize("bw,drop", [(0, 0.0), (3, 0.1), (50, 0.0)])
def test_pnca_attention_one_launch_equals_per_band_launches(bw, drop, monkeypatch):
    """kantts_pnca_attn_fwd / _bwd (both bands, dq and dk/dv passes as roles of ONE launch, D recomputed by the dk/dv
    role) against the per-band launches of the same kernels bodies: identical bits, forward and backward; and against
    the emulated C ABI."""
    import itertools

    import kantts._hip.ops as O_
    from kantts._hip import ops

    B, L, H = 3, 37, 8
    lens = torch.tensor([37, 12, 25], dtype=torch.int32)

    def run(fused):
        if fused:
            monkeypatch.delenv("KANTTS_NO_PNCA_FUSED", raising=False)
        else:
            monkeypatch.setenv("KANTTS_NO_PNCA_FUSED", "1")
        O_._seed_counter = itertools.count(31)
        qkv = _rand(B, L, 3 * H * 16, seed=1, grad=True).cuda().detach().requires_grad_(True)
        hkv = _rand(B, L, 2 * H * 16, seed=2, grad=True).cuda().detach().requires_grad_(True)
        ox, oh, _, _ = ops.pnca_attention(qkv, hkv, lens.cuda(), bw, bw, H, drop_p=drop)
        w = torch.randn(ox.shape, generator=torch.Generator().manual_seed(3)).cuda()
        ((ox * w).sum() + (oh * w.flip(0)).sum()).backward()
        return ox.detach(), oh.detach(), qkv.grad, hkv.grad

    a, b = run(True), run(False)
    for x, y, nm in zip(a, b, ("ctx x", "ctx h", "dqkv", "dhkv")):
        if nm == "dqkv":  # the one-launch form sums the two bands' query gradients in one accumulator chain
            assert torch.equal(x[..., H * 16:], y[..., H * 16:]) and rel_l2(x.cpu(), y.cpu()) < 1e-6, nm
        else:
            assert torch.equal(x, y), nm

    def f(qkv, hkv, lens):
        O_._seed_counter = itertools.count(31)
        return ops.pnca_attention(qkv, hkv, lens, bw, bw, H, drop_p=drop)[:2]

    monkeypatch.delenv("KANTTS_NO_PNCA_FUSED", raising=False)
    go, gg, co, cg = run_both(f, _rand(B, L, 3 * H * 16, seed=1, grad=True), _rand(B, L, 2 * H * 16, seed=2, grad=True), lens)
    assert_close(go[0], co[0], 2e-5, what="x ctx")
    assert_close(go[1], co[1], 2e-5, what="h ctx")
    assert rel_l2(gg[0], cg[0]) < 1e-4 and rel_l2(gg[1], cg[1]) < 1e-4


def test_attention_long_sequence_fallback_kernels():
    """L = 600 does not fit the 64 KB LDS staging -> direct-from-global kernels."""
    from kantts._hip import ops

    B, L, H = 1, 600, 8
    qkv = _rand(B, L, 3 * H * 16, seed=1, grad=True)
    hkv = _rand(B, L, 2 * H * 16, seed=2, grad=True)
    lens = torch.tensor([500], dtype=torch.int32)
    go, gg, co, cg = run_both(lambda a, b, l: ops.pnca_attention(a, b, l, 4, 4, H), qkv, hkv, lens)
    assert_close(go[0], co[0], 2e-5, what="x ctx")
    assert_close(go[1], co[1], 2e-5, what="h ctx")
    assert rel_l2(gg[0], cg[0]) < 1e-4 and rel_l2(gg[1], cg[1]) < 1e-4


@pytest.mark.parametrize("L", [300, 400])
def test_attention_longer_than_a_workgroup_gpu(L):
    """Device twin of tests/test_ops_sweep.py::test_attention_longer_than_a_workgroup: 257 ... 390 rows (a thread owns a second
    query / key row, K / V still staged in LDS) and past the LDS limit (direct-from-global kernels); self-attention and both
    PNCA bands, outputs and gradients against the oracle."""
    import torch_oracle as O
    from kantts._hip import ops

    g = torch.Generator().manual_seed(L)
    B, H = 2, 2
    D = H * 16
    lens = torch.tensor([L, L - 37])
    l32 = lens.to(torch.int32).cuda()
    pad = O.pad_mask(lens, L)
    valid = (~pad)[..., None]
    qkv = torch.randn(B, L, 3 * D, generator=g).requires_grad_(True)
    hkv = torch.randn(B, L, 2 * D, generator=g).requires_grad_(True)
    q, k, v = (O._split_heads(t, H) for t in qkv.chunk(3, -1))
    ro, _ = O._attend(q, k, v, pad[:, None, :].expand(-1, L, -1).repeat(H, 1, 1))
    ro = O._merge_heads(ro, H)
    dq = qkv.detach().cuda().requires_grad_(True)
    dh = hkv.detach().cuda().requires_grad_(True)
    o, _ = ops.self_attention(dq, l32, H)
    cot = torch.randn(B, L, D, generator=g) * valid
    c2 = torch.randn(B, L, D, generator=g) * valid
    assert float(((o.cpu() - ro) * valid).detach().abs().max()) < 2e-5
    (ga,) = torch.autograd.grad(o, dq, cot.cuda())
    (gr,) = torch.autograd.grad(ro, qkv, cot)
    assert rel_l2(ga.cpu(), gr) < 1e-4
    bwx, bwh = 9, 5
    ox, oh, _, _ = ops.pnca_attention(dq, dh, l32, bwx, bwh, H)
    xm, hm = O.pnca_masks(L, bwx, bwh, pad, qkv.device)
    hk, hv = (O._split_heads(t, H) for t in hkv.chunk(2, -1))
    rx, _ = O._attend(q, k, v, xm.expand(B, -1, -1).repeat(H, 1, 1))
    rh, _ = O._attend(q, hk, hv, hm.expand(B, -1, -1).repeat(H, 1, 1))
    rx, rh = O._merge_heads(rx, H), O._merge_heads(rh, H)
    assert float(((ox.cpu() - rx) * valid).detach().abs().max()) < 2e-5
    assert float(((oh.cpu() - rh) * valid).detach().abs().max()) < 2e-5
    got = torch.autograd.grad((ox * cot.cuda()).sum() + (oh * c2.cuda()).sum(), [dq, dh])
    exp = torch.autograd.grad((rx * cot).sum() + (rh * c2).sum(), [qkv, hkv])
    for a, b, nm in zip(got, exp, ("dqkv", "dhkv")):
        assert rel_l2(a.cpu(), b) < 1e-4, nm


# ------------------------------------------------------------------------------------------- LSTM
def test_lstm_uni_bi_and_concat():
    from kantts._hip import ops

    B, T, I, H = 4, 23, 48, 128

    def params(seed, nin):
        return [_rand(4 * H, nin, seed=seed, scale=0.1, grad=True), _rand(4 * H, H, seed=seed + 1, scale=0.1, grad=True),
                _rand(4 * H, seed=seed + 2, scale=0.1, grad=True), _rand(4 * H, seed=seed + 3, scale=0.1, grad=True)]

    x = _rand(B, T, I, seed=1, grad=True)
    p = params(10, I)
    go, gg, co, cg = run_both(lambda x, *p: ops.lstm(x, list(p)), x, *p)
    assert_close(go[0], co[0], 2e-5, what="lstm uni")
    for a, c, nm in zip(gg, cg, ("x", "w_ih", "w_hh", "b_ih", "b_hh")):
        assert rel_l2(a, c) < 2e-4, "uni " + nm
    lens = torch.tensor([23, 7, 15, 1], dtype=torch.int32)
    p2 = params(10, I) + params(20, I)
    go, gg, co, cg = run_both(lambda x, l, *p: ops.lstm(x, list(p), l), x, lens, *p2)
    assert_close(go[0], co[0], 2e-5, what="lstm bi packed")
    names = ["x"] + ["%s_%s" % (n, d) for d in ("f", "r") for n in ("w_ih", "w_hh", "b_ih", "b_hh")]
    for a, c, nm in zip(gg, cg, names):
        assert rel_l2(a, c) < 2e-4, "bi " + nm
    xa, xb = _rand(B, T, 32, seed=3, grad=True), _rand(B, T, 16, seed=4, grad=True)
    go, gg, co, cg = run_both(lambda a, b, *p: ops.lstm([a, b], list(p)), xa, xb, *p)
    assert_close(go[0], co[0], 2e-5, what="lstm concat")
    for a, c in zip(gg, cg):
        assert rel_l2(a, c) < 2e-4


def test_lstm_bf16_recurrence_close_to_fp32():
    """Throughput mode: the recurrent products run on packed bf16 pairs (v_dot2c_f32_bf16).  Outputs / gradients
    must stay within bf16 rounding of the fp32 oracle (emulated ABI) -- a wrong pairing or gate order would be O(1)."""
    import kantts._hip as hip
    from kantts._hip import ops

    B, T, I, H = 3, 40, 32, 128
    x = _rand(B, T, I, seed=1, grad=True)
    p = [_rand(4 * H, I, seed=2, scale=0.1, grad=True), _rand(4 * H, H, seed=3, scale=0.1, grad=True),
         _rand(4 * H, seed=4, scale=0.1, grad=True), _rand(4 * H, seed=5, scale=0.1, grad=True)]
    p2 = p + [_rand(4 * H, I, seed=6, scale=0.1, grad=True), _rand(4 * H, H, seed=7, scale=0.1, grad=True),
              _rand(4 * H, seed=8, scale=0.1, grad=True), _rand(4 * H, seed=9, scale=0.1, grad=True)]
    lens = torch.tensor([40, 11, 29], dtype=torch.int32)
    hip.set_precision("bf16")
    try:
        go, gg, co, cg = run_both(lambda x, l, *p: ops.lstm(x, list(p), l), x, lens, *p2)
    finally:
        hip.set_precision("fp32")
    assert rel_l2(go[0], co[0]) < 2e-2
    for a, c in zip(gg, cg):
        assert rel_l2(a, c) < 5e-2


# ------------------------------------------------------------------------------------------- sequence ops
def test_embedding_and_length_regulator_bit_exact():
    from kantts._hip import ops

    B, T, D = 3, 11, 64
    g = torch.Generator().manual_seed(0)
    ids = torch.stack([torch.randint(0, n, (B, T), generator=g) for n in (20, 5, 4, 6)], -1)
    tabs = [_rand(n, D, seed=k, grad=True) for k, n in enumerate((20, 5, 4, 6))]
    pos = _rand(30, D, seed=9)
    go, gg, co, cg = run_both(lambda i, p, *t: ops.embed_sum(i, list(t), pos=p, scale=3.0, want_scaled=True), ids, pos,
                              *tabs)
    assert_close(go[0], co[0], 1e-5, what="embed")
    assert_close(go[1], co[1], 1e-5, what="embed scaled")
    for a, c in zip(gg, cg):
        assert_close(a, c, 1e-4, what="embed grad")
    # bench-sized call (32 x 64 tokens, 512 columns): tiny tables through the LDS images, a 500-row table through the
    # scatter form, an index tensor with every token on ONE row (maximal collisions)
    ids2 = torch.stack([torch.randint(0, n, (32, 64), generator=g) for n in (150, 7, 500, 5)], -1)
    ids2[:, :, 3] = 2
    tabs2 = [_rand(n, 512, seed=20 + k, grad=True) for k, n in enumerate((150, 7, 500, 5))]
    go, gg, co, cg = run_both(lambda i, *t: ops.embed_sum(i, list(t), scale=2.0)[0], ids2, *tabs2)
    assert_close(go[0], co[0], 1e-5, what="embed (bench size)")
    for a, c in zip(gg, cg):
        assert rel_l2(a, c) < 1e-5, "embed grad (bench size)"
    dur = torch.randint(0, 6, (B, T), generator=g)
    Tp = 60
    outs = {}
    for dev in ("cuda", "cpu"):
        import contextlib

        with (emulation() if dev == "cpu" else contextlib.nullcontext()):
            outs[dev] = [t.cpu() for t in ops.lr_index(dur.to(dev), Tp)]
            outs[dev] += [t.cpu() for t in ops.lr_index((dur.float() * 1.3).to(dev), Tp)]
    for a, c in zip(outs["cuda"], outs["cpu"]):
        assert torch.equal(a, c)  # index tensors: bit-exact
    x = _rand(B, T, 32, seed=5, grad=True)
    valid = torch.tensor([60, 20, 33])

    def f(x, dur, valid):
        idx, pos_, cs, lens = ops.lr_index(dur, Tp)
        return ops.lr_gather(x, idx, cs, valid)

    go, gg, co, cg = run_both(f, x, dur, valid)
    assert torch.equal(go[0], co[0])  # copies: bit-exact
    assert_close(gg[0], cg[0], 1e-5, what="lr bwd")


@pytest.mark.parametrize("C,K,lp,T,B", [(128, 41, 20, 70, 3), (256, 41, 37, 70, 3), (80, 41, 40, 300, 3), (64, 41, 20, 520, 8)])
def test_fsmn_memory(C, K, lp, T, B):
    """The third case: several 128-frame chunks per sequence and a channel count that is no multiple of the filter
    gradient's 64-channel groups; the fourth: 40 partial rows, enough for the reduce kernel's four-loads-per-trip loop."""
    from kantts._hip import ops

    x, w = _rand(B, T, C, seed=1, grad=True), _rand(C, 1, K, seed=2, scale=0.2, grad=True)
    res = _rand(B, T, C, seed=3, grad=True)
    lens = torch.tensor(([T, 33, T - 19] + [T - 7 * i for i in range(B)])[:B])
    go, gg, co, cg = run_both(lambda x, w, r, l: ops.fsmn_memory(x, w, l, lp, res=r), x, w, res, lens)
    assert_close(go[0], co[0], 2e-5, what="fsmn y")
    for a, c, nm in zip(gg, cg, ("dx", "dw", "dres")):
        assert rel_l2(a, c) < 1e-4, nm


def test_dropout2_add_kernel():
    """kantts_dropout2_add on the device against the emulated ABI: identical masks (same counter RNG), so outputs and
    gradients agree to rounding; with and without the residual; p2 = 0."""
    import itertools

    from kantts._hip import ops

    x, r = _rand(32, 612, 256, seed=4, grad=True), _rand(32, 612, 256, seed=5, grad=True)

    def seeded(fn):
        def f(*a):
            ops._seed_counter = itertools.count(300)
            torch.manual_seed(7)
            return fn(*a)
        return f

    for p1, p2, with_res in ((0.1, 0.1, True), (0.25, 0.0, False), (0.0, 0.5, True)):
        if with_res:
            go, gg, co, cg = run_both(seeded(lambda a, b: ops.dropout2_add(a, p1, p2, b)), x, r)
        else:
            go, gg, co, cg = run_both(seeded(lambda a: ops.dropout2_add(a, p1, p2)), x)
        assert_close(go[0], co[0], 1e-6, what="dropout2_add y")
        keep = 1.0 - (go[0] - (r.detach() if with_res else 0) == 0).float().mean().item()
        assert abs(keep - (1 - p1) * (1 - p2)) < 0.01
        for a, c in zip(gg, cg):
            assert_close(a, c, 1e-6, what="dropout2_add grad")


def test_gan_criteria_in_one_launch_each_on_device():
    from test_ops_sweep import _gan_criteria_fused_vs_per_term

    _gan_criteria_fused_vs_per_term("cuda")


def test_mean_many_on_device():
    """kantts_mean_many / kantts_scale_to_many: values, the bf16 LeakyReLU image, one gradient buffer per input."""
    import kantts._hip as hip
    from kantts._hip import ops

    prev = hip.get_precision()
    hip.set_precision("bf16")
    try:
        g = torch.Generator().manual_seed(9)
        for n, shape in ((3, (4, 2048, 64)), (2, (1, 8, 4)), (8, (2, 33, 8))):
            xs = [(torch.randn(shape, generator=g)).cuda().requires_grad_(True) for _ in range(n)]
            cot = torch.randn(shape, generator=g).cuda()
            y = ops.mean_many(xs, image_slope=0.1)
            ref = sum(x.detach() for x in xs) / n
            assert_close(y.detach(), ref, 1e-6, what="mean_many")
            img = ops.get_image(y, 0.1)
            if shape[0] * shape[1] * shape[2] % 8 == 0:
                assert img is not None and img.dtype == torch.bfloat16
                assert_close(img.float(), torch.nn.functional.leaky_relu(y.detach(), 0.1), 1e-6, rtol=8e-3, what="image")
            grads = torch.autograd.grad(y, xs, cot)
            for gr in grads:
                assert_close(gr, cot / n, 3e-7, what="mean_many grad")
            assert len({gr.data_ptr() for gr in grads}) == n
    finally:
        hip.set_precision(prev)


def test_fused_sambert_loss_on_device_equals_the_two_criteria():
    """kantts_masked_l1_many / kantts_scale_many on the GPU against the per-term criteria on the GPU: components, total and
    the five gradients, bench-shaped (B=32, T=640, 80 bins) and ragged small cases."""
    import os

    from kantts._hip import ops
    from kantts.train.loss import MelReconLoss, ProsodyReconLoss, sambert_loss_sum

    ops.zero_pool.enable(1 << 16, "cuda")
    mel_c, pro_c = MelReconLoss(), ProsodyReconLoss()
    for (B, T, N, seed) in ((32, 640, 96, 1), (3, 17, 5, 2), (1, 1, 1, 3)):
        g = torch.Generator().manual_seed(seed)
        ol = torch.randint(1, T + 1, (B,), generator=g)
        il = torch.randint(1, N + 1, (B,), generator=g)
        ol[0], il[-1] = T, N
        batch = dict(output_lengths=ol.cuda(), input_lengths=il.cuda(), mel_targets=torch.randn(B, T, 80, generator=g).cuda())
        leaves = [torch.randn(B, T, 80, generator=g), torch.randn(B, T, 80, generator=g), torch.randn(B, N, generator=g),
                  torch.randn(B, N, generator=g), torch.randn(B, N, generator=g)]
        leaves = [t.cuda().requires_grad_(True) for t in leaves]
        res = dict(zip(("dec_outputs", "postnet_outputs", "log_duration_predictions", "pitch_predictions",
                        "energy_predictions"), leaves))
        res.update(duration_targets=torch.randint(0, 40, (B, N), generator=g).cuda(),
                   pitch_targets=torch.randn(B, N, generator=g).cuda(), energy_targets=torch.randn(B, N, generator=g).cuda())
        out = {}
        for fused in (True, False):
            if not fused:
                os.environ["KANTTS_NO_FUSED_LOSS"] = "1"
            try:
                total, comps = sambert_loss_sum(mel_c, pro_c, batch, res)
            finally:
                os.environ.pop("KANTTS_NO_FUSED_LOSS", None)
            ops.zero_pool.reset()  # the trainer clears the gradients (and with them the accumulator pool) before backward()
            out[fused] = (total.detach().clone(), comps, torch.autograd.grad(total * 0.5, leaves))
        assert abs(float(out[True][0]) - float(out[False][0])) < 2e-5 * max(1.0, abs(float(out[False][0])))
        for k in out[False][1]:
            assert abs(float(out[True][1][k]) - float(out[False][1][k])) < 2e-5 * max(1.0, abs(float(out[False][1][k]))), k
        for a, c in zip(out[True][2], out[False][2]):
            assert_close(a, c, 1e-10, rtol=2e-6, what="fused loss grad")


def test_masked_l1_and_optimizer_kernels():
    from kantts._hip import ops

    B, T, C = 4, 50, 80
    p, t = _rand(B, T, C, seed=1, grad=True), _rand(B, T, C, seed=2)
    lens = torch.tensor([50, 10, 33, 1])
    go, gg, co, cg = run_both(lambda p, t, l: ops.masked_l1(p, t, l) * 3.0, p, t, lens)
    assert abs(float(go[0]) - float(co[0])) < 1e-5
    assert_close(gg[0], cg[0], 1e-8, what="l1 grad")
    n = 100003
    vals = {}
    for dev in ("cuda", "cpu"):
        import contextlib

        with (emulation() if dev == "cpu" else contextlib.nullcontext()):
            prm, g = _rand(n, seed=3).to(dev), _rand(n, seed=4).to(dev)
            m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
            s = torch.zeros((), device=dev)
            for step in (1, 2, 3):
                s.zero_()
                ops.sumsq_into(g, s)
                ops.adam_step(prm, g, m, v, 1e-3, 0.9, 0.98, 1e-9, 0.0, step, gnorm_sq=s, max_norm=1.0)
            vals[dev] = [t_.cpu() for t_ in (prm, m, v, s)]
    assert abs(float(vals["cuda"][3]) - float(vals["cpu"][3])) < 1e-3 * float(vals["cpu"][3])
    for a, c, nm in zip(vals["cuda"], vals["cpu"], ("p", "m", "v")):
        assert_close(a, c, 1e-6, what="adam " + nm)
