"""bf16-operand kernels (csrc/gemm_bf16.hip, ln128 in csrc/norm.hip) on the GPU against the emulated C ABI: the same
product wrapper is run once with device tensors and once, under oracle/cabi_numpy, with host tensors.  The emulation
rounds to bf16 exactly like the kernels, so fp32 outputs agree to accumulation-order noise (1e-5) and bf16 outputs to
one bf16 ulp on a few elements (rel-L2 2e-3).  Shapes include the benchmarked ones (M = 6528 decoder tokens,
128 <-> 1024) and edge cases: ragged M, N < one tile, K with a partial 64-deep tile, conv taps at sequence borders."""
import pytest
import torch

from util import rel_l2, run_both

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def bf16_mode():
    import kantts._hip as hip

    hip.set_precision("bf16")
    yield
    hip.set_precision("fp32")


def _seeded(fn):
    """Dropout seeds come from a process-wide counter: restart it at every evaluation so that the GPU run and the
    emulated run of one case draw the same masks."""
    import itertools

    from kantts._hip import ops

    def wrapped(*a):
        ops._seed_counter = itertools.count(1000)
        torch.manual_seed(5)
        return fn(*a)

    return wrapped


def _cmp(go, gg, co, cg, otol=2e-3, gtol=3e-3):
    for a, b in zip(go, co):
        assert a.shape == b.shape and a.dtype == b.dtype
        assert rel_l2(a.float(), b.float()) <= otol, ("out", rel_l2(a.float(), b.float()))
    assert len(gg) == len(cg)
    for k, (a, b) in enumerate(zip(gg, cg)):
        if b is None:
            continue
        assert a.shape == b.shape and a.dtype == b.dtype
        assert rel_l2(a.float(), b.float()) <= gtol, ("grad", k, rel_l2(a.float(), b.float()))


@pytest.mark.parametrize("M,K,N,x_bf16,out_bf16", [
    (6528, 128, 1024, True, True), (6528, 1024, 128, True, False), (6528, 128, 384, True, False),
    (2048, 512, 384, False, False), (1000, 160, 256, False, False), (77, 80, 240, False, True),
    (19584, 256, 512, False, False), (33, 8, 8, False, False),
])
def test_linear_plain(M, K, N, x_bf16, out_bf16):
    from kantts._hip import ops

    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g)
    if x_bf16:
        x = x.to(torch.bfloat16)
    x.requires_grad_(True)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).requires_grad_(True)
    b = torch.randn(N, generator=g).requires_grad_(True)
    _cmp(*run_both(lambda x_, w_, b_: ops.linear(x_, w_, b_, out_bf16=out_bf16), x, w, b))


def test_linear_epilogues_and_modes():
    from kantts._hip import ops

    g = torch.Generator().manual_seed(3)
    B, T = 6, 204
    M = B * T
    rm = torch.zeros(B, T, dtype=torch.bool)
    rm[2, 150:] = True
    rm[5, 10:] = True
    # relu + dropout, bf16 out (Prenet / FFN up-projection)
    x = torch.randn(B, T, 128, generator=g).to(torch.bfloat16).requires_grad_(True)
    w = (torch.randn(1024, 128, generator=g) * 0.1).requires_grad_(True)
    b = torch.randn(1024, generator=g).requires_grad_(True)
    _cmp(*run_both(_seeded(lambda x_, w_, b_, rm_: ops.linear(x_, w_, b_, relu=True, drop_p=0.1, rowmask=rm_,
                                                              out_bf16=True)), x, w, b, rm))
    # two projections summed + dropout + residual + row zeroing (fc_x + fc_h)
    xa = torch.randn(B, T, 128, generator=g, requires_grad=True)
    xb = torch.randn(B, T, 128, generator=g, requires_grad=True)
    wa = (torch.randn(128, 128, generator=g) * 0.1).requires_grad_(True)
    wb = (torch.randn(128, 128, generator=g) * 0.1).requires_grad_(True)
    ba = torch.randn(128, generator=g).requires_grad_(True)
    bb = torch.randn(128, generator=g).requires_grad_(True)
    res = torch.randn(B, T, 128, generator=g, requires_grad=True)
    _cmp(*run_both(_seeded(lambda a, b_, c, d, e, f, r, m: ops.linear([a, b_], [c, d], e, bias2=f, mode="sum", res=r,
                                                                     rowmask=m, drop_p=0.1)),
                   xa, xb, wa, wb, ba, bb, res, rm))
    # concat of a 160-wide fp32 input and a 128-wide bf16 one (dec_in_proj), alpha scaling
    m1 = torch.randn(B, T, 160, generator=g, requires_grad=True)
    m2 = torch.randn(B, T, 128, generator=g).to(torch.bfloat16).requires_grad_(True)
    wc = (torch.randn(128, 288, generator=g) * 0.1).requires_grad_(True)
    bc = torch.randn(128, generator=g).requires_grad_(True)
    _cmp(*run_both(lambda a, b_, c, d, m: ops.linear([a, b_], c, d, mode="concat", rowmask=m, alpha=128 ** 0.5),
                   m1, m2, wc, bc, rm))


@pytest.mark.parametrize("k1", [3, 1])
def test_conv_mode_and_fused_ffn(k1):
    from kantts._hip import ops

    g = torch.Generator().manual_seed(5 + k1)
    B, T, C, Fh = 5, 64, 128, 1024
    lens = torch.tensor([64, 40, 33, 64, 1])
    pad_rows = torch.arange(T)[None, :] >= lens[:, None]
    x = torch.randn(B, T, C, generator=g, requires_grad=True)
    w1 = (torch.randn(Fh, C, k1, generator=g) * 0.05).requires_grad_(True)
    b1 = torch.randn(Fh, generator=g).requires_grad_(True)
    w2 = (torch.randn(C, Fh, 1, generator=g) * 0.03).requires_grad_(True)
    b2 = torch.randn(C, generator=g).requires_grad_(True)
    if k1 == 3:
        _cmp(*run_both(lambda x_, w_, b_: ops.linear(x_, w_, b_, mode="conv", pad=1), x, w1, b1))
    gam = torch.rand(C, generator=g).add(0.5).requires_grad_(True)
    bet = torch.randn(C, generator=g).requires_grad_(True)

    def block(x_, gam_, bet_, w1_, b1_, w2_, b2_, pr):
        h = ops.layer_norm(x_, gam_, bet_, 1e-6, out_bf16=True)
        assert h.dtype == torch.bfloat16
        return ops.ffn(h, w1_, b1_, w2_, b2_, x_, pad_rows=pr, zero_rows=pr, p_inner=0.1, p_out=0.1)

    _cmp(*run_both(_seeded(block), x, gam, bet, w1, b1, w2, b2, pad_rows), otol=2e-3, gtol=5e-3)


@pytest.mark.parametrize("M,T,F,KT", [(6528, 204, 1024, 1), (2048, 64, 1024, 3), (111, 37, 1024, 1), (95, 19, 1024, 3),
                                      (32, 32, 1024, 5), (19584, 612, 1024, 1)])
def test_ffn_pair_kernel_forward_and_backward_forms(M, T, F, KT):
    """kantts_ffn_pair called directly (it must ACCEPT these shapes: the bench sizes, ragged M, taps at sequence borders,
    M not a multiple of 32): forward form (bias, ReLU, both dropouts, both row masks, residual, bf16 hidden written) and, for KT = 1, the
    backward form (fp32 dy with regenerated dropout, gate by the hidden tensor, transposed weights, bf16 and fp32 dh); for
    KT = 3 the backward form sums the three taps in phase 2 (halo rows, sequence borders)."""
    import kantts._hip as hip
    from kantts._hip.ops_bf16 import frag_major

    g = torch.Generator().manual_seed(M + F + KT)
    bf = torch.bfloat16
    x = torch.randn(M, 128, generator=g).to(bf)
    w1m = torch.randn(KT, F, 128, generator=g) * 0.08
    w2m = torch.randn(128, F, generator=g) * 0.03
    b1, b2 = torch.randn(F, generator=g) * 0.3, torch.randn(128, generator=g)
    res = torch.randn(M, 128, generator=g)
    rm = (torch.rand(M, generator=g) < 0.15)
    dy = torch.randn(M, 128, generator=g)

    img = frag_major                               # fragment-major bf16 image built with tensor ops + the cast kernel

    w1, w2 = w1m.reshape(KT * F, 128), w2m
    w2t = w2m.t().contiguous()                                  # (F, 128)

    def fwd(x_, w1_, w2_, b1_, b2_, res_, rm_):
        hid = torch.zeros(M, F, dtype=bf, device=x_.device)
        y = torch.zeros(M, 128, device=x_.device)
        assert hip.ffn_pair(x_, img(w1_), img(w2_), y, M=M, T=T, F=F, KT=KT, pad=(KT - 1) // 2, bias1=b1_, bias2=b2_, relu=True,
                            drop1_p=0.1, drop1_seed=77, drop2_p=0.2, drop2_seed=78, rowmask1=rm_, rowmask2=rm_,
                            t_out=hid, res=res_)
        return y, hid

    go, _, co, _ = run_both(fwd, x, w1, w2, b1, b2, res, rm)
    assert rel_l2(go[0], co[0]) < 1e-4, rel_l2(go[0], co[0])
    assert rel_l2(go[1].float(), co[1].float()) < 2e-3           # bf16 hidden: an ulp on a few elements
    assert (go[1] != co[1]).float().mean() < 0.02
    if KT not in (1, 3) or M % T:
        return
    hid = co[1]
    # W1[tap]^T stacked over taps: rows tap*128 + c
    w1t = w1m.permute(0, 2, 1).reshape(KT * 128, F).contiguous()

    def bwd(dy_, w2t_, w1t_, hid_, fp32_out):
        dz = torch.zeros(M, F, dtype=bf, device=dy_.device)
        dh = torch.zeros(M, 128, dtype=torch.float32 if fp32_out else bf, device=dy_.device)
        assert hip.ffn_pair(dy_, img(w2t_), img(w1t_), dh, M=M, T=T, F=F, alpha1=1.0 / 0.9, xdrop_p=0.2, xdrop_seed=78, gate=hid_,
                            t_out=dz, KT2=KT, s2_first=(KT - 1) // 2, s2_step=-1)
        return dh, dz

    for fp32_out in (False, True):
        go, _, co, _ = run_both(lambda *a: bwd(*a, fp32_out), dy, w2t, w1t, hid)
        assert rel_l2(go[0].float(), co[0].float()) < (1e-4 if fp32_out else 3e-3)
        assert rel_l2(go[1].float(), co[1].float()) < 2e-3


@pytest.mark.parametrize("M", [6528, 100, 16, 5])
def test_layer_norm128(M):
    from kantts._hip import ops

    g = torch.Generator().manual_seed(M)
    x = (torch.randn(M, 128, generator=g) * 3 + 1).requires_grad_(True)
    gam = torch.rand(128, generator=g).add(0.5).requires_grad_(True)
    bet = torch.randn(128, generator=g).requires_grad_(True)
    for ob in (False, True):
        go, gg, co, cg = run_both(lambda a, b, c: ops.layer_norm(a, b, c, 1e-6, out_bf16=ob), x, gam, bet)
        assert go[0].dtype == (torch.bfloat16 if ob else torch.float32)
        _cmp(go, gg, co, cg, otol=2e-3 if ob else 2e-6, gtol=1e-5)


def test_arena_shadow_matches_master_and_follows_updates():
    """ParamArena(bf16_shadow=True): every shadow view equals the bf16 cast of its parameter (tap-major for Conv1d
    weights with KT > 1) and follows an optimizer step through the forward pre-hook."""
    import kantts._hip as hip
    import torch_oracle as O
    from kantts.models import model_builder

    cfg = O.sambert_config(tiny=True)
    config = {"model_type": "sambert", "Model": {"KanTtsSAMBERT": {
        "params": cfg, "optimizer": {"type": "Adam", "params": {"lr": 1e-2}},
        "scheduler": {"type": "NoamLR", "params": {"warmup_steps": 10}}}}}
    torch.manual_seed(0)
    model, opt, _ = model_builder(config, device="cuda")
    net, o = model["KanTtsSAMBERT"], opt["KanTtsSAMBERT"]

    def check_all():
        n_tap = 0
        for name, p in net.named_parameters():
            if not p.requires_grad:
                continue
            assert torch.equal(p._kantts_bf16, p.detach().to(torch.bfloat16)), name
            if p.dim() == 3 and p.shape[2] > 1:
                assert torch.equal(p._kantts_bf16_tap, p.detach().permute(2, 0, 1).to(torch.bfloat16)), name
                n_tap += 1
        return n_tap

    assert check_all() >= 2
    batch = {k: v.cuda() for k, v in O.synthetic_sambert_batch(B=2, T_in=10, min_len=5, dur_hi=5).items()}
    from test_gpu_sambert import _losses

    hip.set_precision("bf16")
    o.zero_grad()
    _losses(net(**batch), batch).backward()
    o.step()
    net(**batch)  # the pre-hook refreshes the shadow from the updated master
    check_all()
    # a SUB-MODULE called on its own right after an optimizer step (no forward of the whole module in between) must not
    # read the previous step's images: the arena is marked stale by step() and the first image handed out rebuilds it
    o.zero_grad()
    _losses(net(**batch), batch).backward()
    o.step()
    assert o.arena.shadow_stale
    from kantts._hip import ops_bf16

    w = net.text_encoder.ling_enc.fft[0].slf_attn.w_qkv.weight
    assert torch.equal(ops_bf16.bf16_weight(w), w.detach().to(torch.bfloat16))
    assert not o.arena.shadow_stale
    check_all()
    # load_state_dict changes the masters as well
    sd = {k: (v + 1 if v.is_floating_point() else v) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    assert o.arena.shadow_stale
    assert torch.equal(ops_bf16.bf16_weight(w), w.detach().to(torch.bfloat16))


def test_arena_fragment_major_images_follow_the_master():
    """The feed-forward weights of full-size blocks (128 <-> 1024) get fragment-major images in the arena
    (kantts_fragmajor_bf16, one launch for the whole table): they must equal the images built from the parameter with
    tensor ops, for k = 1 and k = 3 (forward and transposed images, one per tap), and follow an update."""
    import torch.nn as nn

    from kantts._hip.ops_bf16 import ffn_frag_weights, frag_major
    from kantts.models.sambert import PositionwiseConvFeedForward
    from kantts.train.optim import ParamArena

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = PositionwiseConvFeedForward(128, 1024, (1, 1))
            self.b = PositionwiseConvFeedForward(128, 1024, (3, 1))
            self.c = PositionwiseConvFeedForward(64, 256, (3, 1))     # unsupported shape: no images

        def forward(self, x):
            return x

    torch.manual_seed(3)
    net = Net().cuda()
    arena = ParamArena(net, bf16_shadow=True)

    def check():
        for blk, kt in ((net.a, 1), (net.b, 3)):
            w1, w2 = blk.w_1.weight, blk.w_2.weight
            f1, f2, t2, t1 = ffn_frag_weights(w1, w2)
            assert f1 is w1._kantts_frag and f2 is w2._kantts_frag
            assert torch.equal(f1, frag_major(w1.detach().permute(2, 0, 1).reshape(kt * 1024, 128)))
            assert torch.equal(f2, frag_major(w2.detach().reshape(128, 1024)))
            assert torch.equal(t2, frag_major(w2.detach().reshape(128, 1024).t()))
            assert torch.equal(t1, frag_major(w1.detach().permute(2, 1, 0).reshape(kt * 128, 1024)))
        assert not hasattr(net.c.w_1.weight, "_kantts_frag")

    check()
    with torch.no_grad():
        arena.flat.mul_(1.5).add_(0.01)
    net(torch.zeros(1, device="cuda"))  # forward pre-hook: refresh
    check()


def test_layernorm_in_the_producer_epilogues_equals_the_separate_kernel(monkeypatch):
    """kantts_bgemm_nt / kantts_ffn_pair ``ln_*``: LayerNorm(128) of the rows the epilogue has just formed (bgemm_nt: same
    arithmetic in the same order as ln128_fwd_kernel; ffn_pair: a token's channels are spread over the eight waves, so
    the two-pass statistics are summed in another order -- fp32 rounding, an occasional bf16 tie in the normalised rows).
    Encoder stack forward + backward with the hand-over on and off, and against the emulated C ABI."""
    from kantts._hip import ops_bf16
    from kantts.models.sambert.kantts_sambert import SelfAttentionEncoder

    def run(on):
        monkeypatch.setitem(ops_bf16.PRENORM, "on", on)
        torch.manual_seed(5)
        enc = SelfAttentionEncoder(3, 128, 128, 8, 16, 1024, 0.0, 0.0, 0.0, position_encoder=None).cuda()
        enc.train()
        x = torch.randn(5, 77, 128, generator=torch.Generator().manual_seed(2)).cuda().requires_grad_(True)
        lens = torch.tensor([77, 9, 50, 33, 64])
        mask = (torch.arange(77)[None, :] >= lens[:, None]).cuda()
        y, _ = enc(x, mask, prescaled=True)
        (y * torch.randn(y.shape, generator=torch.Generator().manual_seed(1)).cuda()).sum().backward()
        return y.detach(), x.grad.clone()

    y_on, g_on = run(True)
    y_off, g_off = run(False)
    # a bf16 tie in one normalised row element moves a block's input by one bf16 ulp: 1e-5 .. 1e-4 at the stack's output
    assert rel_l2(y_on, y_off) < 1e-3, rel_l2(y_on, y_off)
    assert rel_l2(g_on, g_off) < 5e-3, rel_l2(g_on, g_off)

    # the two epilogues on their own against the emulation: rows, statistics
    from kantts._hip import bgemm_nt, ffn_pair
    from kantts._hip.ops_bf16 import frag_major

    def epilogues(x, w, w1, w2, gam, bet, res):
        M = x.shape[0]
        dev = x.device
        outs = []
        for which in ("nt", "pair"):
            y = torch.empty((M, 128), device=dev)
            xn = torch.empty((M, 128), device=dev, dtype=torch.bfloat16)
            mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
            ln = (gam, bet, 1e-6, xn, mean, rstd)
            if which == "nt":
                assert bgemm_nt([(x.to(torch.bfloat16), 128, w.to(torch.bfloat16), 128, 128, 0)], M, 128, y, 128, res=res,
                                ldr=128, ln=ln)
            else:
                hid = torch.empty((M, 1024), device=dev, dtype=torch.bfloat16)
                assert ffn_pair(x.to(torch.bfloat16), frag_major(w1), frag_major(w2), y, M=M, T=M, F=1024, relu=True,
                                t_out=hid, res=res, ln=ln)
            outs += [y, xn.float(), mean, rstd]
        return tuple(outs)

    g = torch.Generator().manual_seed(9)
    M = 300
    args = [torch.randn(M, 128, generator=g), torch.randn(128, 128, generator=g) * 0.1, torch.randn(1024, 128, generator=g) * 0.1,
            torch.randn(128, 1024, generator=g) * 0.05, torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g),
            torch.randn(M, 128, generator=g)]
    go, _, co, _ = run_both(epilogues, *args)
    for k, (a, b) in enumerate(zip(go, co)):
        assert rel_l2(a, b) < (3e-3 if k % 4 == 1 else 2e-4), (k, rel_l2(a, b))


# kantts_bgemm_nt_lnbwd was written blind at the end of round 3 and first ran on a device in round 4 (9 / 9 cases green,
# profiles/r04_runA_*): the tests are unconditional since.  (The feed-forward pair's analogue was slower than the two
# launches it replaced and was removed together with its test.)


@pytest.mark.parametrize("M,a_f32,with_res,with_rows", [(6528, False, True, True), (100, True, True, False), (37, False, False, True),
                                                         (5, True, False, False)])
def test_layernorm_backward_as_the_epilogue_of_the_input_gradient(M, a_f32, with_res, with_rows):
    """kantts_bgemm_nt_lnbwd: dx / dgamma / dbeta of a LayerNorm(128) straight from the launch that computes the gradient
    of its output (the input gradient of a 128 -> 384 projection), against kantts_bgemm_nt (bf16 result) followed by
    kantts_ln128_bwd_rows -- same bf16-rounded dy on both sides, so only summation order differs -- and, through
    run_both, against the numpy model of both entry points."""
    import kantts._hip as hip
    from kantts._hip import ops_bf16

    g = torch.Generator().manual_seed(M)
    N = 384
    dz = torch.randn(M, N, generator=g) * 0.3
    if not a_f32:
        dz = dz.to(torch.bfloat16)
    w = (torch.randn(N, 128, generator=g) * 0.1).to(torch.bfloat16)
    x = torch.randn(M, 128, generator=g) * 2 + 0.5
    gam = torch.rand(128, generator=g) + 0.5
    dres = torch.randn(M, 128, generator=g) if with_res else None
    rows = (torch.arange(M) % 5 == 1).to(torch.uint8) if with_rows else None

    def both(dz, w, x, gam, dres, rows):
        dev = x.device
        mean = x.mean(-1).contiguous()
        rstd = (x.var(-1, unbiased=False) + 1e-6).rsqrt().contiguous()
        seg = [(dz, N, (w, 0), 128, N, 0)]
        # two launches
        dxn = torch.empty(M, 128, device=dev, dtype=torch.bfloat16)
        assert hip.bgemm_nt(seg, M, 128, dxn, 128, b_kn=True)
        dx2, dg2, db2 = torch.empty(M, 128, device=dev), torch.zeros(128, device=dev), torch.zeros(128, device=dev)
        hip.check(hip.lib().kantts_ln128_bwd_rows(hip.ptr(dxn), 1, hip.ptr(x), hip.ptr(gam), hip.ptr(mean), hip.ptr(rstd),
                                                  hip.ptr(dres), hip.ptr(dx2), hip.ptr(dg2), hip.ptr(db2), hip.ptr(rows), M,
                                                  hip.stream()), "ln128_bwd_rows")
        # one launch, dy not stored
        dx1, dg1, db1 = torch.empty(M, 128, device=dev), torch.zeros(128, device=dev), torch.zeros(128, device=dev)
        assert hip.bgemm_nt(seg, M, 128, None, 128, b_kn=True, c_bf16=True,
                            lnb=(x, gam, mean, rstd, dres, rows, dx1, dg1, db1))
        # one launch, dy stored as well
        dxn3 = torch.empty(M, 128, device=dev, dtype=torch.bfloat16)
        dx3, dg3, db3 = torch.empty(M, 128, device=dev), torch.zeros(128, device=dev), torch.zeros(128, device=dev)
        assert hip.bgemm_nt(seg, M, 128, dxn3, 128, b_kn=True, lnb=(x, gam, mean, rstd, dres, rows, dx3, dg3, db3))
        return dx1, dg1, db1, dx2, dg2, db2, dx3, dg3, db3, dxn.float(), dxn3.float()

    go, _, co, _ = run_both(both, dz, w, x, gam, dres, rows)
    for k, (a, b) in enumerate(zip(go, co)):  # device (or kernel source) against the numpy model, entry point by entry point
        assert rel_l2(a, b) <= 3e-3, (k, rel_l2(a, b))
    dx1, dg1, db1, dx2, dg2, db2, dx3, dg3, db3, dxn, dxn3 = go
    assert rel_l2(dxn3, dxn) <= 2e-3  # the result itself, stored by either kernel
    for a, b in ((dx1, dx2), (dx3, dx2)):
        assert rel_l2(a, b) <= 1e-5, rel_l2(a, b)
    for a, b in ((dg1, dg2), (db1, db2), (dg3, dg2), (db3, db2)):
        assert rel_l2(a, b) <= 1e-4, rel_l2(a, b)
    if rows is not None:
        assert float(dx1[rows.bool()].abs().max()) == 0.0




@pytest.mark.gpu
@pytest.mark.parametrize("tile", ["64", "128"])
def test_weight_gradient_contraction_with_both_output_tiles_gpu(tile, monkeypatch):
    """bgemm_tn_kernel with the 64 x 128 and the 128 x 256 output tile (KANTTS_TN_TILE forces one; on its own the launcher
    takes the large tile for M >= 4096 problems with enough tiles) against a float64 contraction of the bf16-rounded
    operands: the decoder feed-forward shape, a ragged one, a 3-tap convolution gradient with a bias gradient."""
    import kantts._hip as hip

    monkeypatch.setenv("KANTTS_TN_TILE", tile)
    g = torch.Generator().manual_seed(int(tile))
    for (M, N, K, a32, b32, taps, T) in ((6528, 128, 1024, True, False, 1, 0), (1000, 136, 264, False, True, 1, 0),
                                          (2048, 1024, 128, False, False, 3, 64), (4500, 256, 512, True, True, 1, 0)):
        a = torch.randn(M, N, generator=g) * 0.5
        b = torch.randn(M, K, generator=g) * 0.5
        ad = (a if a32 else a.to(torch.bfloat16)).cuda()
        bd = (b if b32 else b.to(torch.bfloat16)).cuda()
        c = torch.zeros(taps, N, K, device="cuda")
        db = torch.zeros(N, device="cuda")
        assert hip.bgemm_tn(ad, N, bd, K, M, N, K, c, K, 1, c_ts=N * K, T=T, ntaps=taps, shift0=-(taps // 2), shift_step=1 if taps > 1 else 0,
                            db=db)
        torch.cuda.synchronize()
        A = a.to(torch.bfloat16).double()  # fp32 operands are rounded to bf16 when a tile is staged
        Bq = b.to(torch.bfloat16).double()
        for tap in range(taps):
            sh = tap - taps // 2
            Bs = torch.zeros_like(Bq)
            if sh == 0:
                Bs = Bq
            else:
                idx = torch.arange(M)
                tt = idx % T + sh
                ok = (tt >= 0) & (tt < T)
                Bs[ok] = Bq[idx[ok] + sh]
            ref = A.t() @ Bs
            assert rel_l2(c[tap].cpu().double(), ref) <= 2e-5, (tile, M, N, K, tap)
        assert rel_l2(db.cpu().double(), A.sum(0)) <= 2e-5
