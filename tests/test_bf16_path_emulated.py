"""Host logic of the bf16-storage path (kantts/_hip/ops_bf16.py) under the emulated C ABI: segment / stride / shift
arithmetic, backward formulas, dtype plumbing, the arena's bf16 shadow.  The emulation rounds operands to bf16 exactly
like the kernels (round to nearest even, fp32 accumulate), so comparisons against plain torch on bf16-rounded operands
are tight; against the fp32 oracle they carry the bf16 error of the mode."""
import pytest
import torch
import torch.nn.functional as F

import torch_oracle as O
from util import emulation, rel_l2


def _bf(x):
    return x.to(torch.bfloat16).float()


@pytest.fixture
def bf16_mode():
    import kantts._hip as hip

    hip.set_precision("bf16")
    yield
    hip.set_precision("fp32")


def test_linear_modes_bf16_emulated(bf16_mode):
    from kantts._hip import ops

    g = torch.Generator().manual_seed(0)
    with emulation():
        # ---- concat of two inputs (fp32 + bf16 mixed), bias, fp32 output with residual and row zeroing
        M, K1, K2, N = 24, 16, 8, 32
        x1 = torch.randn(3, 8, K1, generator=g, requires_grad=True)
        x2 = torch.randn(3, 8, K2, generator=g).to(torch.bfloat16).requires_grad_(True)
        w = (torch.randn(N, K1 + K2, generator=g) * 0.2).requires_grad_(True)
        b = torch.randn(N, generator=g, requires_grad=True)
        res = torch.randn(3, 8, N, generator=g, requires_grad=True)
        rm = torch.zeros(3, 8, dtype=torch.bool)
        rm[1, 5:] = True
        y = ops.linear([x1, x2], w, b, mode="concat", res=res, rowmask=rm)
        assert y.dtype == torch.float32
        ref = (F.linear(torch.cat([_bf(x1), x2.float()], -1), _bf(w), b) + res).masked_fill(rm[..., None], 0.0)
        assert rel_l2(y.detach(), ref.detach()) < 1e-5
        cot = torch.randn(y.shape, generator=g)
        gy = torch.autograd.grad((y * cot).sum(), [x1, x2, w, b, res])
        gr = torch.autograd.grad((ref * cot).sum(), [x1, x2, w, b, res])
        assert gy[0].dtype == torch.float32 and gy[1].dtype == torch.bfloat16
        for a, r_, tol in zip(gy, gr, (1e-2, 2e-2, 1e-2, 5e-3, 1e-6)):
            assert rel_l2(a.float(), r_.float()) < tol
        # ---- relu + dropout, bf16 output; the gate and the regenerated mask must agree between forward and backward
        x = torch.randn(M, K1, generator=g, requires_grad=True)
        w2 = (torch.randn(N, K1, generator=g) * 0.3).requires_grad_(True)
        b2 = torch.zeros(N, requires_grad=True)
        y = ops.linear(x, w2, b2, relu=True, drop_p=0.5, out_bf16=True)
        assert y.dtype == torch.bfloat16
        keep = (y.float() > 0)
        z = F.linear(_bf(x), _bf(w2), b2)
        ref = torch.where(keep, z * 2.0, torch.zeros_like(z))  # kept & active elements, scaled by 1/(1-p)
        assert rel_l2(y.float().detach(), ref.detach()) < 1e-2
        frac = float(((z > 0) & ~keep).float().sum() / (z > 0).float().sum())
        assert 0.3 < frac < 0.7  # about half of the active elements were dropped
        cot = torch.randn(M, N, generator=g)
        gx, gw, gb = torch.autograd.grad((y.float() * cot).sum(), [x, w2, b2])
        rx, rw, rb = torch.autograd.grad((ref * cot).sum(), [x, w2, b2])
        assert rel_l2(gx, rx) < 2e-2 and rel_l2(gw, rw) < 2e-2 and rel_l2(gb, rb) < 2e-2
        # ---- "sum" of two projections with dropout on the sum and a residual (fc_x + fc_h)
        xa = torch.randn(M, 16, generator=g, requires_grad=True)
        xb = torch.randn(M, 16, generator=g, requires_grad=True)
        wa = (torch.randn(N, 16, generator=g) * 0.2).requires_grad_(True)
        wb = (torch.randn(N, 16, generator=g) * 0.2).requires_grad_(True)
        ba, bb = torch.randn(N, generator=g, requires_grad=True), torch.randn(N, generator=g, requires_grad=True)
        r2 = torch.randn(M, N, generator=g, requires_grad=True)
        y = ops.linear([xa, xb], [wa, wb], ba, bias2=bb, mode="sum", res=r2, drop_p=0.25)
        z = F.linear(_bf(xa), _bf(wa), ba) + F.linear(_bf(xb), _bf(wb), bb)
        d = (y - r2).detach()
        mask = torch.where(d.abs() > 0, torch.full_like(d, 1 / 0.75), torch.zeros_like(d))
        ref = z * mask + r2
        assert rel_l2(y.detach(), ref.detach()) < 1e-5
        cot = torch.randn(M, N, generator=g)
        gy = torch.autograd.grad((y * cot).sum(), [xa, xb, wa, wb, ba, r2])
        gr = torch.autograd.grad((ref * cot).sum(), [xa, xb, wa, wb, ba, r2])
        for a, r_ in zip(gy, gr):
            assert rel_l2(a, r_) < 1e-2


def test_conv_mode_and_ffn_bf16_emulated(bf16_mode):
    from kantts._hip import ops

    g = torch.Generator().manual_seed(1)
    with emulation():
        B, T, C, Fh = 2, 9, 16, 40
        x = torch.randn(B, T, C, generator=g, requires_grad=True)
        w1 = (torch.randn(Fh, C, 3, generator=g) * 0.2).requires_grad_(True)
        b1 = torch.randn(Fh, generator=g, requires_grad=True)
        y = ops.linear(x, w1, b1, mode="conv", pad=1)
        ref = F.conv1d(_bf(x).transpose(1, 2), _bf(w1), b1, padding=1).transpose(1, 2)
        assert rel_l2(y.detach(), ref.detach()) < 1e-5
        cot = torch.randn(y.shape, generator=g)
        gy = torch.autograd.grad((y * cot).sum(), [x, w1, b1])
        gr = torch.autograd.grad((ref * cot).sum(), [x, w1, b1])
        for a, r_ in zip(gy, gr):
            assert a.shape == r_.shape and rel_l2(a, r_) < 1e-2
        # ---- the fused FFN node (LN output -> conv3 + relu -> conv1 + residual), dropout off, padded rows
        w2 = (torch.randn(C, Fh, 1, generator=g) * 0.2).requires_grad_(True)
        b2 = torch.randn(C, generator=g, requires_grad=True)
        gam, bet = torch.ones(C, requires_grad=True), torch.zeros(C, requires_grad=True)
        pad_rows = torch.zeros(B, T, dtype=torch.bool)
        pad_rows[1, 6:] = True
        for k1 in (3, 1):
            w1k = w1 if k1 == 3 else (torch.randn(Fh, C, 1, generator=g) * 0.2).requires_grad_(True)
            h = _bf(x.detach()).to(torch.bfloat16).requires_grad_(True)
            out = ops.ffn(h, w1k, b1, w2, b2, x, pad_rows=pad_rows, zero_rows=pad_rows)
            hid = F.relu(F.conv1d(h.float().transpose(1, 2), _bf(w1k), b1, padding=(k1 - 1) // 2)).transpose(1, 2)
            hid = _bf(hid.masked_fill(pad_rows[..., None], 0.0))
            ref = (F.conv1d(hid.transpose(1, 2), _bf(w2), b2).transpose(1, 2) + x).masked_fill(pad_rows[..., None], 0.0)
            assert rel_l2(out.detach(), ref.detach()) < 1e-3
            cot = torch.randn(out.shape, generator=g)
            gy = torch.autograd.grad((out * cot).sum(), [h, w1k, b1, w2, b2, x])
            gr = torch.autograd.grad((ref * cot).sum(), [h, w1k, b1, w2, b2, x])
            assert gy[0].dtype == torch.bfloat16
            for a, r_ in zip(gy, gr):
                assert a.shape == r_.shape and rel_l2(a.float(), r_.float()) < 2e-2, k1


def test_tiny_sambert_bf16_mode_emulated_close_to_oracle(bf16_mode, monkeypatch, ln_bwd_epilogue=True):
    """The whole model through the bf16 path (shadow weights cast on demand: no arena on the CPU) stays within the bf16
    error of the oracle and produces every gradient; index outputs stay bit-exact.  ``ln_bwd_epilogue`` (run by
    tests/test_kernel_source_on_cpu.py, where the whole case takes 2 s): with / without the LayerNorm backward of
    every attention sub-layer in the QKV input-gradient launch (ops_bf16.LnBwdToken, on by default)."""
    from kantts._hip import ops_bf16
    from kantts.models.sambert.kantts_sambert import KanTtsSAMBERT

    monkeypatch.setitem(ops_bf16.LNBWD, "on", ln_bwd_epilogue)
    from kantts.train.loss import MelReconLoss, ProsodyReconLoss

    cfg = O.sambert_config(tiny=True)
    cfg = {k: (0.0 if "dropout" in k else v) for k, v in cfg.items()}
    torch.manual_seed(0)
    m = KanTtsSAMBERT(dict(cfg))
    m.eval()
    P = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    batch = O.synthetic_sambert_batch(B=3, T_in=12, seed=10, min_len=6, dur_hi=6)
    with emulation():
        res = m(**batch)
        mel_, mel = MelReconLoss()(batch["output_lengths"], batch["mel_targets"], res["dec_outputs"], res["postnet_outputs"])
        d, p, e = ProsodyReconLoss()(batch["input_lengths"], res["duration_targets"], res["pitch_targets"],
                                     res["energy_targets"], res["log_duration_predictions"], res["pitch_predictions"],
                                     res["energy_predictions"])
        total = mel_ + mel + d + p + e
        total.backward()
    out = O.sambert_forward(P, cfg, **batch)
    L = O.sambert_losses(out, batch["input_lengths"], batch["output_lengths"], batch["mel_targets"])
    L["total"].backward()
    assert torch.equal(res["LR_length_rounded"], out["LR_length_rounded"])
    err = float((res["postnet_outputs"].detach() - out["postnet_outputs"].detach()).abs().mean())
    assert err < 2e-2, err
    assert abs(float(total.detach()) - float(L["total"].detach())) < 5e-2
    num = den = 0.0
    for n, prm in m.named_parameters():
        if prm.requires_grad:
            assert prm.grad is not None and prm.grad.dtype == torch.float32 and prm.grad.shape == prm.shape, n
            num += float((prm.grad.double() - P[n].grad.double()).pow(2).sum())
            den += float(P[n].grad.double().pow(2).sum())
    assert (num / den) ** 0.5 < 0.15, (num / den) ** 0.5


def test_deferred_grouped_weight_gradients_equal_immediate_ones(bf16_mode):
    """kantts._hip.deferred_tn: recording the weight-gradient contractions during backward and issuing them grouped by
    shape afterwards gives bit-identical parameter gradients (same arithmetic per problem), and every p.grad is the
    buffer the deferred launch fills (autograd must adopt it, not clone it)."""
    import kantts._hip as hip
    from kantts._hip import ops
    from kantts.models.sambert.kantts_sambert import KanTtsSAMBERT
    from kantts.train.loss import MelReconLoss, ProsodyReconLoss

    cfg = O.sambert_config(tiny=True)
    cfg = {k: (0.0 if "dropout" in k else v) for k, v in cfg.items()}
    batch = O.synthetic_sambert_batch(B=3, T_in=12, seed=10, min_len=6, dur_hi=6)
    grads = []
    with emulation():
        for on in (False, True):
            torch.manual_seed(0)
            m = KanTtsSAMBERT(dict(cfg))
            m.eval()
            hip.deferred_tn.enabled = on
            try:
                res = m(**batch)
                mel_, mel = MelReconLoss()(batch["output_lengths"], batch["mel_targets"], res["dec_outputs"],
                                           res["postnet_outputs"])
                d, p, e = ProsodyReconLoss()(batch["input_lengths"], res["duration_targets"], res["pitch_targets"],
                                             res["energy_targets"], res["log_duration_predictions"],
                                             res["pitch_predictions"], res["energy_predictions"])
                (mel_ + mel + d + p + e).backward()
                if on:
                    n_prob = sum(len(v) for v in hip.deferred_tn.groups.values())
                    assert len(hip.deferred_tn.groups) < n_prob  # several layers share a launch
                ops.wgrad_overlap.join()
            finally:
                hip.deferred_tn.enabled = False
            grads.append({n: q.grad.clone() for n, q in m.named_parameters() if q.grad is not None})
    assert grads[0].keys() == grads[1].keys()
    for n in grads[0]:
        assert torch.equal(grads[0][n], grads[1][n]), n


@pytest.mark.parametrize("k1,F,B,T", [(1, 1024, 3, 37), (3, 1024, 2, 50), (5, 1024, 2, 33)])
def test_ffn_pair_equals_the_two_launch_form(bf16_mode, k1, F, B, T):
    """csrc/ffn_pair.hip behind ops.ffn: one launch for both feed-forward contractions (forward; backward for k = 1 with
    the transposed weight images) must give what the two-launch form gives -- same dropout masks, same bf16 rounding of
    the hidden tensor; outputs and every gradient agree to accumulation-order noise."""
    import itertools

    from kantts._hip import ops, ops_bf16

    g = torch.Generator().manual_seed(11 + k1)
    C = 128
    lens = torch.tensor([T, max(1, T - 9), 1][:B])
    pr = torch.arange(T)[None, :] >= lens[:, None]
    x = torch.randn(B, T, C, generator=g)
    w1 = torch.randn(F, C, k1, generator=g) * 0.05
    b1 = torch.randn(F, generator=g)
    w2 = torch.randn(C, F, 1, generator=g) * 0.03
    b2 = torch.randn(C, generator=g)
    cot = torch.randn(B, T, C, generator=g)
    res = {}
    with emulation():
        for on in (True, False):
            ops_bf16.PAIR["on"] = on
            ops._seed_counter = itertools.count(500)
            try:
                leaves = [t.clone().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
                h = ops.layer_norm(leaves[0], torch.ones(C), torch.zeros(C), 1e-6, out_bf16=True)
                y = ops.ffn(h, leaves[1], leaves[2], leaves[3], leaves[4], leaves[0], pad_rows=pr, zero_rows=pr,
                            p_inner=0.1, p_out=0.1)
                grads = torch.autograd.grad((y * cot).sum(), leaves)
            finally:
                ops_bf16.PAIR["on"] = True
            res[on] = [y.detach()] + [q.detach() for q in grads]
    for a, b in zip(res[True], res[False]):
        assert a.shape == b.shape
        assert rel_l2(a, b) < 1e-5, rel_l2(a, b)


def test_shared_input_linears_equal_separate_ones(bf16_mode):
    """ops.shared_input_linears (the memory K/V projections of all PNCA blocks): same outputs as n separate fused linears,
    and the single multi-segment input-gradient launch equals the sum autograd forms from n separate ones."""
    import torch.nn as nn

    from kantts._hip import ops

    torch.manual_seed(4)
    lins = [nn.Linear(160, 256) for _ in range(5)]
    x0 = torch.randn(3, 17, 160)
    cots = [torch.randn(3, 17, 256) for _ in lins]
    res = []
    with emulation():
        for shared in (True, False):
            x = x0.clone().requires_grad_(True)
            for m in lins:
                m.zero_grad()
            ys = ops.shared_input_linears(x, lins) if shared else [ops.linear(x, m.weight, m.bias) for m in lins]
            sum((y * c).sum() for y, c in zip(ys, cots)).backward()
            res.append(([y.detach() for y in ys], x.grad.clone(), [m.weight.grad.clone() for m in lins],
                        [m.bias.grad.clone() for m in lins]))
    (ya, gxa, gwa, gba), (yb, gxb, gwb, gbb) = res
    for a, b in zip(ya, yb):
        assert torch.equal(a, b)
    assert rel_l2(gxa, gxb) < 1e-3          # one bf16-operand contraction over 5 x 256 vs five fp32 partial sums
    for a, b in zip(gwa + gba, gwb + gbb):
        assert rel_l2(a, b) < 1e-6


def test_layernorm_in_the_producer_epilogue_equals_the_separate_launch(bf16_mode, monkeypatch):
    """ops_bf16.PreNorm: the attention sub-layer's output GEMM computes the feed-forward sub-layer's LayerNorm in its
    epilogue; the consumer adopts the rows + statistics instead of launching kantts_ln128_fwd.  Same outputs and
    gradients as with the hand-over switched off; the launches are really gone."""
    from kantts._hip import ops_bf16
    from kantts.models.sambert.kantts_sambert import SelfAttentionEncoder

    with emulation() as emu:
        def run(on):
            monkeypatch.setitem(ops_bf16.PRENORM, "on", on)
            calls = []
            orig = emu.kantts_ln128_fwd
            monkeypatch.setattr(emu, "kantts_ln128_fwd", lambda *a: (calls.append(1), orig(*a))[1], raising=False)
            torch.manual_seed(5)
            enc = SelfAttentionEncoder(2, 128, 128, 8, 16, 1024, 0.0, 0.0, 0.0, position_encoder=None)
            enc.train()
            x = torch.randn(3, 21, 128, requires_grad=True)
            mask = torch.arange(21)[None, :] >= torch.tensor([21, 9, 15])[:, None]
            y, _ = enc(x, mask, prescaled=True)
            (y * torch.randn(y.shape, generator=torch.Generator().manual_seed(1))).sum().backward()
            return y.detach(), [x.grad] + [p.grad for p in enc.parameters()], len(calls)

        y_on, g_on, n_on = run(True)
        y_off, g_off, n_off = run(False)
    assert n_off == 5 and n_on == 1  # 2 blocks x 2 sub-layers + the final LayerNorm: all but the very first are adopted
    assert torch.equal(y_on, y_off)
    for a, b in zip(g_on, g_off):
        assert torch.equal(a, b)


def test_weight_image_cache_does_not_outlive_its_tensor(bf16_mode):
    """ops_bf16.bf16_weight / ffn_frag_weights / ops.upsample_forward cache the bf16 images of tensors that live outside
    a parameter arena under id(tensor).  id() is only unique among live objects: a new tensor that inherits the id, the
    address, the shape and the version of a dead one (a second model built in the same process) used to inherit its
    image.  The entries now remember their tensor by weak reference; simulated here by planting a stale entry whose
    signature matches a live tensor."""
    import weakref

    from kantts._hip import ops_bf16

    with emulation():
        old = torch.randn(16, 32)
        stale = ops_bf16.bf16_weight(old)
        new = torch.randn(16, 32)
        ops_bf16._wcache[(id(new), False)] = (new._version, new.data_ptr(), tuple(new.shape), stale, weakref.ref(old))
        got = ops_bf16.bf16_weight(new)
        assert torch.equal(got, new.to(torch.bfloat16)) and not torch.equal(got, stale)
        assert ops_bf16.bf16_weight(new) is got  # and the fresh entry is a hit
        w1o, w2o = torch.randn(1024, 128, 1), torch.randn(128, 1024, 1)
        imgs = ops_bf16.ffn_frag_weights(w1o, w2o)
        w1, w2 = torch.randn(1024, 128, 1), torch.randn(128, 1024, 1)
        sig = (w1._version, w1.data_ptr(), tuple(w1.shape), w2._version, w2.data_ptr(), tuple(w2.shape))
        ops_bf16._wcache[(id(w1), id(w2), "frag")] = (sig, imgs, weakref.ref(w1o), weakref.ref(w2o))
        f1 = ops_bf16.ffn_frag_weights(w1, w2)[0]
        assert torch.equal(f1, ops_bf16.frag_major(w1.permute(2, 0, 1).reshape(1024, 128))) and not torch.equal(f1, imgs[0])
