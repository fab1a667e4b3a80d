"""Randomised sweeps of the SAM-BERT op wrappers (kantts._hip.ops) against plain torch / the oracle's building blocks: fused
linear in its three modes with every epilogue option, self / PNCA band attention over ragged lengths and band widths
(bands above 16 included), (Bi)LSTM with packed-sequence semantics, FSMN memory, length regulator, embedding sum, masked L1,
element losses, weight norm.  Forward and gradients; fixed seeds; small shapes (a few seconds).

EXPECTED VALUES COME FROM THE ORACLE / PLAIN TORCH, never from the numpy model of the C ABI.  Every sweep
runs on two back ends (fixture ``dv``): "emulated" = oracle/cabi_numpy.py on host tensors (CPU suite: pins the numpy model to
the same definitions), "gpu" = the HIP kernels of libkantts_hip.so on device tensors (``-m gpu``: the kernels against the
oracle directly, on shapes the model-level tests do not exercise -- odd extents, bands > 16, widths other than 128)."""
import os
import random

import pytest
import torch
import torch.nn.functional as F

import torch_oracle as O
from util import rel_l2


def _grads(out, cot, leaves):
    return torch.autograd.grad((out * cot).sum(), leaves, allow_unused=True)


def _cmp(got, exp, cfg, tol=2e-5):
    for a, e in zip(got, exp):
        if e is None:
            assert a is None or float(a.abs().max()) == 0.0, cfg
            continue
        assert a is not None, cfg
        assert rel_l2(a, e) < tol or float((a - e).abs().max()) < 1e-6, (cfg, rel_l2(a, e))


def test_fused_linear_modes(dv):
    from kantts._hip import ops

    rnd = random.Random(5)
    g = torch.Generator().manual_seed(5)
    for it in range(30):
        mode = rnd.choice(["concat", "sum", "conv"])
        B, T = rnd.choice([1, 2, 3]), rnd.randint(1, 19)
        N = rnd.choice([1, 3, 8, 33, 64])
        relu, alpha = rnd.random() < 0.4, rnd.choice([1.0, 0.5])
        if mode == "conv":
            cin, kt = rnd.choice([1, 4, 7, 32]), rnd.choice([1, 3, 9])
            pad, dil = (kt - 1) // 2, 1
            xs = [torch.randn(B, T, cin, generator=g).requires_grad_(True)]
            ws = [(torch.randn(N, cin, kt, generator=g) / (cin * kt) ** 0.5).requires_grad_(True)]
            ref = F.conv1d(xs[0].transpose(1, 2), ws[0], None, padding=pad).transpose(1, 2)
            kw = dict(mode="conv", pad=pad, dilation=dil)
        else:
            ks = [rnd.choice([1, 5, 16, 40]) for _ in range(rnd.choice([1, 2, 3]))]
            xs = [torch.randn(B, T, k, generator=g).requires_grad_(True) for k in ks]
            if mode == "concat":
                ws = [(torch.randn(N, sum(ks), generator=g) / sum(ks) ** 0.5).requires_grad_(True)]
                ref = F.linear(torch.cat(xs, -1), ws[0])
            else:
                ws = [(torch.randn(N, k, generator=g) / k ** 0.5).requires_grad_(True) for k in ks]
                ref = sum(F.linear(x, w) for x, w in zip(xs, ws))
            kw = dict(mode=mode)
        bias = torch.randn(N, generator=g).requires_grad_(True) if rnd.random() < 0.7 else None
        bias2 = torch.randn(N, generator=g).requires_grad_(True) if (bias is not None and rnd.random() < 0.3) else None
        res = torch.randn(B, T, N, generator=g).requires_grad_(True) if rnd.random() < 0.4 else None
        rowmask = (torch.rand(B, T, generator=g) < 0.3) if rnd.random() < 0.4 else None
        if bias is not None:
            ref = ref + bias
        if bias2 is not None:
            ref = ref + bias2
        ref = ref * alpha
        if relu:
            ref = torch.relu(ref)
        if res is not None:
            ref = ref + res
        if rowmask is not None:
            ref = ref.masked_fill(rowmask[..., None], 0.0)
        cfg = dict(it=it, mode=mode, B=B, T=T, N=N, relu=relu, alpha=alpha, bias=bias is not None, bias2=bias2 is not None,
                   res=res is not None, rowmask=rowmask is not None, shapes=[tuple(x.shape) for x in xs])
        y = ops.linear(dv(xs if len(xs) > 1 else xs[0]), dv(ws if len(ws) > 1 else ws[0]), dv(bias), bias2=dv(bias2),
                       res=dv(res), rowmask=dv(rowmask), relu=relu, alpha=alpha, **kw)
        assert float((dv.back(y) - ref).detach().abs().max()) <= 2e-5 * max(1.0, float(ref.detach().abs().max())), cfg
        cot = torch.randn(ref.shape, generator=g)
        leaves = [t for t in (*xs, *ws, bias, bias2, res) if t is not None]
        _cmp(dv.back(_grads(y, dv(cot), dv(leaves))), _grads(ref, cot, leaves), cfg)


def test_attention_ragged_lengths_and_bands(dv):
    from kantts._hip import ops

    rnd = random.Random(11)
    g = torch.Generator().manual_seed(11)
    for it in range(12):
        B, L, H = rnd.choice([1, 2, 4]), rnd.randint(1, 23), rnd.choice([1, 2, 8])
        D = H * 16
        lens = torch.tensor([rnd.randint(1, L) for _ in range(B)])
        lens[rnd.randrange(B)] = L
        pad = O.pad_mask(lens, L)
        qkv = torch.randn(B, L, 3 * D, generator=g).requires_grad_(True)
        cfg = dict(it=it, B=B, L=L, H=H, lens=lens.tolist())
        # encoder self-attention: keys beyond the length are masked for every query
        o, _ = ops.self_attention(dv(qkv), dv(lens.to(torch.int32)), H)
        o_dev, o = o, dv.back(o)
        q, k, v = (O._split_heads(t, H) for t in qkv.chunk(3, -1))
        ro, _ = O._attend(q, k, v, pad[:, None, :].expand(-1, L, -1).repeat(H, 1, 1))
        ro = O._merge_heads(ro, H)
        valid = (~pad)[..., None]
        assert float(((o - ro) * valid).detach().abs().max()) < 2e-5, cfg
        cot = torch.randn(B, L, D, generator=g) * valid
        _cmp(dv.back(_grads(o_dev, dv(cot), dv([qkv]))), _grads(ro, cot, [qkv]), cfg)
        # decoder PNCA attention: causal x-band over its own keys, look-ahead h-band over the memory
        bwx, bwh = rnd.randint(0, L + 2), rnd.randint(0, L + 2)
        hkv = torch.randn(B, L, 2 * D, generator=g).requires_grad_(True)
        ox_d, oh_d, _, _ = ops.pnca_attention(dv(qkv), dv(hkv), dv(lens.to(torch.int32)), bwx, bwh, H)
        ox, oh = dv.back(ox_d), dv.back(oh_d)
        xm, hm = O.pnca_masks(L, bwx, bwh, pad, qkv.device)
        hk, hv = (O._split_heads(t, H) for t in hkv.chunk(2, -1))
        rx, _ = O._attend(q, k, v, xm.expand(B, -1, -1).repeat(H, 1, 1))
        rh, _ = O._attend(q, hk, hv, hm.expand(B, -1, -1).repeat(H, 1, 1))
        rx, rh = O._merge_heads(rx, H), O._merge_heads(rh, H)
        cfg.update(bwx=bwx, bwh=bwh)
        assert float(((ox - rx) * valid).detach().abs().max()) < 2e-5, cfg
        assert float(((oh - rh) * valid).detach().abs().max()) < 2e-5, cfg
        c2 = torch.randn(B, L, D, generator=g) * valid
        got = torch.autograd.grad((ox_d * dv(cot)).sum() + (oh_d * dv(c2)).sum(), dv([qkv, hkv]))  # one pass: shared node
        exp = torch.autograd.grad((rx * cot).sum() + (rh * c2).sum(), [qkv, hkv])
        _cmp(dv.back(got), exp, cfg)


@pytest.mark.parametrize("L", [300, 400])
def test_attention_longer_than_a_workgroup(dv, L):
    """More rows than the 256 threads of an attention workgroup: up to 390 rows the K / V tiles still fit the 64 KB of LDS
    and a thread owns a second query / key row (read from global, not from its registers); past that the direct-from-global
    kernels run.  Self-attention and both PNCA bands, outputs and gradients against the oracle (the shipped shapes stop at
    204 rows; the device suite has 600)."""
    from kantts._hip import ops

    g = torch.Generator().manual_seed(L)
    B, H = 2, 1
    D = H * 16
    lens = torch.tensor([L, L - 37])
    pad = O.pad_mask(lens, L)
    qkv = torch.randn(B, L, 3 * D, generator=g).requires_grad_(True)
    cfg = dict(B=B, L=L, H=H)
    o_dev, _ = ops.self_attention(dv(qkv), dv(lens.to(torch.int32)), H)
    o = dv.back(o_dev)
    q, k, v = (O._split_heads(t, H) for t in qkv.chunk(3, -1))
    ro, _ = O._attend(q, k, v, pad[:, None, :].expand(-1, L, -1).repeat(H, 1, 1))
    ro = O._merge_heads(ro, H)
    valid = (~pad)[..., None]
    assert float(((o - ro) * valid).detach().abs().max()) < 2e-5, cfg
    cot = torch.randn(B, L, D, generator=g) * valid
    _cmp(dv.back(_grads(o_dev, dv(cot), dv([qkv]))), _grads(ro, cot, [qkv]), cfg)
    bwx, bwh = 9, 5
    hkv = torch.randn(B, L, 2 * D, generator=g).requires_grad_(True)
    ox_d, oh_d, _, _ = ops.pnca_attention(dv(qkv), dv(hkv), dv(lens.to(torch.int32)), bwx, bwh, H)
    ox, oh = dv.back(ox_d), dv.back(oh_d)
    xm, hm = O.pnca_masks(L, bwx, bwh, pad, qkv.device)
    hk, hv = (O._split_heads(t, H) for t in hkv.chunk(2, -1))
    rx, _ = O._attend(q, k, v, xm.expand(B, -1, -1).repeat(H, 1, 1))
    rh, _ = O._attend(q, hk, hv, hm.expand(B, -1, -1).repeat(H, 1, 1))
    rx, rh = O._merge_heads(rx, H), O._merge_heads(rh, H)
    assert float(((ox - rx) * valid).detach().abs().max()) < 2e-5, cfg
    assert float(((oh - rh) * valid).detach().abs().max()) < 2e-5, cfg
    c2 = torch.randn(B, L, D, generator=g) * valid
    got = torch.autograd.grad((ox_d * dv(cot)).sum() + (oh_d * dv(c2)).sum(), dv([qkv, hkv]))
    exp = torch.autograd.grad((rx * cot).sum() + (rh * c2).sum(), [qkv, hkv])
    _cmp(dv.back(got), exp, cfg)


def test_lstm_uni_and_bidirectional_with_lengths(dv):
    from kantts._hip import ops

    rnd = random.Random(3)
    g = torch.Generator().manual_seed(3)
    for it in range(8):
        B, T, H = rnd.choice([1, 2, 3]), rnd.randint(1, 9), 128
        ndir = rnd.choice([1, 2])
        ks = [rnd.choice([3, 16, 32]) for _ in range(rnd.choice([1, 2]))]
        xs = [torch.randn(B, T, k, generator=g).requires_grad_(True) for k in ks]
        use_len = ndir == 2 or rnd.random() < 0.5
        lens = torch.tensor([rnd.randint(1, T) for _ in range(B)]) if use_len else None
        if lens is not None:
            lens[0] = T
        params = []
        for _ in range(ndir):
            params += [(torch.randn(4 * H, sum(ks), generator=g) * 0.1).requires_grad_(True),
                       (torch.randn(4 * H, H, generator=g) * 0.05).requires_grad_(True),
                       (torch.randn(4 * H, generator=g) * 0.1).requires_grad_(True),
                       (torch.randn(4 * H, generator=g) * 0.1).requires_grad_(True)]
        y_dev = ops.lstm(dv(xs), dv(params), None if lens is None else dv(lens.to(torch.int32)))
        y = dv.back(y_dev)
        xcat = torch.cat(xs, -1)
        ref = torch.cat([O.lstm_layer(xcat, *params[4 * d:4 * d + 4], lengths=lens, reverse=(d == 1))
                         for d in range(ndir)], -1)
        cfg = dict(it=it, B=B, T=T, ndir=ndir, ks=ks, lens=None if lens is None else lens.tolist())
        assert float((y - ref).detach().abs().max()) < 2e-5, cfg
        cot = torch.randn(ref.shape, generator=g)
        _cmp(dv.back(_grads(y_dev, dv(cot), dv(xs + params))), _grads(ref, cot, xs + params), cfg, tol=5e-5)


def test_fsmn_memory_length_regulator_embedding_and_masked_l1(dv):
    from kantts._hip import ops

    rnd = random.Random(8)
    g = torch.Generator().manual_seed(8)
    for it in range(10):
        B, T, C = rnd.choice([1, 2, 3]), rnd.randint(1, 30), rnd.choice([4, 32, 128])
        lens = torch.tensor([rnd.randint(1, T) for _ in range(B)])
        lens[-1] = T
        pad = O.pad_mask(lens, T)
        # FSMN memory: masked input -> depth-wise FIR with (lp, rp) zero padding -> + input -> masked (+ res)
        K = rnd.choice([3, 11, 41])
        lp = rnd.randint(0, K - 1)
        x = torch.randn(B, T, C, generator=g).requires_grad_(True)
        w = (torch.randn(C, 1, K, generator=g) / K ** 0.5).requires_grad_(True)
        res = torch.randn(B, T, C, generator=g).requires_grad_(True) if rnd.random() < 0.5 else None
        y_dev = ops.fsmn_memory(dv(x), dv(w), dv(lens), lp, res=dv(res))
        y = dv.back(y_dev)
        xm = x.masked_fill(pad[..., None], 0.0)
        ref = F.conv1d(F.pad(xm.transpose(1, 2), (lp, K - 1 - lp)), w, groups=C).transpose(1, 2) + xm
        ref = ref.masked_fill(pad[..., None], 0.0)
        if res is not None:
            ref = ref + res
        cfg = dict(it=it, B=B, T=T, C=C, K=K, lp=lp, lens=lens.tolist(), res=res is not None)
        assert float((y - ref).detach().abs().max()) < 2e-5, cfg
        cot = torch.randn(ref.shape, generator=g)
        leaves = [t for t in (x, w, res) if t is not None]
        _cmp(dv.back(_grads(y_dev, dv(cot), dv(leaves))), _grads(ref, cot, leaves), cfg)
        # masked L1: mean |pred - target| over valid rows (all channels)
        p = torch.randn(B, T, C, generator=g).requires_grad_(True)
        t = torch.randn(B, T, C, generator=g)
        loss = ops.masked_l1(dv(p), dv(t), dv(lens))
        valid = (~pad)[..., None].float()
        rloss = ((p - t).abs() * valid).sum() / (valid.sum() * C)
        assert abs(float(loss.detach()) - float(rloss.detach())) < 1e-5, cfg
        _cmp(dv.back(torch.autograd.grad(loss, dv([p]))), torch.autograd.grad(rloss, [p]), cfg)
    for it in range(8):
        # length regulator: token n repeated trunc(dur + 0.5) times, frames beyond the total are empty
        B, N, C = rnd.choice([1, 2, 4]), rnd.randint(1, 12), rnd.choice([4, 32])
        as_float = rnd.random() < 0.5
        dur = torch.randint(0, 5, (B, N), generator=g)
        durs = (dur.float() + (torch.rand(B, N, generator=g) - 0.5) * 0.98) if as_float else dur
        reps = (durs.float() + 0.5).long() if as_float else dur
        Tp = int(reps.sum(1).max()) + rnd.randint(0, 3)
        if Tp == 0:
            continue
        idx_d, pos_d, cs_d, tot_d = ops.lr_index(dv(durs if as_float else durs.long()), Tp)
        idx, pos, tot = dv.back(idx_d), dv.back(pos_d), dv.back(tot_d)
        assert torch.equal(tot, reps.sum(1)), (it, durs, tot)
        for b in range(B):
            exp = torch.repeat_interleave(torch.arange(N), reps[b])
            assert torch.equal(idx[b, :len(exp)].long(), exp) and torch.all(idx[b, len(exp):] == -1)
            within = torch.cat([torch.arange(1, r + 1) for r in reps[b].tolist()] + [torch.zeros(0, dtype=torch.long)])
            assert torch.equal(pos[b, :len(exp)].long(), within)
        x = torch.randn(B, N, C, generator=g).requires_grad_(True)
        vl = tot_d.clamp(max=Tp)
        y_dev = ops.lr_gather(dv(x), idx_d, cs_d, vl)
        y = dv.back(y_dev)
        ref = torch.zeros(B, Tp, C)
        for b in range(B):
            exp = torch.repeat_interleave(torch.arange(N), reps[b])
            ref[b, :len(exp)] = x.detach()[b, exp]
        assert torch.equal(y.detach(), ref), it
        cot = torch.randn(B, Tp, C, generator=g)
        gx = dv.back(torch.autograd.grad((y_dev * dv(cot)).sum(), dv([x])))[0]
        gref = torch.zeros_like(gx)
        for b in range(B):
            exp = torch.repeat_interleave(torch.arange(N), reps[b])
            gref[b].index_add_(0, exp, cot[b, :len(exp)])
        assert float((gx - gref).abs().max()) < 1e-5, it
    for it in range(7):
        # embedding gather-sum with scale and position table
        B, T, D = rnd.choice([1, 3]), rnd.randint(1, 9), rnd.choice([8, 32])
        sizes = [rnd.randint(2, 9) for _ in range(rnd.choice([1, 2, 4]))]
        if it == 6:  # rows spanning several 32-row chunks of the gradient kernel, more distinct ids per chunk than its
            B, T, D, sizes = 2, 75, 32, [40, 3, 1]  # 8-entry accumulator cache holds (evictions), and a one-row table
        tabs = [torch.randn(n, D, generator=g).requires_grad_(True) for n in sizes]
        ids = torch.stack([torch.randint(0, n, (B, T), generator=g) for n in sizes], -1)
        pos = torch.randn(T, D, generator=g) if rnd.random() < 0.5 else None
        scale = rnd.choice([1.0, 11.3])
        out_d, scaled_d = ops.embed_sum(dv(ids), dv(tabs), pos=dv(pos), scale=scale, want_scaled="grad")
        out, scaled = dv.back(out_d), dv.back(scaled_d)
        rs = sum(F.embedding(ids[..., k], tabs[k]) for k in range(len(sizes))) * scale
        ro = rs if pos is None else rs + pos[None]
        assert float((out - ro).detach().abs().max()) < 1e-5 and float((scaled - rs).detach().abs().max()) < 1e-5, it
        c1, c2 = torch.randn(B, T, D, generator=g), torch.randn(B, T, D, generator=g)
        got = torch.autograd.grad((out_d * dv(c1)).sum() + (scaled_d * dv(c2)).sum(), dv(tabs))
        exp = torch.autograd.grad((ro * c1).sum() + (rs * c2).sum(), tabs)
        _cmp(dv.back(got), exp, dict(it=it, sizes=sizes))


def test_elementwise_losses_weight_norm_sin_add(dv):
    from kantts._hip import ops

    rnd = random.Random(13)
    g = torch.Generator().manual_seed(13)
    for it in range(8):
        shape = [rnd.randint(1, 9) for _ in range(rnd.choice([1, 2, 3]))]
        a = torch.randn(shape, generator=g).requires_grad_(True)
        b = torch.randn(shape, generator=g)
        for got, exp in ((ops.l1_mean(dv(a), dv(b)), F.l1_loss(a, b)),
                         (ops.mse_to_const(dv(a), 1.0), F.mse_loss(a, torch.ones_like(a))),
                         (ops.mse_to_const(dv(a), 0.0), F.mse_loss(a, torch.zeros_like(a)))):
            assert abs(float(got.detach()) - float(exp.detach())) < 1e-5 * max(1.0, abs(float(exp.detach()))), (it, shape)
            # a scaled use of the loss (loss weights of the GAN step) scales the gradient
            ga, ge = torch.autograd.grad(got * 3.0, dv([a])), torch.autograd.grad(exp * 3.0, [a])
            _cmp(dv.back(ga), ge, dict(it=it, shape=shape))
        x = torch.randn(shape, generator=g).requires_grad_(True)
        y, ry = ops.sin_add(dv(x)), torch.sin(x) + x
        assert float((dv.back(y) - ry).detach().abs().max()) < 2e-6
        cot = torch.randn(shape, generator=g)
        _cmp(dv.back(_grads(y, dv(cot), dv([x]))), _grads(ry, cot, [x]), dict(it=it, op="sin_add"))
    for it in range(10):
        # weight norm over all dims but 0: parameter layout (Cout, Cin, K) and the kernels' tap-major (K, Cout, Cin)
        cout, cin, K = rnd.choice([1, 4, 32]), rnd.choice([1, 4, 8, 12]), rnd.choice([1, 3, 7, 41])
        v = torch.randn(cout, cin, K, generator=g).requires_grad_(True)
        gg = (torch.rand(cout, 1, 1, generator=g) + 0.5).requires_grad_(True)
        ref = gg * v / v.flatten(1).norm(dim=1).view(-1, 1, 1)
        w = ops.weight_norm(dv(v), dv(gg))
        cfg = dict(it=it, cout=cout, cin=cin, K=K)
        assert float((dv.back(w) - ref).detach().abs().max()) < 1e-5, cfg
        cot = torch.randn(ref.shape, generator=g)
        _cmp(dv.back(_grads(w, dv(cot), dv([v, gg]))), _grads(ref, cot, [v, gg]), cfg)
        if cin % 4 == 0:
            wt = ops.weight_norm_tap(dv(v), dv(gg))
            assert (tuple(wt.shape) == (K, cout, cin)
                    and float((dv.back(wt) - ref.permute(2, 0, 1)).detach().abs().max()) < 1e-5), cfg
            cot_t = cot.permute(2, 0, 1).contiguous()
            ref2 = gg * v / v.flatten(1).norm(dim=1).view(-1, 1, 1)
            _cmp(dv.back(_grads(wt, dv(cot_t), dv([v, gg]))), _grads(ref2, cot, [v, gg]), cfg)


def test_pnca_backward_separate_query_gradients_are_summed_by_the_host(emulated_cabi, monkeypatch):
    """kantts_pnca_attn_bwd returns 1 (not 0) for sequences of more than 256 positions: the two bands' query gradients then
    come back in two buffers and ops._PncaAttention adds them.  The emulation always takes the summed form, so the
    separate form is provided here (per-band emulation, memory band into dqh) and must give the same gradients."""
    from kantts._hip import ops

    emu = emulated_cabi
    B, L, H = 2, 19, 8
    D = H * 16
    g = torch.Generator().manual_seed(3)
    qkv0 = torch.randn(B, L, 3 * D, generator=g)
    hkv0 = torch.randn(B, L, 2 * D, generator=g)
    lens = torch.tensor([19, 11], dtype=torch.int32)
    cot = torch.randn(B, L, D, generator=g)

    def run():
        qkv, hkv = qkv0.clone().requires_grad_(True), hkv0.clone().requires_grad_(True)
        ox, oh, _, _ = ops.pnca_attention(qkv, hkv, lens, 3, 2, H)
        ((ox * cot).sum() + (oh * cot.flip(0)).sum()).backward()
        return qkv.grad.clone(), hkv.grad.clone()

    want = run()

    def separate(qkv, hkv, ldh, ox, oh, d_ox, d_oh, lse_x, lse_h, dqkv, dqh, dhkv, lens_, bw_dev, bw_x, bw_h, B_, H_, L_,
                 d_head, drop_p, seed_x, seed_h, seed_dev, stream):
        qkv, hkv, dqkv, dhkv = int(qkv), int(hkv), int(dqkv), int(dhkv)
        emu.kantts_attn_bwd(qkv, qkv + 4 * D, qkv + 8 * D, 3 * D, 3 * D, 3 * D, ox, D, d_ox, D, lse_x, None, dqkv,
                            dqkv + 4 * D, dqkv + 8 * D, 3 * D, 3 * D, 3 * D, 0, lens_, bw_dev, bw_x, B_, H_, L_, d_head, 1,
                            drop_p, seed_x, seed_dev, stream)
        emu.kantts_attn_bwd(qkv, hkv, hkv + 4 * D, 3 * D, ldh, ldh, oh, D, d_oh, D, lse_h, None, dqh, dhkv, dhkv + 4 * D,
                            D, 2 * D, 2 * D, 0, lens_, bw_dev, bw_h, B_, H_, L_, d_head, 2, drop_p, seed_h, seed_dev, stream)
        return 1

    monkeypatch.setattr(emu, "kantts_pnca_attn_bwd", separate, raising=False)
    got = run()
    for a, b in zip(got, want):
        assert torch.allclose(a, b, rtol=0, atol=1e-6)


def test_fused_sambert_loss_equals_the_two_criteria(emulated_cabi):
    """ops.masked_l1_many (one launch: five masked-L1 terms, their sum, all gradients) against the two criteria called as
    modules (MelReconLoss / ProsodyReconLoss, each term its own launches): same components, same total, same gradients --
    for ragged lengths, one-frame utterances and a non-unit upstream gradient."""
    from kantts.train.loss import MelReconLoss, ProsodyReconLoss, sambert_loss_sum

    rnd = random.Random(21)
    g = torch.Generator().manual_seed(21)
    mel_c, pro_c = MelReconLoss(), ProsodyReconLoss()
    for it in range(6):
        B, T, N, C = rnd.choice([1, 2, 5]), rnd.randint(1, 70), rnd.randint(1, 23), rnd.choice([1, 80])
        ol = torch.tensor([rnd.randint(1, T) for _ in range(B)])
        il = torch.tensor([rnd.randint(1, N) for _ in range(B)])
        ol[rnd.randrange(B)], il[rnd.randrange(B)] = T, N
        batch = dict(output_lengths=ol, input_lengths=il, mel_targets=torch.randn(B, T, C, generator=g))
        leaves = dict(dec_outputs=torch.randn(B, T, C, generator=g), postnet_outputs=torch.randn(B, T, C, generator=g),
                      log_duration_predictions=torch.randn(B, N, generator=g),
                      pitch_predictions=torch.randn(B, N, generator=g), energy_predictions=torch.randn(B, N, generator=g))
        res = {k: v.requires_grad_(True) for k, v in leaves.items()}
        res.update(duration_targets=torch.randint(0, 40, (B, N), generator=g), pitch_targets=torch.randn(B, N, generator=g),
                   energy_targets=torch.randn(B, N, generator=g))
        plens = il if it % 2 == 0 else torch.clamp(il - 1, min=1)  # the trainer hands valid_inter_lengths in
        up = 1.0 if it < 3 else 0.37
        out = {}
        for fused in (True, False):
            if fused:
                os.environ.pop("KANTTS_NO_FUSED_LOSS", None)
            else:
                os.environ["KANTTS_NO_FUSED_LOSS"] = "1"
            try:
                total, comps = sambert_loss_sum(mel_c, pro_c, batch, res, prosody_lengths=plens)
            finally:
                os.environ.pop("KANTTS_NO_FUSED_LOSS", None)
            grads = torch.autograd.grad(total * up, list(leaves.values()))
            out[fused] = (total.detach(), comps, grads)
        cfg = dict(it=it, B=B, T=T, N=N, C=C, ol=ol.tolist(), il=plens.tolist())
        assert abs(float(out[True][0]) - float(out[False][0])) < 2e-5 * max(1.0, abs(float(out[False][0]))), cfg
        for k in out[False][1]:
            assert abs(float(out[True][1][k]) - float(out[False][1][k])) < 1e-5 * max(1.0, abs(float(out[False][1][k]))), (k, cfg)
        _cmp(out[True][2], out[False][2], cfg, tol=1e-7)


def test_mean_of_branch_outputs_in_one_launch(emulated_cabi):
    """ops.mean_many (kantts_mean_many / kantts_scale_to_many) against the ATen chain it replaces in HiFi-GAN's
    multi-receptive-field fusion (sum of the residual stacks / num_kernels): values, every input's gradient, and that each
    input receives its OWN gradient buffer."""
    from kantts._hip import ops

    g = torch.Generator().manual_seed(3)
    for n, shape in ((3, (2, 40, 32)), (2, (1, 8, 4)), (8, (3, 5, 12))):
        xs = [torch.randn(shape, generator=g).requires_grad_(True) for _ in range(n)]
        cot = torch.randn(shape, generator=g)
        y = ops.mean_many(xs)
        ref = sum(x.detach() for x in xs) / n
        assert float((y.detach() - ref).abs().max()) <= 1e-6
        grads = torch.autograd.grad(y, xs, cot)
        for gr in grads:
            assert float((gr - cot / n).abs().max()) <= 3e-7  # g * (1 / n) against g / n: one ulp
        assert len({gr.data_ptr() for gr in grads}) == n
    # bf16 mode: the same launch also leaves bf16(LeakyReLU(mean)) for the convolution that reads it next
    import kantts._hip as hip

    prev = hip.get_precision()
    hip.set_precision("bf16")
    try:
        for n, shape in ((3, (2, 40, 32)), (2, (1, 8, 4))):
            xs = [torch.randn(shape, generator=g).requires_grad_(True) for _ in range(n)]
            y = ops.mean_many(xs, image_slope=0.1)
            img = ops.get_image(y, 0.1)
            assert img is not None and img.dtype == torch.bfloat16
            want = torch.nn.functional.leaky_relu(y.detach(), 0.1)
            assert float((img.float() - want).abs().max()) <= 8e-3 * max(1.0, float(want.abs().max()))
            (gr,) = torch.autograd.grad(y, xs[:1], torch.ones(shape))
            assert float((gr - 1.0 / n).abs().max()) <= 3e-7
    finally:
        hip.set_precision(prev)


def _gan_criteria_fused_vs_per_term(device):
    """The three mean-reduced GAN criteria through ops.elem_loss_many (one launch each) against their per-term forms
    (KANTTS_NO_FUSED_GAN_LOSS=1): values and the gradient of every score / feature map, with channels-last feature maps
    handed out as permuted views (as the discriminators do) and a non-unit upstream gradient."""
    from kantts.train.loss import DiscriminatorAdversarialLoss, FeatureMatchLoss, GeneratorAdversarialLoss

    g = torch.Generator().manual_seed(17)

    def fmap(B, C, T):  # (B, T, C) buffer seen as (B, C, T)
        return torch.randn(B, T, C, generator=g).to(device).transpose(1, 2)

    shapes = [[(2, 8, 50), (2, 16, 17), (2, 1, 9)], [(2, 4, 33), (2, 1, 5)], [(2, 32, 7), (2, 8, 3), (2, 2, 11), (2, 1, 4)]]
    feats_hat = [[fmap(*sh).requires_grad_(True) for sh in d] for d in shapes]
    feats = [[fmap(*sh) for sh in d] for d in shapes]
    leaves = [a for d in feats_hat for a in d]
    out = {}
    for fused in (True, False):
        if not fused:
            os.environ["KANTTS_NO_FUSED_GAN_LOSS"] = "1"
        try:
            fm = FeatureMatchLoss()(feats_hat, feats)
            adv = GeneratorAdversarialLoss()(feats_hat)
            real, fake = DiscriminatorAdversarialLoss()(feats_hat, feats)
            total = 2.0 * fm + 0.7 * adv + 1.3 * fake + 0.0 * real
            grads = torch.autograd.grad(total, leaves, allow_unused=True)
        finally:
            os.environ.pop("KANTTS_NO_FUSED_GAN_LOSS", None)
        out[fused] = ([float(fm), float(adv), float(real), float(fake)], grads)
    for a, b in zip(out[True][0], out[False][0]):
        assert abs(a - b) <= 2e-6 * max(1.0, abs(b)), (out[True][0], out[False][0])
    for a, b in zip(out[True][1], out[False][1]):
        assert (a is None) == (b is None)
        if a is not None:
            assert a.shape == b.shape and float((a - b).abs().max()) <= 1e-6 * max(1e-6, float(b.abs().max())), a.shape


def test_gan_criteria_in_one_launch_each(emulated_cabi):
    _gan_criteria_fused_vs_per_term("cpu")


def test_element_losses_beyond_one_launch_and_empty_terms(emulated_cabi):
    """ops.elem_loss_many with more terms than one launch takes (64) and with empty tensors among them: the same sums and
    gradients as the single-term op."""
    from kantts._hip import ops

    g = torch.Generator().manual_seed(29)
    a_list = [torch.randn(3, 1 + k % 7, generator=g).requires_grad_(True) for k in range(70)]
    a_list[5] = torch.zeros(0, 4).requires_grad_(True)
    b_list = [torch.randn(a.shape, generator=g) for a in a_list]
    terms, ref = [], [0.0, 0.0]
    for k, (a, b) in enumerate(zip(a_list, b_list)):
        out = k % 2
        if k % 3 == 0:
            terms.append((a, None, 0.5, 1, 0.25, out))
            ref[out] = ref[out] + 0.25 * ((a - 0.5) ** 2).sum()
        else:
            terms.append((a, b, 0.0, 0, 0.5, out))
            ref[out] = ref[out] + 0.5 * (a - b).abs().sum()
    got = ops.elem_loss_many(terms, n_out=2)
    for o in range(2):
        assert abs(float(got[o]) - float(ref[o])) <= 2e-5 * max(1.0, abs(float(ref[o]))), o
    live = [a for a in a_list if a.numel()]
    gg = torch.autograd.grad(1.5 * got[0] - 0.5 * got[1], live)
    gr = torch.autograd.grad(1.5 * ref[0] - 0.5 * ref[1], live)
    _cmp(gg, gr, dict(case="70 terms"), tol=1e-6)
