"""Host-logic tests (no GPU): the product's Python layer (segment descriptors, strides, token shifts,
backward formulas, module wiring, optimizer arena) driven through the emulated C ABI
(oracle/cabi_numpy.py) and compared with the oracle / plain torch."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

import torch_oracle as O
from util import ROOT, assert_close, rel_l2


def test_cabi_library_exports_every_declared_symbol():
    """libkantts_hip.so loads and exports exactly what include/kantts_hip.h declares (no compute)."""
    import kantts._hip as hip

    if not hip.available():
        import __graft_entry__ as g

        g.build()
    header = open(os.path.join(ROOT, "include", "kantts_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(kantts_[a-z0-9_]+)\s*\(", header)))
    lib = ctypes.CDLL(hip.LIB_PATH)
    for sym in declared:
        assert hasattr(lib, sym), "missing export: " + sym
    assert sorted(hip.EXPORTED_SYMBOLS) == declared
    assert lib.kantts_abi_version() == 1
    lib.kantts_target_arch.restype = ctypes.c_char_p
    assert lib.kantts_target_arch() == b"gfx950"


def test_product_has_no_cpu_path():
    """Calling an op with host tensors must fail loudly (no silent eager fallback)."""
    from kantts._hip import ops

    with pytest.raises(RuntimeError):
        ops.layer_norm(torch.randn(4, 128), torch.ones(128), torch.zeros(128))


def test_tiny_sambert_forward_backward_matches_oracle(emulated_cabi):
    from kantts.models.sambert.kantts_sambert import KanTtsSAMBERT
    from kantts.train.loss import MelReconLoss, ProsodyReconLoss

    cfg = O.sambert_config(tiny=True)
    torch.manual_seed(0)
    m = KanTtsSAMBERT(dict(cfg))
    m.eval()
    m.return_attns = True
    batch = O.synthetic_sambert_batch(B=3, T_in=12, min_len=6, dur_hi=6)
    res = m(**batch)
    P = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    out = O.sambert_forward(P, cfg, **batch)
    assert torch.equal(res["LR_length_rounded"], out["LR_length_rounded"])
    assert res["x_band_width"] == out["x_band_width"]
    for k in ["dec_outputs", "postnet_outputs", "log_duration_predictions", "pitch_predictions",
              "energy_predictions", "LR_text_outputs", "LR_emo_outputs", "LR_spk_outputs", "ling_embedding"]:
        assert_close(res[k].detach(), out[k].detach(), 2e-5, what=k)
    for key in ["enc_slf_attn_lst", "pnca_x_attn_lst", "pnca_h_attn_lst"]:
        assert len(res[key]) == len(out[key])
        for a, b in zip(res[key], out[key]):
            assert_close(a, b.detach(), 1e-5, what=key)
    mel_, mel = MelReconLoss()(batch["output_lengths"], batch["mel_targets"], res["dec_outputs"], res["postnet_outputs"])
    d, p, e = ProsodyReconLoss()(batch["input_lengths"], res["duration_targets"], res["pitch_targets"],
                                 res["energy_targets"], res["log_duration_predictions"], res["pitch_predictions"],
                                 res["energy_predictions"])
    total = mel_ + mel + d + p + e
    total.backward()
    L = O.sambert_losses(out, batch["input_lengths"], batch["output_lengths"], batch["mel_targets"])
    assert abs(float(total) - float(L["total"])) < 1e-5
    L["total"].backward()
    for n, p_ in m.named_parameters():
        if p_.requires_grad:
            assert p_.grad is not None, n
            assert rel_l2(p_.grad, P[n].grad) < 1e-4, n


def test_dropout_backward_consistent_with_forward_mask(emulated_cabi):
    """Epilogue dropout (with and without ReLU) regenerates the same mask in backward."""
    from kantts._hip import ops

    torch.manual_seed(1)
    x = torch.randn(37, 24, requires_grad=True)
    w = torch.randn(19, 24, requires_grad=True)
    b = torch.randn(19, requires_grad=True)
    for relu in (False, True):
        y = ops.linear(x, w, b, relu=relu, drop_p=0.3)
        dense = torch.relu(x @ w.t() + b) if relu else (x @ w.t() + b)
        keep = (y != 0) if relu else torch.isclose(y, dense / 0.7, atol=1e-5)
        # keep rate 0.7 among the elements dropout can act on (for ReLU only the positive half is observable;
        # ~350-700 draws: +-4 sigma is about +-0.1)
        frac = (keep[dense > 0] if relu else keep).float().mean().item()
        assert 0.58 < frac < 0.82
        mask = torch.where(torch.isclose(y, dense / 0.7, atol=1e-5), 1 / 0.7, 0.0)
        gx, gw, gb = torch.autograd.grad(y.sum(), (x, w, b))
        ex, ew, eb = torch.autograd.grad((dense * mask).sum(), (x, w, b))
        assert_close(gx, ex, 1e-4, what="dx relu=%s" % relu)
        assert_close(gw, ew, 1e-4, what="dw relu=%s" % relu)
        assert_close(gb, eb, 1e-4, what="db relu=%s" % relu)


def test_dropout2_add_and_fsmn_encoder_training_path(emulated_cabi):
    """kantts_dropout2_add: two stacked dropouts (+ residual) in one pass; backward regenerates both masks.  The FSMN
    encoder in training mode uses it for the block dropout, the encoder dropout and "memory + x" (fsmn.py:66-70,114-121)."""
    from kantts._hip import ops
    from kantts.models.sambert.fsmn import FsmnEncoderV2

    torch.manual_seed(2)
    x = torch.randn(5, 33, 64, requires_grad=True)
    r = torch.randn(5, 33, 64, requires_grad=True)
    y = ops.dropout2_add(x, 0.2, 0.3, r)
    d = (y - r).detach()                             # 0 or x / (0.8 * 0.7)
    keep = torch.isclose(d, x.detach() / 0.56, rtol=1e-4, atol=1e-5)
    assert torch.all(keep | (d.abs() < 1e-5))
    assert 0.50 < keep.float().mean().item() < 0.62  # 0.8 * 0.7 = 0.56 over 10 560 draws
    k = keep.float() / 0.56
    gx, gr = torch.autograd.grad((y * y).sum(), (x, r))
    assert_close(gx, (2 * y * k).detach(), 1e-4, what="dx")
    assert_close(gr, (2 * y).detach(), 1e-5, what="dres")
    assert torch.equal(ops.dropout2_add(x, 0.0, 0.0, r), x + r)
    # encoder wiring: eval equals the fused no-dropout path; training runs, keeps shapes, back-propagates
    enc = FsmnEncoderV2(filter_size=11, fsmn_num_layers=3, input_dim=48, num_memory_units=32, ffn_inner_dim=64, dropout=0.1)
    inp = torch.randn(2, 20, 48, requires_grad=True)
    mask = torch.arange(20)[None, :] >= torch.tensor([20, 13])[:, None]
    enc.eval()
    y_eval = enc(inp, mask)
    enc.train()
    y_tr = enc(inp, mask)
    assert y_tr.shape == y_eval.shape and not torch.allclose(y_tr, y_eval)
    y_tr.sum().backward()
    assert inp.grad is not None and torch.isfinite(inp.grad).all()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in enc.parameters())


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_row_mask_handed_to_the_consuming_layernorm_changes_no_gradient(emulated_cabi, monkeypatch, precision):
    """ops_bf16.RowMaskToken: inside a block stack the masking of a sub-layer's incoming gradient is done by the store of
    the next LayerNorm's backward kernel (kantts_ln128_bwd_rows) instead of a masked_fill pass per sub-layer.  Same
    gradients bit for bit with the hand-over switched off, and the elementwise passes are really gone."""
    import kantts._hip as hip
    from kantts._hip import ops_bf16
    from kantts.models.sambert.kantts_sambert import SelfAttentionEncoder

    hip.set_precision(precision)
    try:
        def run(handover):
            if not handover:
                monkeypatch.setattr(ops_bf16, "take_rowmask", lambda x: None)
            torch.manual_seed(5)
            enc = SelfAttentionEncoder(3, 128, 128, 8, 16, 1024, 0.0, 0.0, 0.0, position_encoder=None)
            enc.train()
            x = torch.randn(3, 21, 128, requires_grad=True)
            mask = torch.arange(21)[None, :] >= torch.tensor([21, 9, 15])[:, None]
            calls = []
            orig = torch.Tensor.masked_fill
            monkeypatch.setattr(torch.Tensor, "masked_fill", lambda self, *a, **k: (calls.append(tuple(self.shape)), orig(self, *a, **k))[1])
            y, _ = enc(x, mask, prescaled=True)
            (y * torch.randn_like(y)).sum().backward()
            monkeypatch.setattr(torch.Tensor, "masked_fill", orig)
            return [x.grad] + [p.grad for p in enc.parameters()], [c for c in calls if c == (63, 128)]

        g_on, fills_on = run(True)
        g_off, fills_off = run(False)
        assert len(fills_off) == 6 and len(fills_on) == 0  # 3 blocks x 2 sub-layers
        for a, b in zip(g_on, g_off):
            assert torch.equal(a, b)
        assert torch.all(g_on[0][1, 9:] == 0) and torch.all(g_on[0][2, 15:] == 0)
    finally:
        hip.set_precision("fp32")


def _tiny_training_run(monkeypatch, mutate):
    """One teacher-forced training-mode forward + backward of the tiny SAM-BERT through the emulated ABI; every dropout
    draws the SAME seed (masks then depend on the element index only, not on the order the launches are issued in)."""
    import kantts.models.sambert.kantts_sambert as KS
    from kantts._hip import ops
    from kantts.train.loss import MelReconLoss, ProsodyReconLoss

    monkeypatch.setattr(ops, "next_seed", lambda: 12345)
    cfg = O.sambert_config(tiny=True)
    torch.manual_seed(0)
    m = KS.KanTtsSAMBERT(dict(cfg))
    m.train()
    mutate(m, KS)
    batch = O.synthetic_sambert_batch(B=3, T_in=12, min_len=6, dur_hi=6)
    res = m(**batch)
    mel_, mel = MelReconLoss()(batch["output_lengths"], batch["mel_targets"], res["dec_outputs"], res["postnet_outputs"])
    d, p, e = ProsodyReconLoss()(batch["input_lengths"], res["duration_targets"], res["pitch_targets"],
                                 res["energy_targets"], res["log_duration_predictions"], res["pitch_predictions"],
                                 res["energy_predictions"])
    (mel_ + mel + d + p + e).backward()
    outs = [res[k].detach().clone() for k in ("dec_outputs", "postnet_outputs", "log_duration_predictions",
                                              "pitch_predictions", "energy_predictions")]
    return outs, {n: q.grad.clone() for n, q in m.named_parameters() if q.grad is not None}


def test_teacher_plan_beside_the_encoder_is_the_inline_arithmetic(emulated_cabi, monkeypatch):
    """KanTtsSAMBERT.teacher_forced_plan / _beside_encoder (masks, length-regulator index, duration positions, band width,
    teacher-forcing frames, decoder prenet -- computed ahead of the encoder on a second stream on the device) against the
    same quantities computed where the reference computes them: identical outputs and gradients, dropout on."""
    a_out, a_g = _tiny_training_run(monkeypatch, lambda m, KS: None)
    b_out, b_g = _tiny_training_run(monkeypatch, lambda m, KS: setattr(m, "inline_teacher_plan", True))
    for x, y in zip(a_out, b_out):
        assert torch.equal(x, y)
    assert a_g.keys() == b_g.keys()
    for n in a_g:
        assert torch.equal(a_g[n], b_g[n]), n


def test_variance_predictors_beside_the_postnet_or_the_decoder_same_results(emulated_cabi, monkeypatch):
    """The teacher-forced variance predictors only feed their own losses: started beside the postnet (default) or beside
    the decoder (round 2, KANTTS_PREDICTORS_EARLY) they must give the same predictions and gradients."""
    a_out, a_g = _tiny_training_run(monkeypatch, lambda m, KS: monkeypatch.setattr(KS, "_PREDICTORS_LATE", True))
    b_out, b_g = _tiny_training_run(monkeypatch, lambda m, KS: monkeypatch.setattr(KS, "_PREDICTORS_LATE", False))
    for x, y in zip(a_out, b_out):
        assert torch.equal(x, y)
    # since round 6 the late branch forms its two concatenations itself (two launches fewer on the main chain), so the
    # gradient contributions of the embeddings they read are summed in another order: equal to fp32 rounding, bit-equal
    # when the concatenations stay on the main chain (KANTTS_CATS_EARLY)
    for n in a_g:
        assert rel_l2(a_g[n], b_g[n]) <= 1e-6, (n, rel_l2(a_g[n], b_g[n]))
    c_out, c_g = _tiny_training_run(monkeypatch, lambda m, KS: (monkeypatch.setattr(KS, "_PREDICTORS_LATE", True),
                                                                monkeypatch.setattr(KS, "_CATS_EARLY", True)))
    for x, y in zip(c_out, b_out):
        assert torch.equal(x, y)
    for n in c_g:
        assert torch.equal(c_g[n], b_g[n]), n


def test_arena_adam_matches_torch_adam(emulated_cabi):
    from kantts.train.optim import ArenaAdam, ParamArena

    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    ref = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    ref.load_state_dict(net.state_dict())
    arena = ParamArena(net)
    opt = ArenaAdam(arena, lr=1e-2, betas=(0.9, 0.98), eps=1e-9)
    opt.set_grad_clip(0.05)
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-2, betas=(0.9, 0.98), eps=1e-9)
    for step in range(4):
        x = torch.randn(6, 7)
        for mdl, o in ((net, opt), (ref, ropt)):
            o.zero_grad()
            mdl(x).pow(2).sum().backward()
            if mdl is ref:
                torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.05)
            o.step()
    for a, b in zip(net.parameters(), ref.parameters()):
        assert_close(a.detach(), b.detach(), 1e-6, what="param")
    sd = opt.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}
    assert float(sd["state"][0]["step"]) == 4.0


def test_arena_adam_with_lr_and_step_count_in_device_memory(emulated_cabi):
    """The form a captured step replays (ArenaAdam.enable_device_state: kantts_adam_step reads {lr, step} from a device
    buffer, sync_lr() after the scheduler): same weights as the host-argument form through a changing learning rate."""
    from kantts.train.optim import ArenaAdam, ParamArena

    torch.manual_seed(1)
    nets = [torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3)) for _ in range(2)]
    nets[1].load_state_dict(nets[0].state_dict())
    opts = [ArenaAdam(ParamArena(n), lr=1e-2, betas=(0.9, 0.98), eps=1e-9) for n in nets]
    for o in opts:
        o.set_grad_clip(0.05)
    dyn = opts[1].enable_device_state()
    assert dyn.shape == (2,) and float(dyn[1]) == 0.0
    for step in range(4):
        x = torch.randn(6, 7)
        lr = 1e-2 / (1 + step)
        for n, o in zip(nets, opts):
            o.param_groups[0]["lr"] = lr
            o.sync_lr()
            o.zero_grad()
            n(x).pow(2).sum().backward()
            o.step()
    assert float(dyn[1]) == 4.0 and abs(float(dyn[0]) - 1e-2 / 4) < 1e-9
    for a, b in zip(nets[0].parameters(), nets[1].parameters()):
        assert_close(a.detach(), b.detach(), 1e-7, what="param")
    assert opts[1].enable_device_state() is dyn  # one buffer for the optimizer's lifetime (captured graphs hold its address)


def test_oracle_matches_live_reference_when_present():
    """In the build container the restatement is also compared with the live reference (own process:
    the reference package is also called ``kantts``)."""
    if not os.path.isdir("/root/reference/kantts"):
        pytest.skip("reference tree not present (GPU box)")
    r = subprocess.run(["python", os.path.join(ROOT, "oracle", "check_vs_reference.py")], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def _infer_case(device, B, seed):
    """Product free-running inference vs the oracle's (same weights, duration-head bias 1.5 as in the golden)."""
    from kantts.models.sambert.kantts_sambert import KanTtsSAMBERT

    cfg = O.sambert_config(tiny=True)
    torch.manual_seed(0)
    m = KanTtsSAMBERT(dict(cfg))
    with torch.no_grad():
        m.variance_adaptor.duration_predictor.fc.bias.fill_(1.5)
    P = {k: v.detach().clone() for k, v in m.state_dict().items()}
    batch = O.synthetic_sambert_batch(B=B, T_in=12, seed=seed, min_len=6, dur_hi=6)
    args = {k: batch[k] for k in ("inputs_ling", "inputs_emotion", "inputs_speaker", "input_lengths")}
    m = m.to(device).eval()
    with torch.no_grad():
        res = m(**{k: v.to(device) for k, v in args.items()})
    # the reference (and therefore the oracle) infers one utterance at a time: compare per utterance, and the
    # batched product run must equal its own per-utterance results on the valid frames
    for b in range(B):
        one = {k: v[b:b + 1, : int(args["input_lengths"][b])] if v.dim() > 1 else v[b:b + 1] for k, v in args.items()}
        with torch.no_grad():
            out = O.sambert_forward(P, cfg, **one)
        n = int(out["LR_length_rounded"][0])
        assert int(res["LR_length_rounded"][b]) == n, (b, int(res["LR_length_rounded"][b]), n)
        T = int(args["input_lengths"][b])
        assert_close(res["log_duration_predictions"][b, :T].cpu(), out["log_duration_predictions"][0, :T], 2e-5,
                     what="log dur")
        assert_close(res["postnet_outputs"][b, :n].cpu(), out["postnet_outputs"][0, :n], 1e-4, what="mel %d" % b)
        assert_close(res["dec_outputs"][b, :n].cpu(), out["dec_outputs"][0, :n], 1e-4, what="dec %d" % b)
    return res


def test_free_running_inference_matches_oracle(emulated_cabi):
    _infer_case("cpu", B=1, seed=77)
    _infer_case("cpu", B=3, seed=5)


def test_ctypes_prototypes_match_the_header():
    """Every entry point with declared argtypes: argument count and pointer / int / float / 64-bit kinds agree with the
    C prototype in include/kantts_hip.h (a drifted binding would corrupt the call frame silently)."""
    import kantts._hip as hip

    if not hip.available():
        import __graft_entry__ as g

        g.build()
    L = hip.lib()
    header = open(os.path.join(ROOT, "include", "kantts_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    protos = dict(re.findall(r"\bint\s+(kantts_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", header, flags=re.S))
    kinds = {ctypes.c_void_p: "p", ctypes.c_int: "i", ctypes.c_float: "f", ctypes.c_longlong: "q", ctypes.c_uint64: "q",
             ctypes.c_char_p: "p"}

    def kind_of_c(param):
        param = " ".join(param.split())
        if "*" in param:
            return "p"
        if re.search(r"\b(long long|int64_t|uint64_t|size_t)\b", param):
            return "q"
        if re.search(r"\b(float)\b", param):
            return "f"
        if re.search(r"\b(int|int32_t|unsigned)\b", param):
            return "i"
        raise AssertionError("unclassified C parameter: " + param)

    checked = 0
    for name, params in protos.items():
        fn = getattr(L, name)
        if fn.argtypes is None:
            continue
        want = [kind_of_c(p) for p in params.split(",") if p.strip() and p.strip() != "void"]
        got = ["p" if (isinstance(t, type) and issubclass(t, ctypes._Pointer)) else kinds[t] for t in fn.argtypes]
        assert got == want, (name, "".join(got), "".join(want))
        checked += 1
    assert checked >= 30, checked


def test_ctypes_struct_layouts_match_the_header(tmp_path):
    """The argument structs are filled field by field from Python: size and every field offset must be what a C
    compiler derives from include/kantts_hip.h (the header is plain C, so gcc is the judge)."""
    import kantts._hip as hip

    pairs = [(hip.GemmSeg, "kantts_gemm_seg"), (hip.GemmArgs, "kantts_gemm_args"), (hip.ConvArgs, "kantts_conv_args"),
             (hip.ConvWArgs, "kantts_convw_args"), (hip.ConvC1Args, "kantts_conv_c1_args"),
             (hip.BGemmSeg, "kantts_bgemm_seg"), (hip.BGemmArgs, "kantts_bgemm_args"),
             (hip.BGemmTnArgs, "kantts_bgemm_tn_args"), (hip.TapMajorDesc, "kantts_tapmajor_desc"),
             (hip.FfnArgs, "kantts_ffn_args"), (hip.FragMajorDesc, "kantts_fragmajor_desc"),
             (hip.CConvArgs, "kantts_cconv_args"), (hip.CConvWArgs, "kantts_cconvw_args"),
             (hip.LnBwdArgs, "kantts_lnbwd_args")]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "kantts_hip.h"', 'int main(void) {']
    for cls, cname in pairs:
        lines.append('  printf("%s sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            cfield = fname if fname == "pad_" else fname.rstrip("_")
            lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, cfield))
    lines += ['  return 0;', '}']
    src, exe = tmp_path / "layout.c", tmp_path / "layout"
    src.write_text("\n".join(lines))
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    c_layout = {}
    for ln in out.splitlines():
        cname, field, val = ln.split()
        c_layout[(cname, field)] = int(val)
    for cls, cname in pairs:
        assert ctypes.sizeof(cls) == c_layout[(cname, "sizeof")], (cname, ctypes.sizeof(cls), c_layout[(cname, "sizeof")])
        for fname, _ in cls._fields_:
            assert getattr(cls, fname).offset == c_layout[(cname, fname)], (cname, fname)


def test_layernorm_backward_in_the_consumers_input_gradient_launch(emulated_cabi, monkeypatch):
    """ops_bf16.LnBwdToken (on by default since round 4; KANTTS_NO_LN_BWD_EPILOGUE switches it off): the backward of the
    LayerNorm in front of every attention sub-layer is the epilogue of the QKV projection's input-gradient launch
    (kantts_bgemm_nt_lnbwd); the LayerNorm in front of a feed-forward sub-layer keeps its own launch (the fused form of the
    feed-forward pair was slower on the device and was removed).  Same gradients as with the separate
    kantts_ln128_bwd_rows launches (same bf16-rounded dy on both sides; the model of the ABI evaluates both with the same
    formulas), one launch less per attention sub-layer, and a second consumer of the normalised rows is refused loudly
    instead of silently dropping its gradient."""
    import kantts._hip as hip
    from kantts._hip import ops, ops_bf16
    from kantts.models.sambert.kantts_sambert import SelfAttentionEncoder

    hip.set_precision("bf16")
    try:
        def run(on):
            monkeypatch.setitem(ops_bf16.LNBWD, "on", on)
            counts = {"kantts_ln128_bwd_rows": 0, "kantts_bgemm_nt_lnbwd": 0}
            local = pytest.MonkeyPatch()
            for name in counts:
                orig = getattr(emulated_cabi, name)
                local.setattr(emulated_cabi, name, (lambda o, n: lambda *a: (counts.__setitem__(n, counts[n] + 1), o(*a))[1])(orig, name),
                              raising=False)
            torch.manual_seed(5)
            enc = SelfAttentionEncoder(3, 128, 128, 8, 16, 1024, 0.0, 0.0, 0.0, position_encoder=None)
            enc.train()
            x = torch.randn(3, 21, 128, requires_grad=True)
            mask = torch.arange(21)[None, :] >= torch.tensor([21, 9, 15])[:, None]
            y, _ = enc(x, mask, prescaled=True)
            try:
                (y * torch.randn(y.shape, generator=torch.Generator().manual_seed(1))).sum().backward()
            finally:
                local.undo()
            return [x.grad] + [p.grad for p in enc.parameters()], counts

        g_off, c_off = run(False)
        g_on, c_on = run(True)
        assert c_off["kantts_bgemm_nt_lnbwd"] == 0
        assert c_on["kantts_bgemm_nt_lnbwd"] == 3  # one per block
        assert c_on["kantts_ln128_bwd_rows"] == c_off["kantts_ln128_bwd_rows"] - 3
        for a, b in zip(g_on, g_off):
            assert rel_l2(a, b) <= 1e-6, rel_l2(a, b)
        assert torch.all(g_on[0][1, 9:] == 0) and torch.all(g_on[0][2, 15:] == 0)

        # a second consumer of the normalised rows: autograd then delivers a sum the fused launch has not seen
        monkeypatch.setitem(ops_bf16.LNBWD, "on", True)
        x = torch.randn(4, 128, requires_grad=True)
        gam, bet = torch.ones(128, requires_grad=True), torch.zeros(128, requires_grad=True)
        w = torch.randn(384, 128, requires_grad=True) * 0.1
        xn = ops.layer_norm(x, gam, bet, 1e-6, out_bf16=True)
        q = ops.linear(xn, w)
        with pytest.raises(RuntimeError, match="second consumer"):
            (q.sum() + xn.float().sum()).backward()
    finally:
        hip.set_precision("fp32")


def test_relu_gate_of_a_hidden_gradient_in_its_producers_epilogue(emulated_cabi, monkeypatch):
    """ops_bf16.ReluGateToken (on by default since round 4; KANTTS_NO_RELU_GATE_EPILOGUE switches it off): the gradient of the bf16 hidden activation of an
    FSMN feed-forward net is gated (ReLU, dropout) and scaled by the epilogue of the launch that computes it -- the second
    contraction's input gradient -- instead of a kantts_relu_gate_bf16 pass over it.  Same gradients up to the one bf16
    rounding the pass-through saved; the passes are gone; a second consumer of the hidden activation is refused."""
    import kantts._hip as hip
    from kantts._hip import ops, ops_bf16
    from kantts.models.sambert.fsmn import FsmnEncoderV2

    hip.set_precision("bf16")
    try:
        def run(on):
            monkeypatch.setitem(ops_bf16.RELUGATE, "on", on)
            monkeypatch.setattr(ops, "next_seed", lambda: 4242)
            calls = []
            local = pytest.MonkeyPatch()
            orig = emulated_cabi.kantts_relu_gate_bf16
            local.setattr(emulated_cabi, "kantts_relu_gate_bf16", lambda *a: (calls.append(1), orig(*a))[1], raising=False)
            torch.manual_seed(3)
            enc = FsmnEncoderV2(11, 3, 256, 256, 512, dropout=0.1, shift=2)
            enc.train()
            x = torch.randn(2, 37, 256, requires_grad=True)
            mask = torch.arange(37)[None, :] >= torch.tensor([37, 20])[:, None]
            try:
                y = enc(x, mask)
                (y * torch.randn(y.shape, generator=torch.Generator().manual_seed(1))).sum().backward()
            finally:
                local.undo()
            return [x.grad] + [p.grad for p in enc.parameters()], len(calls)

        g_off, n_off = run(False)
        g_on, n_on = run(True)
        assert n_off == 3 and n_on == 0
        for a, b in zip(g_on, g_off):
            assert rel_l2(a, b) <= 6e-3, rel_l2(a, b)   # one bf16 rounding of the hidden gradient instead of two

        monkeypatch.setitem(ops_bf16.RELUGATE, "on", True)
        x = torch.randn(8, 64, requires_grad=True)
        w1, w2 = (torch.randn(32, 64) * 0.1).requires_grad_(True), (torch.randn(16, 32) * 0.1).requires_grad_(True)
        h = ops.linear(x, w1, None, relu=True, out_bf16=True)
        out = ops.linear(h, w2, None)
        with pytest.raises(RuntimeError, match="second consumer"):
            (out.sum() + h.float().sum()).backward()
    finally:
        hip.set_precision("fp32")


def test_batched_free_running_inference_equals_per_utterance_inference(emulated_cabi):
    """bench.config5_parity on the tiny model through the emulated ABI: length-sorted batches of free-running inference
    against the oracle one utterance at a time.  Sequences whose regulated length is not a multiple of r have r-padding
    frames INSIDE the batch's frame range: their duration-position term must be the padding's (position 0), as in
    per-utterance inference (found at the full configuration on the device: tests/test_config5_inference.py)."""
    import bench
    from kantts.models.sambert.kantts_sambert import KanTtsSAMBERT
    from kantts.utils.synthetic import inference_utterances

    cfg = O.sambert_config(tiny=True)
    torch.manual_seed(0)
    am = KanTtsSAMBERT(dict(cfg))
    with torch.no_grad():  # ~4 frames per symbol (random-init weights predict the bias): lengths 4 * T_in, mostly not
        am.variance_adaptor.duration_predictor.fc.bias.fill_(1.6)  # multiples of r = 3
    am = am.eval()
    utts = inference_utterances(12, seed=99)
    rep = bench.config5_parity(am, cfg, utts, torch.arange(12), batch=4, threads=4)
    assert rep["frame_count_agreement"] == 1.0 and rep["duration_agreement"] == 1.0, rep
    assert rep["mel_max_abs_forced_durations"] <= 2e-5, rep
    assert rep["mel_mean_abs_free_running_where_durations_agree"] <= 1e-5, rep


def test_launch_tuning_is_fed_from_the_environment_by_the_host_layer(emulated_cabi, monkeypatch):
    """KANTTS_TN_TILE / KANTTS_TN_SLICES / KANTTS_C1_WGRAD_WGS are read by kantts._hip (not by the library) and handed over
    through kantts_launch_tuning -- once per change, again when the variables go away."""
    import kantts._hip as hip

    calls = []
    monkeypatch.setattr(type(emulated_cabi), "kantts_launch_tuning",
                        lambda self, a, b, c: calls.append((a, b, c)) or 0, raising=False)
    monkeypatch.setattr(hip, "_launch_tuning", [None])
    for k in ("KANTTS_TN_TILE", "KANTTS_TN_SLICES", "KANTTS_C1_WGRAD_WGS"):
        monkeypatch.delenv(k, raising=False)
    hip.apply_launch_tuning()
    hip.apply_launch_tuning()
    assert calls == [(0, 0, 0)]
    monkeypatch.setenv("KANTTS_TN_TILE", "128256")
    monkeypatch.setenv("KANTTS_C1_WGRAD_WGS", "128")
    hip.apply_launch_tuning()
    hip.apply_launch_tuning()
    assert calls[-1] == (128256, 0, 128) and len(calls) == 2
    monkeypatch.delenv("KANTTS_TN_TILE")
    hip.apply_launch_tuning()
    assert calls[-1] == (0, 0, 128) and len(calls) == 3


def test_adoption_queue_refuses_what_the_launch_did_not_promise():
    """ops_bf16.ADOPT, the hand-over between a fused launch (pnca_block.hip, enc_attn.hip) and the unchanged autograd
    Functions of the block: entries are taken in the order the launch promised them; a Function asking out of order, or a
    result nobody asked for when the block ends, raises instead of silently running with another launch's tensors; scopes
    nest and restore (a fused encoder sub-layer inside an outer scope), also when the body raises."""
    from kantts._hip import ops_bf16
    A = ops_bf16.ADOPT
    assert A.q is None and A.take("linear") is None  # no launch in flight: every Function runs its own kernel
    with ops_bf16._adopting([("linear", 1), ("attn", 2)]):
        assert A.take("linear") == 1
        with ops_bf16._adopting([("ffn", 3)]):
            assert A.take("ffn") == 3
        assert A.q == [("attn", 2)]
        assert A.take("attn") == 2
    assert A.q is None
    with pytest.raises(RuntimeError, match="another order"):
        with ops_bf16._adopting([("linear", 1), ("attn", 2)]):
            A.take("attn")
    assert A.q is None
    with pytest.raises(RuntimeError, match="never adopted"):
        with ops_bf16._adopting([("linear", 1), ("attn", 2)]):
            A.take("linear")
    assert A.q is None
    with pytest.raises(ValueError):
        with ops_bf16._adopting([("linear", 1)]):
            raise ValueError("the block's own error is not masked by the left-over entry")
    assert A.q is None
