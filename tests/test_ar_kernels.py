"""Free-running inference loops as one launch each (csrc/ar_infer.hip, kantts/models/sambert/ar_kernels.py): the mel
decoder's loop (reference kantts/models/sambert/kantts_sambert.py:569-610, :208-253) and the duration predictor's
(adaptors.py:67-83) against the per-launch paths they replace, which are pinned to the reference's own inference run
(tests/test_decode_graph.py, tests/golden/sambert_tiny_infer.pt).  bf16 mode: both sides round the same contraction
operands to bf16 and differ in summation order only -- and in the occasional bf16 tie that this flips."""
import pytest
import torch

import torch_oracle as O
from util import emulation, kernel_source_on_cpu, rel_l2


def _model(device, layers=2):
    from kantts.models.sambert.kantts_sambert import KanTtsSAMBERT

    cfg = O.sambert_config(tiny=True)
    cfg["decoder_num_layers"] = layers
    torch.manual_seed(0)
    m = KanTtsSAMBERT(dict(cfg))
    with torch.no_grad():  # random-init LayerNorm / bias parameters are 1 / 0: make every term of the blobs matter
        for n, p in m.named_parameters():
            if n.endswith("bias") or "layer_norm" in n or n.endswith("ln.weight"):
                p.add_(0.1 * torch.randn_like(p))
    return m.to(device).eval()


def _decode(m, device, mode, B, L, lens, bws, seed=3):
    from kantts.models.utils import get_mask_from_lengths

    g = torch.Generator().manual_seed(seed)
    d_mem = m.mel_decoder.mel_dec.pnca[0].pnca_attn.d_mem
    memory = (0.7 * torch.randn(B, L, d_mem, generator=g)).to(device)
    lens_t = torch.tensor(lens, device=device)
    bw_seq = torch.tensor(bws, device=device, dtype=torch.int32)
    m.mel_decoder.decode_mode = mode
    bw = int(max(bws))
    with torch.no_grad():
        out, _, _ = m.mel_decoder(memory, bw, bw, mask=get_mask_from_lengths(lens_t, L), bw_dev=bw_seq)
    return out.detach().cpu()


def _check_decoder(device, layers, B, L, lens, bws, ref_mode="fused"):
    import kantts._hip as hip

    hip.set_precision("bf16")
    try:
        m = _model(device, layers)
        ref = _decode(m, device, ref_mode, B, L, lens, bws)
        m.mel_decoder._decode_kernel = None
        got = _decode(m, device, "kernel", B, L, lens, bws)
        assert m.mel_decoder._decode_kernel is not None, "the one-launch decoder did not run"
    finally:
        hip.set_precision("fp32")
    assert got.shape == ref.shape and torch.isfinite(got).all()
    scale = float(ref.abs().max())
    for b, n in enumerate(lens):  # the frames of a sequence, then the reference's masked rows behind them
        assert rel_l2(got[b, :n], ref[b, :n]) < 3e-3, (b, rel_l2(got[b, :n], ref[b, :n]))
        assert float((got[b] - ref[b]).abs().max()) < 3e-2 * scale, (b, float((got[b] - ref[b]).abs().max()), scale)


def test_decoder_loop_as_one_launch_emulated():
    """numpy model of kantts_pnca_decode_run against the per-launch step function on the emulated C ABI: ragged lengths, a
    band wider than a sequence, band 0."""
    with emulation():
        _check_decoder("cpu", 2, 3, 9, [9, 5, 1], [2, 7, 0])


def test_decoder_loop_as_one_launch_kernel_source():
    """The kernel SOURCE on the CPU (tests/hipemu) against its own per-launch twins; the second case has a band of more
    than 15 rows (the attentions then read the caches in place instead of the rows parked in LDS)."""
    with kernel_source_on_cpu():
        _check_decoder("cpu", 2, 2, 6, [6, 3], [2, 4])
        _check_decoder("cpu", 1, 2, 24, [24, 19], [20, 17])


def _raw_decode(lib_ctx, d_mel, d_mem, d_out, n_layer, B, L, lens, bws, seed):
    """kantts_pnca_decode_run called directly with random blobs (shapes no shipped configuration has)."""
    import kantts._hip as hip

    g = torch.Generator().manual_seed(seed)
    with lib_ctx():
        nw, nf = hip.decode_blob_sizes(d_mel, d_mem, d_out, n_layer)
        w = (0.08 * torch.randn(nw, generator=g)).to(torch.bfloat16)
        f = 0.1 * torch.randn(nf, generator=g)
        memory = 0.7 * torch.randn(B, L, d_mem, generator=g)
        hkv = 0.7 * torch.randn(B, L, n_layer * 256, generator=g)
        xkv = torch.zeros(n_layer, B, L, 256)
        out = torch.full((B, L, d_out), float("nan"))
        hip.pnca_decode_run(w, f, memory, hkv, xkv, out, torch.tensor(lens, dtype=torch.int32),
                            torch.tensor(bws, dtype=torch.int32), 0, d_mel, n_layer, 128 ** 0.5, 1e-6)
    return out


@pytest.mark.parametrize("d_mel,d_mem,d_out", [(80, 96, 240), (40, 300, 100), (128, 160, 130)])
def test_decoder_kernel_source_equals_its_numpy_model_on_other_shapes(d_mel, d_mem, d_out):
    """The entry-projection widths 256 / 384 / 512 (three instantiations of the product), an output width that is no
    multiple of 16, a frame narrower / as wide as the model: kernel source against the numpy model, same random blobs.
    (The weights here are NOT LayerNorm-friendly model weights; both sides round the same values, so the comparison is
    tight anyway.)"""
    args = (d_mel, d_mem, d_out, 2, 2, 5, [5, 3], [2, 1], 11)
    ref = _raw_decode(emulation, *args)
    got = _raw_decode(kernel_source_on_cpu, *args)
    assert torch.isfinite(got).all() and torch.isfinite(ref).all()
    assert rel_l2(got, ref) < 2e-3, rel_l2(got, ref)


@pytest.mark.parametrize("ctx", ["emulated", "kernel_source"])
def test_decoder_entry_refuses_and_poisons(ctx):
    """What kantts_pnca_decode_run is not compiled for: a host-known band above 127 and an entry projection wider than 512
    are refused with KANTTS_E_UNSUPPORTED (-2); a DEVICE-side band width above 127 cannot be refused at launch -- that
    sequence's output is NaN (never a wrong answer), the others are unaffected; empty batches are a no-op."""
    import ctypes

    import kantts._hip as hip

    lib_ctx = emulation if ctx == "emulated" else kernel_source_on_cpu
    ok = _raw_decode(lib_ctx, 80, 160, 240, 1, 2, 4, [4, 4], [1, 1], 3)
    got = _raw_decode(lib_ctx, 80, 160, 240, 1, 2, 4, [4, 4], [1, 200], 3)
    assert torch.isnan(got[1]).all() and torch.equal(got[0], ok[0])
    with lib_ctx():
        assert hip.decode_blob_sizes(80, 400, 240, 1) is None
        g = hip.DecodeArgs()
        z = torch.zeros(16)
        for name in ("w", "f", "memory", "hkv", "xkv", "out"):
            setattr(g, name, z.data_ptr())
        g.B, g.L, g.d_mem, g.d_mel, g.d_out, g.n_layer, g.bw = 1, 1, 160, 80, 240, 1, 128
        assert hip.lib().kantts_pnca_decode_run(ctypes.byref(g), None) == hip.E_UNSUPPORTED
        g.bw, g.d_mem = 5, 400
        assert hip.lib().kantts_pnca_decode_run(ctypes.byref(g), None) == hip.E_UNSUPPORTED
        g.d_mem, g.B = 160, 0
        assert hip.lib().kantts_pnca_decode_run(ctypes.byref(g), None) == 0
        d = hip.DurArArgs()
        for name in ("w", "f", "gc", "out"):
            setattr(d, name, z.data_ptr())
        d.B, d.T = 0, 5
        assert hip.lib().kantts_dur_ar_run(ctypes.byref(d), None) == 0
        assert hip.lib().kantts_dur_ar_run_f32(ctypes.byref(d), None) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("layers,B,L,lens,bws", [(2, 3, 9, [9, 5, 1], [2, 7, 0]), (12, 4, 70, [70, 33, 64, 8], [6, 3, 40, 2])])
def test_decoder_loop_as_one_launch_gpu(layers, B, L, lens, bws):
    _check_decoder("cuda", layers, B, L, lens, bws, ref_mode="graph")


@pytest.mark.gpu
def test_decoder_loop_refuses_what_it_is_not_compiled_for_gpu():
    """fp32 mode and band widths above 127 keep the replayed graph (the kernel object is never created)."""
    import kantts._hip as hip

    hip.set_precision("fp32")
    m = _model("cuda")
    _decode(m, "cuda", "kernel", 2, 6, [6, 4], [2, 2])
    assert m.mel_decoder._decode_kernel is None
    hip.set_precision("bf16")
    try:
        _decode(m, "cuda", "kernel", 1, 140, [140], [130])
        assert m.mel_decoder._decode_kernel is None
    finally:
        hip.set_precision("fp32")


def _durations(m, device, use_kernel, B, T, lens, seed=5, bf16=False):
    from kantts.models.utils import get_mask_from_lengths

    pred = m.variance_adaptor.duration_predictor
    g = torch.Generator().manual_seed(seed)
    cond = (0.8 * torch.randn(B, T, pred.lstm.input_size - 128, generator=g)).to(device)
    pred.ar_kernel = True if use_kernel else False
    pred.ar_bf16 = bf16
    pred._ar = None
    with torch.no_grad():
        out = pred.infer(cond, masks=None if lens is None else get_mask_from_lengths(torch.tensor(lens, device=device), T))
    assert (pred._ar is not None) == bool(use_kernel)
    if use_kernel:
        assert pred._ar.w.dtype == (torch.bfloat16 if bf16 else torch.float32)
    return out.detach().cpu()


def _check_durations(device, B, T, lens):
    """The one-launch loops against the per-token launches.  fp32 loop (what inference uses in EVERY precision mode: its
    outputs become index tensors) against the fp32 per-token path: 2e-5; it must not depend on the precision mode.  The bf16
    MFMA loop of round 5 (``ar_bf16``) against the bf16 per-token path: 5e-3."""
    import kantts._hip as hip

    m = _model(device)
    with torch.no_grad():
        m.variance_adaptor.duration_predictor.fc.bias.fill_(0.7)  # keep the fed-back value off the ReLU's zero
    hip.set_precision("fp32")
    ref32 = _durations(m, device, False, B, T, lens)
    got32 = _durations(m, device, True, B, T, lens)
    hip.set_precision("bf16")
    try:
        ref = _durations(m, device, False, B, T, lens)
        got = _durations(m, device, True, B, T, lens, bf16=True)
        got32_in_bf16_mode = _durations(m, device, True, B, T, lens)
    finally:
        hip.set_precision("fp32")
    assert got.shape == ref.shape == got32.shape
    assert float(ref.abs().max()) > 0.1
    scale = max(1.0, float(ref32.abs().max()))
    assert float((got32 - ref32).abs().max()) < 2e-5 * scale, float((got32 - ref32).abs().max())
    assert torch.equal(got32, got32_in_bf16_mode)
    assert float((got - ref).abs().max()) < 5e-3 * scale, float((got - ref).abs().max())
    if lens is not None:
        for b, n in enumerate(lens):
            for o in (got, got32):
                assert float(o[b, n:].abs().max() if n < T else 0.0) == 0.0


def test_duration_loop_as_one_launch_emulated():
    with emulation():
        _check_durations("cpu", 3, 7, [7, 4, 1])
        _check_durations("cpu", 2, 5, None)


def test_duration_loop_as_one_launch_kernel_source():
    with kernel_source_on_cpu():
        _check_durations("cpu", 2, 6, [6, 3])


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,lens", [(3, 7, [7, 4, 1]), (32, 80, None), (5, 61, [61, 20, 33, 60, 2])])
def test_duration_loop_as_one_launch_gpu(B, T, lens):
    _check_durations("cuda", B, T, lens)


def test_packed_blobs_follow_raw_optimizer_writes_emulated():
    """ArenaAdam writes the weights with a raw kernel that never bumps ``Tensor._version``; the packed blobs of
    the one-launch loops must still be rebuilt (their key carries ops.weights_epoch).  A step of ArenaAdam between two
    inference calls: the kernel path follows the per-launch path both times."""
    import kantts._hip as hip
    from kantts.train.optim import ArenaAdam, ParamArena

    with emulation():
        hip.set_precision("bf16")
        try:
            m = _model("cpu")
            with torch.no_grad():
                m.variance_adaptor.duration_predictor.fc.bias.fill_(0.7)
            arena = ParamArena(m)
            opt = ArenaAdam(arena, lr=0.05)
            dec_before = _decode(m, "cpu", "kernel", 2, 5, [5, 3], [2, 1])
            dur_before = _durations(m, "cpu", True, 2, 6, [6, 3])
            kern, durk = m.mel_decoder._decode_kernel, m.variance_adaptor.duration_predictor._ar
            versions = [p._version for p in m.parameters()]
            opt.zero_grad()
            for p in m.parameters():
                p.grad = torch.ones_like(p) if p.grad is None else p.grad.fill_(1.0)
            opt.step()
            assert [p._version for p in m.parameters()] == versions  # the raw write is invisible to autograd's counter
            dec_after = _decode(m, "cpu", "kernel", 2, 5, [5, 3], [2, 1])
            assert m.mel_decoder._decode_kernel is kern               # same kernel object, refreshed blobs
            ref_after = _decode(m, "cpu", "graph", 2, 5, [5, 3], [2, 1])
            pred = m.variance_adaptor.duration_predictor
            pred.ar_kernel = True
            pred._ar = durk                                           # keep the object that cached the old blob
            from kantts.models.utils import get_mask_from_lengths
            g = torch.Generator().manual_seed(5)
            cond = 0.8 * torch.randn(2, 6, pred.lstm.input_size - 128, generator=g)
            with torch.no_grad():
                dur_after = pred.infer(cond, masks=get_mask_from_lengths(torch.tensor([6, 3]), 6))
            assert pred._ar is durk
            dur_ref_after = _durations(m, "cpu", False, 2, 6, [6, 3])
        finally:
            hip.set_precision("fp32")
    assert float((dec_after - dec_before).abs().max()) > 1e-2, "the step must move the decoder output"
    assert rel_l2(dec_after, ref_after) < 2e-2, rel_l2(dec_after, ref_after)
    assert float((dur_after - dur_before).abs().max()) > 1e-3
    assert float((dur_after - dur_ref_after).abs().max()) < 5e-3 * max(1.0, float(dur_ref_after.abs().max()))
