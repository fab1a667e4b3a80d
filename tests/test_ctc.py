"""AttentionCTCLoss on kantts_ctc_attn (csrc/ctc.hip) against the REFERENCE'S formulation executed with stock PyTorch: per
utterance slice -> pad a constant blank score -> log_softmax -> torch.nn.CTCLoss(zero_infinity=True) with the target
1..S -> sum / B (kantts/train/loss.py:481-508 of the reference, restated below line by line), values and the gradient
w.r.t. attn_logprob.  The reference side runs in float64 (the exact answer) AND in float32 (what the reference itself
computes).  Tolerances: loss 2e-5 relative; gradient: 1e-4 of its largest entry, or -- long utterances -- four times the
error ATen's own float32 path makes against float64 on the same input: the gradient is exp(alpha + beta + nll - lp), a
difference of log-domain sums whose magnitude grows with T (|nll| ~ 2000 at 612 frames: one fp32 ulp of the exponent is
1e-4), so ANY float32 implementation, the reference's included, is that far from the exact value."""
import pytest
import torch
import torch.nn.functional as F

from util import emulation, kernel_source_on_cpu


def reference_attention_ctc(attn_logprob, in_lens, out_lens, blank_logprob=-1):
    """kantts/train/loss.py:488-508 (the reference's loop, verbatim semantics, stock torch ops on the CPU)."""
    ctc = torch.nn.CTCLoss(zero_infinity=True)
    padded = F.pad(attn_logprob, pad=(1, 0, 0, 0, 0, 0, 0, 0), value=blank_logprob)
    total = 0.0
    for bid in range(attn_logprob.shape[0]):
        target = torch.arange(1, int(in_lens[bid]) + 1).unsqueeze(0)
        cur = padded[bid].permute(1, 0, 2)[: int(out_lens[bid]), :, : int(in_lens[bid]) + 1]
        cur = torch.log_softmax(cur[None], dim=3)[0]
        total = total + ctc(cur, target, input_lengths=out_lens[bid:bid + 1], target_lengths=in_lens[bid:bid + 1])
    return total / attn_logprob.shape[0]


def _case(B, T1, T2, in_lens, out_lens, seed=0, scale=3.0):
    g = torch.Generator().manual_seed(seed)
    x = scale * torch.randn(B, 1, T1, T2, generator=g)
    return x, torch.tensor(in_lens), torch.tensor(out_lens)


def _check(device, B, T1, T2, in_lens, out_lens, seed=0, scale=3.0):
    from kantts.train.loss import AttentionCTCLoss

    x, il, ol = _case(B, T1, T2, in_lens, out_lens, seed, scale)
    xr = x.double().requires_grad_(True)
    ref = reference_attention_ctc(xr, il, ol)
    ref.backward()
    x32 = x.clone().requires_grad_(True)
    reference_attention_ctc(x32, il, ol).backward()
    err_aten = float((x32.grad.double() - xr.grad).abs().max())
    xd = x.to(device).requires_grad_(True)
    got = AttentionCTCLoss()(xd, il.to(device), ol.to(device))
    (2.5 * got).backward()
    assert abs(float(got) - float(ref)) <= 2e-5 * max(1.0, abs(float(ref))), (float(got), float(ref))
    gd = xd.grad.cpu() / 2.5
    gmax = float(xr.grad.abs().max())
    err = float((gd - xr.grad).abs().max())
    assert err <= max(1e-4 * max(gmax, 1e-6), 4.0 * err_aten), (err, gmax, err_aten)
    # frames past the utterance / classes past its phonemes receive no gradient
    for b in range(B):
        assert float(gd[b, 0, int(ol[b]):].abs().max() if int(ol[b]) < T1 else 0.0) == 0.0
        assert float(gd[b, 0, :, int(il[b]):].abs().max() if int(il[b]) < T2 else 0.0) == 0.0


CASES = [(3, 20, 6, [6, 3, 1], [20, 11, 5]),          # ragged, a one-phoneme utterance
         (2, 9, 5, [5, 4], [5, 9]),                   # T == S: only the diagonal path survives
         (2, 7, 6, [6, 2], [4, 7]),                   # T < S: impossible alignment -> zero_infinity (loss 0, gradient 0)
         (1, 40, 17, [17], [40])]


@pytest.mark.parametrize("case", CASES)
def test_attention_ctc_emulated(case):
    """The numpy/torch model of the entry point (oracle/cabi_numpy.py: ATen's CPU ctc_loss per utterance): checks the
    host layer -- slicing, int32 lengths, the mean over the batch, the gradient scaling."""
    with emulation():
        _check("cpu", *case)


@pytest.mark.parametrize("case", CASES + [(2, 70, 140, [140, 90], [70, 300 - 230]), (1, 50, 300, [300], [50])])
def test_attention_ctc_kernel_source(case):
    """The kernel SOURCE on the CPU (tests/hipemu): the recurrences, the register ring of prefetched logits, more than one
    state per thread (2 S + 1 = 281 and 601 > 256 threads)."""
    B, T1, T2, il, ol = case
    il = [min(v, T2) for v in il]
    ol = [min(max(v, 1), T1) for v in ol]
    with kernel_source_on_cpu():
        _check("cpu", B, T1, T2, il, ol)


def test_attention_ctc_limits_kernel_source():
    import ctypes

    import kantts._hip as hip

    with kernel_source_on_cpu():
        g = hip.CtcArgs()
        z = torch.zeros(8)
        for name in ("logits", "ws", "loss", "grad"):
            setattr(g, name, z.data_ptr())
        zi = torch.zeros(2, dtype=torch.int32)
        g.in_lens = g.out_lens = zi.data_ptr()
        g.B, g.T1, g.T2 = 1, 4, 512
        assert hip.lib().kantts_ctc_attn(ctypes.byref(g), None) == hip.E_UNSUPPORTED   # more than 511 phonemes
        g.B = 0
        assert hip.lib().kantts_ctc_attn(ctypes.byref(g), None) == 0
        assert hip.lib().kantts_ctc_attn_workspace(3, 10, 4) == 3 * 2 * 10 * 9


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES + [(32, 612, 64, None, None), (4, 300, 200, [200, 150, 33, 200], [300, 299, 120, 201])])
def test_attention_ctc_gpu(case):
    B, T1, T2, il, ol = case
    if il is None:  # the bench batch's shape: 32 utterances, up to 64 phonemes, up to 612 frames
        g = torch.Generator().manual_seed(1)
        il = torch.randint(32, T2 + 1, (B,), generator=g).tolist()
        ol = torch.randint(300, T1 + 1, (B,), generator=g).tolist()
        il[0], ol[0] = T2, T1
    _check("cuda", B, T1, T2, il, ol, scale=2.0)


@pytest.mark.gpu
def test_attention_ctc_is_capturable_gpu():
    """The point of the kernel: the loss and its backward inside a hipGraph (ATen's ctc_loss synchronises with the host)."""
    from kantts.train.loss import AttentionCTCLoss

    x, il, ol = _case(4, 50, 12, [12, 7, 3, 9], [50, 31, 20, 44])
    xd, il, ol = x.cuda().requires_grad_(True), il.cuda(), ol.cuda()
    crit = AttentionCTCLoss()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            xd.grad = None
            crit(xd, il, ol).backward()
    torch.cuda.current_stream().wait_stream(s)
    eager_loss, eager_grad = float(crit(xd.detach(), il, ol)), xd.grad.clone()
    xd.grad = None
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, capture_error_mode="thread_local"):
        loss = crit(xd, il, ol)
        loss.backward()
    with torch.no_grad():
        xd.grad.zero_()
    gr.replay()
    torch.cuda.synchronize()
    assert abs(float(loss) - eager_loss) < 1e-6 and torch.equal(xd.grad, eager_grad)
