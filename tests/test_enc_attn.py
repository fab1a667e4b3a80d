"""The attention sub-layer of an encoder FFT block forward as ONE launch (csrc/enc_attn.hip, ops_bf16.enc_attn_fused) against
the three-launch chain it replaces (reference: kantts/models/sambert/__init__.py:52-106 inside FFTBlock.forward :152-184):
same block output, same saved tensors, same gradients -- to fp32 summation order, dropout ON (both forms draw the same masks
from the same seeds).

Three executions of the same case: the numpy model of the C ABI (host logic: adoption of the launch's results by the
sub-layer's ops, seed order), the kernel SOURCE on the CPU (tests/hipemu: the wave = head layout, the fp32 MFMA attention
straight from the projection's accumulators), and the device."""
import itertools
import os

import pytest
import torch

from util import emulation, kernel_source_on_cpu, rel_l2

HOSTSIM = os.path.exists(os.environ.get("HIPEMU_CXX", "/opt/rocm/lib/llvm/bin/clang++"))


def _case(device, B, L, lens, drop, private=False, seed=0):
    import kantts._hip as hip
    import kantts._hip.ops as ops
    import kantts._hip.ops_bf16 as ops_bf16
    from kantts.models.sambert import FFTBlock
    from kantts.models.utils import SeqInfo

    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    blk = FFTBlock(128, 128, 8, 16, 1024, (3, 1), drop, drop, drop).to(device)
    with torch.no_grad():  # biases and LayerNorm vectors away from their 0 / 1 initial values
        for n, p in blk.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn(p.shape, generator=g).to(device) * 0.1)
    blk.train()
    nxt = torch.nn.LayerNorm(128, eps=1e-6).to(device)
    x0 = (torch.randn(B, L, 128, generator=g) * 0.7).to(device)
    info = SeqInfo(torch.tensor(lens, dtype=torch.int64, device=device), L)
    cot = torch.randn(B, L, 128, generator=g).to(device)

    def run(fused):
        ops._seed_counter = itertools.count(1000)
        blk.zero_grad()
        x = x0.clone().requires_grad_(True)
        xin = x * 1.0  # a non-leaf, as in the stack
        if private:  # the producer's row mask travels on the tensor (ops_bf16.RowMaskToken)
            xin = ops.linear(x, torch.eye(128, device=device), None, rowmask=info.mask)
        ops_bf16.ENC_ATTN["on"] = fused
        calls = []
        orig = hip.enc_attn_fwd
        hip.enc_attn_fwd = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            out, _ = blk(xin, mask=info, private_input=private, next_ln=nxt)
            pre = out._kantts_prenorm
            (out * cot).sum().backward()
        finally:
            hip.enc_attn_fwd = orig
            ops_bf16.ENC_ATTN["on"] = True
        grads = {n: p.grad.detach().cpu().clone() for n, p in blk.named_parameters() if p.grad is not None}
        return len(calls), out.detach().cpu(), pre.xn.float().cpu(), x.grad.cpu(), grads

    hip.set_precision("bf16")
    try:
        f = run(True)
        c = run(False)
    finally:
        hip.set_precision("fp32")
    assert c[0] == 0
    return f, c


def _compare(f, c, expect_fused=True):
    nf, of, xnf, dxf, gf = f
    _, oc, xnc, dxc, gc = c
    assert nf == (1 if expect_fused else 0)
    assert torch.isfinite(of).all()
    # fp32 summation order differs between the forms; where that moves an element of a bf16 MFMA operand (context, normalised
    # row) across a rounding tie the feed-forward's output moves by ~1e-4: rare, so the l2 bound is the sharp one
    assert float((of - oc).abs().max()) <= 2e-3 * max(1.0, float(oc.abs().max())) and rel_l2(of, oc) <= 1e-4, \
        ("block output", float((of - oc).abs().max()), rel_l2(of, oc))
    assert float((xnf - xnc).abs().max()) <= 4e-2 and rel_l2(xnf, xnc) <= 2e-3, "normalised rows of the consumer"
    # gradients: the backward pass is the SAME code reading the tensors the two forward forms saved
    assert rel_l2(dxf, dxc) <= 2e-3, "dx %g" % rel_l2(dxf, dxc)
    assert set(gf) == set(gc)
    for n in gc:
        assert rel_l2(gf[n], gc[n]) <= 3e-3, "%s %g" % (n, rel_l2(gf[n], gc[n]))


_CASES = [
    dict(B=2, L=37, lens=[37, 20], drop=0.0),                   # a ragged tail inside the third token block
    dict(B=3, L=64, lens=[30, 64, 1], drop=0.1),                # the largest sequence a workgroup holds; one-key attention
    dict(B=2, L=16, lens=[16, 9], drop=0.1, private=True),      # one token block; input row mask travelling on the tensor
    dict(B=1, L=5, lens=[5], drop=0.0),
    dict(B=2, L=100, lens=[100, 77], drop=0.1),                 # the 8-token-block instantiation (65 .. 128 tokens)
]
_ID = lambda k: "B%d-L%d-p%g" % (k["B"], k["L"], k["drop"])  # noqa: E731


@pytest.mark.parametrize("kw", _CASES, ids=_ID)
def test_fused_sublayer_equals_the_chain_emulated(kw):
    with emulation():
        _compare(*_case("cpu", **kw))


@pytest.mark.skipif(not HOSTSIM, reason="the host build of the kernel sources needs the ROCm clang")
@pytest.mark.parametrize("kw", _CASES, ids=_ID)
def test_fused_sublayer_equals_the_chain_kernel_source(kw):
    with kernel_source_on_cpu():
        _compare(*_case("cpu", **kw))


def test_longer_sequences_keep_the_chain_emulated():
    """More than 128 tokens per sequence: the launch does not apply, the sub-layer runs its three launches."""
    with emulation():
        f, c = _case("cpu", B=1, L=130, lens=[129], drop=0.0)
        _compare(f, c, expect_fused=False)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", _CASES + [dict(B=32, L=64, lens=[64 - i for i in range(32)], drop=0.1)], ids=_ID)
def test_fused_sublayer_equals_the_chain_gpu(kw):
    _compare(*_case("cuda", **kw))
