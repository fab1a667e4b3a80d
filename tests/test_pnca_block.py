"""One PNCA decoder block forward as ONE launch (csrc/pnca_block.hip, ops_bf16.pnca_block_fused) against the five-launch
chain it replaces (reference: kantts/models/sambert/__init__.py:212-348 + :134-149): same outputs, same saved tensors, same
gradients -- to fp32 summation order, dropout ON (both forms draw the same masks from the same seeds).

Three executions of the same case: the numpy model of the C ABI (host logic: adoption of the launch's results by the block's
ops, seed order), the kernel SOURCE on the CPU (tests/hipemu), and the device."""
import contextlib
import itertools
import os

import pytest
import torch

from util import emulation, kernel_source_on_cpu, rel_l2

HOSTSIM = os.path.exists(os.environ.get("HIPEMU_CXX", "/opt/rocm/lib/llvm/bin/clang++"))


def _case(device, B, L, lens, bw, drop, bw_on_device=False, private=False, final_ln_fp32=False, seed=0):
    import kantts._hip as hip
    import kantts._hip.ops as ops
    import kantts._hip.ops_bf16 as ops_bf16
    from kantts.models.sambert import PNCABlock
    from kantts.models.utils import SeqInfo

    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    blk = PNCABlock(128, 160, 8, 16, 1024, (1, 1), drop, drop, drop).to(device)
    with torch.no_grad():  # biases and LayerNorm vectors away from their 0 / 1 initial values
        for n, p in blk.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn(p.shape, generator=g).to(device) * 0.1)
    blk.train()
    nxt = torch.nn.LayerNorm(128, eps=1e-6).to(device)
    if final_ln_fp32:
        nxt._kantts_out_bf16 = False
    x0 = (torch.randn(B, L, 128, generator=g) * 0.7).to(device)
    hkv0 = torch.randn(B, L, 3 * 256, generator=g).to(device)  # this block's K | V = a column block of a wider buffer
    lens_t = torch.tensor(lens, dtype=torch.int64, device=device)
    info = SeqInfo(lens_t, L)
    cot = torch.randn(B, L, 128, generator=g).to(device)
    bw_dev = torch.tensor([bw], dtype=torch.int32, device=device) if bw_on_device else None

    def run(fused):
        ops._seed_counter = itertools.count(1000)
        blk.zero_grad()
        x = x0.clone().requires_grad_(True)
        hk = hkv0.clone().requires_grad_(True)
        hkv = hk[:, :, 256:512]
        xin = x * 1.0  # a non-leaf, as in the stack
        if private:  # the producer's row mask travels on the tensor (ops_bf16.RowMaskToken)
            xin = ops.linear(x, torch.eye(128, device=device), None, rowmask=info.mask)
        ops_bf16.PNCA_BLOCK["on"] = fused
        ops_bf16.BAND_BOUND["max"] = bw if bw_on_device else None
        calls, bcalls, ccalls = [], [], []
        real, real_b, real_c = ops_bf16.pnca_block_fwd, ops_bf16.pnca_block_bwd, hip.pnca_attn_qkv_bwd
        ops_bf16.pnca_block_fwd = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        ops_bf16.pnca_block_bwd = lambda *a, **k: (bcalls.append(1), real_b(*a, **k))[1]
        hip.pnca_attn_qkv_bwd = lambda *a, **k: (ccalls.append(1), real_c(*a, **k))[1]
        try:
            out, _, _ = blk(xin, None, mask=info, x_band_width=0 if bw_on_device else bw,
                            h_band_width=0 if bw_on_device else bw, bw_dev=bw_dev, hkv=hkv, private_input=private,
                            next_ln=nxt)
            pre = out._kantts_prenorm
            (out * cot).sum().backward()
        finally:
            ops_bf16.pnca_block_fwd, ops_bf16.pnca_block_bwd, hip.pnca_attn_qkv_bwd = real, real_b, real_c
            ops_bf16.PNCA_BLOCK["on"] = True
            ops_bf16.BAND_BOUND["max"] = None
        # whenever the forward was one launch, the backward is three: its row-local half, its cross-row half (attention
        # backward + QKV input gradient + LayerNorm backward) and nothing else but weight gradients
        assert len(bcalls) == len(calls) and len(ccalls) == len(calls)
        grads = {n: p.grad.detach().cpu().clone() for n, p in blk.named_parameters() if p.grad is not None}
        return (len(calls), out.detach().cpu(), pre.xn.float().cpu(), pre.mean.cpu(), pre.rstd.cpu(), x.grad.cpu(),
                hk.grad.cpu(), grads)

    hip.set_precision("bf16")
    try:
        nf, of, xnf, mf, rf, dxf, dhf, gf = run(True)
        nc, oc, xnc, mc, rc, dxc, dhc, gc = run(False)
    finally:
        hip.set_precision("fp32")
    assert nc == 0
    return nf, (of, xnf, mf, rf, dxf, dhf, gf), (oc, xnc, mc, rc, dxc, dhc, gc)


def _compare(nf, f, c, expect_fused=True):
    assert nf == (1 if expect_fused else 0)
    of, xnf, mf, rf, dxf, dhf, gf = f
    oc, xnc, mc, rc, dxc, dhc, gc = c
    assert torch.isfinite(of).all()
    # fp32 summation order differs between the forms; where that moves a normalised row's element across a bf16 rounding tie
    # (one element in ~10^4) the feed-forward's output moves by ~1e-4: rare, so the l2 bound is the sharp one
    assert float((of - oc).abs().max()) <= 2e-3 * max(1.0, float(oc.abs().max())) and rel_l2(of, oc) <= 1e-4, "block output"
    # the consumer's statistics are means over the 128 channels of those rows: a row with a flipped tie moves its mean by
    # (what the element moved) / 128 -- the bound follows the output's own difference (on the device the dropout masks, and
    # with them which rows hold a tie, depend on how many draws the tests before this one made)
    d_out = float((of - oc).abs().max())
    assert float((mf - mc).abs().max()) <= 1e-5 + d_out / 32 and rel_l2(rf, rc) <= 1e-5 + 1e-2 * d_out, \
        "LayerNorm statistics of the consumer"
    # normalised rows are bf16 (or fp32 for the stack's final LayerNorm): one ulp where the fp32 value sits on a tie
    assert float((xnf - xnc).abs().max()) <= 4e-2 and rel_l2(xnf, xnc) <= 2e-3, "normalised rows"
    # gradients: the backward pass is the SAME code reading the tensors the two forward forms saved (bf16 operands round
    # the occasional element differently)
    assert rel_l2(dxf, dxc) <= 2e-3, "dx %g" % rel_l2(dxf, dxc)
    assert rel_l2(dhf, dhc) <= 2e-3, "dhkv %g" % rel_l2(dhf, dhc)
    for n in gc:
        assert rel_l2(gf[n], gc[n]) <= 3e-3, "%s %g" % (n, rel_l2(gf[n], gc[n]))


_CASES = [
    # B, L, lens, bw, drop
    dict(B=2, L=37, lens=[37, 20], bw=3, drop=0.0),                      # tiles cross the sequence boundary, ragged tail
    dict(B=3, L=37, lens=[30, 37, 1], bw=5, drop=0.1),                   # 111 rows, dropout everywhere
    dict(B=1, L=70, lens=[64], bw=16, drop=0.1, private=True),           # widest band the launch holds, input row mask
    dict(B=2, L=33, lens=[33, 33], bw=0, drop=0.0, final_ln_fp32=True),  # band of one key; fp32 normalised rows out
    dict(B=2, L=40, lens=[40, 25], bw=4, drop=0.1, bw_on_device=True),   # band width in device memory (captured step)
]


@pytest.mark.parametrize("kw", _CASES, ids=lambda k: "B%d-L%d-bw%d-p%g" % (k["B"], k["L"], k["bw"], k["drop"]))
def test_fused_block_equals_the_chain_emulated(kw):
    with emulation():
        _compare(*_case("cpu", **kw))


# 66 row tiles: from 64 tiles up the three launches walk the tiles in XCD-band order (pb_tile; 66 = 8 * 8 + 2: uneven bands)
_BANDED = dict(B=11, L=190, lens=[190 - 7 * i for i in range(11)], bw=6, drop=0.1)


@pytest.mark.skipif(not HOSTSIM, reason="the host build of the kernel sources needs the ROCm clang")
@pytest.mark.parametrize("kw", _CASES + [_BANDED], ids=lambda k: "B%d-L%d-bw%d-p%g" % (k["B"], k["L"], k["bw"], k["drop"]))
def test_fused_block_equals_the_chain_kernel_source(kw):
    with kernel_source_on_cpu():
        _compare(*_case("cpu", **kw))


@pytest.mark.gpu
@pytest.mark.parametrize("kw", _CASES + [dict(B=32, L=204, lens=[204 - 3 * i for i in range(32)], bw=5, drop=0.1)],
                         ids=lambda k: "B%d-L%d-bw%d-p%g" % (k["B"], k["L"], k["bw"], k["drop"]))
def test_fused_block_equals_the_chain_gpu(kw):
    _compare(*_case("cuda", **kw))


def _declines(device):
    """Band widths the launch cannot hold keep the chain (host-known), and an unknown bound with the band width on the device
    keeps it too; a device band width ABOVE a wrongly promised bound poisons the result instead of computing something else."""
    import kantts._hip.ops_bf16 as ops_bf16

    nf, f, c = _case(device, B=1, L=50, lens=[50], bw=17, drop=0.0)
    _compare(nf, f, c, expect_fused=False)
    old = ops_bf16.PB_MAX_BAND
    try:
        ops_bf16.PB_MAX_BAND = 64  # the host is lied to: the kernel must not return finite numbers
        nf, f, c = _case(device, B=1, L=50, lens=[50], bw=17, drop=0.0, bw_on_device=True)
    finally:
        ops_bf16.PB_MAX_BAND = old
    assert nf == 1 and not torch.isfinite(f[0]).all()


def test_fused_block_declines_wide_bands_emulated():
    with emulation():
        _declines("cpu")


@pytest.mark.skipif(not HOSTSIM, reason="the host build of the kernel sources needs the ROCm clang")
def test_fused_block_declines_wide_bands_kernel_source():
    with kernel_source_on_cpu():
        _declines("cpu")


@pytest.mark.gpu
def test_fused_block_declines_wide_bands_gpu():
    _declines("cuda")
