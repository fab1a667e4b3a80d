"""The target-only preparation of a teacher-forced SAM-BERT step as one launch (csrc/seq.hip: kantts_teacher_plan,
KanTtsSAMBERT.teacher_forced_plan_fused) against the stock-operator form it replaces (teacher_forced_plan + SeqInfo;
reference kantts/models/utils.py:13-23, kantts_sambert.py:466-468, 556-559, 736-750, 981-985, positions.py:83-98): masks,
clamped lengths, the length-regulator plan and the band width bit-exact, sinusoids / logs to fp32 rounding."""
import os

import pytest
import torch

import torch_oracle as O
from util import emulation, kernel_source_on_cpu

HOSTSIM = os.path.exists(os.environ.get("HIPEMU_CXX", "/opt/rocm/lib/llvm/bin/clang++"))


def _case(device):
    from kantts.models.sambert.kantts_sambert import KanTtsSAMBERT
    from kantts.models.utils import SeqInfo

    cfg = O.sambert_config(tiny=True)
    torch.manual_seed(0)
    m = KanTtsSAMBERT(dict(cfg)).to(device).train()
    for B, T_in, seed, dev_bw in ((3, 16, 5, False), (5, 24, 9, True), (32, 64, 1234, True)):
        m.device_band_width = dev_bw
        b = O.synthetic_sambert_batch(B=B, T_in=T_in, min_len=T_in // 2, dur_hi=7 if T_in < 64 else 17, seed=seed)
        b["output_lengths"][1] -= 2  # a sequence that ends inside its last decoder step
        b = {k: v.to(device) for k, v in b.items()}
        in_info = SeqInfo(b["input_lengths"], T_in)
        ref = m.teacher_forced_plan(in_info, b["output_lengths"], b["mel_targets"], b["duration_targets"])
        got_in, got = m.teacher_forced_plan_fused(b["input_lengths"], T_in, b["output_lengths"], b["mel_targets"],
                                                  b["duration_targets"])
        for a, c in ((got_in, in_info), (got["out_info"], ref["out_info"]), (got["lfr_info"], ref["lfr_info"])):
            assert a.mask.dtype == torch.bool and a.lens64.dtype == torch.int64 and a.lens32.dtype == torch.int32
            assert torch.equal(a.mask, c.mask) and torch.equal(a.lens64, c.lens64) and torch.equal(a.lens32, c.lens32)
        for k in range(7):
            x, y = got["lr_plan"][k], ref["lr_plan"][k]
            assert (torch.equal(x, y) if torch.is_tensor(x) else x == y), k
        for k in ("pos_enc", "prev", "dec_input"):
            assert got[k].shape == ref[k].shape and got[k].dtype == ref[k].dtype, k
            assert float((got[k] - ref[k]).abs().max()) <= 1e-6, k
        assert torch.equal(got["dec_input"], ref["dec_input"])
        assert float(got["bw_val"]) == float(ref["bw_val"])
        if dev_bw:
            assert torch.equal(got["bw_dev"], ref["bw_dev"])


def test_teacher_plan_kernel_equals_stock_operators_emulated():
    with emulation():
        _case("cpu")


@pytest.mark.skipif(not HOSTSIM, reason="the host build of the kernel sources needs the ROCm clang")
def test_teacher_plan_kernel_equals_stock_operators_kernel_source():
    with kernel_source_on_cpu():
        _case("cpu")


@pytest.mark.gpu
def test_teacher_plan_kernel_equals_stock_operators_gpu():
    _case("cuda")
