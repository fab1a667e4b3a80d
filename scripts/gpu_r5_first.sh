#!/bin/bash
# First visit of the next round (written at the end of round 4, when the GPU minutes were spent): the device twins of the
# tests the line-coverage run of the kernel sources added on the CPU side, then the usual suite + bench.  Nothing here changes
# a default; each block prints PASS / FAIL and goes on.
T=${1:-r5a}
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
run_py() {  # name, python source
  timeout 300 python -c "import sys; sys.path[:0] = ['tests', 'kan-tts_amd', 'oracle']; $2" > gpurun_out/${T}_$1.log 2>&1 \
    && echo "PASS $1" || { echo "FAIL $1"; tail -n 15 gpurun_out/${T}_$1.log; }
}
# 1. melspec_reg_kernel: several pairs of frames per wave (grid capped) == one pair per wave, on the device
run_py mel_pairs "import test_melspec as t; t._several_pairs_per_wave_case('cuda')"
# 2. the > 256-chunk filterbank fallback of the same kernel
run_py mel_fallback "import test_melspec as t; t._wide_filterbank_case('cuda')"
# 3. the 67 584-frame launch (the roofline_saturating leg of bench.py) against the radix-2 kernel: values, not time
run_py mel_saturating "
import os, torch
from kantts.utils.audio_torch import MelSpectrogram
ms = MelSpectrogram().cuda(); x = torch.randn(2048, 8192, device='cuda') * 0.1
a = ms(x[:, None, :]); os.environ['KANTTS_MEL_GENERIC'] = '1'; b = ms(x[:, None, :]); os.environ.pop('KANTTS_MEL_GENERIC')
d = float((a - b).abs().max()); print('max |register - generic| over 67584 frames', d); assert d < 5e-5"
# 4. attention between 257 and 390 rows (second row per thread) and past the LDS limit, on the device
run_py attn_long "
import torch, kantts._hip as hip, torch_oracle as O
from kantts._hip import ops
hip.set_precision('fp32')
for L in (300, 400):
    g = torch.Generator().manual_seed(L); B, H, D = 2, 2, 32
    lens = torch.tensor([L, L - 37]); pad = O.pad_mask(lens, L)
    qkv = torch.randn(B, L, 3 * D, generator=g).requires_grad_(True)
    q, k, v = (O._split_heads(t, H) for t in qkv.chunk(3, -1))
    ro, _ = O._attend(q, k, v, pad[:, None, :].expand(-1, L, -1).repeat(H, 1, 1)); ro = O._merge_heads(ro, H)
    dq = qkv.detach().cuda().requires_grad_(True)
    o, _ = ops.self_attention(dq, lens.to(torch.int32).cuda(), H)
    valid = (~pad)[..., None]; cot = torch.randn(B, L, D, generator=g) * valid
    assert float(((o.cpu() - ro) * valid).detach().abs().max()) < 2e-5, L
    (ga,) = torch.autograd.grad(o, dq, cot.cuda()); (gr,) = torch.autograd.grad(ro, qkv, cot)
    assert float((ga.cpu() - gr).abs().max()) < 1e-4 * max(1.0, float(gr.abs().max())), L
print('ok')"
# 5. every MAS kernel variant is already a device test; the window-form cconv variants and the weight-gradient slices too
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -n 3 gpurun_out/${T}_pytest_gpu.log
timeout 1500 python bench.py > gpurun_out/${T}_bench_full.log 2> gpurun_out/${T}_bench_full.err; echo "bench exit $?"
tail -c 600 gpurun_out/${T}_bench_full.log
# If 1-4 pass: move the four checks into tests/ as @pytest.mark.gpu twins (tests/test_melspec.py, tests/test_gpu_ops.py).
