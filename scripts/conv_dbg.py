"""Debug: conv_cl (window kernel / GEMM route) vs ATen conv1d on the GPU, per-output error report."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
import torch
import torch.nn.functional as F

import kantts._hip as hip
from kantts._hip import ops

hip.set_precision("fp32")
cases = [(2, 512, 16, 16, 11, 5, 25), (2, 512, 16, 16, 11, 1, 5), (2, 512, 16, 16, 3, 1, 1), (2, 2048, 4, 4, 11, 5, 25),
         (2, 128, 32, 32, 7, 3, 9), (2, 64, 64, 64, 3, 1, 1)]
for route in ("win", "gemm"):
    if route == "gemm":
        os.environ["KANTTS_NO_CONVWIN"] = "1"
    for (B, T, Ci, Co, K, dil, pad) in cases:
        for rep in range(3):
            g = torch.Generator().manual_seed(rep)
            x = torch.randn(B, T, Ci, generator=g).cuda().requires_grad_(True)
            w = (torch.randn(Co, Ci, K, generator=g) * 0.1).cuda().requires_grad_(True)
            b = torch.randn(Co, generator=g).cuda().requires_grad_(True)
            res = torch.randn(B, T, Co, generator=g).cuda().requires_grad_(True)
            y = ops.conv_cl(x, w, b, dilation=dil, pad=2 * pad, in_leaky=0.1, res=res)  # causal: all padding on the left
            ref = F.conv1d(F.pad(F.leaky_relu(x, 0.1).transpose(1, 2), (2 * pad, 0)), w, b, dilation=dil).transpose(1, 2) + res
            cot = torch.randn(ref.shape, generator=g).cuda()
            gy = torch.autograd.grad((y * cot).sum(), (x, w, b))
            gr = torch.autograd.grad((ref * cot).sum(), (x, w, b))
            errs = [float((y - ref).abs().max())] + [float((a - c).abs().max()) for a, c in zip(gy, gr)]
            print(route, (B, T, Ci, Co, K, dil), "rep", rep, "max|err| y/dx/dw/db = %.2e %.2e %.2e %.2e" % tuple(errs),
                  "db bad idx", (gy[2] - gr[2]).abs().gt(1e-3).nonzero().flatten().tolist())
