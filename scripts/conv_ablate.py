"""Times single conv_win launches of representative generator shapes (forward, bf16 MFMA) -- run once per ablation mask
(KANTTS_CW_DBG with the CWDBG experiment build) to see which phase of the kernel the time belongs to."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
import torch

import kantts._hip as hip
from kantts._hip import ops

hip.set_precision("bf16")
shapes = [(32, 8192, 32, 3), (32, 8192, 32, 11), (32, 2048, 128, 3), (32, 2048, 128, 11), (32, 256, 256, 7)]
out = []
for B, T, C, K in shapes:
    x = torch.randn(B, T, C, device="cuda")
    w = torch.randn(K, C, C, device="cuda") * 0.05
    b = torch.randn(C, device="cuda")
    with torch.no_grad():
        f = lambda: ops.conv_cl(x, w, b, pad=(K - 1) // 2, in_leaky=0.1, tap_major=True)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        # same bytes through a plain device copy (read x, write y): what the memory system gives a trivial kernel
        y = torch.empty_like(x)
        for _ in range(3):
            y.copy_(x)
        e0.record()
        for _ in range(20):
            y.copy_(x)
        e1.record()
        torch.cuda.synchronize()
        cp = e0.elapsed_time(e1) / 20 * 1e3
    out.append("C=%d K=%d T=%d: %.1f us (copy %.1f us)" % (C, K, T, us, cp))
print("mask=%s  " % os.environ.get("KANTTS_CW_DBG", "0") + " | ".join(out))
