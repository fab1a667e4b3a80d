"""cconv_wgrad launch (+ its reduce) over the number of token slices, for the generator's wide residual layers and a few
discriminator layers at batch 32 (bf16 operands).  Usage (GPU box): python scripts/cconv_wgrad_slices.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
import torch

import kantts._hip as hip

bf = torch.bfloat16
# (B, T, inner, Cin, Cout, groups, K, stride, dil)
SHAPES = [(32, 2048, 1, 128, 128, 1, 11, 1, 1), (32, 2048, 1, 128, 128, 1, 7, 1, 1), (32, 2048, 1, 128, 128, 1, 3, 1, 1),
          (32, 256, 1, 256, 256, 1, 11, 1, 1), (32, 256, 1, 256, 256, 1, 3, 1, 1), (64, 61, 5, 128, 512, 1, 5, 3, 1),
          (64, 304, 3, 32, 128, 1, 5, 3, 1)]


def ev(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (B, T, P, Cin, Cout, G, K, stride, dil) in SHAPES:
    Td = (T - 1) // stride + 1 if stride > 1 else T
    pad = (K - 1) * dil // 2
    x = torch.randn(B, T, P, Cin, device="cuda").to(bf)
    dy = torch.randn(B, Td, P, Cout, device="cuda").to(bf)
    dw = torch.zeros(K, Cout, Cin // G, device="cuda")
    db = torch.zeros(Cout, device="cuda")
    flops = 2.0 * B * Td * P * Cout * (Cin // G) * K
    res = []
    for sl in (0, 8, 16, 23, 32, 36, 42, 46, 47, 51, 64, 73, 74, 85, 102, 128, 170, 256):
        def run():
            assert hip.cconv_wgrad(x, dy, dw, db, B=B, Tsrc=T, Tdst=Td, groups=G, CR=Cin // G, NG=Cout // G, K=K, stride=stride,
                                   dil=dil, pad=pad, inner=P, up=1, slices=sl)
        res.append((sl, ev(run)))
    best = min(res, key=lambda r: r[1])
    print("B %d T %d inner %d %d->%d g%d K %d s%d d%d (%.1f GFLOP): " % (B, T, P, Cin, Cout, G, K, stride, dil, flops / 1e9) +
          "  ".join("s%d %.0f" % r for r in res) + "   best s%d %.0f us = %.0f TFLOP/s" % (best[0], best[1], flops / best[1] / 1e6),
          flush=True)
