"""Recurrence kernels alone (csrc/lstm.hip) at the shapes of the SAM-BERT step: postnet LSTM (B=32, T=612, 1 direction)
and a predictor BiLSTM (B=32, T=64, 2 directions); both numerics modes.  Usage (GPU box): python scripts/lstm_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
import torch  # noqa: E402

import kantts._hip as hip  # noqa: E402
from kantts._hip import check, lib, ptr, stream  # noqa: E402


def ev(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    hip.lib()
    dev, H, G = "cuda", 128, 512
    for B, T, ndir in ((32, 612, 1), (32, 64, 2), (32, 204, 1)):
        gx = torch.randn(B, T, ndir * G, device=dev)
        whh = torch.randn(ndir, G, H, device=dev) * 0.08
        bhh = torch.randn(ndir, G, device=dev) * 0.1
        out = torch.empty(B, T, ndir * H, device=dev)
        gates = torch.empty(ndir, B, T, G, device=dev)
        cst = torch.empty(ndir, B, T, H, device=dev)
        dout = torch.randn(B, T, ndir * H, device=dev)
        dg = torch.empty(ndir, B, T, G, device=dev)
        for prec in (1, 0):
            f = ev(lambda: check(lib().kantts_lstm_fwd(ptr(gx), ptr(whh), ptr(bhh), None, ptr(out), ptr(gates), ptr(cst),
                                                        B, T, H, ndir, 0, prec, stream()), "fwd"))
            b = ev(lambda: check(lib().kantts_lstm_bwd(ptr(dout), ptr(whh), None, ptr(gates), ptr(cst), ptr(dg), B, T, H,
                                                        ndir, 0, prec, stream()), "bwd"))
            print("B=%d T=%d ndir=%d %s: fwd %7.1f us (%.3f us/step)  bwd %7.1f us (%.3f us/step)"
                  % (B, T, ndir, "bf16" if prec else "fp32", f, f / T, b, b / T))


if __name__ == "__main__":
    main()
