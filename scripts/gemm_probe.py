"""Probe of the dominant GEMM shapes: tile-height / split-K sweep, timed inside captured hipGraphs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
import torch

import kantts._hip as hip
from kantts._hip import gemm, make_seg


def graph_time(fn, reps=20, replays=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * replays)


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    dev = "cuda"
    M, C, F = 32 * 204, 128, 1024
    p = hip.PREC_BF16
    x, h = torch.randn(M, C, device=dev), torch.randn(M, F, device=dev)
    w1, w2 = torch.randn(F, C, device=dev) * 0.05, torch.randn(C, F, device=dev) * 0.05
    yh, yx = torch.empty(M, F, device=dev), torch.empty(M, C, device=dev)
    dw1, dw2 = torch.zeros(F, C, device=dev), torch.zeros(C, F, device=dev)
    fl = 2.0 * M * C * F
    cases = {
        "fwd128->1024": lambda sk: gemm([make_seg(x, C, 1, w1, C, 1, C)], M, F, yh, F, 1, precision=p),
        "fwd1024->128": lambda sk: gemm([make_seg(h, F, 1, w2, F, 1, F)], M, C, yx, C, 1, precision=p),
        "dgrad128->1024": lambda sk: gemm([make_seg(h, F, 1, w1, 1, C, F)], M, C, yx, C, 1, precision=p),
        "dgrad1024->128": lambda sk: gemm([make_seg(x, C, 1, w2, 1, F, C)], M, F, yh, F, 1, precision=p),
        "wgrad128->1024": lambda sk: gemm([make_seg(h, 1, F, x, 1, C, M)], F, C, dw1, C, 1, accumulate=True, splitk=sk,
                                          precision=p),
        "wgrad1024->128": lambda sk: gemm([make_seg(x, 1, C, h, 1, F, M)], C, F, dw2, F, 1, accumulate=True, splitk=sk,
                                          precision=p),
    }
    for name, fn in cases.items():
        if only and only not in name:
            continue
        sks = (1, 2, 4, 8, 16, 32) if name.startswith("wgrad") else (1,)
        for sk in sks:
            us = graph_time(lambda: fn(sk))
            print("%-16s BM=%s splitk=%2d : %7.2f us  %6.1f TF" % (name, os.environ.get("KANTTS_GEMM_BM", "auto"), sk, us,
                                                              fl / us / 1e6), flush=True)
    a = torch.randn(M, F, device=dev)
    b = torch.empty_like(a)
    print("copy 26.7MB (read+write): %.2f us" % graph_time(lambda: b.copy_(a)))
    print("torch.matmul fp32 6528x1024x128: %.2f us" % graph_time(lambda: torch.matmul(x, w1.t())))
    xb, wb = x.bfloat16(), w1.bfloat16()
    print("torch.matmul bf16 6528x1024x128: %.2f us" % graph_time(lambda: torch.matmul(xb, wb.t())))


if __name__ == "__main__":
    main()
