"""Microbenchmark of the bf16-operand contractions (csrc/gemm_bf16.hip) against the round-1 segmented GEMM at the
shapes of the SAM-BERT step; every case is launched `reps` times inside a captured hipGraph and timed with HIP events.
Usage (GPU box): python scripts/bgemm_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
import torch  # noqa: E402

import kantts._hip as hip  # noqa: E402
from kantts._hip import bgemm_nt, bgemm_tn, gemm, make_seg, ops  # noqa: E402

dev = "cuda"


def timed(fn, reps=20, replays=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
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
    hip.lib()
    print("%-34s %9s %9s %9s  %s" % ("case", "old us", "new us", "GB/s new", "bytes (new layout)"))
    for M, K, N in ((6528, 128, 1024), (6528, 1024, 128), (6528, 128, 384), (6528, 256, 128), (2048, 128, 1024),
                    (2048, 1024, 128), (19584, 80, 512), (19584, 512, 256), (19584, 256, 512), (19584, 128, 80)):
        xf = torch.randn(M, K, device=dev)
        xb = xf.to(torch.bfloat16)
        wf = torch.randn(N, K, device=dev) * 0.05
        wb = wf.to(torch.bfloat16)
        yf = torch.empty(M, N, device=dev)
        yb = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        dyf, dyb = torch.randn(M, N, device=dev), torch.randn(M, N, device=dev).to(torch.bfloat16)
        dxb = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
        dxf = torch.empty(M, K, device=dev)
        dw = torch.zeros(N, K, device=dev)
        old_f = timed(lambda: gemm([make_seg(xf, K, 1, wf, K, 1, K)], M, N, yf, N, 1, precision=hip.PREC_BF16))
        new_f = timed(lambda: bgemm_nt([(xb, K, wb, K, K, 0)], M, N, yb, N))
        by = 2.0 * (M * K + N * K + M * N)
        print("%-34s %9.2f %9.2f %9.0f  %.1f MB" % ("fwd   %dx%d->%d bf16->bf16" % (M, K, N), old_f, new_f, by / new_f / 1e3, by / 1e6))
        new_f2 = timed(lambda: bgemm_nt([(xf, K, wb, K, K, 0)], M, N, yf, N))
        by2 = 4.0 * M * K + 2.0 * N * K + 4.0 * M * N
        print("%-34s %9s %9.2f %9.0f  %.1f MB" % ("fwd   same, fp32 A -> fp32 C", "", new_f2, by2 / new_f2 / 1e3, by2 / 1e6))
        old_d = timed(lambda: gemm([make_seg(dyf, N, 1, wf, 1, K, N)], M, K, dxf, K, 1, precision=hip.PREC_BF16))
        new_d = timed(lambda: bgemm_nt([(dyb, N, wb, K, N, 0)], M, K, dxb, K, b_kn=True))
        print("%-34s %9.2f %9.2f %9.0f" % ("dgrad (b_kn) bf16->bf16", old_d, new_d, by / new_d / 1e3))
        old_w = timed(lambda: gemm([make_seg(dyf, 1, N, xf, 1, K, M)], N, K, dw, K, 1, accumulate=True,
                                   splitk=ops._splitk_for(N, K, M), precision=hip.PREC_BF16))
        new_w = timed(lambda: bgemm_tn(dyb, N, xb, K, M, N, K, dw, K, 1))
        byw = 2.0 * (M * K + M * N) + 4.0 * N * K
        print("%-34s %9.2f %9.2f %9.0f" % ("wgrad bf16 x bf16", old_w, new_w, byw / new_w / 1e3))
    # LayerNorm(128)
    for M in (6528, 2048, 19584):
        x = torch.randn(M, 128, device=dev, requires_grad=True)
        gam, bet = torch.ones(128, device=dev, requires_grad=True), torch.zeros(128, device=dev, requires_grad=True)
        hip.set_precision("bf16")
        y = ops.layer_norm(x, gam, bet, 1e-6, out_bf16=True)
        dy = torch.randn_like(y)
        t_f = timed(lambda: ops.layer_norm(x.detach(), gam.detach(), bet.detach(), 1e-6, out_bf16=True))
        print("ln128 fwd M=%d: %.2f us (%.0f GB/s)" % (M, t_f, M * 128 * 6 / t_f / 1e3))


if __name__ == "__main__":
    main()
