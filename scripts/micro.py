"""GPU micro-benchmarks of single kernels (HIP events on the launch stream)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
import torch

import kantts._hip as hip
from kantts._hip import ops, gemm, make_seg


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def gemm_case(M, N, K, prec, kind="nt"):
    dev = "cuda"
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev)
    y = torch.empty(M, N, device=dev)
    dy = torch.randn(M, N, device=dev)
    dx = torch.empty(M, K, device=dev)
    dw = torch.zeros(N, K, device=dev)
    p = {"fp32": hip.PREC_FP32, "bf16": hip.PREC_BF16}[prec]
    if kind == "nt":
        f = lambda: gemm([make_seg(x, K, 1, w, K, 1, K)], M, N, y, N, 1, precision=p)
        fl = 2.0 * M * N * K
    elif kind == "nn":
        f = lambda: gemm([make_seg(dy, N, 1, w, 1, K, N)], M, K, dx, K, 1, precision=p)
        fl = 2.0 * M * N * K
    else:
        sk = ops._splitk_for(N, K, M)
        f = lambda: gemm([make_seg(dy, 1, N, x, 1, K, M)], N, K, dw, K, 1, accumulate=True, splitk=sk, precision=p)
        fl = 2.0 * M * N * K
    us = timeit(f)
    tm = timeit(lambda: torch.matmul(x, w.t()) if kind == "nt" else (torch.matmul(dy, w) if kind == "nn" else torch.matmul(dy.t(), x)))
    print("gemm %-3s %-4s M=%5d N=%4d K=%4d : %8.1f us  %7.2f TF   (torch.matmul fp32 %7.1f us)" % (kind, prec, M, N, K, us, fl / us / 1e6, tm), flush=True)


def attn_case(B, L, H, mode, bw, drop):
    qkv = torch.randn(B, L, 3 * H * 16, device="cuda")
    hkv = torch.randn(B, L, 2 * H * 16, device="cuda")
    lens = torch.full((B,), L - 3, dtype=torch.int32, device="cuda")
    if mode == 0:
        f = lambda: ops.self_attention(qkv, lens, H, drop_p=drop)
    else:
        f = lambda: ops.pnca_attention(qkv, hkv, lens, bw, bw, H, drop_p=drop)
    print("attn fwd mode=%d B=%d L=%d bw=%d drop=%.1f : %8.1f us" % (mode, B, L, bw, drop, timeit(f)), flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "gemm"):
        for prec in ("bf16", "fp32"):
            for (M, N, K) in [(6528, 128, 128), (6528, 1024, 128), (6528, 128, 1024), (2048, 128, 1024), (2048, 384, 512),
                              (6528, 384, 128), (8192, 1024, 1024)]:
                gemm_case(M, N, K, prec, "nt")
            gemm_case(6528, 1024, 128, prec, "nn")
            gemm_case(6528, 1024, 128, prec, "tn")
            gemm_case(6528, 128, 1024, prec, "tn")
    if which in ("all", "attn"):
        attn_case(32, 64, 8, 0, 0, 0.0)
        attn_case(32, 64, 8, 0, 0, 0.1)
        attn_case(32, 204, 8, 1, 5, 0.0)
        attn_case(32, 204, 8, 1, 5, 0.1)
    if which in ("all", "misc"):
        x = torch.randn(6528, 128, device="cuda")
        g = torch.ones(128, device="cuda")
        print("layernorm fwd 6528x128: %.1f us" % timeit(lambda: ops.layer_norm(x, g, g)))
        print("torch empty+fill 1MB: %.1f us" % timeit(lambda: torch.zeros(262144, device="cuda")))
        print("launch floor (tiny gemm 64x64x32): %.1f us" % timeit(lambda: gemm([make_seg(x, 128, 1, x, 128, 1, 32)], 64, 64, torch.empty(64, 64, device="cuda"), 64, 1)))
