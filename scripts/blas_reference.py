"""Library GEMM times (torch.matmul -> hipBLASLt / rocBLAS, bf16 in / bf16 out, and fp32) at the contraction shapes of the
two hot paths, as a yardstick for DESIGN section 8: what a tuned MFMA kernel reaches on the same M x N x K when nothing is
fused.  Not used by the product.  Usage (GPU box): python scripts/blas_reference.py"""
import torch


def timed(fn, reps=20, replays=5):
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
    dev = "cuda"
    shapes = [
        ("HiFi-GAN MPD 1024->1024 k5, B=64 x 110 tokens: forward", 7040, 1024, 5120),
        ("  same layer, weight gradient (tokens are the reduction)", 1024, 5120, 7040),
        ("HiFi-GAN MPD 512->1024 k5 stride 3, B=64: forward", 7040, 1024, 2560),
        ("HiFi-GAN generator 256->256 k11, B=32 x 256: forward", 8192, 256, 2816),
        ("HiFi-GAN generator 128->128 k11, B=32 x 2048: forward", 65536, 128, 1408),
        ("HiFi-GAN generator 32->32 k11, B=32 x 8192: forward", 262144, 32, 352),
        ("SAM-BERT QKV projection 128->384, M=6528", 6528, 384, 128),
        ("SAM-BERT FFN up 128->1024, M=6528", 6528, 1024, 128),
        ("SAM-BERT FFN down 1024->128, M=6528", 6528, 128, 1024),
        ("SAM-BERT weight gradient 128x1024, M=6528", 128, 1024, 6528),
        ("SAM-BERT postnet 256->512, M=19584", 19584, 512, 256),
    ]
    print("%-62s %8s %8s %8s   %8s %8s" % ("shape (M x N x K)", "bf16 us", "TFLOP/s", "GB/s", "fp32 us", "TFLOP/s"))
    for name, M, N, K in shapes:
        row = []
        for dt in (torch.bfloat16, torch.float32):
            a = torch.randn(M, K, device=dev, dtype=dt)
            b = torch.randn(K, N, device=dev, dtype=dt)
            us = timed(lambda: torch.matmul(a, b))
            esz = 2 if dt == torch.bfloat16 else 4
            row.append((us, 2.0 * M * N * K / us / 1e6, esz * (M * K + K * N + M * N) / us / 1e3))
        print("%-62s %8.1f %8.1f %8.0f   %8.1f %8.1f   (%d x %d x %d)" % (name, row[0][0], row[0][1], row[0][2], row[1][0],
                                                                       row[1][1], M, N, K))


if __name__ == "__main__":
    main()
