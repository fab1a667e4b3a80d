"""The postnet-sized contractions (M = 19584 = 32 x 612 frames, 256 <-> 512) on kantts_bgemm_nt, per row-tile height
(KANTTS_BGEMM_BM is read once per process: run once per value).  Usage: KANTTS_BGEMM_BM=128 python scripts/postnet_gemm_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402

import kantts._hip as hip  # noqa: E402
from kantts._hip import bgemm_nt  # noqa: E402
from bgemm_bench import timed  # noqa: E402


def main():
    hip.lib()
    dev, bf = "cuda", torch.bfloat16
    print("KANTTS_BGEMM_BM =", os.environ.get("KANTTS_BGEMM_BM", "(default)"))
    for M, K, N in ((19584, 512, 256), (19584, 256, 512), (19584, 80, 512), (6528, 256, 256), (6528, 384, 128), (6528, 256, 128)):
        xf = torch.randn(M, K, device=dev)
        xb = xf.to(bf)
        wb = (torch.randn(N, K, device=dev) * 0.05).to(bf)
        yf, yb = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev, dtype=bf)
        dyf = torch.randn(M, N, device=dev)
        dxf = torch.empty(M, K, device=dev)
        t1 = timed(lambda: bgemm_nt([(xf, K, wb, K, K, 0)], M, N, yf, N))
        t2 = timed(lambda: bgemm_nt([(xf, K, wb, K, K, 0)], M, N, yb, N, relu=True))
        t3 = timed(lambda: bgemm_nt([(xb, K, wb, K, K, 0)], M, N, yf, N))
        t4 = timed(lambda: bgemm_nt([(dyf, N, wb, K, N, 0)], M, K, dxf, K, b_kn=True))
        print("M %5d  %3d -> %3d   fp32->fp32 %6.2f us | fp32->bf16 relu %6.2f | bf16->fp32 %6.2f | dgrad fp32->fp32 %6.2f"
              % (M, K, N, t1, t2, t3, t4))


if __name__ == "__main__":
    main()
