"""LayerNorm backward as the epilogue of the launch that produces its output gradient, against the two-launch forms, per
launch (written in round 3 without a GPU at hand: run it first thing in round 4):
  * input gradient of the QKV projection (M x 384 -> M x 128) + kantts_ln128_bwd_rows   vs   kantts_bgemm_nt_lnbwd
    (the feed-forward pair's analogue, kantts_ffn_pair_lnbwd, measured 35.1 us against 23.0 us in round 4 --
    profiles/r04_runA_lnbwd_per_launch.log -- and was removed)
at the decoder's M = 6528 (T = 204) and the encoder's M = 2048 (T = 64, k = 3).  Usage (GPU box): python scripts/lnbwd_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402

import kantts._hip as hip  # noqa: E402
from kantts._hip import lib, ptr, stream  # noqa: E402
from kantts._hip.ops_bf16 import frag_major  # noqa: E402
from bgemm_bench import timed  # noqa: E402


def main():
    hip.lib()
    hip.set_precision("bf16")
    dev, bf = "cuda", torch.bfloat16
    for M, T, KT in ((6528, 204, 1), (2048, 64, 3)):
        x = torch.randn(M, 128, device=dev)
        gam = torch.rand(128, device=dev) + 0.5
        mean, rstd = x.mean(-1).contiguous(), (x.var(-1, unbiased=False) + 1e-6).rsqrt().contiguous()
        dres = torch.randn(M, 128, device=dev)
        rows = (torch.arange(M, device=dev) % 7 == 2).to(torch.uint8)
        dx, dg, db = torch.empty(M, 128, device=dev), torch.zeros(128, device=dev), torch.zeros(128, device=dev)
        lnb = (x, gam, mean, rstd, dres, rows, dx, dg, db)

        def ln_bwd(dy):
            hip.check(lib().kantts_ln128_bwd_rows(ptr(dy), 1, ptr(x), ptr(gam), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(dg),
                                                  ptr(db), ptr(rows), M, stream()), "ln128_bwd_rows")

        # ---- attention sub-layer: dz (M, 384) fp32 (what the attention backward writes) x W_qkv (384, 128) bf16
        dz = torch.randn(M, 384, device=dev)
        w = (torch.randn(384, 128, device=dev) * 0.1).to(bf)
        dxn = torch.empty(M, 128, device=dev, dtype=bf)
        seg = [(dz, 384, (w, 0), 128, 384, 0)]
        two = lambda: (hip.bgemm_nt(seg, M, 128, dxn, 128, b_kn=True), ln_bwd(dxn))  # noqa: E731
        one = lambda: hip.bgemm_nt(seg, M, 128, None, 128, b_kn=True, c_bf16=True, lnb=lnb)  # noqa: E731
        gemm = lambda: hip.bgemm_nt(seg, M, 128, dxn, 128, b_kn=True)  # noqa: E731
        two(), one()
        print("M %5d  QKV input gradient: GEMM alone %6.2f us | + LayerNorm backward, two launches %6.2f us | one launch %6.2f us"
              % (M, timed(gemm), timed(two), timed(one)))


if __name__ == "__main__":
    main()
