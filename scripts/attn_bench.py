"""Band attention of a PNCA block (B=32, H=8, L=204, band 5) and encoder self-attention (L=64): time per launch of the
one-launch forward / backward forms, with and without attention dropout.  Usage (GPU box): python scripts/attn_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402

import kantts._hip as hip  # noqa: E402
from kantts._hip import lib, ptr, rng_state, stream  # noqa: E402
from bgemm_bench import timed  # noqa: E402


def main():
    hip.lib()
    dev = "cuda"
    B, H, D = 32, 8, 128
    for L, bw in ((204, 5), (204, 20), (64, 5)):
        qkv, hkv = torch.randn(B * L, 3 * D, device=dev), torch.randn(B * L, 2 * D, device=dev)
        ox, oh = torch.empty(B * L, D, device=dev), torch.empty(B * L, D, device=dev)
        lx, lh = torch.empty(B, H, L, device=dev), torch.empty(B, H, L, device=dev)
        dox, doh = torch.randn(B * L, D, device=dev), torch.randn(B * L, D, device=dev)
        dqkv, dqh, dhkv = torch.empty_like(qkv), torch.empty(B * L, D, device=dev), torch.empty_like(hkv)
        lens = torch.randint(L // 2, L + 1, (B,), device=dev, dtype=torch.int32)
        bwd = torch.tensor([bw], device=dev, dtype=torch.int32)
        for p in (0.0, 0.1):
            rs = ptr(rng_state(torch.device("cuda", 0))) if p > 0 else None

            def fwd():
                assert lib().kantts_pnca_attn_fwd(ptr(qkv), ptr(hkv), 2 * D, ptr(ox), ptr(oh), ptr(lx), ptr(lh), ptr(lens), ptr(bwd),
                                                  0, 0, B, H, L, 16, p, 11, 12, rs, stream()) == 0

            def bwd_():
                assert lib().kantts_pnca_attn_bwd(ptr(qkv), ptr(hkv), 2 * D, ptr(ox), ptr(oh), ptr(dox), ptr(doh), ptr(lx), ptr(lh),
                                                  ptr(dqkv), ptr(dqh), ptr(dhkv), ptr(lens), ptr(bwd), 0, 0, B, H, L, 16, p, 11,
                                                  12, rs, stream()) in (0, 1)

            def enc_f():
                assert lib().kantts_attn_fwd(ptr(qkv), ptr(qkv) + 4 * D, ptr(qkv) + 8 * D, 3 * D, 3 * D, 3 * D, ptr(ox), D, ptr(lx),
                                             None, ptr(lens), None, 0, B, H, L, 16, 0, p, 11, rs, stream()) == 0

            def enc_b():
                assert lib().kantts_attn_bwd(ptr(qkv), ptr(qkv) + 4 * D, ptr(qkv) + 8 * D, 3 * D, 3 * D, 3 * D, ptr(ox), D, ptr(dox),
                                             D, ptr(lx), ptr(lh), ptr(dqkv), ptr(dqkv) + 4 * D, ptr(dqkv) + 8 * D, 3 * D, 3 * D,
                                             3 * D, 0, ptr(lens), None, 0, B, H, L, 16, 0, p, 11, rs, stream()) == 0

            fwd()
            print("L %3d band %2d dropout %.1f: PNCA fwd %6.2f us  bwd %6.2f us | key-padding fwd %6.2f us  bwd %6.2f us"
                  % (L, bw, p, timed(fwd), timed(bwd_), timed(enc_f), timed(enc_b)))


if __name__ == "__main__":
    main()
