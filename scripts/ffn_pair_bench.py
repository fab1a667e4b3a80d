"""Fused feed-forward pair (csrc/ffn_pair.hip) against the two-launch form (csrc/gemm_bf16.hip) at the shapes of the
SAM-BERT step: decoder blocks M = 6528, k = 1 (forward and backward form) and encoder blocks M = 2048, k = 3 (forward).
Every case is launched `reps` times inside a captured hipGraph and timed with HIP events.
Usage (GPU box): python scripts/ffn_pair_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
import torch  # noqa: E402

import kantts._hip as hip  # noqa: E402
from kantts._hip import bgemm_nt, ffn_pair  # noqa: E402
from kantts._hip.ops_bf16 import frag_major  # noqa: E402

dev, bf = "cuda", torch.bfloat16


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
    C = 128
    for M, T, F, KT in ((6528, 204, 1024, 1), (2048, 64, 1024, 3), (19584, 612, 1024, 1)):
        pad = (KT - 1) // 2
        x = torch.randn(M, C, device=dev).to(bf)
        w1 = (torch.randn(KT, F, C, device=dev) * 0.05).to(bf)
        w2 = (torch.randn(C, F, device=dev) * 0.03).to(bf)
        b1, b2 = torch.zeros(F, device=dev), torch.zeros(C, device=dev)
        res, dy = torch.randn(M, C, device=dev), torch.randn(M, C, device=dev)
        hid = torch.empty(M, F, device=dev, dtype=bf)
        y = torch.empty(M, C, device=dev)
        f1, f2 = frag_major(w1.reshape(KT * F, C)), frag_major(w2)
        for p in (0.0, 0.1):
            def two():
                segs = [(x, C, (w1, tap * F * C), C, C, tap - pad) for tap in range(KT)]
                bgemm_nt(segs, M, F, hid, F, T=T, bias=b1, relu=True, drop_p=p, drop_seed=5)
                bgemm_nt([(hid, F, w2, F, F, 0)], M, C, y, C, bias=b2, drop_p=p, drop_seed=6, res=res, ldr=C)

            def one():
                assert ffn_pair(x, f1, f2, y, M=M, T=T, F=F, KT=KT, pad=pad, bias1=b1, bias2=b2, relu=True, drop1_p=p,
                                drop1_seed=5, drop2_p=p, drop2_seed=6, t_out=hid, res=res)

            t2, t1 = timed(two), timed(one)
            by = 2 * M * C + 2 * KT * F * C + 2 * C * F + 2 * M * F + 8 * M * C
            print("fwd  M=%5d k=%d F=%d dropout %.1f: two launches %7.2f us   one launch %7.2f us  (%.0f GB/s at %.1f MB)"
                  % (M, KT, F, p, t2, t1, by / t1 / 1e3, by / 1e6))
        if KT == 1:
            w2t, w1t = frag_major(w2.t().contiguous()), frag_major(w1[0].t().contiguous())
            dz = torch.empty(M, F, device=dev, dtype=bf)
            dh = torch.empty(M, C, device=dev, dtype=bf)
            one()
            for p in (0.0, 0.1):
                def two_b():
                    bgemm_nt([(dy, C, w2, F, C, 0)], M, F, dz, F, b_kn=True, gate=hid, ldg=F, a_drop_p=p, a_drop_seed=6,
                             a_drop_ld=C)
                    bgemm_nt([(dz, F, w1, C, F, 0)], M, C, dh, C, b_kn=True)

                def one_b():
                    assert ffn_pair(dy, w2t, w1t, dh, M=M, T=T, F=F, xdrop_p=p, xdrop_seed=6, gate=hid, t_out=dz)

                t2, t1 = timed(two_b), timed(one_b)
                by = 4 * M * C + 4 * C * F + 2 * M * F + 2 * M * F + 2 * M * C
                print("bwd  M=%5d k=%d F=%d dropout %.1f: two launches %7.2f us   one launch %7.2f us  (%.0f GB/s at %.1f MB)"
                      % (M, KT, F, p, t2, t1, by / t1 / 1e3, by / 1e6))


if __name__ == "__main__":
    main()
