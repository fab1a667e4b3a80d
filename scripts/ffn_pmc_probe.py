"""The fused feed-forward launches (csrc/ffn_pair.hip) and the six bf16 contractions bench.py's roofline leg times (M=6528, 128<->1024), each launched 30 times
outside any graph so that `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) sees every dispatch.
Distinct output buffers are rotated so the Infinity Cache does not hide the writes of the previous launch.
Usage (GPU box): rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -o pmc -- python scripts/ffn_pmc_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
import torch  # noqa: E402

import kantts._hip as hip  # noqa: E402
from kantts._hip import bgemm_nt, bgemm_tn, ffn_pair, ops  # noqa: E402
from kantts._hip.ops_bf16 import frag_major  # noqa: E402


def main():
    hip.lib()
    ops.wgrad_overlap.enable(False)
    dev, bf = "cuda", torch.bfloat16
    M, C, F = 32 * 204, 128, 1024
    xb, hb = torch.randn(M, C, device=dev).to(bf), torch.randn(M, F, device=dev).relu().to(bf)
    w1b, w2b = (torch.randn(F, C, device=dev) * 0.05).to(bf), (torch.randn(C, F, device=dev) * 0.05).to(bf)
    b1, b2 = torch.zeros(F, device=dev), torch.zeros(C, device=dev)
    res, dy = torch.randn(M, C, device=dev), torch.randn(M, C, device=dev)
    yh, dz = torch.empty(M, F, device=dev, dtype=bf), torch.empty(M, F, device=dev, dtype=bf)
    yx, dh = torch.empty(M, C, device=dev), torch.empty(M, C, device=dev, dtype=bf)
    dw1, dw2 = torch.zeros(F, C, device=dev), torch.zeros(C, F, device=dev)
    marker = torch.zeros(1, device=dev)
    f1, f2 = frag_major(w1b.float()), frag_major(w2b.float())
    t2, t1 = frag_major(w2b.float().t().contiguous()), frag_major(w1b.float().t().contiguous())

    def pair_fwd():
        assert ffn_pair(xb, f1, f2, yx, M=M, T=204, F=F, bias1=b1, bias2=b2, relu=True, t_out=yh, res=res)

    def pair_bwd():
        assert ffn_pair(dy, t2, t1, dh, M=M, T=204, F=F, gate=hb, t_out=dz)

    cases = [
        pair_fwd, pair_bwd,
        lambda: bgemm_nt([(xb, C, w1b, C, C, 0)], M, F, yh, F, bias=b1, relu=True),
        lambda: bgemm_nt([(hb, F, w2b, F, F, 0)], M, C, yx, C, bias=b2, res=res, ldr=C),
        lambda: bgemm_nt([(dy, C, w2b, F, C, 0)], M, F, dz, F, b_kn=True, gate=hb, ldg=F),
        lambda: bgemm_nt([(dz, F, w1b, C, F, 0)], M, C, dh, C, b_kn=True),
        lambda: bgemm_tn(dy, C, hb, F, M, C, F, dw2, F, 1),
        lambda: bgemm_tn(dz, F, xb, C, M, F, C, dw1, C, 1),
    ]
    for i, fn in enumerate(cases):
        for _ in range(30):
            fn()
        # a recognisable separator between the cases in the dispatch list: (i + 1) tiny fills
        for _ in range(i + 1):
            marker.fill_(float(i))
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
