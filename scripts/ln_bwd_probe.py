"""LayerNorm(128) backward: time per launch against the cap on workgroups (KANTTS_LN_BWD_BLOCKS, read once per process).
Usage: KANTTS_LN_BWD_BLOCKS=256 python scripts/ln_bwd_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402

import kantts._hip as hip  # noqa: E402
from kantts._hip import check, lib, ptr, stream  # noqa: E402
from bgemm_bench import timed  # noqa: E402


def main():
    hip.lib()
    out = []
    for M in (6528, 2048, 19584):
        dy = torch.randn(M, 128, device="cuda").to(torch.bfloat16)
        x, dres, dx = torch.randn(M, 128, device="cuda"), torch.randn(M, 128, device="cuda"), torch.empty(M, 128, device="cuda")
        g, mean, rstd = torch.ones(128, device="cuda"), torch.zeros(M, device="cuda"), torch.ones(M, device="cuda")
        dg, db = torch.zeros(128, device="cuda"), torch.zeros(128, device="cuda")
        t = timed(lambda: check(lib().kantts_ln128_bwd_rows(ptr(dy), 1, ptr(x), ptr(g), ptr(mean), ptr(rstd), ptr(dres), ptr(dx),
                                                          ptr(dg), ptr(db), None, M, stream()), "ln"))
        out.append("M %5d: %.2f us" % (M, t))
    print("blocks cap %s: %s" % (os.environ.get("KANTTS_LN_BWD_BLOCKS", "128 (default)"), " | ".join(out)))


if __name__ == "__main__":
    main()
