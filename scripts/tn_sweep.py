"""Grouped weight-gradient launches of one SAM-BERT step (shapes at batch 32) over output tiles and token slices.
KANTTS_TN_TILE / KANTTS_TN_SLICES are read per launch by the library.  Usage (GPU box): python scripts/tn_sweep.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
import torch

import kantts._hip as hip

bf = torch.bfloat16
# (M, N, K, problems, A fp32, B fp32, taps, T)
SHAPES = [(19584, 256, 512, 4, 1, 0, 1, 0), (19584, 512, 256, 3, 0, 1, 1, 0), (19584, 512, 256, 1, 1, 1, 1, 0),
          (19584, 512, 128, 1, 1, 1, 1, 0), (6528, 128, 1024, 12, 1, 0, 1, 0), (6528, 1024, 128, 12, 0, 0, 1, 0),
          (6528, 128, 128, 16, 1, 1, 1, 0), (6528, 384, 128, 12, 1, 0, 1, 0), (6528, 256, 160, 12, 1, 1, 1, 0),
          (2048, 128, 1024, 8, 1, 0, 1, 0), (2048, 1024, 128, 8, 0, 0, 3, 64), (2048, 384, 128, 7, 1, 0, 1, 0)]


def run_group(ops_):
    hip.deferred_tn.enabled = True
    for (a, b, c, M, N, K, taps, T) in ops_:
        assert hip.bgemm_tn(a, N, b, K, M, N, K, c, K * taps, taps, c_ts=1, T=T, ntaps=taps, shift0=-(taps // 2),
                            shift_step=1 if taps > 1 else 0)
    hip.deferred_tn.flush()
    hip.deferred_tn.enabled = False


def ev(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (M, N, K, n, a32, b32, taps, T) in SHAPES:
    probs = []
    for _ in range(n):
        a = torch.randn(M, N, device="cuda")
        b = torch.randn(M, K, device="cuda")
        probs.append((a if a32 else a.to(bf), b if b32 else b.to(bf), torch.zeros(N, K * taps, device="cuda"), M, N, K, taps, T))
    uniq = n * M * (N * (4 if a32 else 2) + K * (4 if b32 else 2)) / 1e6
    line = "M %5d N %4d K %4d x%2d %s%s taps %d (%.0f MB):" % (M, N, K, n, "f" if a32 else "b", "f" if b32 else "b", taps, uniq)
    best = None
    # code + 1 = the same tile on the 3-D grid of rounds 2-5 (without the XCD-aware workgroup mapping of round 6)
    for tile in (64128, 64129, 128128, 128129, 128256, 128257):
        os.environ["KANTTS_TN_TILE"] = str(tile)
        res = []
        for sl in (0, 1, 2, 3, 4, 6, 8, 12):
            if sl:
                os.environ["KANTTS_TN_SLICES"] = str(sl)
            else:
                os.environ.pop("KANTTS_TN_SLICES", None)
            t = ev(lambda: run_group(probs))
            res.append((sl, t))
            if best is None or t < best[0]:
                best = (t, tile, sl)
        line += "\n    tile %6d  " % tile + "  ".join("s%d %.0f" % r for r in res)
    os.environ.pop("KANTTS_TN_SLICES", None)
    print(line)
    print("    best: %.0f us  tile %d  slices %s   (%.2f TB/s of unique bytes)" % (best[0], best[1], best[2] or "auto",
                                                                                  uniq / best[0]))
