"""How much does a dependent kernel node cost inside a replayed hipGraph?  Chains of N trivial launches (a 1 KB elementwise
op) and of N small-but-real launches (LayerNorm over 6528 x 128) are captured and replayed; time per node is printed.
Usage (GPU box): python scripts/graph_gap_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
import torch  # noqa: E402

import kantts._hip as hip  # noqa: E402
from kantts._hip import ops  # noqa: E402


def timed(fn, n, replays=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (replays * n)


def main():
    hip.lib()
    hip.set_precision("bf16")
    tiny = torch.zeros(256, device="cuda")
    x = torch.randn(6528, 128, device="cuda")
    gam, bet = torch.ones(128, device="cuda"), torch.zeros(128, device="cuda")
    big = torch.randn(6528, 1024, device="cuda")
    with torch.no_grad():
        for n in (100, 600):
            print("chain of %4d: tiny add_ %.2f us/node | LayerNorm 6528x128 %.2f us/node | add_ on 6528x1024 (26.7 MB r+w) %.2f us/node"
                  % (n, timed(lambda: tiny.add_(1.0), n), timed(lambda: ops.layer_norm(x, gam, bet, 1e-6, out_bf16=True), n),
                     timed(lambda: big.add_(1.0), n)))
        # eager (no graph) for comparison
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(600):
            tiny.add_(1.0)
        e1.record()
        torch.cuda.synchronize()
        print("eager chain of 600 tiny add_: %.2f us/launch" % (e0.elapsed_time(e1) * 1e3 / 600))


if __name__ == "__main__":
    main()
