"""TIMING EXPERIMENT ONLY: how much of the SAM-BERT step's critical path a group of launches accounts for.

Named C-ABI entry points are replaced by no-ops (their outputs stay uninitialised, so every number the step computes is
garbage -- only the step time means anything), then bench.py's SAM-BERT leg runs as usual.  ``masked_fill`` can be
ablated too (the ATen row-mask passes of the backward).  The difference to the un-ablated time is the upper bound of what
fusing those launches into their neighbours can gain.

Usage (GPU box): python scripts/ablate_bench.py kantts_ln128_fwd,kantts_ln128_bwd [bench.py arguments]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
import torch  # noqa: E402


def main():
    names = [n for n in sys.argv[1].split(",") if n]
    sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
    import kantts._hip as hip
    import kantts._hip.ops as ops
    import kantts._hip.ops_bf16 as ops_bf16

    real = hip.lib()

    class Ablated:
        def __getattr__(self, name):
            f = getattr(real, name)
            if name in names:
                return lambda *a, **k: 0
            return f

    proxy = Ablated()
    for mod in (hip, ops, ops_bf16):
        mod.lib = lambda: proxy
    if "masked_fill" in names:
        orig = torch.Tensor.masked_fill
        # only the (tokens, 128) row-mask passes of the backward, not the small forward masks that steer band widths
        torch.Tensor.masked_fill = lambda self, mask, value: (
            self if (self.dim() == 2 and self.shape[1] == 128 and self.shape[0] > 512) else orig(self, mask, value))
    import bench

    bench.main()


if __name__ == "__main__":
    main()
