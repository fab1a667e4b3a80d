"""Device diagnosis for tests/test_gpu_sambert.py::test_wgrad_side_stream_gives_the_same_gradients: repeats the test's two
backward passes (weight gradients on the main stream / on the side stream) and says which gradient tensors differ, and where."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("kan-tts_amd", "oracle", "tests", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import torch_oracle as O
import kantts._hip as hip
from kantts._hip import ops
from test_gpu_sambert import _build, _losses

hip.set_precision("fp32")
cfg = O.sambert_config(tiny=True)
batch = {k: v.cuda() for k, v in O.synthetic_sambert_batch(B=3, T_in=12, seed=5, min_len=6, dur_hi=6).items()}
ref = None
for rep in range(10):
    for on in (False, True):
        m, _ = _build(cfg)
        ops.wgrad_overlap.enable(on)
        try:
            _losses(m(**batch), batch).backward()
            ops.wgrad_overlap.join()
            torch.cuda.synchronize()
        finally:
            ops.wgrad_overlap.enable(False)
        g = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
        if ref is None:
            ref = g
            continue
        bad = []
        for n in ref:
            d = (g[n] - ref[n]).norm() / (ref[n].norm() + 1e-30)
            if float(d) > 1e-5:
                e = (g[n] - ref[n]).abs()
                idx = (e > 1e-6 * float(ref[n].abs().max())).nonzero()
                bad.append("%s rel %.3g, %d of %d elements, first %s last %s" % (n, float(d), idx.shape[0], e.numel(),
                                                                               idx[0].tolist(), idx[-1].tolist()))
        print("rep %d side=%s: %s" % (rep, on, "all equal" if not bad else "; ".join(bad)), flush=True)
