"""Device diagnosis for tests/test_trainer.py::test_direct_gradients_equal_packed_gradients_gpu: gradient differences between
a direct-gradient trainer (a) and two packing trainers (b, c) fed the same batches; b-c is the run-to-run noise floor."""
import sys, tempfile, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "kan-tts_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import kantts._hip as hip
from test_trainer import _sambert_setup

hip.set_precision("fp32")
d = tempfile.mkdtemp()
tr = {k: _sambert_setup("cuda", os.path.join(d, k), use_arena=True) for k in "abc"}
batches = tr["a"][1]
tr = {k: v[0] for k, v in tr.items()}
ar = {k: t.optimizer["KanTtsSAMBERT"].arena for k, t in tr.items()}
ar["a"].enable_direct_grads()
names = [n for n, _ in tr["a"].model["KanTtsSAMBERT"].named_parameters()]
for k, b in enumerate(batches):
    losses = {x: float(t.train_step(b)) for x, t in tr.items()}
    gmax = float(ar["b"].grad.abs().max())
    for x, y in (("a", "b"), ("b", "c")):
        diff = (ar[x].grad - ar[y].grad).abs()
        worst, wn = 0.0, None
        for n, vx, vy in zip(names, ar[x].grad_views, ar[y].grad_views):
            e = float((vx - vy).abs().max())
            if e > worst:
                worst, wn = e, (n, float(vy.abs().max()))
        wdiff = max(float((p - q).abs().max()) for p, q in zip(tr[x].model["KanTtsSAMBERT"].parameters(),
                                                              tr[y].model["KanTtsSAMBERT"].parameters()))
        print("step %d  %s-%s  grad max|diff| %.3e (max |g| %.3e)  worst %s  weights max|diff| %.3e  losses %r"
              % (k, x, y, float(diff.max()), gmax, wn, wdiff, losses))
