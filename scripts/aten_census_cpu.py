"""Launch census of one SAM-BERT training step WITHOUT a GPU: the product's Python layer runs on the emulated C ABI
(oracle/cabi_numpy.py, test infrastructure), every C-ABI call and every stock ATen operator the step issues is counted
(operators executed inside the emulation itself are filtered out).  Counts do not depend on the batch size, so a batch
of 2 is used (B=...).  Eager order, weight gradients ungrouped: in the captured step the ~148 kantts_bgemm_tn calls become
~15 grouped launches and the per-parameter gradient copies one multi-tensor launch.  The 381 ``zeros`` of ops.py:take are
the accumulator pool, which is one memset on the device.
Usage: python scripts/aten_census_cpu.py            (PREC=fp32 for the parity path)"""
import collections, os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("kan-tts_amd", "oracle", "tests", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import conftest

class MP:
    def setattr(self, mod, name, val, raising=True):
        setattr(mod, name, val)

emu = conftest._emulate(MP())
CALLS = collections.Counter()
class Proxy:
    def __getattr__(self, name):
        f = getattr(emu, name)
        if not callable(f): return f
        def w(*a, **k):
            CALLS[name] += 1
            return f(*a, **k)
        return w
proxy = Proxy()
import kantts._hip as _h, kantts._hip.ops as _o, kantts._hip.ops_bf16 as _ob
for m in (_h, _o, _ob):
    m.lib = lambda: proxy
import kantts._hip as hip
from kantts._hip import ops
sys.argv = ["x"]
import importlib.util
spec = importlib.util.spec_from_file_location("aten_census", os.path.join(ROOT, "scripts/aten_census.py"))
ac = importlib.util.module_from_spec(spec); spec.loader.exec_module(ac)
import bench, torch_oracle as O
from kantts.models import model_builder
from kantts.train.loss import MelReconLoss, ProsodyReconLoss
prec = os.environ.get("PREC", "bf16")
hip.set_precision(prec)
cfg = O.sambert_config(tiny=False)
torch.manual_seed(1234)
from kantts.models import sambert_model_builder
model, opt, _ = sambert_model_builder(bench.sambert_yaml_config(cfg), "cpu", 0, False, use_arena=True)
net, optimizer = model["KanTtsSAMBERT"], opt["KanTtsSAMBERT"]
net.train()
B = int(os.environ.get("B", "2"))
batch = {k: v for k, v in O.synthetic_sambert_batch(B=B, T_in=64, seed=1234).items()}
mel_crit, pros_crit = MelReconLoss(), ProsodyReconLoss()
def step():
    optimizer.zero_grad()
    res = net(**batch)
    a, b = mel_crit(batch["output_lengths"], batch["mel_targets"], res["dec_outputs"], res["postnet_outputs"])
    d, p, e = pros_crit(batch["input_lengths"], res["duration_targets"], res["pitch_targets"], res["energy_targets"],
                        res["log_duration_predictions"], res["pitch_predictions"], res["energy_predictions"])
    (a + b + d + p + e).backward()
    ops.wgrad_overlap.join()
    optimizer.step()
t=time.time(); step(); print("step", time.time()-t, type(optimizer).__name__)
class Census2(ac.Census):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        st = traceback.extract_stack(limit=40)
        if any("cabi_numpy" in fr.filename for fr in st):
            return func(*args, **(kwargs or {}))
        return super().__torch_dispatch__(func, types, args, kwargs)
CALLS.clear()
with Census2() as c:
    step()
print("C-ABI calls:", sum(CALLS.values())); print(CALLS.most_common())
tot = sum(c.count.values())
print("%d ATen operator calls" % tot)
by_op = collections.Counter()
for (name, where, shape), n in c.count.items():
    by_op[name] += n
print(", ".join("%s x%d" % (k.replace("aten.", ""), v) for k, v in by_op.most_common()))
for (name, where, shape), n in sorted(c.count.items(), key=lambda kv: -kv[1]):
    print("x%-4d %-34s %-70s %s" % (n, name.replace("aten.", ""), where, shape))
