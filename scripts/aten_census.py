"""Which stock ATen operators one SAM-BERT training step (bf16 mode, eager) still launches, and from where: a
TorchDispatchMode records every operator with its first stack frame inside kan-tts_amd/ (built-in autograd nodes have
none).  Usage (GPU box): python scripts/aten_census.py"""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

import kantts._hip as hip  # noqa: E402
from kantts._hip import ops  # noqa: E402

SKIP = ("aten.view", "aten._unsafe_view", "aten.detach", "aten.alias", "aten.t.", "aten.transpose", "aten.permute",
        "aten.expand", "aten.slice", "aten.select", "aten.unsqueeze", "aten.squeeze", "aten.as_strided", "aten.reshape",
        "aten.empty", "aten.new_empty", "aten.empty_like", "aten.empty_strided", "aten.unbind", "aten.split",
        "aten._local_scalar_dense", "aten.lift_fresh", "aten.is_", "aten.sym_", "aten.set_", "aten.chunk", "aten.narrow",
        "aten.unfold", "aten.view_as", "aten.resize_")


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.count = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            where = "(autograd / library)"
            for fr in reversed(traceback.extract_stack(limit=30)):
                if "kan-tts_amd" in fr.filename and "aten_census" not in fr.filename:
                    where = "%s:%d %s" % (fr.filename.split("kan-tts_amd/")[-1], fr.lineno, fr.name)
                    break
            shape = next((tuple(a.shape) for a in args if torch.is_tensor(a)), ())
            self.count[(name, where, shape if len(shape) < 4 else shape[:4])] += 1
        return func(*args, **(kwargs or {}))


def main():
    import bench
    import torch_oracle as O
    from kantts.models import model_builder
    from kantts.train.loss import MelReconLoss, ProsodyReconLoss

    hip.set_precision("bf16")
    cfg = O.sambert_config(tiny=False)
    torch.manual_seed(1234)
    model, opt, _ = model_builder(bench.sambert_yaml_config(cfg), device="cuda")
    net, optimizer = model["KanTtsSAMBERT"], opt["KanTtsSAMBERT"]
    net.train()
    batch = {k: v.cuda() for k, v in O.synthetic_sambert_batch(B=32, T_in=64, seed=1234).items()}
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

    ops.wgrad_overlap.enable(False)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with Census() as c:
        step()
    torch.cuda.synchronize()
    tot = sum(c.count.values())
    print("%d ATen operator calls that launch work (views / allocations not counted)" % tot)
    by_op = collections.Counter()
    for (name, where, shape), n in c.count.items():
        by_op[name] += n
    print("by operator:", ", ".join("%s x%d" % (k.replace("aten.", ""), v) for k, v in by_op.most_common(25)))
    for (name, where, shape), n in c.count.most_common(70):
        print("x%-4d %-34s %-70s %s" % (n, name.replace("aten.", ""), where, shape))


if __name__ == "__main__":
    main()
