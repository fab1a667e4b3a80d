"""Does one SAM-BERT training step read memory it never wrote?  Step-1 loss from identical weights / seeds, before and after
poisoning the caching allocator's free blocks with NaN / large values."""
import itertools, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
import torch
import kantts._hip as hip
from kantts._hip import ops
from kantts.models import model_builder
from kantts.train.loss import MelReconLoss, ProsodyReconLoss
from kantts.utils import synthetic

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
hip.set_precision(prec)
dev = torch.device("cuda")
cfg = synthetic.sambert_16k_config()
yaml_cfg = {"model_type": "sambert", "Model": {"KanTtsSAMBERT": {"params": cfg,
    "optimizer": {"type": "Adam", "params": {"lr": 0.001, "betas": [0.9, 0.98], "eps": 1.0e-9, "weight_decay": 0.0}},
    "scheduler": {"type": "NoamLR", "params": {"warmup_steps": 4000}}}}, "grad_norm": 1.0, "batch_size": 32}
batch = {k: v.to(dev) for k, v in synthetic.sambert_batch(B=32, T_in=64, seed=1234).items()}
mel_crit, pros_crit = MelReconLoss(), ProsodyReconLoss()
rng0 = hip.rng_state(dev).clone()

def one(poison=None, train=True):
    torch.manual_seed(0)
    model, opt, sch = model_builder(yaml_cfg, device=dev)
    net, o = model["KanTtsSAMBERT"], opt["KanTtsSAMBERT"]
    net.train(train)
    if poison is not None:
        junk = [torch.full((64 << 20,), poison, device=dev) for _ in range(6)]
        del junk
    hip.rng_state(dev).copy_(rng0)
    ops._seed_counter = itertools.count(1)
    ops.advance_rng(dev)
    o.zero_grad()
    res = net(**batch)
    mel_, mel = mel_crit(batch["output_lengths"], batch["mel_targets"], res["dec_outputs"], res["postnet_outputs"])
    d, p, e = pros_crit(batch["input_lengths"], res["duration_targets"], res["pitch_targets"], res["energy_targets"],
                        res["log_duration_predictions"], res["pitch_predictions"], res["energy_predictions"])
    loss = mel_ + mel + d + p + e
    loss.backward()
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in net.parameters() if p.grad is not None)))
    return [float(v.detach()) for v in (mel_, mel, d, p, e)], gn

for train in (False, True):
    print("train" if train else "eval (dropout off)")
    for poison in (None, 0.0, float("nan"), 1e30, None):
        print("   poison", poison, one(poison, train))
