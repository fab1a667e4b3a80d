"""Debug helper: graph vs eager SAM-BERT step with identical dropout seeds (see tests/test_trainer.py)."""
import itertools, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
import torch
import kantts._hip as hip
from kantts._hip import ops
from kantts.models import model_builder
from kantts.train.graph_step import GraphedSambertStep
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

def build():
    torch.manual_seed(0)
    model, opt, sch = model_builder(yaml_cfg, device=dev)
    net, o, s = model["KanTtsSAMBERT"], opt["KanTtsSAMBERT"], sch["KanTtsSAMBERT"]
    o.set_grad_clip(1.0); net.train()
    return net, o, s

def eager_step(net, o, s):
    ops.advance_rng(dev)
    o.zero_grad()
    res = net(**batch)
    mel_, mel = mel_crit(batch["output_lengths"], batch["mel_targets"], res["dec_outputs"], res["postnet_outputs"])
    d, p, e = pros_crit(batch["input_lengths"], res["duration_targets"], res["pitch_targets"], res["energy_targets"],
                        res["log_duration_predictions"], res["pitch_predictions"], res["energy_predictions"])
    loss = mel_ + mel + d + p + e
    loss.backward(); o.step(); s.step()
    return float(loss.detach())

rng0 = hip.rng_state(dev).clone()
def eager_run(start):
    net, o, s = build()
    hip.rng_state(dev).copy_(rng0)
    out = []
    for _ in range(3):
        ops._seed_counter = itertools.count(start)
        out.append(eager_step(net, o, s))
    return out, o.arena.flat.detach().clone()

def graph_run():
    ops._seed_counter = itertools.count(1)
    net, o, s = build()
    step = GraphedSambertStep(net, o, s, mel_crit, pros_crit, batch, warmup=1)
    used = next(ops._seed_counter) - 1
    hip.rng_state(dev).copy_(rng0)
    out = [float(step().detach()) for _ in range(3)]
    return out, o.arena.flat.detach().clone(), used

for tag, env in (("graph, attention streams", {}), ("graph, no attention streams", {"KANTTS_NO_ATTN_STREAMS": "1"})):
    for k, v in env.items(): os.environ[k] = v
    gl, gf, used = graph_run()
    for k in env: os.environ.pop(k)
    print(tag, "used", used, gl)
    os.environ["KANTTS_NO_ATTN_STREAMS"] = "1"
    el, ef = eager_run(used // 2 + 1)
    el2, ef2 = eager_run(used // 2 + 1)
    os.environ.pop("KANTTS_NO_ATTN_STREAMS")
    print("  eager", el, "repeat equal:", el == el2, float((ef - ef2).norm() / ef.norm()))
    print("  graph-eager rel", float((gf - ef).norm() / ef.norm()))
    el3, ef3 = eager_run(1)
    print("  eager with warm-up seeds", el3)
