"""Per-shape table of the HiFi-GAN conv launches inside one GAN training step (V1, B=32 x 8192, bf16 MFMA):
count, total ms, achieved TFLOP/s and algorithmic GB/s -- the map of where the step's conv time goes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch

import kantts._hip as hip
from hifigan_bench import v1_config
from kantts.models import model_builder
from kantts.train.gan_step import gan_train_step
from kantts.train.loss import criterion_builder

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
hip.set_precision("bf16")
config = v1_config()
torch.manual_seed(0)
model, optimizer, scheduler = model_builder(config, device="cuda")
crit = criterion_builder(config, device="cuda")
x = torch.randn(B, 80, 32, device="cuda")
y = torch.randn(B, 1, 8192, device="cuda").clamp(-1, 1)
for _ in range(2):
    gan_train_step(model, optimizer, scheduler, crit, config, y, x, steps=1)
torch.cuda.synchronize()
hip.profile_begin()
gan_train_step(model, optimizer, scheduler, crit, config, y, x, steps=1)
tab = hip.profile_end_by_shape()
rows = sorted(tab.items(), key=lambda kv: -kv[1]["ms"])
tot = sum(v["ms"] for v in tab.values())
print("conv launches total %.2f ms over %d shapes" % (tot, len(rows)))
print("%7s %4s %8s %8s %8s  shape" % ("ms", "n", "us/call", "TFLOP/s", "GB/s"))
for k, v in rows[:int(os.environ.get('ROWS', '60'))]:
    us = v["ms"] / v["n"] * 1e3
    print("%7.2f %4d %8.1f %8.1f %8.0f  %s" % (v["ms"], v["n"], us, v["flops"] / (us * 1e-6) / 1e12,
                                                v["bytes"] / (us * 1e-6) / 1e9, k))
