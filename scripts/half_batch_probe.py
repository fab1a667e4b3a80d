"""Timing probe (numbers only, the arithmetic of the pair is NOT a training step): would two half-batch chains on two
streams finish sooner than one full-batch chain?  Two independent SAM-BERT models + optimizers at batch B/2, each captured
as its own hipGraph, replayed concurrently on two streams, against one model at batch B."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
sys.path.insert(0, ROOT)
import torch

import bench as BN
import kantts._hip as hip
import kantts._hip.ops  # noqa: F401
from kantts.models import model_builder
from kantts.train.graph_step import GraphedSambertStep
from kantts.train.loss import MelReconLoss, ProsodyReconLoss
from kantts.utils import synthetic

hip.set_precision("bf16")
dev = torch.device("cuda", 0)
cfg = synthetic.sambert_16k_config()


def build(B, seed):
    torch.manual_seed(0)
    model, opt, sch = model_builder(BN.sambert_yaml_config(cfg), device=dev, rank=0, distributed=False)
    net, optimizer, scheduler = model["KanTtsSAMBERT"], opt["KanTtsSAMBERT"], sch["KanTtsSAMBERT"]
    optimizer.set_grad_clip(1.0)
    net.train()
    full = synthetic.sambert_batch(B=32, T_in=64, seed=1234)
    sl = slice(0, B) if seed == 0 else slice(32 - B, 32)
    batch = {k: v[sl].contiguous().to(dev) for k, v in full.items()}
    step = GraphedSambertStep(net, optimizer, scheduler, MelReconLoss(), ProsodyReconLoss(), batch)
    return step


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


full = build(32, 0)
t_full = timeit(full)
print("one chain, batch 32: %.3f ms" % t_full)
del full
a, b = build(16, 0), build(16, 1)
t_a = timeit(a)
print("one chain, batch 16 alone: %.3f ms" % t_a)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def pair():
    with torch.cuda.stream(s1):
        a()
    with torch.cuda.stream(s2):
        b()


t_pair = timeit(pair)
print("two chains of batch 16 on two streams: %.3f ms  (%.2f x the batch-32 chain)" % (t_pair, t_pair / t_full))


def serial():
    a()
    b()


print("the same two chains one after the other: %.3f ms" % timeit(serial))
