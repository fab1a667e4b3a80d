"""Decoder-FFN contractions (M = 6528, 128 <-> 1024, bf16 MFMA) timed inside a captured hipGraph (host out of the picture) --
run once per ablation mask (KANTTS_GEMM_DBG with the GDBG experiment build: 1 no operand loads, 2 no output stores, 8 no MFMA)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
sys.path.insert(0, ROOT)
import torch

import kantts._hip as hip
import bench

hip.set_precision("bf16")
tf, per, flops, gbps, nbytes = bench.dominant_gemm_roofline(hip, "bf16")
y, h = torch.empty(6528, 1024, device="cuda"), torch.randn(6528, 1024, device="cuda")
g = torch.cuda.CUDAGraph()
y.copy_(h)
torch.cuda.synchronize()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    for _ in range(20):
        y.copy_(h)
g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    g.replay()
e1.record()
torch.cuda.synchronize()
print("mask=%s  %s | copy 2x26.7MB %.1f us" % (os.environ.get("KANTTS_GEMM_DBG", "0"), per, e0.elapsed_time(e1) * 10))
