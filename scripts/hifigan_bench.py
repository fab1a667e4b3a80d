"""HiFi-GAN V1 (class defaults, 22.05 kHz) measurements: GAN training step, generator forward, transposed-conv
upsampling stack (BASELINE config 3; SURVEY 8d byte/flop counts)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
import torch

import kantts._hip as hip
from kantts.models import model_builder
from kantts.train.gan_step import gan_train_step
from kantts.train.loss import criterion_builder


def v1_config(channels=512):
    opt = {"type": "Adam", "params": {"lr": 2e-4, "betas": [0.5, 0.9], "weight_decay": 0.0}}
    sch = {"type": "MultiStepLR", "params": {"gamma": 0.5, "milestones": [200000, 400000, 600000, 800000]}}
    return {"model_type": "hifigan", "Model": {
        "Generator": {"params": {"channels": channels}, "optimizer": opt, "scheduler": sch},
        "MultiScaleDiscriminator": {"params": {}, "optimizer": opt, "scheduler": sch},
        "MultiPeriodDiscriminator": {"params": {}, "optimizer": opt, "scheduler": sch}},
        "Loss": {"generator_adv_loss": {"enable": True, "params": {}, "weights": 1.0},
                 "discriminator_adv_loss": {"enable": True, "params": {}, "weights": 1.0},
                 "mel_loss": {"enable": True, "params": {}, "weights": 45.0},
                 "feat_match_loss": {"enable": True, "params": {}, "weights": 2.0}},
        "generator_grad_norm": -1, "discriminator_grad_norm": -1, "discriminator_train_start_steps": 0,
        "generator_train_start_steps": 0}


def ev_time(fn, n):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
    hip.set_precision(prec)
    config = v1_config()
    torch.manual_seed(0)
    model, optimizer, scheduler = model_builder(config, device="cuda")
    crit = criterion_builder(config, device="cuda")
    x = torch.randn(B, 80, 32, device="cuda")
    y = torch.randn(B, 1, 8192, device="cuda").clamp(-1, 1)
    res = {"B": B, "precision": prec}
    G = model["generator"]
    with torch.no_grad():
        ms = ev_time(lambda: G(x), 3)
        res["generator_forward_eager_ms"] = ms
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            G(x)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gg, capture_error_mode="thread_local"):
            yg = G(x)
        ms = ev_time(gg.replay, 20)
        del gg, yg
    res["generator_forward_ms"] = ms
    res["generator_forward_samples_per_s"] = B * 8192 / (ms * 1e-3)
    res["generator_forward_tflops"] = 696.5e9 * B / 32 / (ms * 1e-3) / 1e12
    # transposed-conv upsampling stack alone (HBM roofline target): inputs of the four stages
    with torch.no_grad():
        hs, T, C = [], 32, 512
        for s in (8, 8, 2, 2):
            hs.append(torch.randn(B, T, C, device="cuda"))
            T, C = T * s, C // 2

        def up():
            for i, h in enumerate(hs):
                G.transpose_upsamples[i][1].forward_cl(h, in_leaky=0.1)

        ms = ev_time(up, 5)
    elems = 49324032 * B / 32
    res["upsampling_ms"] = ms
    res["upsampling_GBps_fp32_algorithmic"] = elems * 4 / (ms * 1e-3) / 1e9
    res["upsampling_frac_of_8TBps"] = res["upsampling_GBps_fp32_algorithmic"] / 8000.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = gan_train_step(model, optimizer, scheduler, crit, config, y, x, steps=1)
    torch.cuda.synchronize()
    res["first_step_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(steps):
        out = gan_train_step(model, optimizer, scheduler, crit, config, y, x, steps=1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    res["gan_step_ms"] = dt * 1e3
    res["gan_step_samples_per_s"] = B * 8192 / dt
    res["gan_step_tflops"] = 8.3e12 * B / 32 / dt / 1e12
    res["losses"] = {k: float(v.detach()) for k, v in out.items()}
    if not os.environ.get("KANTTS_NO_GAN_GRAPH"):
        from kantts.train.gan_graph_step import GraphedGanStep

        t0 = time.perf_counter()
        gstep = GraphedGanStep(model, optimizer, scheduler, crit, config, y, x)
        torch.cuda.synchronize()
        res["graph_capture_s"] = time.perf_counter() - t0
        gstep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(max(steps, 8)):
            out = gstep()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / max(steps, 8)
        res["gan_step_graph_ms"] = dt * 1e3
        res["gan_step_graph_samples_per_s"] = B * 8192 / dt
        res["gan_step_graph_tflops"] = 8.3e12 * B / 32 / dt / 1e12
        res["graph_losses"] = {k: float(v.detach()) for k, v in out.items()}
    res["max_mem_GB"] = torch.cuda.max_memory_allocated() / 1e9
    print(json.dumps(res))


if __name__ == "__main__":
    main()
