"""bench.py -- SAM-BERT training-step throughput on MI355X (BASELINE.json metric / config).

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A "step" = one full training step of the hot path on one seeded synthetic batch per rank
(B=32 phoneme sequences, T_in=64, ~612 mel frames): forward, the five masked-L1
losses, backward, global-norm clip (1.0), fused Adam, NoamLR -- i.e. Sambert_Trainer.train_step
(reference kantts/train/trainer.py:898-1005) with dropout on as shipped.  Inputs are resident in HBM
before the timed region.  value = valid mel frames of all ranks / max-over-ranks wall time.
Extra objects on the JSON line:
  "roofline"     dominant kernel of the step = the MFMA GEMM on the decoder-FFN contractions.  With fp32 activations in
                 HBM these are bandwidth-bound (57 flop/byte against a machine balance of ~310), so the binding roof
                 is HBM: achieved = algorithmic bytes (A + B + C once, fp32) / mean launch duration, each contraction
                 replayed from a captured hipGraph and timed with HIP events on the replay stream.  The MFMA view
                 (flop/s against the dense bf16 peak) is reported beside it as "mfma_frac".
  "cpu_baseline" the CPU oracle port timed on the host cores on a bounded sample (rank 0, N=1 only).
  "hifigan"      the second half of BASELINE.json's metric (audio-samples/s): HiFi-GAN V1 at batch 32 x 8192 samples --
                 full GAN training step, generator forward, and the transposed-conv upsampling stack against the HBM
                 roofline (rank 0, N=1 only; --no-hifigan skips it).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "kan-tts_amd"), os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}  # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBPS = 8000.0
# mean over the FFN contractions profiled in profiles/r01_gemm_ffn_pmc_v2.txt (34.7 / 35.1 MB vs 30.6 MB algorithmic)
MEASURED_TRAFFIC_BYTES = {"bf16": 34.9e6}


def sambert_yaml_config(cfg):
    return {"model_type": "sambert", "Model": {"KanTtsSAMBERT": {
        "params": cfg,
        "optimizer": {"type": "Adam", "params": {"lr": 0.001, "betas": [0.9, 0.98], "eps": 1.0e-9, "weight_decay": 0.0}},
        "scheduler": {"type": "NoamLR", "params": {"warmup_steps": 4000}}}},
        "grad_norm": 1.0, "batch_size": 32}


def cpu_baseline(cfg, sample_B=8, iters=3, max_threads=16):
    """CPU oracle ("port" of the reference path, pinned against it by tests/golden) fwd+bwd on a
    bounded sample of the same workload; Adam omitted (negligible next to fwd+bwd on CPU)."""
    import torch_oracle as O

    torch.manual_seed(0)
    from kantts.models.sambert.kantts_sambert import KanTtsSAMBERT

    m = KanTtsSAMBERT(dict(cfg))
    P = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    batch = O.synthetic_sambert_batch(B=sample_B, T_in=64, seed=1234)
    frames = int(batch["output_lengths"].sum())
    # the port is many small ops + python LSTM loops: more than ~16 threads only adds fork/join overhead
    cores = min(os.cpu_count() or 1, max_threads)
    torch.set_num_threads(cores)

    def one():
        for p in P.values():
            p.grad = None
        out = O.sambert_forward(P, cfg, **batch)
        O.sambert_losses(out, batch["input_lengths"], batch["output_lengths"], batch["mel_targets"])["total"].backward()

    one()
    t0 = time.time()
    for _ in range(iters):
        one()
    dt = (time.time() - t0) / iters
    return {"value": frames / dt, "unit": "mel-frames/s", "cores": cores, "kind": "port",
            "sample": "oracle/torch_oracle.py fwd+bwd, fp32, dropout off, B=%d of the same seeded batch (%d valid frames), "
                      "%d timed iters, %.2f s/iter" % (sample_B, frames, iters, dt)}


def dominant_gemm_roofline(hip, precision, reps=20, replays=5):
    """Roofline of the dominant kernel: the MFMA GEMM on the six decoder-FFN contractions (forward, dgrad, wgrad
    of Conv1d(128->1024,k=1) and Conv1d(1024->128,k=1) at M = 32*204 decoder tokens; 12 layers each = 31 % of the
    step's GEMM flops).  Each contraction is launched `reps` times inside a captured hipGraph (so the host is out
    of the picture) and timed with HIP events on the replay stream; achieved = algorithmic flops / mean duration."""
    from kantts._hip import gemm, make_seg, ops

    dev = "cuda"
    M, C, F = 32 * 204, 128, 1024
    p = {"fp32": hip.PREC_FP32, "bf16": hip.PREC_BF16}[precision]
    x, h = torch.randn(M, C, device=dev), torch.randn(M, F, device=dev)
    w1, w2 = torch.randn(F, C, device=dev) * 0.05, torch.randn(C, F, device=dev) * 0.05
    yh, yx = torch.empty(M, F, device=dev), torch.empty(M, C, device=dev)
    dw1, dw2 = torch.zeros(F, C, device=dev), torch.zeros(C, F, device=dev)
    cases = {
        "fwd 128->1024": lambda: gemm([make_seg(x, C, 1, w1, C, 1, C)], M, F, yh, F, 1, precision=p),
        "fwd 1024->128": lambda: gemm([make_seg(h, F, 1, w2, F, 1, F)], M, C, yx, C, 1, precision=p),
        "dgrad 128->1024": lambda: gemm([make_seg(h, F, 1, w1, 1, C, F)], M, C, yx, C, 1, precision=p),
        "dgrad 1024->128": lambda: gemm([make_seg(x, C, 1, w2, 1, F, C)], M, F, yh, F, 1, precision=p),
        "wgrad 128->1024": lambda: gemm([make_seg(h, 1, F, x, 1, C, M)], F, C, dw1, C, 1, accumulate=True,
                                        splitk=ops._splitk_for(F, C, M), precision=p),
        "wgrad 1024->128": lambda: gemm([make_seg(x, 1, C, h, 1, F, M)], C, F, dw2, F, 1, accumulate=True,
                                        splitk=ops._splitk_for(C, F, M), precision=p),
    }
    flops = 2.0 * M * C * F
    per = {}
    tot_us = 0.0
    for name, fn in cases.items():
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(replays):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (reps * replays)
        per[name] = round(us, 2)
        tot_us += us
    bytes_per_launch = 4.0 * (M * C + C * F + M * F)  # both operands read once + the output written once, fp32
    return flops * len(cases) / (tot_us * 1e-6) / 1e12, per, flops, bytes_per_launch * len(cases) / (tot_us * 1e-6) / 1e9, \
        bytes_per_launch


def hifigan_v1_config(channels=512):
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


def _event_ms(fn, n):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def hifigan_leg(hip, precision, B=32, T_wav=8192, steps=3):
    """HiFi-GAN V1 (reference hifigan.py class defaults, the hifigan_v1 yaml loss weights) at batch 32:
    GAN training step (generator + MPD + MSD, mel/adv/feature-matching losses, 3 Adam steps), generator forward,
    and the four causal transposed-conv upsampling layers alone (x8, x8, x2, x2; SURVEY 8d: algorithmic traffic =
    input + output activations, fp32)."""
    from kantts.models import model_builder
    from kantts.train.gan_step import gan_train_step
    from kantts.train.loss import criterion_builder

    hip.set_precision(precision)
    config = hifigan_v1_config()
    torch.manual_seed(0)
    model, optimizer, scheduler = model_builder(config, device="cuda")
    crit = criterion_builder(config, device="cuda")
    frames = T_wav // 256
    x = torch.randn(B, 80, frames, device="cuda")
    y = torch.randn(B, 1, T_wav, device="cuda").clamp(-1, 1)
    G = model["generator"]
    res = {"workload": "HiFi-GAN V1 (512 ch, up 8/8/2/2, MPD 2/3/5/7/11, MSD x3), batch %d x %d samples" % (B, T_wav),
           "dtype": precision}
    with torch.no_grad():
        ms = _event_ms(lambda: G(x), 5)
        res["generator_forward_ms"] = ms
        res["generator_forward_samples_per_s"] = B * T_wav / (ms * 1e-3)
        hs, T, C, elems, flops = [], frames, 512, 0, 0.0
        for s_ in (8, 8, 2, 2):
            hs.append(torch.randn(B, T, C, device="cuda"))
            elems += B * T * C + B * T * s_ * (C // 2)
            flops += 2.0 * B * T * C * (C // 2) * 2 * s_
            T, C = T * s_, C // 2

        def up():
            for i, h in enumerate(hs):
                G.transpose_upsamples[i][1].forward_cl(h, in_leaky=0.1)

        ms = _event_ms(up, 10)
        per_stage = []
        for i, h in enumerate(hs):
            per_stage.append(round(_event_ms(lambda: G.transpose_upsamples[i][1].forward_cl(h, in_leaky=0.1), 10) * 1e3, 1))
    gbps = elems * 4 / (ms * 1e-3) / 1e9
    res["upsampling"] = {"ms": ms, "bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                         "frac": gbps / PEAK_HBM_GBPS, "algorithmic_bytes": elems * 4, "tflops": flops / (ms * 1e-3) / 1e12,
                         "stage_us": per_stage,
                         "note": "4 launches (one polyphase GEMM per layer); at fp32 storage the first two layers are "
                                 "MFMA-bound (410 / 200 flop per byte), the last two HBM-bound"}
    out = gan_train_step(model, optimizer, scheduler, crit, config, y, x, steps=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = gan_train_step(model, optimizer, scheduler, crit, config, y, x, steps=1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    res["gan_step_ms"] = dt * 1e3
    res["gan_step_samples_per_s"] = B * T_wav / dt
    res["value"] = res["gan_step_samples_per_s"]
    res["unit"] = "audio-samples/s (GAN training step)"
    res["losses"] = {k: float(v) for k, v in out.items()}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hifigan", action="store_true")
    ap.add_argument("--no-wgrad-overlap", action="store_true", help="weight gradients on the main stream (A/B switch)")
    ap.add_argument("--mode", default="graph", choices=["graph", "eager"],
                    help="graph: whole step captured once in a hipGraph and replayed; eager: launch per op")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", init_method="env://", device_id=dev)

    import kantts._hip as hip
    import kantts._hip.ops  # noqa: F401
    from kantts.models import model_builder
    from kantts.utils import synthetic
    from kantts.train.loss import MelReconLoss, ProsodyReconLoss

    hip.lib()
    hip.set_precision(args.precision)
    cfg = synthetic.sambert_16k_config()
    torch.manual_seed(0)
    model, opt, sch = model_builder(sambert_yaml_config(cfg), device=dev, rank=local_rank, distributed=distributed)
    net, optimizer, scheduler = model["KanTtsSAMBERT"], opt["KanTtsSAMBERT"], sch["KanTtsSAMBERT"]
    optimizer.set_grad_clip(1.0)
    net.train()
    mel_crit, pros_crit = MelReconLoss(), ProsodyReconLoss()
    batch = {k: v.to(dev) for k, v in synthetic.sambert_batch(B=args.batch, T_in=64, seed=1234 + rank).items()}
    frames = int(batch["output_lengths"].sum())

    def eager_step():
        hip.ops.advance_rng(dev)
        optimizer.zero_grad()
        res = net(**batch)
        mel_, mel = mel_crit(batch["output_lengths"], batch["mel_targets"], res["dec_outputs"], res["postnet_outputs"])
        d, p, e = pros_crit(batch["input_lengths"], res["duration_targets"], res["pitch_targets"],
                            res["energy_targets"], res["log_duration_predictions"], res["pitch_predictions"],
                            res["energy_predictions"])
        loss = mel_ + mel + d + p + e
        loss.backward()
        optimizer.step()
        scheduler.step()
        return loss

    mode = args.mode
    step = eager_step
    if mode == "graph":
        try:
            from kantts.train.graph_step import GraphedSambertStep

            step = GraphedSambertStep(net, optimizer, scheduler, mel_crit, pros_crit, batch,
                                      overlap_wgrad=not args.no_wgrad_overlap)
            step()  # first replay (and, data-parallel, the first all-reduce between the two graphs) inside the guard
            torch.cuda.synchronize()
        except Exception as exc:  # capture is an optimisation; say so loudly and measure the eager path
            print("[bench] hipGraph capture failed (%s: %s); falling back to eager launches" % (
                type(exc).__name__, str(exc)[:300]), file=sys.stderr)
            mode = "eager"
            step = eager_step
            hip.ops.wgrad_overlap.enable(False)
            net.device_band_width = False
            optimizer.dyn = None

    for _ in range(args.warmup):
        step()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt, float(frames)], device=dev, dtype=torch.float64)
    if distributed:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_max, total_frames = float(tmax[0]), float(tsum[1])
    else:
        dt_max, total_frames = dt, float(frames)

    # ---- roofline of the dominant kernel: HIP events around every GEMM launch of one extra step
    roof = None
    if rank == 0:
        hip.profile_begin()
    net.device_band_width = False
    hip.ops.wgrad_overlap.enable(False)
    eager_step()  # instrumented launches are issued eagerly on every rank (the step holds a collective)
    torch.cuda.synchronize()
    if rank == 0:
        prof = hip.profile_end()
        tf, per_launch_us, flops_per_launch, gbps, bytes_per_launch = dominant_gemm_roofline(hip, args.precision)
        peak = PEAK_TFLOPS[args.precision]
        roof = {"bound": "hbm", "kernel": "gemm_fast_kernel<%s> (decoder FFN contractions, M=6528, 128<->1024)" % args.precision,
                "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS,
                # memory-side bytes per launch from rocprofv3 PMC passes of the same contractions (2*FETCH_SIZE +
                # WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md); collected offline, see
                # profiles/r01_gemm_ffn_pmc_v2.txt -- bench.py cannot run the profiler on itself
                "traffic": MEASURED_TRAFFIC_BYTES.get(args.precision),
                "bytes_per_launch": bytes_per_launch, "flops_per_launch": flops_per_launch, "launch_us": per_launch_us,
                "mfma_tflops": tf, "mfma_peak": peak, "mfma_frac": tf / peak,
                "gemm_launches_per_step": prof["launches"], "gemm_gflop_per_step": prof["flops"] / 1e9,
                "gemm_ms_per_step_eager_events": prof["ms"],
                "whole_step_algorithmic_tflops": 456.9e9 * args.batch / 32 / (dt_max / args.steps) / 1e12}

    if rank == 0:
        out = {
            "metric": "mel-frames/sec (SAM-BERT train)", "value": total_frames * args.steps / dt_max,
            "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt_max / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "SAM-BERT full (sambert_16k.yaml zhcn) fwd+bwd+clip+Adam, batch %d/GPU, T_in 64, "
                                   "%d valid mel frames on rank 0, dropout on" % (args.batch, frames),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world, "launch": mode,
                       "final_loss": float(loss)},
            "roofline": roof,
        }
        if world == 1 and not args.no_hifigan:
            try:
                del step, net, optimizer, model, opt
                torch.cuda.empty_cache()
                out["hifigan"] = hifigan_leg(hip, args.precision)
            except Exception as exc:  # the SAM-BERT line must survive a failure of the secondary leg
                out["hifigan"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
