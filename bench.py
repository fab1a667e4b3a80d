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
                 "cold_cache": the same launches over ten disjoint buffer sets (every operand from HBM); "traffic": memory-side
                 bytes per launch from this round's rocprofv3 --pmc passes, read from profiles/ffn_block_pmc.json
                 (scripts/ffn_pmc_probe.py + scripts/pmc_to_json.py); "forward_ms" / "forward_mfma_frac": the forward pass alone.
  "cpu_baseline" the CPU oracle port timed on the host cores on a bounded sample (rank 0, N=1 only): >= 5 timed iterations at
                 the default thread count + "all_cores": one attempt at every host core in a time-limited child process.
  "hifigan"      the second half of BASELINE.json's metric (audio-samples/s): HiFi-GAN V1 at batch 32 x 8192 samples --
                 full GAN training step, generator forward, and the transposed-conv upsampling stack against the HBM
                 roofline (rank 0, N=1 only; --no-hifigan skips it).
Progress markers go to stderr ("[bench  12.3 s] ...").  --backend gloo --share-device runs the N-rank code path on ONE GPU (a
functional check of spawn / rendezvous / barrier / max-over-ranks / the exchange between the two graphs, not a scaling
figure); --no-roofline / --no-forward-only / --no-fp32 / --no-inference / --no-cpu-baseline trim legs for experiment runs.
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
# memory-side bytes per launch (2*FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes of scripts/ffn_pmc_probe.py):
# bf16: mean of the four launches of a feed-forward block (ffn_pair forward 26.4, backward 36.2, weight gradients
# 47.3 / 33.8 MB against 22.2 / 32.3 / 17.2 / 15.6 MB algorithmic); fp32: the round-1 fp32-storage kernel
# -- read from the committed summary scripts/pmc_to_json.py writes (bench.py cannot run the profiler on itself)
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "ffn_block_pmc.json")


def port_over_reference(leg):
    """Speed of the oracle ("port") relative to the untouched reference on one host, measured in the build container by
    oracle/time_vs_reference.py (the reference cannot travel to the GPU box): seconds per iteration of the port / of the
    reference for ``leg`` ("sambert_b32_dropout_off", "sambert_b32_dropout_on", "hifigan_gan_step", "inference"); > 1 = the
    port is slower, i.e. ``cpu_baseline.value`` understates the reference by that factor.  None if the file is missing."""
    try:
        with open(os.path.join(ROOT, "profiles", "r06_time_vs_reference.json")) as f:
            doc = json.load(f)
        return {"ratio": doc["port_over_reference"][leg], "source": "profiles/r06_time_vs_reference.json "
                "(oracle/time_vs_reference.py, %d threads of the build container, reference and port interleaved)"
                % doc["host_threads"]}
    except (OSError, KeyError, ValueError):
        return None


def measured_traffic(precision):
    """(bytes per launch or None, source string or None, per-launch dict) from profiles/ffn_block_pmc.json."""
    try:
        with open(TRAFFIC_FILE) as f:
            doc = json.load(f)
        rec = doc[precision]
        return rec["mean_traffic_bytes"], "profiles/ffn_block_pmc.json (%s)" % rec["source"], rec.get("launches")
    except (OSError, KeyError, ValueError):
        return None, None, None


def sambert_yaml_config(cfg):
    return {"model_type": "sambert", "Model": {"KanTtsSAMBERT": {
        "params": cfg,
        "optimizer": {"type": "Adam", "params": {"lr": 0.001, "betas": [0.9, 0.98], "eps": 1.0e-9, "weight_decay": 0.0}},
        "scheduler": {"type": "NoamLR", "params": {"warmup_steps": 4000}}}},
        "grad_norm": 1.0, "batch_size": 32}


CPU_THREADS = {"n": None}


def _cpu_threads():
    """Host threads of the CPU legs.  The oracle ports are chains of small ops (the postnet LSTM alone is 612 time steps
    of eager ops per direction of autograd): beyond ~16 threads every op pays more fork/join than it gains -- round 1
    measured that on the same box class, and this round's first bench visit, run with every core, did not finish its
    CPU legs inside 13 minutes.  So: min(cores, 16) unless --cpu-threads says otherwise; both numbers are reported."""
    n = CPU_THREADS["n"] or min(os.cpu_count() or 1, 16)
    torch.set_num_threads(n)
    return n


def _time_iters(fn, warmup, iters, budget_s, min_iters=3):
    """>= ``min_iters`` timed iterations, up to ``iters``, stopping once ``budget_s`` of timed work is spent."""
    for _ in range(warmup):
        fn()
    ts = []
    t_all = time.time()
    for _ in range(iters):
        t0 = time.time()
        fn()
        ts.append(time.time() - t0)
        if len(ts) >= min_iters and time.time() - t_all > budget_s:
            break
    return sum(ts) / len(ts), len(ts)


def _note(msg):
    """Progress marker on stderr (the JSON line on stdout stays the only stdout output)."""
    print("[bench %7.1f s] %s" % (time.time() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.time()


def oracle_probe_main(threads, B):
    """Child-process body of ``_oracle_probe``: the CPU oracle step (dropout off) at ``threads`` host threads, one
    warm-up + one timed iteration; prints one JSON line."""
    import torch_oracle as O
    from kantts.models.sambert.kantts_sambert import KanTtsSAMBERT
    from kantts.utils import synthetic

    torch.set_num_threads(threads)
    cfg0 = dict(synthetic.sambert_16k_config())
    torch.manual_seed(0)
    m = KanTtsSAMBERT(cfg0)
    P = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    batch = O.synthetic_sambert_batch(B=B, T_in=64, seed=1234)
    frames = int(batch["output_lengths"].sum())
    O.DROP["on"] = False

    def one():
        for p in P.values():
            p.grad = None
        out = O.sambert_forward(P, cfg0, **batch)
        O.sambert_losses(out, batch["input_lengths"], batch["output_lengths"], batch["mel_targets"])["total"].backward()

    dt, n = _time_iters(one, 1, 1, 1e9, min_iters=1)
    print(json.dumps({"cores": threads, "value": frames / dt, "timed": n, "s_per_iter": dt}))


def _usable_cores():
    """Host cores this process may actually run on: the affinity mask, cut by the cgroup CPU quota when there is one
    (a container that sees 256 cores but owns 32 of them makes an "all cores" thread count an oversubscription test)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def _oracle_probe(threads, B, limit_s, ref_s=float('nan')):
    import subprocess

    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--oracle-probe", str(threads), "--batch", str(B)],
                           capture_output=True, text=True, timeout=limit_s)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if lines:
            return json.loads(lines[-1])
        return {"cores": threads, "error": (r.stderr or "no output")[-200:]}
    except subprocess.TimeoutExpired:
        return {"cores": threads, "value": None,
                "note": "1 warm-up + 1 timed iteration did not finish within %.0f s at %d threads (%.1f s per iteration "
                        "at the default thread count)" % (limit_s, threads, ref_s)}


def cpu_baseline(cfg, hip, B=32, budget_s=20.0):
    """CPU oracle ("port" of the reference path, pinned against it by tests/golden + oracle/check_vs_reference.py)
    forward + losses + backward on the SAME seeded batch as the GPU line (B=32, 14 797 valid frames), all host cores,
    fp32; Adam omitted (12 M parameters: negligible next to fwd+bwd on CPU).  Timed with dropout off (>= 2 warm-up
    + <= 5 timed) and "as shipped" with dropout on (1 warm-up + <= 3 timed).  Its dropout-off result doubles as the
    checker for "parity_error": the HIP path (fp32 mode and bf16 mode) on the same batch and weights."""
    import torch_oracle as O

    torch.manual_seed(0)
    from kantts.models.sambert.kantts_sambert import KanTtsSAMBERT

    cfg0 = dict(cfg)
    m = KanTtsSAMBERT(cfg0)
    P = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    batch = O.synthetic_sambert_batch(B=B, T_in=64, seed=1234)
    frames = int(batch["output_lengths"].sum())
    cores = _cpu_threads()
    keep = {}

    def one():
        for p in P.values():
            p.grad = None
        out = O.sambert_forward(P, cfg0, **batch)
        L = O.sambert_losses(out, batch["input_lengths"], batch["output_lengths"], batch["mel_targets"])
        L["total"].backward()
        keep["out"], keep["L"] = out, L

    O.DROP["on"] = False
    dt_off, n_off = _time_iters(one, 2, 6, budget_s, min_iters=5)  # >= 2 warm-up + >= 5 timed
    ref_out = {k: keep["out"][k].detach().clone() for k in ("dec_outputs", "postnet_outputs")}
    ref_loss = float(keep["L"]["total"].detach())
    ref_grads = {k: p.grad.detach().clone() for k, p in P.items() if p.grad is not None}
    O.DROP["on"] = True
    try:
        dt_on, n_on = _time_iters(one, 1, 3, budget_s * 0.6, min_iters=2)
    finally:
        O.DROP["on"] = False
    # every host core, beside the default thread count (the chain of small ops gets SLOWER beyond ~16 threads on this box
    # class; reported so that the choice is visible).  Runs in a child process under a hard time limit: one iteration at
    # 256 threads must not be able to eat the bench's time budget.
    all_cores = _usable_cores()
    all_core = eight = None
    if all_cores > cores and CPU_THREADS["n"] is None:
        # a quarter of the batch (B = 8: the same model, 8 of the 32 seeded utterances' worth of frames) so that the
        # probe FINISHES inside its limit -- at B = 32 one iteration on 256 threads did not end within 90 s in round 3
        all_core = _oracle_probe(all_cores, 8, limit_s=60.0, ref_s=dt_off)
        all_core["visible_cores"] = os.cpu_count() or 1
        all_core["batch"] = 8
    elif CPU_THREADS["n"] is None:
        # the default thread count IS every core this process may use (cgroup quota / affinity): the headline figure is the
        # all-core figure
        all_core = {"cores": cores, "value": frames / dt_off, "batch": B, "visible_cores": os.cpu_count() or 1,
                    "note": "usable cores (cgroup quota / affinity) = the default thread count: same run as `value`"}
    if cores != 8 and CPU_THREADS["n"] is None:
        eight = _oracle_probe(8, 8, limit_s=60.0, ref_s=dt_off)  # N = 8 for comparability with its probes
        eight["batch"] = 8

    # ---- parity of the HIP path at the benchmarked shape (dropout forced to 0 on both sides)
    parity = {}
    from kantts.train.loss import MelReconLoss, ProsodyReconLoss

    cfg_nodrop = {k: (0.0 if "dropout" in k else v) for k, v in cfg0.items()}
    gb = {k: v.cuda() for k, v in batch.items()}
    hip.ops.wgrad_overlap.enable(False)  # plain launches: gradients are read right after backward() here
    for mode in ("fp32", "bf16"):
        hip.set_precision(mode)
        torch.manual_seed(0)
        g = KanTtsSAMBERT(dict(cfg_nodrop)).cuda()
        g.eval()  # Prenet's hard-wired Dropout(0.5) off, like the oracle
        res = g(**gb)
        mel_, mel = MelReconLoss()(gb["output_lengths"], gb["mel_targets"], res["dec_outputs"], res["postnet_outputs"])
        d, p_, e = ProsodyReconLoss()(gb["input_lengths"], res["duration_targets"], res["pitch_targets"],
                                      res["energy_targets"], res["log_duration_predictions"], res["pitch_predictions"],
                                      res["energy_predictions"])
        total = mel_ + mel + d + p_ + e
        total.backward()
        hip.ops.wgrad_overlap.join()
        torch.cuda.synchronize()
        dm = (res["postnet_outputs"].detach().cpu() - ref_out["postnet_outputs"]).abs()
        dd = (res["dec_outputs"].detach().cpu() - ref_out["dec_outputs"]).abs()
        worst, wname, num, den = 0.0, "", 0.0, 0.0
        for n, prm in g.named_parameters():
            if prm.grad is not None and n in ref_grads:
                e2 = float((prm.grad.detach().cpu().double() - ref_grads[n].double()).pow(2).sum())
                r2 = float(ref_grads[n].double().pow(2).sum())
                num, den = num + e2, den + r2
                if (e2 / (r2 + 1e-60)) ** 0.5 > worst:
                    worst, wname = (e2 / (r2 + 1e-60)) ** 0.5, n
        parity[mode] = {"mel_mean_abs": float(dm.mean()), "mel_max_abs": float(dm.max()),
                        "dec_mean_abs": float(dd.mean()), "loss_abs": abs(float(total.detach()) - ref_loss),
                        "grad_rel_l2_global": (num / (den + 1e-60)) ** 0.5, "grad_rel_l2_worst_tensor": worst,
                        "worst_tensor": wname,
                        "lr_length_bit_exact": bool(torch.equal(res["LR_length_rounded"].cpu(),
                                                                keep["out"]["LR_length_rounded"]))}
        del g, res, total
    base = {"value": frames / dt_off, "unit": "mel-frames/s", "cores": cores, "kind": "port",
            "value_dropout_on": frames / dt_on, "all_cores": all_core, "threads_8": eight,
            "port_over_reference": port_over_reference("sambert_b32_dropout_off"),
            "port_over_reference_dropout_on": port_over_reference("sambert_b32_dropout_on"),
            "sample": "oracle/torch_oracle.py fwd+losses+bwd, fp32, the full seeded batch B=%d (%d valid frames): "
                      "dropout off 2 warm-up + %d timed, %.2f s/iter; dropout on (as shipped) 1 warm-up + %d timed, "
                      "%.2f s/iter; torch.set_num_threads(%d) of %d host cores" % (B, frames, n_off, dt_off, n_on, dt_on,
                                                                                   cores, os.cpu_count() or 1)}
    return base, parity


def _ffn_cases(hip, precision):
    """One fresh set of operand / output buffers of the decoder feed-forward block and the launches over it:
    name -> (launch, algorithmic bytes)."""
    from kantts._hip import bgemm_nt, bgemm_tn, gemm, make_seg, ops

    dev = "cuda"
    M, C, F = 32 * 204, 128, 1024
    if precision == "bf16":
        bf = torch.bfloat16
        xb, hb = torch.randn(M, C, device=dev).to(bf), torch.randn(M, F, device=dev).relu().to(bf)
        w1b, w2b = (torch.randn(F, C, device=dev) * 0.05).to(bf), (torch.randn(C, F, device=dev) * 0.05).to(bf)
        b1, b2 = torch.zeros(F, device=dev), torch.zeros(C, device=dev)
        res, dy = torch.randn(M, C, device=dev), torch.randn(M, C, device=dev)
        yh, dz = torch.empty(M, F, device=dev, dtype=bf), torch.empty(M, F, device=dev, dtype=bf)
        yx, dh = torch.empty(M, C, device=dev), torch.empty(M, C, device=dev, dtype=bf)
        dw1, dw2 = torch.zeros(F, C, device=dev), torch.zeros(C, F, device=dev)
        from kantts._hip import ffn_pair
        from kantts._hip.ops_bf16 import frag_major

        f1, f2 = frag_major(w1b.float()), frag_major(w2b.float())              # (F, C), (C, F) images: forward
        t2, t1 = frag_major(w2b.float().t().contiguous()), frag_major(w1b.float().t().contiguous())  # backward

        def pair_fwd():
            assert ffn_pair(xb, f1, f2, yx, M=M, T=204, F=F, bias1=b1, bias2=b2, relu=True, t_out=yh, res=res)

        def pair_bwd():
            assert ffn_pair(dy, t2, t1, dh, M=M, T=204, F=F, gate=hb, t_out=dz)

        wbytes = 2 * 2 * C * F
        cases = {
            # one launch: x (bf16) -> hidden (bf16, written for the weight gradients) -> out (fp32) + residual (fp32)
            "ffn_pair forward (LN-out bf16 -> hidden bf16 + out fp32; bias, ReLU, residual)": (
                pair_fwd, 2 * M * C + wbytes + 2 * M * F + 8 * M * C),
            # one launch: dy (fp32), gate = hidden (bf16) -> dz (bf16, written for dW1) -> dh (bf16)
            "ffn_pair input gradients (dy fp32, gate by hidden -> dz bf16 -> dh bf16)": (
                pair_bwd, 4 * M * C + 2 * M * F + wbytes + 2 * M * F + 2 * M * C),
            "wgrad 128x1024 (fp32 dy x bf16 hidden)": (
                lambda: bgemm_tn(dy, C, hb, F, M, C, F, dw2, F, 1), 4 * M * C + 2 * M * F + 4 * C * F),
            "wgrad 1024x128 (bf16 dz x bf16 ln-out)": (
                lambda: bgemm_tn(dz, F, xb, C, M, F, C, dw1, C, 1), 2 * M * F + 2 * M * C + 4 * C * F),
            # the two-launch form of the same block (round 2 before csrc/ffn_pair.hip; still the fallback for other shapes)
            "[two-launch form] fwd 128->1024 (bf16 -> bf16, bias+relu)": (
                lambda: bgemm_nt([(xb, C, w1b, C, C, 0)], M, F, yh, F, bias=b1, relu=True), 2 * M * C + 2 * C * F + 2 * M * F),
            "[two-launch form] fwd 1024->128 (bf16 -> fp32, bias+residual)": (
                lambda: bgemm_nt([(hb, F, w2b, F, F, 0)], M, C, yx, C, bias=b2, res=res, ldr=C),
                2 * M * F + 2 * C * F + 8 * M * C),
            "[two-launch form] dgrad 1024<-128 (fp32 dy, gate by hidden -> bf16)": (
                lambda: bgemm_nt([(dy, C, w2b, F, C, 0)], M, F, dz, F, b_kn=True, gate=hb, ldg=F),
                4 * M * C + 2 * C * F + 4 * M * F),
            "[two-launch form] dgrad 128<-1024 (bf16 -> bf16)": (
                lambda: bgemm_nt([(dz, F, w1b, C, F, 0)], M, C, dh, C, b_kn=True), 2 * M * F + 2 * C * F + 2 * M * C),
        }
        kernel, bytes_dtype = ("ffn_pair_kernel (csrc/ffn_pair.hip) + bgemm_tn_kernel (csrc/gemm_bf16.hip)",
                               "bf16 activations + weights, fp32 residual stream")
    else:
        p = hip.PREC_FP32
        x, h = torch.randn(M, C, device=dev), torch.randn(M, F, device=dev)
        w1, w2 = torch.randn(F, C, device=dev) * 0.05, torch.randn(C, F, device=dev) * 0.05
        yh, yx = torch.empty(M, F, device=dev), torch.empty(M, C, device=dev)
        dw1, dw2 = torch.zeros(F, C, device=dev), torch.zeros(C, F, device=dev)
        by = 4 * (M * C + C * F + M * F)
        cases = {
            "fwd 128->1024": (lambda: gemm([make_seg(x, C, 1, w1, C, 1, C)], M, F, yh, F, 1, precision=p), by),
            "fwd 1024->128": (lambda: gemm([make_seg(h, F, 1, w2, F, 1, F)], M, C, yx, C, 1, precision=p), by),
            "dgrad 128->1024": (lambda: gemm([make_seg(h, F, 1, w1, 1, C, F)], M, C, yx, C, 1, precision=p), by),
            "dgrad 1024->128": (lambda: gemm([make_seg(x, C, 1, w2, 1, F, C)], M, F, yh, F, 1, precision=p), by),
            "wgrad 128->1024": (lambda: gemm([make_seg(h, 1, F, x, 1, C, M)], F, C, dw1, C, 1, accumulate=True,
                                             splitk=ops._splitk_for(F, C, M), precision=p), by),
            "wgrad 1024->128": (lambda: gemm([make_seg(x, 1, C, h, 1, F, M)], C, F, dw2, F, 1, accumulate=True,
                                             splitk=ops._splitk_for(C, F, M), precision=p), by),
        }
        kernel, bytes_dtype = "gemm_fast_kernel<fp32> (csrc/gemm_fast.hip)", "fp32"
    return cases, kernel, bytes_dtype


def dominant_gemm_roofline(hip, precision, reps=20, replays=5, cold_sets=10):
    """Roofline of the dominant kernel: the MFMA contractions of the decoder feed-forward block (forward, input
    gradient, weight gradient of Conv1d(128->1024,k=1) and Conv1d(1024->128,k=1) at M = 32*204 decoder tokens; 12 layers
    each = 31 %% of the step's contraction flops).  Each contraction is launched `reps` times inside a captured hipGraph
    (the host is out of the picture) and timed with HIP events on the replay stream.  Algorithmic bytes = every operand
    read once + the output written once AT ITS STORAGE DTYPE: bf16 mode stores the LayerNorm output, the hidden
    activation and the weights as bf16 and keeps the residual stream / its gradient fp32 (kantts/_hip/ops_bf16.py).
    Two figures per launch: WARM (the same buffers every repetition -- at 15-30 MB per launch the operands stay in the
    256 MB Infinity Cache, as they partly do inside the training step, where the producer ran just before) and COLD
    (``cold_sets`` disjoint buffer sets visited round-robin, so every operand comes from HBM)."""
    from kantts._hip import ops

    ops.wgrad_overlap.enable(False)  # every contraction is launched (and timed) on its own here
    sets = []
    for _ in range(cold_sets):
        cases, kernel, bytes_dtype = _ffn_cases(hip, precision)
        sets.append(cases)

    def timed(launches):
        for fn in launches[:2]:
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for fn in launches:
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(replays):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (len(launches) * replays)

    per, per_gbps, per_cold, per_cold_gbps = {}, {}, {}, {}
    tot_us, tot_cold_us, tot_bytes, n_live = 0.0, 0.0, 0.0, 0
    for name, (fn, nbytes) in sets[0].items():
        us = timed([fn] * reps)
        per[name] = round(us, 2)
        per_gbps[name] = round(nbytes / us / 1e3, 1)
        if not name.startswith("["):  # the launches the step actually issues
            # cold: consecutive launches walk `cold_sets` disjoint buffer sets (~55 MB each, > 2x the 256 MB Infinity
            # Cache in total), so no operand of a launch is left in a cache by the previous visit of its set
            cold = timed([sets[r % cold_sets][name][0] for r in range(max(reps, 2 * cold_sets))])
            per_cold[name] = round(cold, 2)
            per_cold_gbps[name] = round(nbytes / cold / 1e3, 1)
            tot_us += us
            tot_cold_us += cold
            tot_bytes += nbytes
            n_live += 1
    M, C, F = 32 * 204, 128, 1024
    flops = 2.0 * M * C * F
    # the block's contraction flops: forward 2, input gradients 2, weight gradients 2 products of 2*M*C*F each
    return {"tflops": flops * 6 / (tot_us * 1e-6) / 1e12, "launch_us": per, "launch_gbps": per_gbps,
            "flops_per_launch": flops * 6 / n_live, "gbps": tot_bytes / (tot_us * 1e-6) / 1e9,
            "bytes_per_launch": tot_bytes / n_live, "mean_launch_us": tot_us / n_live, "kernel": kernel,
            "bytes_dtype": bytes_dtype,
            "cold": {"launch_us": per_cold, "launch_gbps": per_cold_gbps, "mean_launch_us": tot_cold_us / n_live,
                     "gbps": tot_bytes / (tot_cold_us * 1e-6) / 1e9, "tflops": flops * 6 / (tot_cold_us * 1e-6) / 1e12,
                     "buffer_sets": cold_sets}}


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


def _replay_ms(fn, n, reps=1):
    """Device time of ``fn`` (a short chain of launches) replayed from a hipGraph -- how the model issues them (the
    generator runs from a captured graph).  Eager issue of 10-20 us launches from Python is host-bound: the stage times of
    the upsampling chain measured that way were up to twice the kernels' own durations (profiles/r04_runF_*).  The graph
    holds ``reps`` copies of the chain so that a replay is long against its own launch cost.  Falls back to eager timing
    when capture fails."""
    try:
        fn()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            keep = [fn() for _ in range(reps)]
        ms = _event_ms(g.replay, n) / reps
        del g, keep
        return ms
    except Exception:
        return _event_ms(fn, n)


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
        res["generator_forward_eager_ms"] = ms  # ~300 launches issued from Python: host-bound
        try:  # the same forward replayed from a hipGraph (what an inference server runs: fixed-shape vocoder chunks)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                G(x)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            gg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gg, capture_error_mode="thread_local"):
                yg = G(x)
            ms = _event_ms(gg.replay, 20)
            res["generator_forward_schedule"] = "hipGraph replay"
            del gg, yg
        except Exception as exc:
            res["generator_forward_schedule"] = "eager (capture failed: %s)" % type(exc).__name__
        res["generator_forward_ms"] = ms
        res["generator_forward_samples_per_s"] = B * T_wav / (ms * 1e-3)
        res["generator_forward_tflops"] = 696.5e9 * B / 32 / (ms * 1e-3) / 1e12
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
    res["upsampling_fp32_storage"] = {
        "ms": ms, "achieved": gbps, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS, "algorithmic_bytes": elems * 4,
        "tflops": flops / (ms * 1e-3) / 1e12, "stage_us": per_stage,
        "note": "round-1 path (fp32 activations in HBM, one windowed-convolution launch per layer), kept for comparison"}
    # ---- the same four layers on bf16 activations: two-segment bf16 contractions for the wide layers (512->256, 256->128),
    # the LDS-free streaming kernel for the narrow ones (128->64, 64->32).  SURVEY 8(d): algorithmic traffic = input +
    # output + weights, each touched once = 49.3 M elements = 98.6 MB at bf16.
    if precision == "bf16":
        from kantts._hip import ops as _ops
        from kantts.models.hifigan.layers import effective_weight

        with torch.no_grad():
            acts, ws, T, C = [], [], frames, 512
            belems = 0
            for i, s_ in enumerate((8, 8, 2, 2)):
                layer = G.transpose_upsamples[i][1]
                acts.append(torch.randn(B, T, C, device="cuda").to(torch.bfloat16))
                w_ = effective_weight(layer.deconv).detach().contiguous()
                ws.append((w_, layer.deconv.bias.detach(), s_, _ops.upsample_weights(w_, s_, layer.deconv.bias)))
                belems += B * T * C + B * T * s_ * (C // 2) + C * (C // 2) * 2 * s_
                T, C = T * s_, C // 2

            def up_b(i):
                w, b, s_, prep = ws[i]
                y = _ops.upsample_forward(acts[i], w, b, s_, out_bf16=True, in_slope=0.1 if acts[i].shape[2] <= 128 else 1.0,
                                          prepared=prep)
                assert y is not None
                return y

            stage_b = [round(_replay_ms(lambda i=i: up_b(i), 20, reps=8) * 1e3, 1) for i in range(4)]
            msb = _replay_ms(lambda: [up_b(i) for i in range(4)], 20, reps=4)
            prep_ms = _event_ms(lambda: [_ops.upsample_weights(w, s_) for w, _, s_, _ in ws], 5)
            # the whole dual-path stage (transposed convolution + k = 7 convolution over the repeated signal) as the model
            # runs it since round 4: ONE polyphase contraction with the summed weights (Generator._dual_path_weight)
            dws = []
            for i, s_ in enumerate((8, 8, 2, 2)):
                wd_, bd_ = G._dual_path_weight(i, s_)
                wd_ = wd_.detach().contiguous()
                dws.append((wd_, bd_.detach(), s_, _ops.upsample_weights(wd_, s_, bd_)))

            def dual_b(i):
                w, b, s_, prep = dws[i]
                y = _ops.upsample_forward(acts[i], w, b, s_, out_bf16=True, prepared=prep)
                assert y is not None
                return y

            dual_stage = [round(_replay_ms(lambda i=i: dual_b(i), 20, reps=8) * 1e3, 1) for i in range(4)]
            ms_dual = _replay_ms(lambda: [dual_b(i) for i in range(4)], 20, reps=4)
            # the copy roof of the same chain: per stage a launch that reads (input + weights) and writes (output) bytes once,
            # whole chip, nothing else -- the achievable time of a 10-34 MB problem INCLUDING its launch ramp, measured in
            # the same 4-launch graph form as the real kernels (kantts_copy_roof)
            import kantts._hip as _hipmod

            cbuf = []
            T_, C_ = frames, 512
            for s_ in (8, 8, 2, 2):
                rd = (B * T_ * C_ + C_ * (C_ // 2) * 2 * s_) * 2
                wr = (B * T_ * s_ * (C_ // 2)) * 2
                cbuf.append((torch.empty((rd + 15) // 16 * 16, device="cuda", dtype=torch.uint8),
                             torch.empty((wr + 15) // 16 * 16, device="cuda", dtype=torch.uint8), rd, wr))
                T_, C_ = T_ * s_, C_ // 2

            def copy_b(i):
                src, dst, rd, wr = cbuf[i]
                _hipmod.check(_hipmod.lib().kantts_copy_roof(_hipmod.ptr(src), rd, _hipmod.ptr(dst), wr, _hipmod.stream()),
                              "copy_roof")

            copy_stage = [round(_replay_ms(lambda i=i: copy_b(i), 20, reps=8) * 1e3, 1) for i in range(4)]
            ms_copy = _replay_ms(lambda: [copy_b(i) for i in range(4)], 20, reps=4)
        gb = belems * 2 / (msb * 1e-3) / 1e9
        res["upsampling"] = {"ms": msb, "bound": "hbm", "achieved": gb, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                             "frac": gb / PEAK_HBM_GBPS, "algorithmic_bytes": belems * 2, "bytes_dtype": "bf16",
                             "tflops": flops / (msb * 1e-3) / 1e12, "stage_us": stage_b,
                             "kernel": "cconv_kernel, 2-tap polyphase form (layers 0-1) + upsample_stream_kernel (layers 2-3)",
                             "weight_prep_ms_not_included": prep_ms,
                             "copy_roof_us": round(ms_copy * 1e3, 1), "copy_roof_stage_us": copy_stage,
                             "copy_roof_gbps": belems * 2 / (ms_copy * 1e-3) / 1e9,
                             "time_over_copy_roof": msb / ms_copy,
                             "stage_time_over_copy_roof": [round(a / max(c, 1e-9), 2) for a, c in zip(stage_b, copy_stage)],
                             "timing": "hipGraph replay of the four launches (device time; eager issue from Python is host-bound)",
                             "note": "4 launches; bf16 activations in and out; the polyphase re-layout + bf16 cast of the "
                                     "weights (6.7 MB, once per optimizer step / once for inference) is timed separately"}
        dual_bytes = 50.5e6 * 2 * B / 32  # in + out + both weights of the fused dual-path stage, bf16
        res["upsampling_dual_path"] = {
            "ms": ms_dual, "bound": "hbm", "achieved": dual_bytes / (ms_dual * 1e-3) / 1e9, "peak": PEAK_HBM_GBPS,
            "unit": "GB/s", "frac": dual_bytes / (ms_dual * 1e-3) / 1e9 / PEAK_HBM_GBPS, "algorithmic_bytes": dual_bytes,
            "stage_us": dual_stage, "taps_per_stage": [int(w.shape[2] // s_) for w, _, s_, _ in dws],
            "algorithmic_gflop_two_launch_form": 87.0 * B / 32,
            "note": "transposed convolution + repeat convolution of a stage as ONE polyphase contraction with summed "
                    "weights (one output write, `rep` never formed): 2 taps at the x8 stages, 4 at the x2 stages; the "
                    "weight combination (a few small operations per optimizer step) is not included"}
    else:
        res["upsampling"] = dict(res["upsampling_fp32_storage"], bound="hbm", peak=PEAK_HBM_GBPS)
    out = gan_train_step(model, optimizer, scheduler, crit, config, y, x, steps=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = gan_train_step(model, optimizer, scheduler, crit, config, y, x, steps=1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    res["gan_step_eager_ms"] = dt * 1e3
    res["losses"] = {k: float(v.detach()) if torch.is_tensor(v) else float(v) for k, v in out.items()}
    # the arithmetic the step EXECUTES, per convolution family: HIP events around every convolution launch of one eager
    # step (launches back to back on one stream: the times are those of each kernel alone, their sum exceeds the captured
    # step, which overlaps launches).  The single-input-channel layers (conv_c1 / conv_n1) carry no tag: < 0.5 % of the flops.
    hip.profile_begin()
    gan_train_step(model, optimizer, scheduler, crit, config, y, x, steps=1)
    fams = {}
    for key, v in hip.profile_end_by_shape().items():
        o = fams.setdefault(key.split(" ", 1)[0], dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
        o["launches"] += v["n"]
        o["ms"] += v["ms"]
        o["flops"] += v["flops"] * v["n"]
        o["bytes"] += v["bytes"] * v["n"]
    peak = PEAK_TFLOPS["bf16" if precision == "bf16" else "fp32"]
    for o in fams.values():
        t = max(o["ms"], 1e-9) * 1e-3
        o.update(tflops=o["flops"] / t / 1e12, mfma_frac=o["flops"] / t / 1e12 / peak, gbps=o["bytes"] / t / 1e9,
                 hbm_frac=o["bytes"] / t / 1e9 / PEAK_HBM_GBPS)
    res["gan_step_families"] = fams
    executed = sum(o["flops"] for o in fams.values())
    # the step as training runs it: both phases + the three Adam updates replayed from one hipGraph
    # (kantts/train/gan_graph_step.py; graph == eager is tests/test_hifigan.py::test_graphed_gan_step_matches_eager_gpu)
    from kantts.train.gan_graph_step import GraphedGanStep

    gstep = GraphedGanStep(model, optimizer, scheduler, crit, config, y, x)
    gstep()
    torch.cuda.synchronize()
    n = max(steps, 10)
    t0 = time.perf_counter()
    for _ in range(n):
        out = gstep()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    res["gan_step_ms"] = dt * 1e3
    res["gan_step_schedule"] = "hipGraph replay of the whole step (generator phase + discriminator phase + 3 Adam updates)"
    res["gan_step_samples_per_s"] = B * T_wav / dt
    res["gan_step_tflops"] = 8.3e12 * B / 32 / dt / 1e12
    res["gan_step_mfma_frac"] = res["gan_step_tflops"] / PEAK_TFLOPS["bf16"] if precision == "bf16" else None
    # beside the reference count (8.3 TFLOP: the dense arithmetic of the reference's modules) the flops the convolution
    # launches of a step execute (polyphase transposed convolutions, grouped layers on their real group width)
    res["gan_step_executed_tflop"] = executed / 1e12
    res["gan_step_mfma_frac_executed"] = executed / dt / 1e12 / peak
    res["value"] = res["gan_step_samples_per_s"]
    res["unit"] = "audio-samples/s (GAN training step)"
    res["graph_losses"] = {k: float(v.detach()) if torch.is_tensor(v) else float(v) for k, v in out.items()}
    return res


def hifigan_cpu_baseline(B=4, T_wav=8192, budget_s=14.0):
    """CPU oracle port of the HiFi-GAN V1 GAN step (oracle/hifigan_oracle.py: generator forward + mel / adversarial /
    feature-matching generator loss + backward, generator re-run, discriminator loss + backward; optimizer updates
    omitted) and of the generator forward alone (no grad), at batch ``B`` x 8192 samples on all host cores."""
    import audio_oracle as A
    import hifigan_oracle as H
    from kantts.models.hifigan.hifigan import Generator, MultiPeriodDiscriminator, MultiScaleDiscriminator

    torch.manual_seed(0)
    mods = (Generator(), MultiPeriodDiscriminator(), MultiScaleDiscriminator())
    PG, PP, PS = ({k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
                  for m in mods)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 80, T_wav // 256, generator=g)
    y = torch.randn(B, 1, T_wav, generator=g).clamp(-1, 1)
    cores = _cpu_threads()

    def zero():
        for P in (PG, PP, PS):
            for p in P.values():
                p.grad = None

    def gan_step():
        zero()
        y_ = H.generator(PG, x)
        mel = torch.nn.functional.l1_loss(A.mel_spectrogram(y_), A.mel_spectrogram(y))
        adv, fm = 0.0, 0.0
        for P, f in ((PP, H.mpd), (PS, H.msd)):
            o_, f_ = f(P, y_)
            with torch.no_grad():
                _, fr = f(P, y)
            adv = adv + H.gen_adv_loss(o_)
            fm = fm + H.feat_match_loss(fr, f_)
        (45.0 * mel + adv + 2.0 * fm).backward()
        zero()
        with torch.no_grad():
            y2 = H.generator(PG, x)
        dl = 0.0
        for P, f in ((PP, H.mpd), (PS, H.msd)):
            o, _ = f(P, y)
            o_, _ = f(P, y2)
            real, fake = H.dis_adv_loss(o_, o)
            dl = dl + real + fake
        dl.backward()

    def gen_fwd():
        with torch.no_grad():
            H.generator(PG, x)

    dt_step, n_step = _time_iters(gan_step, 1, 3, budget_s, min_iters=1)
    dt_fwd, n_fwd = _time_iters(gen_fwd, 1, 5, budget_s * 0.3, min_iters=1)
    return {"value": B * T_wav / dt_step, "unit": "audio-samples/s (GAN training step)", "cores": cores, "kind": "port",
            "port_over_reference": port_over_reference("hifigan_gan_step"),
            "batch_note": "CPU batch %d against the GPU leg's 32: samples/s normalises the batch, and the CPU rate does not "
                          "grow with the batch (every layer already fills the cores at B = 4)" % B,
            "generator_forward_samples_per_s": B * T_wav / dt_fwd,
            "sample": "oracle/hifigan_oracle.py V1 (512 ch) GAN step fwd+bwd without optimizer updates, batch %d x %d "
                      "samples, fp32: 1 warm-up + %d timed, %.2f s/iter; generator forward (no grad): %d timed, %.2f "
                      "s/iter" % (B, T_wav, n_step, dt_step, n_fwd, dt_fwd)}


def melspec_leg(B=32, T_wav=8192, reps=20):
    """Mel-STFT feature extractor (third part of the north-star path): V1 loss settings (22.05 kHz, n_fft 1024, hop 256,
    80 mels) on batch 32 x 8192 samples, forward and forward+backward; HBM roofline at SURVEY 8(d)'s 1344 algorithmic
    bytes per frame (hop * 4 B read + 80 * 4 B written); CPU oracle (oracle/audio_oracle.py) beside it."""
    import audio_oracle as A
    from kantts.utils.audio_torch import MelSpectrogram

    ms = MelSpectrogram().cuda()
    x = torch.randn(B, T_wav, device="cuda") * 0.1
    frames = B * (1 + T_wav // 256)
    with torch.no_grad():
        ms_fwd = _event_ms(lambda: ms(x), reps)
    xg = x.clone().requires_grad_(True)
    cot = torch.randn_like(ms(x))

    def fb():
        xg.grad = None
        (ms(xg) * cot).sum().backward()

    ms_fb = _event_ms(fb, reps)
    # the same kernel at a size that can show bandwidth: 2048 x 8192 samples = 67 584 frames, 90.8 MB algorithmic (the
    # batch-32 launch moves 1.4 MB and measures launch latency, not the kernel)
    Bs = 2048
    xs = torch.randn(Bs, T_wav, device="cuda") * 0.1
    frames_s = Bs * (1 + T_wav // 256)
    with torch.no_grad():
        ms_s = _event_ms(lambda: ms(xs), 10)
    xsg = xs.clone().requires_grad_(True)
    cot_s = torch.randn_like(ms(xs))

    def fb_s():
        xsg.grad = None
        (ms(xsg) * cot_s).sum().backward()

    ms_s_fb = _event_ms(fb_s, 5)
    gbps_s = frames_s * 1344.0 / (ms_s * 1e-3) / 1e9
    del xs, xsg, cot_s
    xc = x.cpu()
    cores = _cpu_threads()
    dt_cpu, n_cpu = _time_iters(lambda: A.mel_spectrogram(xc), 2, 10, 3.0)
    err = float((ms(x).cpu() - A.mel_spectrogram(xc)).abs().max())
    gbps = frames * 1344.0 / (ms_fwd * 1e-3) / 1e9
    return {"workload": "mel-STFT 22.05 kHz n_fft 1024 hop 256 80 mels, batch %d x %d samples (%d frames)" % (B, T_wav, frames),
            "dtype": "fp32", "forward_ms": ms_fwd, "forward_frames_per_s": frames / (ms_fwd * 1e-3),
            "forward_backward_ms": ms_fb,
            "roofline": {"bound": "hbm", "kernel": "melspec_reg_kernel", "achieved": gbps, "peak": PEAK_HBM_GBPS,
                         "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS, "algorithmic_bytes_per_frame": 1344,
                         "note": "one launch of %d frames = %.2f MB algorithmic: launch-latency bound at this size"
                                 % (frames, frames * 1344 / 1e6)},
            "roofline_saturating": {"bound": "hbm", "kernel": "melspec_reg_kernel", "workload": "%d x %d samples (%d frames)"
                                    % (Bs, T_wav, frames_s), "forward_ms": ms_s, "forward_backward_ms": ms_s_fb,
                                    "achieved": gbps_s, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps_s / PEAK_HBM_GBPS,
                                    "algorithmic_bytes": frames_s * 1344.0},
            "cpu_baseline": {"value": frames / dt_cpu, "unit": "frames/s", "cores": cores, "kind": "port",
                             "sample": "oracle/audio_oracle.py mel_spectrogram on the same batch, %d timed" % n_cpu},
            "parity_error": {"max_abs_vs_oracle": err}}


def fp32_leg(hip, cfg, batch, frames, mel_crit, pros_crit, dev, steps=5, warmup=2):
    """The same training step in fp32 mode (MFMA 16x16x4 f32 = exact fp32 FMA chains: the mode every 1e-5 parity
    assertion runs in), graph-replayed like the headline line."""
    from kantts.models import model_builder
    from kantts.train.graph_step import GraphedSambertStep

    hip.set_precision("fp32")
    torch.manual_seed(0)
    model, opt, sch = model_builder(sambert_yaml_config(cfg), device=dev)
    net, optimizer, scheduler = model["KanTtsSAMBERT"], opt["KanTtsSAMBERT"], sch["KanTtsSAMBERT"]
    optimizer.set_grad_clip(1.0)
    net.train()
    step = GraphedSambertStep(net, optimizer, scheduler, mel_crit, pros_crit, batch)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"dtype": "fp32", "ms_per_step": dt * 1e3, "value": frames / dt, "unit": "mel-frames/s", "steps": steps,
            "whole_step_mfma_frac": 456.9e9 * batch["mel_targets"].shape[0] / 32 / dt / 1e12 / PEAK_TFLOPS["fp32"]}


def mas_leg(hip, cfg, precision, dev, steps=10, warmup=3, B=32):
    """The Monotonic-Alignment-Search training step (configs/sambert_16k_MAS.yaml: no duration targets; alignment learner,
    device-side MAS, CTC + binarisation terms; reference trainer.py:862-1005) on the full model at batch 32, replayed from a
    hipGraph -- capturable since round 6 (the CTC criterion is one launch with device-side lengths: csrc/ctc.hip).  The
    eager step (round 5's only form: ATen's ctc_loss synchronises with the host) is timed beside it."""
    from kantts.models import model_builder
    from kantts.train.graph_step import GraphedSambertStep
    from kantts.train.loss import AttentionBinarizationLoss, AttentionCTCLoss, MelReconLoss, ProsodyReconLoss, sambert_loss_sum
    from kantts.utils.synthetic import sambert_mas_batch

    hip.set_precision(precision)
    torch.manual_seed(0)
    cfg = dict(cfg, MAS=True)
    model, opt, sch = model_builder(sambert_yaml_config(cfg), device=dev)
    net, optimizer, scheduler = model["KanTtsSAMBERT"], opt["KanTtsSAMBERT"], sch["KanTtsSAMBERT"]
    optimizer.set_grad_clip(1.0)
    net.train()
    batch = {k: v.to(dev) for k, v in sambert_mas_batch(B=B, T_in=64, seed=1234).items() if v is not None}
    frames = int(batch["output_lengths"].sum())
    mc, pc = MelReconLoss(), ProsodyReconLoss()
    mas = {"AttentionCTCLoss": AttentionCTCLoss(), "AttentionBinarizationLoss": AttentionBinarizationLoss()}

    def eager():
        hip.ops.advance_rng(torch.device(dev))
        optimizer.zero_grad(set_to_none=True)
        res = net(**batch)
        total, _ = sambert_loss_sum(mc, pc, batch, res, prosody_lengths=res["valid_inter_lengths"])
        total = (total + mas["AttentionCTCLoss"](res["attn_logprob"], batch["input_lengths"], batch["output_lengths"])
                 + mas["AttentionBinarizationLoss"](50, res["attn_hard"], res["attn_soft"]))
        total.backward()
        optimizer.step()
        return total

    def timed(fn):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            last = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps, float(last.detach())

    dt_e, loss_e = timed(eager)
    step = GraphedSambertStep(net, optimizer, scheduler, mc, pc, batch, mas_criteria=mas)
    step.set_epoch(50)
    dt_g, loss_g = timed(step)
    return {"dtype": precision, "workload": "SAM-BERT full, MAS: True (alignment learner + device-side MAS + CTC + binarisation "
            "terms), batch %d, %d valid frames" % (B, frames), "ms_per_step": dt_g * 1e3, "value": frames / dt_g,
            "unit": "mel-frames/s", "launch": "one hipGraph per step (wide-band decoder form)", "steps": steps,
            "eager_ms_per_step": dt_e * 1e3, "final_loss": loss_g, "eager_final_loss": loss_e}


def inference_leg(hip, cfg, precision, n_utt=128, batch=32, loop_sample=4, parity_utts=32, cpu_threads=0, wav_utts=6):
    """BASELINE config 5: end-to-end SAM-BERT -> HiFi-GAN inference on 128 synthetic utterances (T_in uniform
    20..80, the training id distributions, duration head biased to ~3.5 frames per symbol -- random-init weights predict
    zero durations otherwise), free-running: AR duration predictor, AR mel decoder, postnet, HiFi-GAN V1 generator with
    weight norm folded.  Decoder modes: "loop" = one launch per op and step issued from Python (round 1; the reference's
    structure), "graph" = one decoder step captured in a hipGraph with the step index in device memory and replayed
    (kantts/models/sambert/decode_graph.py).  Batch 1 (the reference's only mode) and length-sorted batches."""
    from kantts.models.hifigan.hifigan import Generator
    from kantts.models.sambert.kantts_sambert import KanTtsSAMBERT

    hip.set_precision(precision)
    dev = "cuda"
    torch.manual_seed(0)
    am = KanTtsSAMBERT(dict(cfg))
    with torch.no_grad():
        am.variance_adaptor.duration_predictor.fc.bias.fill_(1.5)
    am = am.to(dev).eval()
    voc = Generator()
    voc_P = {k: v.detach().clone() for k, v in voc.state_dict().items()}  # weight_g / weight_v: what the oracle reads
    voc = voc.to(dev).eval()
    voc.remove_weight_norm()
    from kantts.utils.synthetic import inference_utterances

    lens, ling, emo, spk = inference_utterances(n_utt)
    order = torch.argsort(lens, descending=True)

    def synth(idx, mode, voc_group=0):
        """One acoustic-model pass over the utterances ``idx`` and the vocoder over its mels.  ``voc_group`` > 0: the
        vocoder runs over consecutive groups of that many utterances, each trimmed to ITS longest mel (the utterances are
        length-sorted, so a group's frame counts are close) -- the acoustic model's autoregressive loops cost the same
        wall time for 128 sequences as for 32 (a workgroup per sequence: 128 of 256 CUs), the vocoder's cost is the frames
        it is given."""
        am.mel_decoder.decode_mode = mode
        ln = lens[idx]
        Tm = int(ln.max())
        args = dict(inputs_ling=ling[idx, :Tm].to(dev), inputs_emotion=emo[idx, :Tm].to(dev),
                    inputs_speaker=spk[idx, :Tm].to(dev), input_lengths=ln.to(dev))
        with torch.no_grad():
            res = am(**args)
            mel = res["postnet_outputs"].transpose(1, 2).contiguous()  # (B, 80, frames)
            n = res["LR_length_rounded"].clamp(max=mel.shape[2])
            if voc_group and len(idx) > voc_group:
                n_host = n.tolist()  # (the only host read: the groups' frame counts)
                wav = [voc(mel[g0:g0 + voc_group, :, :max(1, max(n_host[g0:g0 + voc_group]))].contiguous())
                       for g0 in range(0, len(idx), voc_group)]
                return sum(n_host), sum(n_host) * 256, wav
            wav = voc(mel)
        return int(n.sum()), int(n.sum()) * 256, wav

    def timed(groups, mode, voc_group=0):
        synth(groups[0], mode, voc_group)  # warm-up / capture for the first shape
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        frames = samples = 0
        for idx in groups:
            f, s_, _ = synth(idx, mode, voc_group)
            frames += f
            samples += s_
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {"utterances": sum(len(i) for i in groups), "seconds": dt, "mel_frames_per_s": frames / dt,
                "audio_samples_per_s": samples / dt, "rtf_22k": dt / (samples / 22050.0),
                "utterances_per_s": sum(len(i) for i in groups) / dt, "mel_frames": frames}

    singles = [order[i:i + 1] for i in range(n_utt)]
    batches = [order[i:i + batch] for i in range(0, n_utt, batch)]
    out = {"workload": "%d utterances, T_in 20..80 (mean %.1f symbols), SAM-BERT full free-running -> HiFi-GAN V1 (22.05 kHz, "
                       "hop 256)" % (n_utt, float(lens.float().mean())), "dtype": precision}
    # the Python-loop decoder on a bounded sample (it is ~10x slower); everything else on all utterances
    out["batch1_loop"] = timed(singles[::max(1, n_utt // loop_sample)][:loop_sample], "loop")
    dp = am.variance_adaptor.duration_predictor
    dp.ar_kernel = False  # the round-4 configuration: duration tokens and decoder steps issued / replayed from the host
    out["batch1_graph"] = timed(singles, "graph")
    out["batch%d_loop" % batch] = timed(batches[:1], "loop")
    out["batch%d_graph" % batch] = timed(batches, "graph")
    dp.ar_kernel = None  # round 5: each autoregressive loop is ONE launch (csrc/ar_infer.hip; bf16 mode)
    out["batch1_kernel"] = timed(singles, "kernel")
    out["batch%d_kernel" % batch] = timed(batches, "kernel")
    one_launch = getattr(am.mel_decoder, "_decode_kernel", None) is not None
    best = "batch%d_kernel" % batch if one_launch else "batch%d_graph" % batch
    how = "in length-sorted batches of %d" % batch
    if one_launch and n_utt > batch:
        # round 6: the acoustic model over ALL utterances at once (its autoregressive loops are one workgroup per sequence:
        # 128 sequences take the wall time of 32), the vocoder over length-sorted groups of ``batch`` trimmed to their own
        # longest mel
        key = "batch%d_am_voc%d_kernel" % (n_utt, batch)
        out[key] = timed([order], "kernel", voc_group=batch)
        # every sequence is its own workgroup / its own rows: the batch size must not change a single frame count
        out[key]["frame_counts_equal_batch%d" % batch] = out[key]["mel_frames"] == out["batch%d_kernel" % batch]["mel_frames"]
        if out[key]["audio_samples_per_s"] > out[best]["audio_samples_per_s"]:
            best, how = key, "acoustic model over all %d at once, vocoder in length-sorted groups of %d" % (n_utt, batch)
    out["value"] = out[best]["audio_samples_per_s"]
    out["headline_leg"] = best
    out["unit"] = ("audio-samples/s, symbols -> wav, %d utterances %s, %s" % (
        n_utt, how, "duration predictor and mel decoder loops as one launch each" if one_launch else "graph-replayed decoder"))
    if parity_utts:
        # the benchmarked configuration checked against the CPU oracle's free-running inference (outside the timed regions),
        # whose wall time is the CPU baseline of this leg
        try:
            par = config5_parity(am, cfg, (lens, ling, emo, spk), order[:: max(1, n_utt // parity_utts)][:parity_utts],
                                 batch=batch, threads=cpu_threads, voc=voc, voc_P=voc_P, wav_utts=wav_utts)
            out["cpu_baseline"] = par.pop("cpu_baseline")
            out["parity_error"] = par
        except Exception as exc:  # noqa: BLE001
            out["parity_error"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
    return out


def config5_parity(am, cfg, utts, idx, batch=32, threads=0, mode="kernel", voc=None, voc_P=None, wav_utts=0):
    """BASELINE config 5 checked at the configuration that is timed: the product's batched free-running (decoder mode
    ``mode``: "kernel" = each autoregressive loop one launch in bf16 mode / the replayed graph in fp32 mode, "graph")
    inference of the utterances ``idx`` (length-sorted batches of ``batch``) against oracle/torch_oracle.py's free-running
    inference of the same utterances one at a time (the reference's only mode, kantts/bin/infer_sambert.py:58-227;
    kantts_sambert.py:569-610; adaptors.py:67-83).  Reports (i) agreement of the frame counts and of the rounded durations
    (free-running: a 1-ulp change of exp(log_dur) - 1 + 0.5 can flip a duration), (ii) the mel error
    with the durations FORCED to the oracle's, over the valid frames, (iii) the oracle's speed as this leg's CPU baseline,
    (iv) with ``voc`` (the product's Generator) and ``voc_P`` (its state_dict before weight norm was folded): the VOCODER half
    (kantts/bin/infer_hifigan.py:66-139) on ``wav_utts`` of the utterances -- the product's generator on the product's
    FREE-RUNNING mel against oracle/hifigan_oracle.py's generator on the oracle's mel, over the utterance's samples.
    Checker only: nothing here runs inside a timed region."""
    import torch_oracle as O

    lens, ling, emo, spk = utts
    idx = torch.as_tensor(idx)
    r = cfg["outputs_per_step"]
    P = {k: v.detach().cpu().clone() for k, v in am.state_dict().items()}
    nthr = torch.get_num_threads()
    if threads:
        torch.set_num_threads(threads)
    ref = {}
    t0 = time.perf_counter()
    try:
        with torch.no_grad():
            for b in idx.tolist():
                n = int(lens[b])
                o = O.sambert_forward(P, cfg, ling[b:b + 1, :n], emo[b:b + 1, :n], spk[b:b + 1, :n], lens[b:b + 1])
                ref[b] = dict(frames=int(o["LR_length_rounded"][0]), mel=o["postnet_outputs"][0], bw=int(o["x_band_width"]),
                              dur=(torch.exp(o["log_duration_predictions"][0]) - 1 + 0.5).long())
    finally:
        cpu_s = time.perf_counter() - t0
        used = torch.get_num_threads()
        torch.set_num_threads(nthr)
    dev = next(am.parameters()).device
    am.mel_decoder.decode_mode = mode
    order = idx[torch.argsort(lens[idx], descending=True)]
    same_frames = same_dur = n_sym = same_bw = 0
    abs_sum = abs_max = 0.0
    n_el = 0
    free_abs_sum, free_n = 0.0, 0
    wav = {"n": 0, "abs": 0.0, "max": 0.0, "ref_sq": 0.0, "el": 0, "cpu_s": 0.0}
    wav_set = set(idx.tolist()[:wav_utts]) if (voc is not None and wav_utts) else set()
    if wav_set:
        import hifigan_oracle as H
    for g0 in range(0, len(order), batch):
        grp = order[g0:g0 + batch]
        ln = lens[grp]
        Tm = int(ln.max())
        args = dict(inputs_ling=ling[grp, :Tm].to(dev), inputs_emotion=emo[grp, :Tm].to(dev),
                    inputs_speaker=spk[grp, :Tm].to(dev), input_lengths=ln.to(dev))
        dur_ref = torch.zeros(len(grp), Tm, dtype=torch.long)
        for i, b in enumerate(grp.tolist()):
            dur_ref[i, : int(lens[b])] = ref[b]["dur"]
        with torch.no_grad():
            free = am(**args)
            forced = am(**args, duration_targets=dur_ref.to(dev))
            wav_free = (voc(free["postnet_outputs"].transpose(1, 2).contiguous()).float().cpu()
                        if wav_set.intersection(grp.tolist()) else None)
        fl = free["LR_length_rounded"].cpu()
        dur_free = (torch.exp(free["log_duration_predictions"].cpu()) - 1 + 0.5).long()
        bw_free = free["band_width_per_sequence"].cpu() if "band_width_per_sequence" in free else None
        for i, b in enumerate(grp.tolist()):
            n, nf = int(lens[b]), ref[b]["frames"]
            same_frames += int(int(fl[i]) == nf)
            same_dur += int((dur_free[i, :n] == ref[b]["dur"]).sum())
            same_bw += int(bw_free is not None and int(bw_free[i]) == ref[b]["bw"])
            n_sym += n
            assert int(forced["LR_length_rounded"][i]) == nf, "forced durations must reproduce the oracle's frame count"
            d = (forced["postnet_outputs"][i, :nf].cpu() - ref[b]["mel"][:nf]).abs()
            abs_sum += float(d.sum())
            abs_max = max(abs_max, float(d.max()))
            n_el += d.numel()
            if int(fl[i]) == nf and bool((dur_free[i, :n] == ref[b]["dur"]).all()):
                df = (free["postnet_outputs"][i, :nf].cpu() - ref[b]["mel"][:nf]).abs()
                free_abs_sum += float(df.sum())
                free_n += df.numel()
            if b in wav_set and int(fl[i]) == nf:
                t1 = time.perf_counter()
                if threads:
                    torch.set_num_threads(threads)
                try:
                    with torch.no_grad():
                        w_ref = H.generator(voc_P, ref[b]["mel"][:nf].t().unsqueeze(0).contiguous())[0, 0]
                finally:
                    torch.set_num_threads(nthr)
                wav["cpu_s"] += time.perf_counter() - t1
                dw = (wav_free[i, 0, : w_ref.numel()] - w_ref).abs()
                wav["n"] += 1
                wav["abs"] += float(dw.sum())
                wav["max"] = max(wav["max"], float(dw.max()))
                wav["ref_sq"] += float((w_ref.double() ** 2).sum())
                wav["el"] += dw.numel()
    frames = sum(v["frames"] for v in ref.values())
    used_kernel = getattr(am.mel_decoder, "_decode_kernel", None) is not None
    return {"utterances": len(ref),
            "decoder": "%s, length-sorted batches of %d" % ("one launch per autoregressive loop" if used_kernel else
                                                            ("graph" if mode != "loop" else "loop"), batch),
            "frame_count_agreement": same_frames / len(ref), "duration_agreement": same_dur / n_sym,
            "band_width_agreement": same_bw / len(ref),
            "mel_mean_abs_forced_durations": abs_sum / n_el, "mel_max_abs_forced_durations": abs_max,
            "mel_mean_abs_free_running_where_durations_agree": (free_abs_sum / free_n) if free_n else None,
            "wav": None if not wav["n"] else {
                "utterances": wav["n"], "samples": wav["el"], "mean_abs": wav["abs"] / wav["el"], "max_abs": wav["max"],
                "reference_rms": (wav["ref_sq"] / wav["el"]) ** 0.5, "oracle_generator_cpu_s": wav["cpu_s"],
                "what": "product generator on the product's free-running mel vs oracle generator on the oracle's mel"},
            "cpu_baseline": {"value": len(ref) / cpu_s, "unit": "utterances/s (symbols -> mel, batch 1, free-running)",
                             "mel_frames_per_s": frames / cpu_s, "cores": used, "kind": "port",
                             "port_over_reference": port_over_reference("inference"),
                             "sample": "%d of the leg's utterances through oracle/torch_oracle.py, %.1f s" % (len(ref), cpu_s)}}


def _spawn_ranks(n, argv):
    """`python bench.py --gpus N` outside a torch.distributed.run launch: re-exec under it (one rank per GPU of this
    node, rendezvous on 127.0.0.1) and pass its JSON line through."""
    import socket
    import subprocess

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hifigan", action="store_true")
    ap.add_argument("--no-fp32", action="store_true", help="skip the fp32 (parity-path) throughput figure")
    ap.add_argument("--no-inference", action="store_true", help="skip the end-to-end inference leg (BASELINE config 5)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="host threads of the CPU-oracle legs (default min(cores, 16))")
    ap.add_argument("--no-wgrad-overlap", action="store_true", help="weight gradients on the main stream (A/B switch)")
    ap.add_argument("--no-wgrad-group", action="store_true", help="one launch per weight gradient (A/B switch)")
    ap.add_argument("--mode", default="graph", choices=["graph", "eager"],
                    help="graph: whole step captured once in a hipGraph and replayed; eager: launch per op")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo + --share-device runs the N-rank path on a one-GPU box")
    ap.add_argument("--no-forward-only", action="store_true", help="skip the forward-only capture (kernel-trace runs)")
    ap.add_argument("--forward-replays", type=int, default=20,
                    help="timed replays of the forward-only graph (raise it for a kernel trace of the forward pass)")
    ap.add_argument("--no-roofline", action="store_true", help="skip the roofline microbench (timing experiments only)")
    ap.add_argument("--oracle-probe", type=int, default=0, help=argparse.SUPPRESS)  # child process of cpu_baseline
    ap.add_argument("--share-device", action="store_true",
                    help="every rank uses cuda:0 (a functional check of the distributed path, not a scaling figure)")
    ap.add_argument("--rccl-world1", action="store_true",
                    help="one rank, but with a REAL world-size-1 nccl (= RCCL) process group: the data-parallel code path "
                         "(arena buckets, graph segments, the all-reduce between their replays, RCCL's init and its watchdog "
                         "thread beside relaxed-mode captures) executes on a one-GPU box; not a scaling figure")
    args = ap.parse_args()

    CPU_THREADS["n"] = args.cpu_threads or None
    if args.oracle_probe:
        return oracle_probe_main(args.oracle_probe, args.batch)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_spawn_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("[bench] --gpus %d but the launcher started %d rank(s): reporting n_gpus=%d" % (args.gpus, world, world),
              file=sys.stderr)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or args.rccl_world1
    if args.rccl_world1 and world == 1 and "MASTER_PORT" not in os.environ:
        import socket

        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(so.getsockname()[1])
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", KANTTS_DP_FORCE="1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    device_index = 0 if args.share_device else local_rank
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":  # RCCL refuses two ranks on one device: --share-device needs gloo
            dist.init_process_group("nccl", init_method="env://", device_id=dev)
        else:
            dist.init_process_group("gloo", init_method="env://")

    import kantts._hip as hip
    import kantts._hip.ops  # noqa: F401
    from kantts.models import model_builder
    from kantts.utils import synthetic
    from kantts.train.loss import MelReconLoss, ProsodyReconLoss, sambert_loss_sum

    hip.lib()
    hip.set_precision(args.precision)
    cfg = synthetic.sambert_16k_config()
    torch.manual_seed(0)
    model, opt, sch = model_builder(sambert_yaml_config(cfg), device=dev, rank=device_index, distributed=distributed)
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
        loss, _ = sambert_loss_sum(mel_crit, pros_crit, batch, res)
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
                                      overlap_wgrad=not args.no_wgrad_overlap, group_wgrads=not args.no_wgrad_group)
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

    _note("model built, step %s" % mode)
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
    _note("timed %d steps: %.3f ms/step on this rank" % (args.steps, 1e3 * dt / args.steps))
    t = torch.tensor([dt, float(frames)], device=dev, dtype=torch.float64)
    per_rank_ms = [1e3 * dt / args.steps]
    if distributed:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_max, total_frames = float(tmax[0]), float(tsum[1])
        tg = t if args.backend == "nccl" else t.cpu()  # gloo gathers host tensors only
        allt = [torch.zeros_like(tg) for _ in range(world)]
        dist.all_gather(allt, tg)
        per_rank_ms = [1e3 * float(a[0]) / args.steps for a in allt]
    else:
        dt_max, total_frames = dt, float(frames)

    # ---- data-parallel runs: how much of the gradient exchange is NOT hidden behind backward.  The same captured step
    # replayed without its collectives (weights diverge between the replicas from here on: nothing after this point reads
    # them) -- exposed = step with the exchange - step without it, max over ranks.
    dp_info = None
    final_loss = float(loss.detach())
    if distributed and mode == "graph":
        dp_info = {"form": ("%d graph segments, one all-reduce per gradient bucket issued between the replays "
                            "(kantts/train/segments.py)" % len(step.segments.segments)) if step.segments is not None
                   else "two graphs: backward | bucketed all-reduce | update",
                   "gradient_bytes": int(optimizer.arena.numel) * 4,
                   "buckets": len(getattr(optimizer.arena, "buckets", None) or []) or optimizer.arena.n_buckets}
        try:
            step.skip_exchange = True
            for _ in range(2):
                step()
            dist.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            dt_free = time.perf_counter() - t1
            tf = torch.tensor([dt_free], device=dev, dtype=torch.float64)
            dist.all_reduce(tf, op=dist.ReduceOp.MAX)
            dp_info["ms_per_step_without_exchange"] = 1e3 * float(tf[0]) / args.steps
            dp_info["allreduce_ms_exposed"] = 1e3 * (dt_max - float(tf[0])) / args.steps
        except Exception as exc:
            dp_info["error"] = "%s: %s" % (type(exc).__name__, str(exc)[:200])
        finally:
            step.skip_exchange = False

    # ---- forward pass alone, as training runs it (dropout on, activations kept for backward, predictors on their side
    # branch), replayed from its own hipGraph: SURVEY 8(d) prices the MFMA roofline on the forward pass
    # (152.3 GFLOP at batch 32: roofline.forward_mfma_frac = 152.3e9 / t_fwd / peak)
    fwd_ms = None
    if rank == 0 and world == 1 and mode == "graph" and not args.no_forward_only:
        try:
            def fwd_only():
                hip.ops.advance_rng(dev)
                # the band width of this (static) batch, as the captured training step knows it (train/graph_step.py): the
                # decoder blocks take the same form in both graphs
                net.band_width_bound = getattr(step, "_bound", None)
                try:
                    res = net(**batch)
                finally:
                    net.band_width_bound = None
                return sambert_loss_sum(mel_crit, pros_crit, batch, res)[0]

            hip.ops.wgrad_overlap.enable(True)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fwd_only()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            fg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(fg, capture_error_mode="thread_local"):
                keep = fwd_only()
            hip.ops.wgrad_overlap.join()
            fg.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(max(1, args.forward_replays)):
                fg.replay()
            e1.record()
            torch.cuda.synchronize()
            fwd_ms = e0.elapsed_time(e1) / max(1, args.forward_replays)
            del keep, fg
        except Exception as exc:
            print("[bench] forward-only capture failed (%s: %s)" % (type(exc).__name__, str(exc)[:200]), file=sys.stderr)
        finally:
            hip.ops.wgrad_overlap.enable(False)

    # ---- roofline of the dominant kernel: HIP events around every GEMM launch of one extra step
    roof = None
    rank_roof = rank == 0 and not args.no_roofline
    if rank_roof:
        hip.profile_begin()
    net.device_band_width = False
    hip.ops.wgrad_overlap.enable(False, group_wgrads=not args.no_wgrad_group)
    if not args.no_roofline:
        eager_step()  # instrumented launches are issued eagerly on every rank (the step holds a collective)
    torch.cuda.synchronize()
    if rank == 0 and args.no_roofline and fwd_ms is not None:
        roof = {"forward_ms": fwd_ms}
    if rank_roof:
        prof = hip.profile_end()
        fam = hip.profile_families().get("bgemm_nt")
        _note("forward-only pass and eager instrumented step done")
        dg = dominant_gemm_roofline(hip, args.precision)
        _note("roofline microbench (warm + cold) done")
        peak = PEAK_TFLOPS[args.precision]
        traffic, traffic_src, traffic_launches = measured_traffic(args.precision)
        roof = {"bound": "hbm", "kernel": dg["kernel"] + ": decoder FFN contractions, M=6528, 128<->1024",
                "achieved": dg["gbps"], "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": dg["gbps"] / PEAK_HBM_GBPS,
                # memory-side bytes per launch come from rocprofv3 --pmc passes of scripts/bgemm_bench.py (2*FETCH_SIZE +
                # WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md), collected offline per round under profiles/
                # -- bench.py cannot run the profiler on itself; null until this round's pass exists
                "traffic": traffic, "traffic_source": traffic_src, "traffic_per_launch": traffic_launches,
                "bytes_per_launch": dg["bytes_per_launch"], "bytes_dtype": dg["bytes_dtype"],
                "flops_per_launch": dg["flops_per_launch"], "launch_us": dg["launch_us"], "launch_gbps": dg["launch_gbps"],
                "mean_launch_us": dg["mean_launch_us"],
                "cold_cache": dict(dg["cold"], frac=dg["cold"]["gbps"] / PEAK_HBM_GBPS,
                                   mfma_frac=dg["cold"]["tflops"] / peak),
                "mfma_tflops": dg["tflops"], "mfma_peak": peak, "mfma_frac": dg["tflops"] / peak,
                "gemm_launches_per_step": prof["launches"], "gemm_gflop_per_step": prof["flops"] / 1e9,
                "gemm_ms_per_step_eager_events": prof["ms"],
                "whole_step_algorithmic_tflops": 456.9e9 * args.batch / 32 / (dt_max / args.steps) / 1e12}
        roof["whole_step_mfma_frac"] = roof["whole_step_algorithmic_tflops"] / peak
        # per kernel family of the step (HIP events around every launch of one eager, instrumented training step -- launches
        # inside a real step, not a replay micro-benchmark; the rocprofv3 averages of the captured step are under profiles/)
        fams = {}
        for name, f in hip.profile_families().items():
            if not f["launches"] or f["ms"] <= 0:
                continue
            t = f["ms"] * 1e-3
            fams[name] = {"launches_per_step": f["launches"], "mean_launch_us": 1e6 * t / f["launches"], "ms_per_step": f["ms"],
                          "gflop_per_step": f["flops"] / 1e9, "algorithmic_mb_per_step": f["bytes"] / 1e6,
                          "tflops": f["flops"] / t / 1e12, "mfma_frac": f["flops"] / t / 1e12 / peak,
                          "gbps": f["bytes"] / t / 1e9, "hbm_frac": f["bytes"] / t / 1e9 / PEAK_HBM_GBPS}
        roof["families"] = fams
        if fam and fam["launches"]:
            # the kernel family with the largest share of the step's kernel time (20 % in round 3's kernel trace): every
            # Linear / Conv1d forward and input gradient outside the feed-forward pair.  HIP events around each launch of
            # one eager step; algorithmic bytes = A + weight slices + result (+ epilogue operands) at storage dtype.
            nt_gbps = fam["bytes"] / (fam["ms"] * 1e-3) / 1e9
            roof["nt_contractions"] = {
                "kernel": "bgemm_nt_kernel / bgemm_nt_lnb_kernel (all shapes of one training step)", "bound": "hbm",
                "launches_per_step": fam["launches"], "mean_launch_us": 1e3 * fam["ms"] / fam["launches"],
                "ms_per_step": fam["ms"], "algorithmic_bytes_per_step": fam["bytes"],
                "mean_bytes_per_launch": fam["bytes"] / fam["launches"], "achieved": nt_gbps, "peak": PEAK_HBM_GBPS,
                "unit": "GB/s", "frac": nt_gbps / PEAK_HBM_GBPS, "tflops": fam["flops"] / (fam["ms"] * 1e-3) / 1e12,
                "mfma_frac": fam["flops"] / (fam["ms"] * 1e-3) / 1e12 / peak,
                "note": "launch-latency sized: 6528 x 128 x 384 is 0.64 GFLOP / 5 MB; the rocprofv3 averages of the same "
                        "command are under profiles/ (r05_runFINAL_bench_command_kernel_stats_top.csv)"}
        if fwd_ms is not None:
            roof["forward_ms"] = fwd_ms
            roof["forward_algorithmic_tflops"] = 152.3e9 * args.batch / 32 / (fwd_ms * 1e-3) / 1e12
            roof["forward_mfma_frac"] = roof["forward_algorithmic_tflops"] / peak
            # SAM-BERT is priced on MFMA -- achieved = 152.3 GFLOP (the dense-algorithmic forward count at batch
            # 32) / the forward pass's duration (its own hipGraph, replayed, HIP events) against the dense bf16 (fp32-mode:
            # fp32) matrix peak.  The HBM view of the feed-forward contractions, the headline until round 4, is kept below
            # as ``ffn_hbm``; ``families`` has flops, bytes and time of every contraction family of the step.
            roof = dict(roof, ffn_hbm={k: roof[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                                          "traffic_source", "traffic_per_launch", "bytes_per_launch",
                                                          "bytes_dtype", "flops_per_launch", "launch_us", "launch_gbps",
                                                          "mean_launch_us", "cold_cache", "mfma_tflops", "mfma_frac")})
            for k in ("traffic_source", "traffic_per_launch", "bytes_per_launch", "bytes_dtype", "launch_us", "launch_gbps",
                      "mean_launch_us", "cold_cache", "mfma_tflops", "mfma_frac"):
                roof.pop(k, None)
            roof.update({"bound": "mfma", "kernel": "SAM-BERT forward pass (all kernels of the forward-only hipGraph: dropout on, "
                                                  "activations kept for backward, batch %d)" % args.batch,
                         "achieved": roof["forward_algorithmic_tflops"], "peak": peak, "unit": "TFLOP/s",
                         "frac": roof["forward_mfma_frac"], "flops_per_launch": 152.3e9 * args.batch / 32,
                         "launch_ms": fwd_ms, "traffic": None, "mfma_peak": peak,
                         "note": "one 'launch' = one forward pass; dense-algorithmic flop count of SURVEY 8(d) (band-limited "
                                 "attention skips masked tiles, the count does not)"})

    if rank == 0:
        out = {
            "metric": "mel-frames/sec (SAM-BERT train)", "value": total_frames * args.steps / dt_max,
            "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt_max / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "SAM-BERT full (sambert_16k.yaml zhcn) fwd+bwd+clip+Adam, batch %d/GPU, T_in 64, "
                                   "%d valid mel frames on rank 0, dropout on" % (args.batch, frames),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world, "launch": mode,
                       "scheduling": ("one hipGraph per step: variance predictors as a parallel branch, deferred weight "
                                      "gradients grouped by shape and issued from flush points in backward"
                                      if mode == "graph" else "eager launches, one stream"),
                       "final_loss": final_loss, "per_rank_ms_per_step": per_rank_ms,
                       "collective_world_size": dist.get_world_size() if distributed else 1,
                       "collective_backend": (("nccl (RCCL)" if args.backend == "nccl" else "gloo") if distributed else None),
                       "shared_device": bool(args.share_device and distributed),
                       "data_parallel": dp_info},
            "roofline": roof,
        }
        del step, net, optimizer, model, opt
        torch.cuda.empty_cache()
        if world == 1 and args.precision != "fp32" and not args.no_fp32:
            try:  # the parity-proven path (fp32 MFMA) on the same step, beside the throughput (bf16) line
                out["fp32_path"] = fp32_leg(hip, cfg, batch, frames, mel_crit, pros_crit, dev)
                hip.set_precision(args.precision)
                _note("fp32 leg done")
            except Exception as exc:
                out["fp32_path"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
        if world == 1 and not args.no_fp32:
            try:  # the MAS training step, captured since round 6
                out["mas_step"] = mas_leg(hip, cfg, args.precision, dev)
                hip.set_precision(args.precision)
                _note("MAS leg done")
            except Exception as exc:
                out["mas_step"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
            torch.cuda.empty_cache()
        if world == 1 and not args.no_hifigan:
            try:
                out["hifigan"] = hifigan_leg(hip, args.precision)
                _note("HiFi-GAN leg done")
                if not args.no_cpu_baseline:
                    out["hifigan"]["cpu_baseline"] = hifigan_cpu_baseline()
                    _note("HiFi-GAN CPU baseline done")
            except Exception as exc:  # the SAM-BERT line must survive a failure of the secondary leg
                out["hifigan"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
            try:
                out["melspec"] = melspec_leg()
                _note("mel-STFT leg done")
            except Exception as exc:
                out["melspec"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
        if world == 1 and not args.no_inference:
            try:
                out["inference"] = inference_leg(hip, cfg, args.precision, parity_utts=0 if args.no_cpu_baseline else 32,
                                                 cpu_threads=_cpu_threads())
                _note("inference leg done")
            except Exception as exc:
                out["inference"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"], out["parity_error"] = cpu_baseline(cfg, hip)
                _note("SAM-BERT CPU baseline + parity done")
            except Exception as exc:
                out["cpu_baseline"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
