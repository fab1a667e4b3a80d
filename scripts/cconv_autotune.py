"""Per-shape tile sweep of kantts_cconv_launch (csrc/cconv.hip) on the device -- an EXPERIMENT of round 6, measured and not
adopted: the table it writes is not read by the product.

The launcher picks its output tile (128x128 ... 256x32) and LDS ring depth (2 / 3 / 4 stages) with a heuristic that round 4
tuned on a handful of shapes; a HiFi-GAN V1 training step issues ~100 distinct convolution shapes through it, vocoder
inference another few dozen.  This script records every distinct shape of (a) one GAN training step at batch 32 x 8192 and
(b) one generator forward at an inference shape, times each of them under every tile variant (a replayed hipGraph of 8
back-to-back launches on synthetic operands; results are bit-identical for every tile) and writes, for the shapes where a
variant beats the heuristic by more than 4 %, the winning tile code.

Result (profiles/r06_runN_cconv_autotune.log, r06_runO_gan_step_tile_table_ab.log): 397 shapes, 100 of them with a variant
more than 4 % faster IN ISOLATION (1024 -> 1024, k = 5 at 65 rows: 136 -> 75 us with 64 x 128 tiles), 2.4 of 65 ms of
serialised launch time -- but with the table consulted by kantts._hip.cconv the captured GAN step did not move (26.59 /
26.75 / 26.99 ms without, 26.73 / 26.97 ms with, interleaved on one box) and the generator forward got slower (2.11 -> 2.20 ms):
what a tile gains back to back against itself it does not gain between other kernels of a step that already overlaps 1.5
launches on average.  The lookup was removed again; the launcher's heuristic stays.

Usage (GPU box):  python scripts/cconv_autotune.py [--out PATH] [--min-gain 0.04]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402

import kantts._hip as hip  # noqa: E402
import kantts._hip.ops as ops  # noqa: E402

def shape_key(B, Tsrc, Tdst, inner, groups, CR, NG, K, in_mul, in_div, in_kstep, phases, up):
    """What the speed of a kantts_cconv_launch depends on (and nothing its result depends on)."""
    return "%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d" % (B, Tsrc, Tdst, inner, groups, CR, NG, K, in_mul, in_div, in_kstep,
                                                       phases, up)


TILES = [128128, 256064, 128064, 64128, 64064, 256032]
STAGES = {128128: (0, 3, 4), 256064: (0, 3), 128064: (0, 3, 4), 64128: (0, 3, 4), 64064: (0, 3, 4), 256032: (0, 3)}


def record_shapes():
    from hifigan_bench import v1_config
    from kantts.models import model_builder
    from kantts.train.gan_step import gan_train_step
    from kantts.train.loss import criterion_builder

    seen = {}
    real = hip.cconv

    def spy(x_bf, w_bf, **kw):
        key = shape_key(kw["B"], kw["Tsrc"], kw["Tdst"], kw.get("inner", 1), kw["groups"], kw["CR"], kw["NG"],
                                  kw["K"], kw["in_mul"], kw["in_div"], kw["in_kstep"], kw["phases"], kw.get("up", 1))
        if key not in seen and not kw.get("tile"):
            meta = {k: v for k, v in kw.items() if not torch.is_tensor(v) and k != "tile"}
            for name in ("out", "out_bf", "bias", "res", "out_gate"):
                t = kw.get(name)
                meta[name] = None if t is None else (tuple(t.shape), str(t.dtype))
            meta["x"] = (tuple(x_bf.shape), str(x_bf.dtype))
            meta["w"] = (tuple(w_bf.shape), str(w_bf.dtype))
            seen[key] = [meta, 0]
        if key in seen:
            seen[key][1] += 1
        return real(x_bf, w_bf, **kw)

    hip.cconv = ops.cconv = spy
    try:
        hip.set_precision("bf16")
        config = v1_config()
        torch.manual_seed(0)
        model, optimizer, scheduler = model_builder(config, device="cuda")
        crit = criterion_builder(config, device="cuda")
        x = torch.randn(32, 80, 32, device="cuda")
        y = torch.randn(32, 1, 8192, device="cuda").clamp(-1, 1)
        for _ in range(2):
            gan_train_step(model, optimizer, scheduler, crit, config, y, x, steps=1)
        n_train = len(seen)
        G = model["generator"].eval()
        G.remove_weight_norm()
        with torch.no_grad():
            for frames in (320, 256, 192, 128):  # the length-sorted groups of the inference leg
                G(torch.randn(32, 80, frames, device="cuda"))
        torch.cuda.synchronize()
    finally:
        hip.cconv = ops.cconv = real
    return seen, n_train


def graph_us(fn, reps=8, n=5):
    s_ = torch.cuda.Stream()
    s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        ok = fn()
    torch.cuda.current_stream().wait_stream(s_)
    torch.cuda.synchronize()
    if not ok:
        return None
    g_ = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_, capture_error_mode="thread_local"):
        for _ in range(reps):
            fn()
    g_.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g_.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "cconv_tiles_gfx950.json"))
    ap.add_argument("--min-gain", type=float, default=0.04)
    a = ap.parse_args()
    seen, n_train = record_shapes()
    print("%d distinct shapes (%d of the training step)" % (len(seen), n_train), flush=True)
    dt = {"torch.bfloat16": torch.bfloat16, "torch.float32": torch.float32, "torch.uint8": torch.uint8, "torch.bool": torch.bool}
    tiles, report = {}, []
    for key, (meta, calls) in seen.items():
        def mk(spec, rnd=False):
            if spec is None:
                return None
            shape, d = spec
            t = (torch.randn(shape, device="cuda") * 0.1) if rnd else torch.zeros(shape, device="cuda")
            return t.to(dt[d])

        x, w = mk(meta["x"], True), mk(meta["w"], True)
        tens = {n: mk(meta[n], n in ("res", "out_gate", "bias")) for n in ("out", "out_bf", "bias", "res", "out_gate")}
        kw = {k: v for k, v in meta.items() if k not in ("x", "w", "out", "out_bf", "bias", "res", "out_gate")}

        def run(tile):
            try:
                return hip.cconv(x, w, tile=tile, **tens, **kw)
            except RuntimeError:
                return False

        base = graph_us(lambda: run(0))
        if base is None:
            continue
        best_t, best_us = 0, base
        for t in TILES:
            for st in STAGES[t]:
                code = st * 1000000 + t
                us = graph_us(lambda: run(code))
                if us is not None and us < best_us:
                    best_t, best_us = code, us
        gain = 1.0 - best_us / base
        if best_t and gain > a.min_gain:
            tiles[key] = best_t
        report.append((calls * (base - best_us if (best_t and gain > a.min_gain) else 0.0), key, calls, base, best_us, best_t))
        print("%-48s calls %3d  heuristic %7.1f us  best %7.1f us  tile %8d  %s" % (
            key, calls, base, best_us, best_t, "TABLE" if key in tiles else ""), flush=True)
    saved = sum(r[0] for r in report)
    total = sum(r[2] * r[3] for r in report)
    print("serialised launch time of the recorded calls: %.2f ms with the heuristic, %.2f ms saved by %d table entries" % (
        total / 1e3, saved / 1e3, len(tiles)))
    with open(a.out, "w") as f:
        json.dump({"device": torch.cuda.get_device_name(0), "how": "scripts/cconv_autotune.py: replayed graphs of 8 launches, best "
                   "of 5; entries where a variant beats the launcher's heuristic by more than %.0f %%" % (100 * a.min_gain),
                   "key": "B,Tsrc,Tdst,inner,groups,CR,NG,K,in_mul,in_div,in_kstep,phases,up", "tiles": tiles}, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
