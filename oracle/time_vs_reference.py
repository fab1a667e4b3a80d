"""TEST INFRASTRUCTURE -- SPEED of the oracle ("port") against the untouched reference, in one process (build container
only: the GPU box has no /root/reference).  ``bench.py`` times the port as ``cpu_baseline`` (kind "port") because the
reference cannot travel; this script pins how far the port's speed is from the reference's so that the bench line can say
``cpu_baseline.port_over_reference`` (seconds per iteration of the port / of the reference, same host, same threads; > 1 =
the port is slower, i.e. the baseline UNDERSTATES the reference).

Legs:
  sambert      B = 32 seeded batch (the bench's), forward + losses + backward, dropout off and on, 8 threads and all threads
  hifigan      V1 512 channels, B = 4 x 8192 GAN step (generator phase + discriminator phase, no optimizer) -- the same
               arithmetic as bench.py::hifigan_cpu_baseline
  inference    free-running symbols -> mel, one utterance at a time (the reference's only mode)

Usage:  python oracle/time_vs_reference.py [--out profiles/r06_time_vs_reference.json] [--quick]
Reference call sites: kantts/models/sambert/kantts_sambert.py:862-1044, kantts/train/loss.py:7-85,108-310,
kantts/train/trainer.py:469-589, kantts/bin/infer_sambert.py:58-227.
"""
import argparse
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

ref_harness.import_reference()
import audio_oracle as A  # noqa: E402,F401
import hifigan_oracle as H  # noqa: E402
import torch_oracle as O  # noqa: E402
from kantts.models.hifigan.hifigan import Generator, MultiPeriodDiscriminator, MultiScaleDiscriminator  # noqa: E402
from kantts.models.sambert.kantts_sambert import KanTtsSAMBERT  # noqa: E402
from kantts.train.loss import (DiscriminatorAdversarialLoss, FeatureMatchLoss, GeneratorAdversarialLoss,  # noqa: E402
                               MelReconLoss, MelSpectrogramLoss, ProsodyReconLoss)


def _time(fn, warm, n):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], ts


def _pair(ref_fn, port_fn, warm, n):
    """Interleaved medians: reference, port, reference, port ... so that drift of the host hits both alike."""
    for _ in range(warm):
        ref_fn()
        port_fn()
    tr, tp = [], []
    for _ in range(n):
        t0 = time.perf_counter()
        ref_fn()
        t1 = time.perf_counter()
        port_fn()
        t2 = time.perf_counter()
        tr.append(t1 - t0)
        tp.append(t2 - t1)
    tr.sort()
    tp.sort()
    r, p = tr[len(tr) // 2], tp[len(tp) // 2]
    # the ratio of the BEST iterations is the headline (a shared 8-core container: single iterations vary by +-30 %, always
    # upwards); the medians are kept beside it
    return {"reference_s": tr[0], "port_s": tp[0], "port_over_reference": tp[0] / tr[0], "iters": n,
            "reference_median_s": r, "port_median_s": p, "port_over_reference_medians": p / r,
            "reference_min_max_s": [tr[0], tr[-1]], "port_min_max_s": [tp[0], tp[-1]]}


def sambert_leg(B, threads, dropout, warm, n):
    cfg = O.sambert_config(tiny=False)
    torch.manual_seed(0)
    m = KanTtsSAMBERT(dict(cfg))
    m.train(dropout)
    batch = O.synthetic_sambert_batch(B=B, T_in=64, seed=1234)
    P = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    mc, pc = MelReconLoss(), ProsodyReconLoss()

    def ref_step():
        m.zero_grad(set_to_none=True)
        r = m(**batch)
        mel_, mel = mc(batch["output_lengths"], batch["mel_targets"], r["dec_outputs"], r["postnet_outputs"])
        d, p, e = pc(batch["input_lengths"], r["duration_targets"], r["pitch_targets"], r["energy_targets"],
                     r["log_duration_predictions"], r["pitch_predictions"], r["energy_predictions"])
        (mel_ + mel + d + p + e).backward()

    def port_step():
        for v in P.values():
            v.grad = None
        out = O.sambert_forward(P, cfg, **batch)
        O.sambert_losses(out, batch["input_lengths"], batch["output_lengths"], batch["mel_targets"])["total"].backward()

    torch.set_num_threads(threads)
    O.DROP["on"] = bool(dropout)
    try:
        res = _pair(ref_step, port_step, warm, n)
    finally:
        O.DROP["on"] = False
    res.update(threads=threads, batch=B, dropout=bool(dropout), frames=int(batch["output_lengths"].sum()))
    return res


def hifigan_leg(B, threads, warm, n, T_wav=8192):
    torch.manual_seed(0)
    G, D1, D2 = Generator(), MultiPeriodDiscriminator(), MultiScaleDiscriminator()
    PG = {k: v.detach().clone().requires_grad_(True) for k, v in G.state_dict().items()}
    PP = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in D1.state_dict().items()}
    PS = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in D2.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 80, T_wav // 256, generator=g)
    y = torch.randn(B, 1, T_wav, generator=g).clamp(-1, 1)
    gadv, dadv, fmc, melc = GeneratorAdversarialLoss(), DiscriminatorAdversarialLoss(), FeatureMatchLoss(), MelSpectrogramLoss()

    def ref_step():  # kantts/train/trainer.py:469-589 without logging / optimizer updates
        for mod in (G, D1, D2):
            mod.zero_grad(set_to_none=True)
        y_ = G(x)
        gen = 45.0 * melc(y_, y)
        adv, fm_ = 0.0, []
        for D in (D1, D2):
            p_, f_ = D(y_)
            fm_.append(f_)
            adv = adv + gadv(p_)
        fm = 0.0
        for D, f_ in zip((D1, D2), fm_):
            with torch.no_grad():
                _, f = D(y)
            fm = fm + fmc(f_, f)
        (gen + adv + 2.0 * fm).backward()
        for mod in (G, D1, D2):
            mod.zero_grad(set_to_none=True)
        with torch.no_grad():
            y2 = G(x)
        dl = 0.0
        for D in (D1, D2):
            p, _ = D(y)
            p_, _ = D(y2.detach())
            real, fake = dadv(p_, p)
            dl = dl + real + fake
        dl.backward()

    def port_step():  # bench.py::hifigan_cpu_baseline
        for P in (PG, PP, PS):
            for p in P.values():
                p.grad = None
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
        for P in (PG, PP, PS):
            for p in P.values():
                p.grad = None
        with torch.no_grad():
            y2 = H.generator(PG, x)
        dl = 0.0
        for P, f in ((PP, H.mpd), (PS, H.msd)):
            o, _ = f(P, y)
            o_, _ = f(P, y2)
            real, fake = H.dis_adv_loss(o_, o)
            dl = dl + real + fake
        dl.backward()

    torch.set_num_threads(threads)
    res = _pair(ref_step, port_step, warm, n)
    res.update(threads=threads, batch=B, samples=B * T_wav)
    return res


def inference_leg(n_utt, threads, warm, n):
    cfg = O.sambert_config(tiny=False)
    torch.manual_seed(0)
    m = KanTtsSAMBERT(dict(cfg))
    with torch.no_grad():
        m.variance_adaptor.duration_predictor.fc.bias.fill_(1.5)  # as bench.py: ~3.5 frames per symbol
    m.eval()
    P = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(7)
    utts = []
    for _ in range(n_utt):
        T = int(torch.randint(20, 81, (1,), generator=g))
        utts.append(dict(inputs_ling=torch.stack([torch.randint(0, hi, (1, T), generator=g) for hi in (
            cfg["sy"], cfg["tone"], cfg["syllable_flag"], cfg["word_segment"])], -1),
            inputs_emotion=torch.randint(0, cfg["emotion"], (1, T), generator=g),
            inputs_speaker=torch.randint(0, cfg["speaker"], (1, T), generator=g), input_lengths=torch.tensor([T])))
    frames = {}

    def ref_run():
        with torch.no_grad():
            frames["ref"] = sum(int(m(**u)["LR_length_rounded"][0]) for u in utts)

    def port_run():
        with torch.no_grad():
            frames["port"] = sum(int(O.sambert_forward(P, cfg, **u)["LR_length_rounded"][0]) for u in utts)

    torch.set_num_threads(threads)
    res = _pair(ref_run, port_run, warm, n)
    assert frames["ref"] == frames["port"], frames
    res.update(threads=threads, utterances=n_utt, mel_frames=frames["ref"])
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--quick", action="store_true", help="B = 8 / 1 timed iteration: a smoke run of this script")
    a = ap.parse_args()
    all_thr = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    B = 8 if a.quick else 32
    warm, n = (1, 1) if a.quick else (1, 4)
    out = {"host_threads": all_thr, "torch": torch.__version__, "legs": {}}
    for thr in sorted({8, all_thr}):
        for drop in (False, True):
            key = "sambert_b%d_t%d_dropout_%s" % (B, thr, "on" if drop else "off")
            out["legs"][key] = sambert_leg(B, thr, drop, warm, n)
            print(key, json.dumps(out["legs"][key]), flush=True)
    out["legs"]["hifigan_b4_gan_step_t%d" % all_thr] = hifigan_leg(2 if a.quick else 4, all_thr, 1, 1 if a.quick else 3)
    print("hifigan", json.dumps(out["legs"]["hifigan_b4_gan_step_t%d" % all_thr]), flush=True)
    out["legs"]["inference_t%d" % all_thr] = inference_leg(2 if a.quick else 6, all_thr, 1, 1 if a.quick else 3)
    print("inference", json.dumps(out["legs"]["inference_t%d" % all_thr]), flush=True)
    ratios = [v["port_over_reference"] for v in out["legs"].values()]
    out["port_over_reference"] = {"min": min(ratios), "max": max(ratios),
                                  "sambert_b32_dropout_on": next(v["port_over_reference"] for k, v in out["legs"].items()
                                                                 if k.startswith("sambert") and k.endswith("dropout_on")),
                                  "sambert_b32_dropout_off": next(v["port_over_reference"] for k, v in out["legs"].items()
                                                                  if k.startswith("sambert") and k.endswith("dropout_off")),
                                  "hifigan_gan_step": out["legs"]["hifigan_b4_gan_step_t%d" % all_thr]["port_over_reference"],
                                  "inference": out["legs"]["inference_t%d" % all_thr]["port_over_reference"]}
    print(json.dumps(out["port_over_reference"]))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
