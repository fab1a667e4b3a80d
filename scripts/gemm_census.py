"""Where the contraction time of one SAM-BERT training step goes, by launch shape: the step is run eagerly with every
bgemm_nt / bgemm_tn (grouped weight gradients are issued one by one) / ffn_pair launch bracketed by HIP events.
Usage (GPU box): python scripts/gemm_census.py"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import kantts._hip as hip  # noqa: E402
from kantts._hip import ops, ops_bf16  # noqa: E402

records = []


def wrap(mod, name, describe):
    orig = getattr(mod, name)

    def f(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(*a, **k)
        e1.record()
        records.append((describe(a, k), e0, e1))
        return r

    for m in (hip, ops, ops_bf16):
        if getattr(m, name, None) is orig:
            setattr(m, name, f)


def d_nt(a, k):
    segs, M, N = a[0], a[1], a[2]
    at = segs[0][0][0] if isinstance(segs[0][0], tuple) else segs[0][0]
    c = a[3][0] if isinstance(a[3], tuple) else a[3]
    return "nt  M=%-6d N=%-5d K=%-5d segs=%d A=%s C=%s%s%s" % (
        M, N, sum(s[4] for s in segs), len(segs), str(at.dtype)[6:], str(c.dtype)[6:], " kn" if k.get("b_kn") else "",
        " drop" if (k.get("drop_p", 0) > 0 or k.get("a_drop_p", 0) > 0) else "")


def d_tn(a, k):
    at = a[0][0] if isinstance(a[0], tuple) else a[0]
    bt = a[2][0] if isinstance(a[2], tuple) else a[2]
    return "tn  M=%-6d N=%-5d K=%-5d taps=%d A=%s B=%s" % (a[4], a[5], a[6], k.get("ntaps", 1), str(at.dtype)[6:],
                                                            str(bt.dtype)[6:])


def d_ffn(a, k):
    return "ffn M=%-6d KT=%d %s" % (k["M"], k.get("KT", 1), "bwd" if k.get("gate") is not None else "fwd")


def main():
    import bench
    import torch_oracle as O
    from kantts.models import model_builder
    from kantts.train.loss import MelReconLoss, ProsodyReconLoss

    hip.set_precision("bf16")
    cfg = O.sambert_config(tiny=False)
    torch.manual_seed(1234)
    model, opt, _ = model_builder(bench.sambert_yaml_config(cfg), device="cuda")
    net, optimizer = model["KanTtsSAMBERT"], opt["KanTtsSAMBERT"]
    net.train()
    batch = {k: v.cuda() for k, v in O.synthetic_sambert_batch(B=32, T_in=64, seed=1234).items()}
    mel_crit, pros_crit = MelReconLoss(), ProsodyReconLoss()

    def step():
        optimizer.zero_grad()
        res = net(**batch)
        a, b = mel_crit(batch["output_lengths"], batch["mel_targets"], res["dec_outputs"], res["postnet_outputs"])
        d, p, e = pros_crit(batch["input_lengths"], res["duration_targets"], res["pitch_targets"], res["energy_targets"],
                            res["log_duration_predictions"], res["pitch_predictions"], res["energy_predictions"])
        (a + b + d + p + e).backward()
        ops.wgrad_overlap.join()
        optimizer.step()

    ops.wgrad_overlap.enable(False)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    wrap(hip, "bgemm_nt", d_nt)
    wrap(hip, "bgemm_tn", d_tn)
    wrap(hip, "ffn_pair", d_ffn)
    step()
    torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for key, e0, e1 in records:
        v = agg.setdefault(key, [0, 0.0])
        v[0] += 1
        v[1] += e0.elapsed_time(e1) * 1e3
    tot = sum(v[1] for v in agg.values())
    print("total %.2f ms in %d launches (eager, event-bracketed: includes ~3 us of launch gap each)" % (tot / 1e3, len(records)))
    for key, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        print("%-72s x%-3d %8.1f us total %7.1f us each  %4.1f%%" % (key, n, us, us / n, 100 * us / tot))


if __name__ == "__main__":
    main()
