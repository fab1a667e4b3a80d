"""TEST INFRASTRUCTURE -- live comparison of oracle/torch_oracle.py with the untouched reference
(build container only).  Exit code 0 = restatement matches (outputs <= 2e-6, grads rel-L2 <= 1e-5)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_harness  # noqa: E402

ref_harness.import_reference()
import audio_oracle as A  # noqa: E402
import torch_oracle as O  # noqa: E402
from kantts.models.sambert.kantts_sambert import KanTtsSAMBERT  # noqa: E402
from kantts.train.loss import MelReconLoss, ProsodyReconLoss  # noqa: E402
from kantts.utils.audio_torch import MelSpectrogram  # noqa: E402

ok = True
for tiny, B, T_in, min_len in ((True, 4, 24, 10), (False, 4, 48, 20)):
    cfg = O.sambert_config(tiny=tiny)
    torch.manual_seed(0)
    m = KanTtsSAMBERT(dict(cfg))
    m.eval()
    batch = O.synthetic_sambert_batch(B=B, T_in=T_in, min_len=min_len)
    ref = m(**batch)
    P = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    out = O.sambert_forward(P, cfg, **batch)
    for k in ["dec_outputs", "postnet_outputs", "log_duration_predictions", "pitch_predictions",
              "energy_predictions", "LR_text_outputs"]:
        err = (ref[k] - out[k]).abs().max().item()
        ok &= err <= 2e-6
        print(k, err)
    ok &= torch.equal(ref["LR_length_rounded"], out["LR_length_rounded"]) and ref["x_band_width"] == out["x_band_width"]
    for key in ["enc_slf_attn_lst", "pnca_x_attn_lst", "pnca_h_attn_lst"]:
        ok &= max((a - b).abs().max().item() for a, b in zip(ref[key], out[key])) <= 1e-6
    mel_, mel = MelReconLoss()(batch["output_lengths"], batch["mel_targets"], ref["dec_outputs"], ref["postnet_outputs"])
    d, p, e = ProsodyReconLoss()(batch["input_lengths"], ref["duration_targets"], ref["pitch_targets"],
                                 ref["energy_targets"], ref["log_duration_predictions"], ref["pitch_predictions"],
                                 ref["energy_predictions"])
    (mel_ + mel + d + p + e).backward()
    O.sambert_losses(out, batch["input_lengths"], batch["output_lengths"], batch["mel_targets"])["total"].backward()
    worst = max(((p_.grad - P[n].grad).norm() / (p_.grad.norm() + 1e-12)).item()
                for n, p_ in m.named_parameters() if p_.grad is not None)
    print("worst grad rel", worst)
    ok &= worst <= 1e-5
# ---- HiFi-GAN: oracle/hifigan_oracle.py against the untouched reference (V1 class defaults at a small batch, and a
# narrow generator), forward and parameter gradients
import hifigan_oracle as H  # noqa: E402
from kantts.models.hifigan.hifigan import Generator, MultiPeriodDiscriminator, MultiScaleDiscriminator  # noqa: E402

for channels, B, frames in ((512, 1, 8), (64, 2, 8)):
    torch.manual_seed(0)
    G, D1, D2 = Generator(channels=channels), MultiPeriodDiscriminator(), MultiScaleDiscriminator()
    g = torch.Generator().manual_seed(1)
    xm = torch.randn(B, 80, frames, generator=g)
    yw = torch.randn(B, 1, frames * 256, generator=g).clamp(-1, 1)
    PG = {k: v.detach().clone().requires_grad_(True) for k, v in G.state_dict().items()}
    a, b = G(xm), H.generator(PG, xm)
    err = (a - b).abs().max().item()
    a.sum().backward()
    b.sum().backward()
    gw = max(((p_.grad - PG[n].grad).norm() / (p_.grad.norm() + 1e-12)).item() for n, p_ in G.named_parameters())
    print("hifigan G ch=%d" % channels, err, "grad", gw)
    ok &= err <= 2e-6 and gw <= 1e-4
    for D, f, nm in ((D1, H.mpd, "mpd"), (D2, H.msd, "msd")):
        P = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in D.state_dict().items()}
        o, fm = D(yw)
        o2, fm2 = f(P, yw)
        err = max((u - v).abs().max().item() for u, v in zip(o, o2))
        errf = max((u - v).abs().max().item() for fa, fb in zip(fm, fm2) for u, v in zip(fa, fb))
        (sum((u * u).mean() for u in o) + sum(u.abs().mean() for fa in fm for u in fa)).backward()
        (sum((u * u).mean() for u in o2) + sum(u.abs().mean() for fa in fm2 for u in fa)).backward()
        gw = max(((p_.grad - P[n].grad).norm() / (p_.grad.norm() + 1e-12)).item() for n, p_ in D.named_parameters())
        print("hifigan", nm, err, errf, "grad", gw)
        ok &= err <= 2e-6 and errf <= 2e-5 and gw <= 1e-4
x = torch.randn(3, 4096) * 0.1
ok &= (MelSpectrogram()(x[:, None]) - A.mel_spectrogram(x)).abs().max().item() <= 1e-6
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
