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
x = torch.randn(3, 4096) * 0.1
ok &= (MelSpectrogram()(x[:, None]) - A.mel_spectrogram(x)).abs().max().item() <= 1e-6
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
