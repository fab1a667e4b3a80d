"""TEST INFRASTRUCTURE -- generates tests/golden/*.pt by running the UNTOUCHED reference on CPU.

Run in the build container only (needs /root/reference):  python oracle/make_golden.py
The fixtures hold seeded inputs, reference outputs / losses / selected gradients and a checksum per
state_dict entry (the weights themselves are reproduced from the seed: the product and the reference
construct their parameters in the same order, so ``torch.manual_seed(s); Model(cfg)`` gives identical
tensors -- tests/test_oracle_golden.py verifies that through the checksums).
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

ref_harness.import_reference()
import torch_oracle as O  # noqa: E402
from kantts.models.sambert.kantts_sambert import KanTtsSAMBERT  # noqa: E402
from kantts.train.loss import MelReconLoss, ProsodyReconLoss  # noqa: E402
from kantts.utils.audio_torch import MelSpectrogram, stft  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def checksums(sd):
    return {k: (tuple(v.shape), float(v.double().sum()), float(v.double().abs().sum())) for k, v in sd.items()}


def sambert_case(name, tiny, B, T_in, min_len, dur_hi, seed_w=0, seed_b=1234, grad_keys=()):
    cfg = O.sambert_config(tiny=tiny)
    torch.manual_seed(seed_w)
    m = KanTtsSAMBERT(dict(cfg))
    m.eval()  # dropout off (incl. the hard-coded Prenet Dropout(0.5)); grads still flow
    batch = O.synthetic_sambert_batch(B=B, T_in=T_in, seed=seed_b, min_len=min_len, dur_hi=dur_hi)
    res = m(**batch)
    mel_, mel = MelReconLoss()(batch["output_lengths"], batch["mel_targets"], res["dec_outputs"], res["postnet_outputs"])
    d, p, e = ProsodyReconLoss()(batch["input_lengths"], res["duration_targets"], res["pitch_targets"],
                                 res["energy_targets"], res["log_duration_predictions"], res["pitch_predictions"],
                                 res["energy_predictions"])
    total = mel_ + mel + d + p + e
    total.backward()
    grads = {n: p_.grad.clone() for n, p_ in m.named_parameters() if p_.grad is not None and n in grad_keys}
    gsum = {n: (float(p_.grad.double().sum()), float(p_.grad.double().norm())) for n, p_ in m.named_parameters()
            if p_.grad is not None}
    keep = ["dec_outputs", "postnet_outputs", "LR_length_rounded", "log_duration_predictions", "pitch_predictions",
            "energy_predictions", "LR_text_outputs", "LR_emo_outputs", "LR_spk_outputs"]
    fix = dict(
        cfg=cfg, seed_w=seed_w, batch_args=dict(B=B, T_in=T_in, seed=seed_b, min_len=min_len, dur_hi=dur_hi),
        outputs={k: res[k].detach().clone() for k in keep},
        x_band_width=res["x_band_width"], h_band_width=res["h_band_width"],
        attn_checks=dict(enc0=res["enc_slf_attn_lst"][0].detach()[:2].clone(),
                         pnca_x_last=res["pnca_x_attn_lst"][-1].detach()[:2].clone(),
                         pnca_h_last=res["pnca_h_attn_lst"][-1].detach()[:2].clone()),
        losses=dict(mel_loss_=float(mel_), mel_loss=float(mel), dur_loss=float(d), pitch_loss=float(p),
                    energy_loss=float(e), total=float(total)),
        grads=grads, grad_summaries=gsum, weight_checksums=checksums(m.state_dict()),
        torch_version=torch.__version__,
    )
    torch.save(fix, os.path.join(OUT, name + ".pt"))
    print(name, "loss", float(total), "bytes", os.path.getsize(os.path.join(OUT, name + ".pt")))


def sambert_infer_case(name, B, T_in, min_len, seed_w=0, seed_b=77, dur_bias=1.5):
    """Free-running inference of the untouched reference (no targets): AR duration loop, predicted-duration length
    regulation, AR decoder loop.  With random-init weights the duration ReLU outputs 0 and the reference crashes on
    an empty memory, so the duration head's bias is set to ``dur_bias`` (~3.5 frames / token);
    the test applies the same override."""
    cfg = O.sambert_config(tiny=True)
    torch.manual_seed(seed_w)
    m = KanTtsSAMBERT(dict(cfg))
    m.eval()
    with torch.no_grad():
        m.variance_adaptor.duration_predictor.fc.bias.fill_(dur_bias)
    batch = O.synthetic_sambert_batch(B=B, T_in=T_in, seed=seed_b, min_len=min_len, dur_hi=6)
    args = {k: batch[k] for k in ("inputs_ling", "inputs_emotion", "inputs_speaker", "input_lengths")}
    with torch.no_grad():
        res = m(**args)
    keep = ["dec_outputs", "postnet_outputs", "LR_length_rounded", "log_duration_predictions", "pitch_predictions",
            "energy_predictions", "LR_text_outputs"]
    fix = dict(cfg=cfg, seed_w=seed_w, dur_bias=dur_bias,
               batch_args=dict(B=B, T_in=T_in, seed=seed_b, min_len=min_len, dur_hi=6),
               outputs={k: res[k].detach().clone() for k in keep}, x_band_width=res["x_band_width"],
               weight_checksums=checksums(m.state_dict()), torch_version=torch.__version__)
    torch.save(fix, os.path.join(OUT, name + ".pt"))
    print(name, "frames", res["LR_length_rounded"].tolist(), "xbw", res["x_band_width"], "bytes",
          os.path.getsize(os.path.join(OUT, name + ".pt")))


def collate_case():
    """Reference batch assembly (AM_Dataset.collate_fn with its Padder, Voc_Dataset.collate_fn) on random items."""
    import numpy as np
    import kantts.datasets.dataset as D

    rs = np.random.RandomState(11)
    pad_ids = [146, 9, 7, 7, 35, 3]
    items = []
    for n_sym, frames in ((7, 31), (12, 44), (3, 9), (9, 45)):
        ling = [rs.randint(0, pad_ids[k], size=n_sym + 1).astype(np.int64) for k in range(6)]  # incl. the "~" slot
        dur = rs.randint(1, 8, size=n_sym).astype(np.int64)
        dur[-1] += frames - dur.sum() if frames > dur.sum() else 0
        frames = int(dur.sum())
        items.append((ling, rs.randn(frames, 80).astype(np.float32), dur, rs.randn(n_sym + 1).astype(np.float32),
                      rs.randn(n_sym + 1).astype(np.float32), None, None, None))

    class _LU:
        _lfeat_type_list = ["sy", "tone", "syllable_flag", "word_segment", "emo_category", "speaker_category"]
        _sub_unit_pad = dict(zip(_lfeat_type_list, pad_ids))

        def using_byte(self):
            return False

    class _Self:
        ling_unit, padder, with_duration, se_enable, fp_enable, r = _LU(), D.Padder(), True, False, False, 3

    am = D.AM_Dataset.collate_fn(_Self(), items)

    class _V:
        hop_length, batch_max_steps, batch_max_frames, aux_context_window = 200, 1600, 8, 0
        start_offset, end_offset = 0, -8

    vitems = [(rs.randn(f * 200).astype(np.float32), rs.randn(f, 80).astype(np.float32)) for f in (20, 9, 33)]
    np.random.seed(5)
    wav_b, mel_b = D.Voc_Dataset.collate_fn(_V(), vitems)
    torch.save(dict(items=items, pad_ids=pad_ids, r=3, am={k: v for k, v in am.items()}, vitems=vitems, voc_seed=5,
                    voc=(wav_b, mel_b)), os.path.join(OUT, "collate.pt"))
    print("collate bytes", os.path.getsize(os.path.join(OUT, "collate.pt")))


def melspec_case():
    g = torch.Generator().manual_seed(7)
    x = torch.randn(4, 2048, generator=g) * 0.1
    fix = dict(wav=x, mel_v1=MelSpectrogram()(x[:, None, :]).clone(),
               mel_16k=MelSpectrogram(fs=16000, fft_size=2048, hop_size=200, win_length=1000, fmin=0, fmax=8000)(
                   x[:, None, :]).clone(),
               stft_1024_120_600=stft(x, 1024, 120, 600, torch.hann_window(600)).clone())
    torch.save(fix, os.path.join(OUT, "melspec.pt"))
    print("melspec bytes", os.path.getsize(os.path.join(OUT, "melspec.pt")))


if __name__ == "__main__":
    gk = ("text_encoder.ling_proj.weight", "mel_decoder.mel_dec.dec_out_proj.weight", "mel_postnet.fc.weight",
          "variance_adaptor.duration_predictor.fc.weight", "emo_tokenizer.weight")
    sambert_case("sambert_tiny", True, B=3, T_in=12, min_len=6, dur_hi=6, grad_keys=gk)
    sambert_case("sambert_tiny16", True, B=16, T_in=32, min_len=12, dur_hi=9, grad_keys=gk)
    # the reference's free-running path only works at batch 1 (its band masks are built without a batch axis)
    sambert_infer_case("sambert_tiny_infer", B=1, T_in=12, min_len=6)
    melspec_case()
    collate_case()
