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


def mas_batch(B, T_in, min_len, dur_hi, seed_b):
    """Duration-free batch as AM_Dataset produces with MAS: True (dataset.py:498-503, 798-827): frame-level pitch /
    energy (with unvoiced zeros) and the beta-binomial prior over len+1 symbols (the trailing "~")."""
    from kantts.datasets.dataset import beta_binomial_prior_distribution

    batch = O.synthetic_sambert_batch(B=B, T_in=T_in, seed=seed_b, min_len=min_len, dur_hi=dur_hi)
    g = torch.Generator().manual_seed(seed_b + 1)
    T_mel = batch["mel_targets"].shape[1]
    valid = torch.arange(T_mel)[None, :] < batch["output_lengths"][:, None]
    pitch = torch.randn(B, T_mel, generator=g) * (torch.rand(B, T_mel, generator=g) > 0.3) * valid
    energy = torch.randn(B, T_mel, generator=g) * valid
    pri = torch.zeros(B, T_mel, T_in)
    for b in range(B):
        p = beta_binomial_prior_distribution(int(batch["input_lengths"][b]) + 1, int(batch["output_lengths"][b]))
        pri[b, :p.shape[0], :p.shape[1]] = p
    batch.update(duration_targets=None, pitch_targets=pitch, energy_targets=energy, attn_priors=pri)
    return batch


def sambert_mas_case(name, B, T_in, min_len, dur_hi, seed_w=0, seed_b=4321, epoch=50):
    from kantts.train.loss import AttentionBinarizationLoss, AttentionCTCLoss

    cfg = O.sambert_config(tiny=True)
    cfg["MAS"] = True
    torch.manual_seed(seed_w)
    m = KanTtsSAMBERT(dict(cfg))
    m.eval()
    batch = mas_batch(B, T_in, min_len, dur_hi, seed_b)
    # binarize_attention_parallel ends with .to(attn.get_device()), which is -1 (invalid) for CPU tensors: give it the
    # tensor's device for the duration of the call -- no arithmetic is touched
    orig_get_device = torch.Tensor.get_device
    torch.Tensor.get_device = lambda self: self.device
    try:
        res = m(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
    finally:
        torch.Tensor.get_device = orig_get_device
    mel_, mel = MelReconLoss()(batch["output_lengths"], batch["mel_targets"], res["dec_outputs"], res["postnet_outputs"])
    d, p, e = ProsodyReconLoss()(batch["input_lengths"], res["duration_targets"], res["pitch_targets"],
                                 res["energy_targets"], res["log_duration_predictions"], res["pitch_predictions"],
                                 res["energy_predictions"])
    ctc = AttentionCTCLoss()(res["attn_logprob"], batch["input_lengths"], batch["output_lengths"])
    kl = AttentionBinarizationLoss()(epoch, res["attn_hard"], res["attn_soft"])
    total = mel_ + mel + d + p + e + ctc + kl
    total.backward()
    grads = {n: p_.grad.clone() for n, p_ in m.named_parameters() if n.startswith("align_attention.") and p_.grad is not None and p_.numel() <= 20000}
    gsum = {n: (float(p_.grad.double().sum()), float(p_.grad.double().norm())) for n, p_ in m.named_parameters()
            if p_.grad is not None}
    keep = ["dec_outputs", "postnet_outputs", "LR_length_rounded", "log_duration_predictions", "attn_soft", "attn_hard",
            "attn_logprob", "duration_targets", "pitch_targets", "energy_targets"]
    fix = dict(cfg=cfg, seed_w=seed_w, epoch=epoch, batch=batch,
               outputs={k: res[k].detach().clone() for k in keep}, x_band_width=res["x_band_width"],
               losses=dict(mel_loss_=float(mel_), mel_loss=float(mel), dur_loss=float(d), pitch_loss=float(p),
                           energy_loss=float(e), attn_ctc_loss=float(ctc), attn_kl_loss=float(kl), total=float(total)),
               grads=grads, grad_summaries=gsum, weight_checksums=checksums(m.state_dict()),
               torch_version=torch.__version__)
    torch.save(fix, os.path.join(OUT, name + ".pt"))
    print(name, "loss", float(total), "ctc", float(ctc), "kl", float(kl), "bytes",
          os.path.getsize(os.path.join(OUT, name + ".pt")))


def sambert_se_case(name, B, T_in, min_len, dur_hi, seed_w=0, seed_b=555):
    """SE: True (configs/sambert_se_nsf_global_16k.yaml): the speaker stream is a 192-d embedding per token."""
    cfg = O.sambert_config(tiny=True)
    cfg["SE"] = True
    cfg["speaker_units"] = 192
    torch.manual_seed(seed_w)
    m = KanTtsSAMBERT(dict(cfg))
    m.eval()
    batch = O.synthetic_sambert_batch(B=B, T_in=T_in, seed=seed_b, min_len=min_len, dur_hi=dur_hi)
    g = torch.Generator().manual_seed(seed_b + 7)
    batch["inputs_speaker"] = torch.randn(B, 1, 192, generator=g).repeat(1, T_in, 1)
    res = m(**batch)
    mel_, mel = MelReconLoss()(batch["output_lengths"], batch["mel_targets"], res["dec_outputs"], res["postnet_outputs"])
    d, p, e = ProsodyReconLoss()(batch["input_lengths"], res["duration_targets"], res["pitch_targets"],
                                 res["energy_targets"], res["log_duration_predictions"], res["pitch_predictions"],
                                 res["energy_predictions"])
    total = mel_ + mel + d + p + e
    total.backward()
    gsum = {n: (float(p_.grad.double().sum()), float(p_.grad.double().norm())) for n, p_ in m.named_parameters()
            if p_.grad is not None}
    keep = ["dec_outputs", "postnet_outputs", "LR_length_rounded", "log_duration_predictions", "LR_spk_outputs"]
    fix = dict(cfg=cfg, seed_w=seed_w, batch=batch, outputs={k: res[k].detach().clone() for k in keep},
               x_band_width=res["x_band_width"], losses=dict(total=float(total)), grad_summaries=gsum,
               weight_checksums=checksums(m.state_dict()), state_keys=sorted(m.state_dict().keys()))
    torch.save(fix, os.path.join(OUT, name + ".pt"))
    print(name, "loss", float(total), "bytes", os.path.getsize(os.path.join(OUT, name + ".pt")))


def nsf_generator_case():
    """NSF HiFi-GAN generator (hifigan.py:22-197 with nsf_params, layers.py:229-290): forward + backward of the reference
    at a small width; the excitation's random phase / noise come from torch's global CPU generator, seeded right before
    the forward -- the product draws through the same torch.distributions calls, so a CPU run reproduces them."""
    from kantts.models.hifigan.hifigan import Generator

    res = {}
    for causal in (True, False):
        torch.manual_seed(3)
        G = Generator(in_channels=80, channels=32, upsample_scales=[4, 4, 2, 2], upsample_kernal_sizes=[8, 8, 4, 4],
                      causal=causal, nsf_params={"nb_harmonics": 7, "sampling_rate": 16000})
        g = torch.Generator().manual_seed(11)
        frames = 6
        mel = torch.randn(2, 80, frames, generator=g)
        f0 = 80 + 300 * torch.rand(2, 1, frames, generator=g)
        uv = (torch.rand(2, 1, frames, generator=g) > 0.3).float()
        x = torch.cat([mel, f0 * uv, uv], dim=1)
        torch.manual_seed(1234)
        y = G(x)
        cot = torch.randn(y.shape, generator=g)
        (y * cot).sum().backward()
        gsum = {n: float(p.grad.double().norm()) for n, p in G.named_parameters() if p.grad is not None}
        res["causal" if causal else "noncausal"] = dict(
            x=x, y=y.detach().clone(), cot=cot, grad_norms=gsum, state_keys=sorted(G.state_dict().keys()),
            weight_checksums=checksums(G.state_dict()))
    torch.save(res, os.path.join(OUT, "hifigan_nsf.pt"))
    print("hifigan_nsf bytes", os.path.getsize(os.path.join(OUT, "hifigan_nsf.pt")), float(res["causal"]["y"].abs().mean()))


def sambert_curve_case(steps=6):
    """Loss curve of the reference's own training step (Sambert_Trainer.train_step, trainer.py:898-1005: forward, five
    losses, backward, clip_grad_norm_(1.0), Adam(1e-3, (0.9, 0.98), eps 1e-9), NoamLR) over `steps` steps on cycling
    batches, dropout off (eval mode; incl. the Prenet's hard-wired Dropout(0.5)), NoamLR warm-up shortened to 4 steps so
    that the updates are large enough to bend the curve."""
    from kantts.train.scheduler import NoamLR

    cfg = O.sambert_config(tiny=True)
    torch.manual_seed(0)
    m = KanTtsSAMBERT(dict(cfg))
    m.eval()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=0.0)
    sch = NoamLR(opt, warmup_steps=4)
    batches = [O.synthetic_sambert_batch(B=3, T_in=12, seed=10 + s, min_len=6, dur_hi=6) for s in range(3)]
    losses, lrs = [], []
    for it in range(steps):
        b = batches[it % len(batches)]
        res = m(**b)
        mel_, mel = MelReconLoss()(b["output_lengths"], b["mel_targets"], res["dec_outputs"], res["postnet_outputs"])
        d, p, e = ProsodyReconLoss()(b["input_lengths"], res["duration_targets"], res["pitch_targets"],
                                     res["energy_targets"], res["log_duration_predictions"], res["pitch_predictions"],
                                     res["energy_predictions"])
        total = mel_ + mel + d + p + e
        opt.zero_grad()
        total.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
        losses.append(float(total))
    # parameter checksums after the last update: the whole optimisation trajectory in a few numbers
    fix = dict(cfg=cfg, steps=steps, losses=losses, lrs=lrs, final_checksums=checksums(m.state_dict()))
    torch.save(fix, os.path.join(OUT, "sambert_tiny_curve.pt"))
    print("curve", [round(x, 4) for x in losses], "bytes",
          os.path.getsize(os.path.join(OUT, "sambert_tiny_curve.pt")))


GAN_CURVE_CONFIG = {
    "model_type": "hifigan",
    "Model": {
        "Generator": {"params": {"channels": 32, "out_channels": 1},
                      "optimizer": {"type": "Adam", "params": {"lr": 2e-3, "betas": [0.5, 0.9], "weight_decay": 0.0}},
                      "scheduler": {"type": "MultiStepLR", "params": {"gamma": 0.5, "milestones": [3]}}},
        "MultiPeriodDiscriminator": {
            "params": {"periods": [2, 3]},
            "optimizer": {"type": "Adam", "params": {"lr": 2e-3, "betas": [0.5, 0.9], "weight_decay": 0.0}},
            "scheduler": {"type": "MultiStepLR", "params": {"gamma": 0.5, "milestones": [3]}}},
        "MultiScaleDiscriminator": {
            "params": {"scales": 2, "discriminator_params": {
                "in_channels": 1, "out_channels": 1, "kernel_sizes": [15, 41, 5, 3], "channels": 16,
                "max_downsample_channels": 64, "max_groups": 4, "bias": True, "downsample_scales": [2, 2, 4, 4, 1],
                "nonlinear_activation": "LeakyReLU", "nonlinear_activation_params": {"negative_slope": 0.1}}},
            "optimizer": {"type": "Adam", "params": {"lr": 2e-3, "betas": [0.5, 0.9], "weight_decay": 0.0}},
            "scheduler": {"type": "MultiStepLR", "params": {"gamma": 0.5, "milestones": [3]}}}},
    "Loss": {"generator_adv_loss": {"enable": True, "params": {}, "weights": 1.0},
             "discriminator_adv_loss": {"enable": True, "params": {}, "weights": 1.0},
             "mel_loss": {"enable": True, "params": {}, "weights": 45.0},
             "feat_match_loss": {"enable": True, "params": {}, "weights": 2.0}},
    "generator_grad_norm": 10.0, "discriminator_grad_norm": -1,  # > 0 crashes the reference (trainer.py:583: dict.parameters())
     "discriminator_train_start_steps": 0,
    "generator_train_start_steps": 0,
}


def gan_curve_case(steps=4):
    """Loss curve of the reference's GAN_Trainer.train_step (trainer.py:469-589) over `steps` steps: generator update
    (mel + adversarial + feature matching), generator re-run, discriminator updates, gradient clipping on both sides,
    MultiStepLR with a milestone inside the window; small widths (G 32 ch, MPD 2/3, MSD 2 scales x 16 ch).  The trainer's
    own method is executed on a bare instance (its logging / IO shell is not constructed)."""
    import copy
    from collections import defaultdict

    from kantts.models import model_builder
    from kantts.train.loss import criterion_builder
    from kantts.train.trainer import GAN_Trainer

    config = copy.deepcopy(GAN_CURVE_CONFIG)
    torch.manual_seed(0)
    model, optimizer, scheduler = model_builder(config, "cpu", 0, False)
    criterion = criterion_builder(config, "cpu")
    g = torch.Generator().manual_seed(3)
    batches = [(torch.randn(2, 1, 2048, generator=g).clamp(-1, 1) * 0.5, torch.randn(2, 80, 8, generator=g))
               for _ in range(2)]
    tr = GAN_Trainer.__new__(GAN_Trainer)
    tr.model, tr.optimizer, tr.scheduler, tr.criterion, tr.config = model, optimizer, scheduler, criterion, config
    tr.device = torch.device("cpu")
    tr.total_train_loss = defaultdict(float)
    curve = []
    init = {"generator": checksums(model["generator"].state_dict())}
    for it in range(steps):
        tr.steps = it + 1
        before = dict(tr.total_train_loss)
        tr.train_step(batches[it % 2])
        curve.append({k.split("/")[1]: tr.total_train_loss[k] - before.get(k, 0.0) for k in tr.total_train_loss})
    fix = dict(config=GAN_CURVE_CONFIG, steps=steps, curve=curve, init_checksums=init,
               final_checksums={"generator": checksums(model["generator"].state_dict()),
                                **{k: checksums(d.state_dict()) for k, d in model["discriminator"].items()}})
    torch.save(fix, os.path.join(OUT, "hifigan_curve.pt"))
    print("gan curve", [(round(c["generator_loss"], 4), round(c["discriminator_loss"], 4)) for c in curve], "bytes",
          os.path.getsize(os.path.join(OUT, "hifigan_curve.pt")))


def voc_dataset_case():
    """Items of the reference's Voc_Dataset (datasets/dataset.py:88-276) over a small synthetic data directory: plain and
    NSF configuration, an utterance shorter than a crop (zero-padded branch) and longer ones (reflect-padded branch).
    The raw file contents travel in the fixture so that the test can rebuild the directory."""
    import tempfile

    import numpy as np
    from scipy.io import wavfile

    import kantts.datasets.dataset as D

    rs = np.random.RandomState(21)
    hop, sr = 200, 16000
    utts = {}
    for name, frames in (("a01", 30), ("a02", 9), ("a03", 21), ("a04", 16)):
        n = frames * hop - rs.randint(0, hop)  # the wav is a little shorter than frames * hop, as after trimming
        utts[name] = dict(wav=np.clip(np.round(rs.randn(n) * 3000), -32768, 32767).astype(np.int16),
                          mel=rs.randn(frames, 80).astype(np.float32), f0=rs.randn(frames).astype(np.float32),
                          uv=(rs.rand(frames) > 0.4).astype(np.float32))
    f0_mean, f0_std = 187.25, 41.5
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for sub in ("wav", "mel", "frame_f0", "frame_uv", "f0"):
            os.makedirs(os.path.join(d, sub))
        for name, u in utts.items():
            wavfile.write(os.path.join(d, "wav", name + ".wav"), sr, u["wav"])
            np.save(os.path.join(d, "mel", name + ".npy"), u["mel"])
            np.save(os.path.join(d, "frame_f0", name + ".npy"), u["f0"])
            np.save(os.path.join(d, "frame_uv", name + ".npy"), u["uv"])
        np.savetxt(os.path.join(d, "f0", "f0_mean.txt"), np.array([f0_mean]))
        np.savetxt(os.path.join(d, "f0", "f0_std.txt"), np.array([f0_std]))
        with open(os.path.join(d, "train.lst"), "w") as f:
            f.write("\n".join(sorted(utts)) + "\n")
        for tag, nsf in (("plain", None), ("nsf", {"nb_harmonics": 7, "sampling_rate": sr})):
            config = {"audio_config": {"sampling_rate": sr, "n_fft": 2048, "hop_length": hop}, "batch_max_steps": 3200,
                      "allow_cache": False, "Model": {"Generator": {"params": {"nsf_params": nsf}}}}
            ds = D.Voc_Dataset([os.path.join(d, "train.lst")], [d], config)
            items = [ds[i] for i in range(len(ds))]
            np.random.seed(9)
            wav_b, mel_b = ds.collate_fn(items)
            out[tag] = dict(items=[(np.asarray(w), np.asarray(m)) for w, m in items], batch=(wav_b, mel_b))
    torch.save(dict(utts=utts, f0_mean=f0_mean, f0_std=f0_std, hop=hop, sr=sr, n_fft=2048, batch_max_steps=3200,
                    collate_seed=9, expected=out), os.path.join(OUT, "voc_dataset.pt"))
    print("voc_dataset bytes", os.path.getsize(os.path.join(OUT, "voc_dataset.pt")),
          [tuple(m.shape) for _, m in out["nsf"]["items"]])


def am_dataset_case():
    """The reference's KanTtsLinguisticUnit (utils/ling_unit/ling_unit.py:56-398) and AM_Dataset (datasets/dataset.py:
    391-827) on a small synthetic data directory: symbol lines over the PinYin inventory (incl. a symbol outside it, which
    the sy stream drops), three model variants (durations; NSF with globally re-normalised f0; MAS = no durations +
    alignment prior), items, one collated batch each, and the seeded train / valid split of gen_metafile.  The phone /
    tone inventories and every input file travel in the fixture so that the test can rebuild language and data
    directories anywhere."""
    import tempfile

    import numpy as np

    import kantts.datasets.dataset as D
    from kantts.utils.ling_unit.lang_symbols import get_language_symbols
    from kantts.utils.ling_unit.ling_unit import KanTtsLinguisticUnit

    phones, tones, _, _ = get_language_symbols("PinYin")
    rs = np.random.RandomState(5)
    flags = ["s_begin", "s_end", "s_none", "s_both", "s_middle"]
    segs = ["word_begin", "word_end", "word_middle", "word_both", "word_none"]
    emos = ["emotion_neutral", "emotion_happy", "emotion_none"]
    spks = ["F7", "M3"]

    def line(n, spk):
        groups = []
        for i in range(n):
            ph = phones[rs.randint(len(phones))] if rs.rand() > 0.08 else "not_a_phone"
            groups.append("{%s$%s$%s$%s$%s$%s}" % (ph, tones[rs.randint(len(tones))], flags[rs.randint(5)],
                                                  segs[rs.randint(5)], emos[rs.randint(3)], spk))
        return " ".join(groups)

    utts = {}
    for k in range(12):
        name = "u%02d" % k
        n_sym = int(rs.randint(4, 11))
        ling = line(n_sym, spks[k % 2])
        dur = rs.randint(1, 6, size=n_sym).astype(np.int64)          # one duration per symbol (the "~" slot is the collate's)
        frames = int(dur.sum())
        utts[name] = dict(ling=ling, mel=rs.randn(frames, 80).astype(np.float32), dur=dur,
                          f0=rs.randn(n_sym).astype(np.float32), energy=rs.randn(n_sym).astype(np.float32),
                          frame_f0=rs.randn(frames).astype(np.float32), frame_uv=(rs.rand(frames) > 0.4).astype(np.float32))
    f0_mean, f0_std = 201.5, 37.25
    base_params = {"outputs_per_step": 3}
    unit = {"cleaners": "english_cleaners", "speaker_list": "F7,M3",
            "lfeat_type_list": "sy,tone,syllable_flag,word_segment,emo_category,speaker_category"}
    variants = {"plain": {}, "nsf_global": {"NSF": True, "nsf_norm_type": "global", "nsf_f0_global_minimum": 30.0,
                                            "nsf_f0_global_maximum": 730.0}, "mas": {"MAS": True}}
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for sub in ("mel", "duration", "f0", "energy", "frame_f0", "frame_uv"):
            os.makedirs(os.path.join(d, sub))
        for name, u in utts.items():
            for sub, key in (("mel", "mel"), ("duration", "dur"), ("f0", "f0"), ("energy", "energy"),
                             ("frame_f0", "frame_f0"), ("frame_uv", "frame_uv")):
                np.save(os.path.join(d, sub, name + ".npy"), u[key])
        np.savetxt(os.path.join(d, "f0", "f0_mean.txt"), np.array([f0_mean]))
        np.savetxt(os.path.join(d, "f0", "f0_std.txt"), np.array([f0_std]))
        raw = os.path.join(d, "raw_metafile.txt")
        with open(raw, "w") as f:
            for name in sorted(utts):
                f.write("%s\t%s\n" % (name, utts[name]["ling"]))
        os.remove(os.path.join(d, "duration", "u07.npy"))                  # dropped by gen_metafile (no duration file)
        tr, va = os.path.join(d, "am_train.lst"), os.path.join(d, "am_valid.lst")
        D.AM_Dataset.gen_metafile(raw, d, tr, va, badlist=["u03"], split_ratio=0.8)
        split = dict(train=open(tr).read(), valid=open(va).read())
        for tag, extra in variants.items():
            config = {"linguistic_unit": dict(unit), "Model": {"KanTtsSAMBERT": {"params": dict(base_params, **extra)}}}
            ds = D.AM_Dataset(config, tr, d, allow_cache=False)
            items = [ds[i] for i in range(len(ds))]
            batch = ds.collate_fn(items[:5])
            out[tag] = dict(items=items, batch=batch, with_duration=ds.with_duration)
        lu = KanTtsLinguisticUnit({"linguistic_unit": dict(unit), "Model": {"KanTtsSAMBERT": {"params": dict(base_params)}}})
        sizes, pads = lu.get_unit_size(), dict(lu._sub_unit_pad)
    torch.save(dict(phones=phones[:-4], tones_file=[t[4:] if t != "tone_none" else "" for t in tones], utts=utts,
                    f0_mean=f0_mean, f0_std=f0_std, unit=unit, base_params=base_params, variants=variants, split=split,
                    missing_duration="u07", badlist=["u03"], split_ratio=0.8, unit_size=sizes, pad_ids=pads, expected=out),
               os.path.join(OUT, "am_dataset.pt"))
    print("am_dataset bytes", os.path.getsize(os.path.join(OUT, "am_dataset.pt")), sizes,
          [len(out[k]["items"]) for k in out], split["valid"].count("\n"))


def mas_dp_case():
    """b_mas (alignment.py:63-71; numba replaced by the identity jit of ref_harness, i.e. its plain-Python semantics)
    on random soft maps, on maps with exact ties (uniform rows) and with zeros (log -> -inf)."""
    import numpy as np
    from kantts.models.sambert.alignment import b_mas

    g = torch.Generator().manual_seed(99)
    B, To, Ti = 6, 57, 19
    logits = torch.randn(B, 1, To, Ti, generator=g) * 3
    attn = torch.softmax(logits, dim=3)
    attn[1] = 1.0 / Ti                                   # every comparison is a tie
    attn[2] = attn[2] * (torch.rand(1, To, Ti, generator=g) > 0.3)   # zeros -> -inf scores
    attn[3, :, :, 5:] = torch.round(attn[3, :, :, 5:] * 8) / 8       # coarse values: many ties, some zeros
    in_lens = np.array([19, 19, 17, 12, 1, 7], dtype=np.int64)
    out_lens = np.array([57, 40, 57, 33, 20, 7], dtype=np.int64)
    hard = b_mas(attn.numpy().copy(), in_lens, out_lens, width=1)
    fix = dict(attn=attn, in_lens=torch.from_numpy(in_lens), out_lens=torch.from_numpy(out_lens),
               hard=torch.from_numpy(hard))
    torch.save(fix, os.path.join(OUT, "mas_dp.pt"))
    print("mas_dp durations", hard.sum(2)[:, 0, :8].tolist()[0], "bytes", os.path.getsize(os.path.join(OUT, "mas_dp.pt")))


def hifigan_v1_case():
    """The reference's HiFi-GAN V1 at its own class defaults (Generator 512 channels, MPD 2/3/5/7/11, MSD x3 with DWT
    pooling -- BASELINE config 3's architecture): generator forward on 2 x 8 mel frames, one MPD and one MSD pass on
    2 x 2048 samples, gradient norms of a generator backward.  Weights are reproduced from the seed (checksums)."""
    from kantts.models.hifigan.hifigan import Generator, MultiPeriodDiscriminator, MultiScaleDiscriminator

    torch.manual_seed(0)
    G, D1, D2 = Generator(), MultiPeriodDiscriminator(), MultiScaleDiscriminator()
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 80, 8, generator=g)
    y = torch.randn(2, 1, 2048, generator=g).clamp(-1, 1)
    cot = torch.randn(2, 1, 2048, generator=g)
    wav = G(x)
    (wav * cot).sum().backward()
    fix = dict(x=x, y=y, cot=cot, wav=wav.detach().clone(),
               G_grad_norms={n: float(p.grad.double().norm()) for n, p in G.named_parameters()},
               G_checksums=checksums(G.state_dict()))
    for nm, D in (("mpd", D1), ("msd", D2)):
        o, fm = D(y)
        fix[nm + "_out"] = [a.detach().clone() for a in o]
        fix[nm + "_fmap_sums"] = [[(tuple(a.shape), float(a.double().sum()), float(a.double().abs().sum())) for a in fa]
                                  for fa in fm]
        fix[nm + "_checksums"] = checksums(D.state_dict())
    torch.save(fix, os.path.join(OUT, "hifigan_v1.pt"))
    print("hifigan_v1 bytes", os.path.getsize(os.path.join(OUT, "hifigan_v1.pt")), float(wav.abs().mean()))


def hifigan_v1_b32_case():
    """BASELINE config 3's size -- HiFi-GAN V1 at batch 32 x 8192 samples -- recorded from the reference: generator
    output, gradient norms of a generator backward, MPD / MSD outputs and feature-map sums on the real batch.  Inputs are
    regenerated from the seed by the test (they are 8 MB); the GPU box never needs a CPU oracle run at this size."""
    from kantts.models.hifigan.hifigan import Generator, MultiPeriodDiscriminator, MultiScaleDiscriminator

    torch.manual_seed(0)
    G, D1, D2 = Generator(), MultiPeriodDiscriminator(), MultiScaleDiscriminator()
    g = torch.Generator().manual_seed(77)
    x = torch.randn(32, 80, 32, generator=g)
    y = torch.randn(32, 1, 8192, generator=g).clamp(-1, 1)
    cot = torch.randn(32, 1, 8192, generator=g)
    wav = G(x)
    (wav * cot).sum().backward()
    fix = dict(seed=77, wav=wav.detach().clone(),
               G_grad_norms={n: float(p.grad.double().norm()) for n, p in G.named_parameters()},
               G_grad_samples={n: p.grad.flatten()[:64].clone() for n, p in G.named_parameters()
                               if n in ("conv_pre.conv1d.weight_v", "conv_post.conv1d.weight_v",
                                        "conv_blocks.0.convs1.0.conv1d.weight_v", "conv_blocks.11.convs2.2.conv1d.weight_v",
                                        "transpose_upsamples.1.1.deconv.weight_v")})
    with torch.no_grad():
        for nm, D in (("mpd", D1), ("msd", D2)):
            o, fm = D(y)
            fix[nm + "_out"] = [a.detach().clone() for a in o]
            fix[nm + "_fmap_sums"] = [[(tuple(a.shape), float(a.double().sum()), float(a.double().abs().sum()))
                                       for a in fa] for fa in fm]
    torch.save(fix, os.path.join(OUT, "hifigan_v1_b32.pt"))
    print("hifigan_v1_b32 bytes", os.path.getsize(os.path.join(OUT, "hifigan_v1_b32.pt")), float(wav.abs().mean()))


def multiband_case():
    """SURVEY row f4 pieces recorded from the reference: PQMF analysis / synthesis (pqmf.py:50-148), the multi-resolution
    STFT loss with its input gradient (loss.py:312-441; default resolutions, and a small one on sub-band shaped input),
    SpecDiscriminator / MultiSpecDiscriminator outputs and feature maps (hifigan.py:481-617) with seeded weights."""
    from kantts.models.pqmf import PQMF
    from kantts.models.hifigan.hifigan import MultiSpecDiscriminator
    from kantts.train.loss import MultiResolutionSTFTLoss

    g = torch.Generator().manual_seed(31)
    fix = {}
    pq = PQMF()
    x = torch.randn(2, 1, 2048, generator=g) * 0.3
    z = pq.analysis(x)
    fix["pqmf"] = dict(x=x, analysis=z.clone(), synthesis=pq.synthesis(z).clone(),
                       filters={k: v.clone() for k, v in pq.state_dict().items()})
    crit = MultiResolutionSTFTLoss()
    yh = (torch.randn(2, 1, 4096, generator=g) * 0.2).requires_grad_(True)
    y = torch.randn(2, 1, 4096, generator=g) * 0.2
    sc, mag = crit(yh, y)
    (sc + mag).backward()
    fix["mrstft"] = dict(y_hat=yh.detach().clone(), y=y, sc=float(sc), mag=float(mag), grad=yh.grad.clone())
    crit2 = MultiResolutionSTFTLoss(fft_sizes=[384, 683, 171], hop_sizes=[30, 60, 10], win_lengths=[150, 300, 60])
    sh = (torch.randn(2, 4, 1024, generator=g) * 0.2).requires_grad_(True)
    sy = torch.randn(2, 4, 1024, generator=g) * 0.2
    try:
        sc2, mag2 = crit2(sh, sy)
        (sc2 + mag2).backward()
        fix["mrstft_subband"] = dict(y_hat=sh.detach().clone(), y=sy, sc=float(sc2), mag=float(mag2), grad=sh.grad.clone())
    except Exception as exc:  # non-power-of-two FFT sizes are legal for torch.stft; keep the record either way
        fix["mrstft_subband"] = dict(error=repr(exc))
    params = {"channels": 16, "init_kernel": 15, "kernel_size": 11, "stride": 2, "use_spectral_norm": False,
              "window": "hann_window", "nonlinear_activation": "LeakyReLU",
              "nonlinear_activation_params": {"negative_slope": 0.1}}
    torch.manual_seed(4)
    D = MultiSpecDiscriminator(fft_sizes=[256, 512], hop_sizes=[60, 120], win_lengths=[240, 400], discriminator_params=params)
    yw = (torch.randn(2, 1, 2400, generator=g) * 0.3).requires_grad_(True)
    outs, fmaps = D(yw)
    loss = sum((o * o).mean() for o in outs) + sum(a.abs().mean() for fm in fmaps for a in fm)
    loss.backward()
    fix["multispec"] = dict(params=params, y=yw.detach().clone(), outs=[o.detach().clone() for o in outs],
                            fmap_sums=[[(tuple(a.shape), float(a.double().sum()), float(a.double().abs().sum())) for a in fm]
                                       for fm in fmaps],
                            input_grad_is_none=yw.grad is None, loss=float(loss),
                            grad_norms={n: float(p.grad.double().norm()) for n, p in D.named_parameters() if p.grad is not None},
                            checksums=checksums(D.state_dict()))
    try:
        MultiSpecDiscriminator()
        fix["multispec_default_ctor"] = "ok"
    except TypeError as exc:
        fix["multispec_default_ctor"] = "TypeError"
    torch.save(fix, os.path.join(OUT, "multiband.pt"))
    print("multiband bytes", os.path.getsize(os.path.join(OUT, "multiband.pt")), fix["mrstft"]["sc"], fix["mrstft"]["mag"],
          fix["multispec_default_ctor"], fix["mrstft_subband"].get("error"))


def loss_variants_case():
    """The criteria variants no shipped yaml selects (loss.py:7-85 with loss_type="mse"; :108-216 with loss_type="hinge"):
    values and input gradients of the reference classes on seeded inputs (padded batch; discriminator outputs as plain
    tensors, as lists, and as lists of feature-map lists)."""
    from kantts.train.loss import DiscriminatorAdversarialLoss, GeneratorAdversarialLoss

    g = torch.Generator().manual_seed(11)
    B, T, C, N = 3, 17, 8, 9
    out_len, in_len = torch.tensor([17, 9, 13]), torch.tensor([9, 4, 7])
    fix = dict(output_lengths=out_len, input_lengths=in_len,
               mel_targets=torch.randn(B, T, C, generator=g), dec=torch.randn(B, T, C, generator=g),
               post=torch.randn(B, T, C, generator=g), dur=torch.randint(0, 9, (B, N), generator=g),
               pitch=torch.randn(B, N, generator=g), energy=torch.randn(B, N, generator=g),
               logdur_p=torch.randn(B, N, generator=g), pitch_p=torch.randn(B, N, generator=g),
               energy_p=torch.randn(B, N, generator=g))
    dec, post = fix["dec"].clone().requires_grad_(True), fix["post"].clone().requires_grad_(True)
    a, b = MelReconLoss("mse")(out_len, fix["mel_targets"], dec, post)
    (a + 2 * b).backward()
    fix["mel_mse"] = (a.detach(), b.detach(), dec.grad.clone(), post.grad.clone())
    fix["mel_mse_no_postnet"] = MelReconLoss("mse")(out_len, fix["mel_targets"], fix["dec"])[0]
    lp, pp, ep = (fix[k].clone().requires_grad_(True) for k in ("logdur_p", "pitch_p", "energy_p"))
    d, p_, e = ProsodyReconLoss("mse")(in_len, fix["dur"], fix["pitch"], fix["energy"], lp, pp, ep)
    (d + 2 * p_ + 3 * e).backward()
    fix["prosody_mse"] = (d.detach(), p_.detach(), e.detach(), lp.grad.clone(), pp.grad.clone(), ep.grad.clone())
    # discriminator outputs: 3 discriminators, each a list of 2 feature maps + the score
    shapes = [[(2, 4, 30), (2, 8, 10), (2, 1, 10)], [(2, 4, 6, 5), (2, 8, 3, 5), (2, 1, 3, 5)], [(2, 6, 12), (2, 6, 6), (2, 1, 6)]]
    fake = [[torch.randn(*sh, generator=g) for sh in d_] for d_ in shapes]
    real = [[torch.randn(*sh, generator=g) for sh in d_] for d_ in shapes]
    fix["d_fake"], fix["d_real"] = fake, real
    hinge = {}
    for avg in (True, False):
        scores = [f[-1].clone().requires_grad_(True) for f in fake]
        v = GeneratorAdversarialLoss(average_by_discriminators=avg, loss_type="hinge")(scores)
        v.backward()
        hinge[("g_list", avg)] = (v.detach(), [s_.grad.clone() for s_ in scores])
        fk = [[t.clone().requires_grad_(True) for t in f] for f in fake]
        rl = [[t.clone().requires_grad_(True) for t in f] for f in real]
        r_, f_ = DiscriminatorAdversarialLoss(average_by_discriminators=avg, loss_type="hinge")(fk, rl)
        (r_ + 2 * f_).backward()
        hinge[("d_nested", avg)] = (r_.detach(), f_.detach(), [f[-1].grad.clone() for f in fk], [f[-1].grad.clone() for f in rl])
    one = fake[0][-1].clone().requires_grad_(True)
    v = GeneratorAdversarialLoss(loss_type="hinge")(one)
    v.backward()
    hinge["g_tensor"] = (v.detach(), one.grad.clone())
    r_, f_ = DiscriminatorAdversarialLoss(loss_type="hinge")(fake[1][-1], real[1][-1])
    hinge["d_tensor"] = (r_, f_)
    fix["hinge"] = hinge
    torch.save(fix, os.path.join(OUT, "loss_variants.pt"))
    print("loss_variants.pt", float(a), float(d), float(hinge[("g_list", True)][0]))


def msd_avgpool_case():
    """MultiScaleDiscriminator with the pooling no shipped yaml selects (hifigan.py:456-471: AvgPool1d(4, 2, padding=2),
    no auxiliary convolutions): outputs, feature-map sums and an input gradient of the reference at a small width."""
    from kantts.models.hifigan.hifigan import MultiScaleDiscriminator

    dp = {"in_channels": 1, "out_channels": 1, "kernel_sizes": [15, 41, 5, 3], "channels": 16,
          "max_downsample_channels": 64, "max_groups": 4, "bias": True, "downsample_scales": [2, 2, 4, 4, 1],
          "nonlinear_activation": "LeakyReLU", "nonlinear_activation_params": {"negative_slope": 0.1}}
    torch.manual_seed(21)
    msd = MultiScaleDiscriminator(scales=3, downsample_pooling="AvgPool1d", discriminator_params=dp)
    x = torch.randn(2, 1, 1537, generator=torch.Generator().manual_seed(22)).requires_grad_(True)
    outs, fmaps = msd(x)
    sum(o.pow(2).mean() for o in outs).backward()
    fix = dict(discriminator_params=dp, x=x.detach().clone(), outs=[o.detach().clone() for o in outs],
               fmap_sums=[[(tuple(f.shape), float(f.double().sum()), float(f.double().abs().sum())) for f in fm] for fm in fmaps],
               dx=x.grad.clone(), keys=sorted(msd.state_dict().keys()), checksums=checksums(msd.state_dict()))
    torch.save(fix, os.path.join(OUT, "msd_avgpool.pt"))
    print("msd_avgpool.pt", [tuple(o.shape) for o in outs])


def relu_generator_case():
    """Generator with nonlinear_activation="ReLU" (hifigan.py:69-71, layers.py:209-211; no shipped yaml): forward +
    gradient norms of the reference at a small width, causal and non-causal."""
    from kantts.models.hifigan.hifigan import Generator

    res = {}
    for causal in (True, False):
        torch.manual_seed(5)
        G = Generator(in_channels=80, channels=32, upsample_scales=[4, 4, 2, 2], upsample_kernal_sizes=[8, 8, 4, 4],
                      causal=causal, nonlinear_activation="ReLU", nonlinear_activation_params={})
        g = torch.Generator().manual_seed(12)
        x = torch.randn(2, 80, 7, generator=g)
        y = G(x)
        cot = torch.randn(y.shape, generator=g)
        (y * cot).sum().backward()
        res["causal" if causal else "noncausal"] = dict(
            x=x, y=y.detach().clone(), cot=cot, weight_checksums=checksums(G.state_dict()),
            grad_norms={n: float(p.grad.double().norm()) for n, p in G.named_parameters() if p.grad is not None})
    torch.save(res, os.path.join(OUT, "hifigan_relu.pt"))
    print("hifigan_relu.pt", float(res["causal"]["y"].abs().mean()))


def masks_case():
    """get_mask_from_lengths (kantts/models/utils.py:13-23) and get_lfr_mask_from_lengths' ceil(len / r) rule on
    seeded lengths, with and without an explicit max_len."""
    from kantts.models.utils import get_mask_from_lengths

    g = torch.Generator().manual_seed(5)
    cases = []
    for B, hi in ((1, 5), (7, 40), (32, 613)):
        lens = torch.randint(1, hi, (B,), generator=g)
        cases.append(dict(lengths=lens, mask=get_mask_from_lengths(lens).clone(),
                          mask_maxlen=get_mask_from_lengths(lens, max_len=hi + 3).clone(), max_len=hi + 3))
    torch.save(cases, os.path.join(OUT, "masks.pt"))


def melspec_case():
    g = torch.Generator().manual_seed(7)
    x = torch.randn(4, 2048, generator=g) * 0.1
    fix = dict(wav=x, mel_v1=MelSpectrogram()(x[:, None, :]).clone(),
               mel_16k=MelSpectrogram(fs=16000, fft_size=2048, hop_size=200, win_length=1000, fmin=0, fmax=8000)(
                   x[:, None, :]).clone(),
               stft_1024_120_600=stft(x, 1024, 120, 600, torch.hann_window(600)).clone())
    torch.save(fix, os.path.join(OUT, "melspec.pt"))
    print("melspec bytes", os.path.getsize(os.path.join(OUT, "melspec.pt")))


def dsp_melspec_case():
    """the reference's OWN numpy chain ``dsp.melspectrogram`` (preprocess/audio_processor/core/
    dsp.py:165-201: _stft -> abs -> _linear_to_mel -> _amp_to_db -> - ref_level_db -> _normalize -> .T) and its OWN
    ``AudioProcessor.mel_extract`` (audio_processor.py:317-387: per-utterance mel, corpus mean / std text files, normed
    .npy per utterance) executed here; of librosa only ``librosa.stft`` and ``librosa.filters.mel`` run below them,
    served by the scipy- / transformers-pinned restatements of oracle/thirdparty.py."""
    import tempfile

    import numpy as np
    from scipy.io import wavfile

    import kantts.preprocess.audio_processor.audio_processor as AP
    import kantts.preprocess.audio_processor.core.dsp as dsp

    rs = np.random.RandomState(31)
    t = np.arange(7000) / 16000.0
    y = (0.25 * np.sin(2 * np.pi * 180 * t) + 0.1 * np.sin(2 * np.pi * 1333 * t) + 0.03 * rs.randn(7000)).astype(np.float32)
    cfgs = {
        "audio_config_16k": dict(sample_rate=16000, n_fft=2048, hop_length=200, win_length=1000, n_mels=80, max_norm=1.0,
                                 min_level_db=-100, ref_level_db=20, fmin=0.0, fmax=8000.0, symmetric=False,
                                 preemphasize=False),
        "defaults_24k": dict(sample_rate=24000),  # the function's own defaults: n_fft 1024, hop 256, fmin 50, fmax 8000
        "symmetric_4": dict(sample_rate=22050, n_fft=1024, hop_length=256, win_length=1024, n_mels=80, max_norm=4.0,
                            min_level_db=-100, ref_level_db=20, fmin=80, fmax=7600, symmetric=True, preemphasize=False),
        "preemphasis": dict(sample_rate=16000, n_fft=1024, hop_length=160, win_length=800, n_mels=40, fmin=50, fmax=7600,
                            preemphasize=True),
    }
    fix = dict(wav=y, configs=cfgs, mel={})
    for name, kw in cfgs.items():
        dsp._mel_basis = None  # the reference caches the basis of its FIRST call in a module global (dsp.py:143-151)
        out = dsp.melspectrogram(y, **kw)
        fix["mel"][name] = np.asarray(out).copy()
        print("dsp_melspec", name, out.shape, out.dtype, float(out.mean()))
    dsp._mel_basis = None

    # AudioProcessor.mel_extract over a small 16 kHz corpus (one utterance under 0.5 s -> bad case)
    cfg = {"sampling_rate": 16000, "hop_length": 200, "win_length": 1000, "n_mels": 80, "n_fft": 2048, "fmin": 0.0,
           "fmax": 8000.0, "min_level_db": -100, "ref_level_db": 20, "max_norm": 1.0, "symmetric": False,
           "preemphasize": False, "num_workers": 2}
    pcm16 = {}
    for i, n in enumerate([9000, 12345, 8000, 16001, 4000]):
        tt = np.arange(n) / 16000.0
        x = 0.3 * np.sin(2 * np.pi * (110 + 40 * i) * tt) + 0.05 * rs.randn(n)
        pcm16["utt%02d" % i] = np.clip(np.round(x * 32768), -32768, 32767).astype(np.int16)
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "wav"))
        for k, q in pcm16.items():
            wavfile.write(os.path.join(d, "wav", k + ".wav"), 16000, q)
        ap = AP.AudioProcessor(dict(cfg))
        assert ap.mel_extract(os.path.join(d, "wav"), os.path.join(d, "mel"))
        fix["extract"] = dict(config=cfg, pcm16=pcm16, badcases=list(ap.badcase_list),
                              mel_dict={k: np.asarray(v).copy() for k, v in ap.mel_dict.items()},
                              mel_mean_txt=open(os.path.join(d, "mel", "mel_mean.txt")).read(),
                              mel_std_txt=open(os.path.join(d, "mel", "mel_std.txt")).read(),
                              normed={k: np.load(os.path.join(d, "mel", k + ".npy")) for k in ap.mel_dict})
    dsp._mel_basis = None
    torch.save(fix, os.path.join(OUT, "dsp_melspec.pt"))
    print("dsp_melspec bytes", os.path.getsize(os.path.join(OUT, "dsp_melspec.pt")), fix["extract"]["badcases"],
          {k: (v.shape, v.dtype) for k, v in fix["extract"]["mel_dict"].items()})


if __name__ == "__main__":
    if len(sys.argv) > 1:  # python oracle/make_golden.py hifigan_v1_b32_case ...: only the named cases (no arguments)
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    gk = ("text_encoder.ling_proj.weight", "mel_decoder.mel_dec.dec_out_proj.weight", "mel_postnet.fc.weight",
          "variance_adaptor.duration_predictor.fc.weight", "emo_tokenizer.weight")
    sambert_case("sambert_tiny", True, B=3, T_in=12, min_len=6, dur_hi=6, grad_keys=gk)
    sambert_case("sambert_tiny16", True, B=16, T_in=32, min_len=12, dur_hi=9, grad_keys=gk)
    # the reference's free-running path only works at batch 1 (its band masks are built without a batch axis)
    sambert_infer_case("sambert_tiny_infer", B=1, T_in=12, min_len=6)
    melspec_case()
    collate_case()
    mas_dp_case()
    sambert_mas_case("sambert_tiny_mas", B=3, T_in=12, min_len=6, dur_hi=6)
    sambert_se_case("sambert_tiny_se", B=2, T_in=10, min_len=5, dur_hi=5)
    nsf_generator_case()
    sambert_curve_case()
    gan_curve_case()
    voc_dataset_case()
    am_dataset_case()
    hifigan_v1_case()
    hifigan_v1_b32_case()
    masks_case()
    multiband_case()
    loss_variants_case()
    msd_avgpool_case()
    relu_generator_case()
    dsp_melspec_case()
