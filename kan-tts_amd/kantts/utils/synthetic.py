"""Synthetic workloads of the benchmark / CLI smoke runs (no dataset ships with the hot path).

``sambert_16k_config`` spells out configs/sambert_16k.yaml:6-52 with the vocabulary sizes of the PinYin front-end
(SURVEY.md section 8); ``sambert_batch`` draws a seeded batch of that shape: lengths, four linguistic id streams,
emotion ids, integer durations (with the r-padding frames parked on token ``len`` as Padder._pad_durations does,
kantts/datasets/dataset.py:47-64), mel / pitch / energy targets.  tests/test_oracle_golden.py checks that this
generator and the oracle's own produce identical tensors."""
import math

import torch

SAMBERT_VOCAB = dict(sy=147, tone=10, syllable_flag=8, word_segment=8, emotion=36, speaker=4)


def sambert_16k_config(tiny=False):
    cfg = dict(
        max_len=800, embedding_dim=512, encoder_num_layers=8, encoder_num_heads=8, encoder_num_units=128,
        encoder_ffn_inner_dim=1024, encoder_dropout=0.1, encoder_attention_dropout=0.1, encoder_relu_dropout=0.1,
        encoder_projection_units=32, speaker_units=32, emotion_units=32, predictor_filter_size=41,
        predictor_fsmn_num_layers=3, predictor_num_memory_units=128, predictor_ffn_inner_dim=256,
        predictor_dropout=0.1, predictor_shift=0, predictor_lstm_units=128, dur_pred_prenet_units=[128, 128],
        dur_pred_lstm_units=128, decoder_prenet_units=[256, 256], decoder_num_layers=12, decoder_num_heads=8,
        decoder_num_units=128, decoder_ffn_inner_dim=1024, decoder_dropout=0.1, decoder_attention_dropout=0.1,
        decoder_relu_dropout=0.1, outputs_per_step=3, num_mels=80, postnet_filter_size=41, postnet_fsmn_num_layers=4,
        postnet_num_memory_units=256, postnet_ffn_inner_dim=512, postnet_dropout=0.1, postnet_shift=17,
        postnet_lstm_units=128, MAS=False)
    cfg.update(SAMBERT_VOCAB)
    if tiny:
        cfg["encoder_num_layers"] = cfg["decoder_num_layers"] = 2
    return cfg


def sambert_batch(B=32, T_in=64, seed=1234, min_len=32, dur_hi=17, r=3, num_mels=80):
    g = torch.Generator().manual_seed(seed)
    vocab = (147, 10, 8, 8)
    lens = torch.randint(min_len, T_in, (B,), generator=g)
    lens[0] = T_in - 1
    ling = torch.stack([torch.randint(0, vocab[k] - 3, (B, T_in), generator=g) for k in range(4)], -1)
    emo = torch.randint(0, 33, (B, T_in), generator=g)
    spk = torch.zeros(B, T_in, dtype=torch.long)
    dur = torch.randint(2, dur_hi, (B, T_in), generator=g)
    dur = dur * (torch.arange(T_in)[None, :] < lens[:, None])
    out_lens = dur.sum(1)
    T_mel = int(math.ceil(int(out_lens.max()) / r) * r)
    for b in range(B):
        dur[b, lens[b]] = T_mel - out_lens[b]
    mel = torch.randn(B, T_mel, num_mels, generator=g)
    mel = mel * (torch.arange(T_mel)[None, :, None] < out_lens[:, None, None])
    pitch = torch.randn(B, T_in, generator=g)
    energy = torch.randn(B, T_in, generator=g)
    return dict(inputs_ling=ling, inputs_emotion=emo, inputs_speaker=spk, input_lengths=lens, output_lengths=out_lens,
                mel_targets=mel, duration_targets=dur, pitch_targets=pitch, energy_targets=energy)


def sambert_mas_batch(B=32, T_in=64, seed=1234, min_len=32, dur_hi=17, r=3, num_mels=80):
    """Duration-free batch of the MAS configuration (configs/sambert_16k_MAS.yaml): same draws as ``sambert_batch``, then
    frame-level pitch (30 % unvoiced zeros) / energy and the beta-binomial alignment prior over len+1 symbols (the
    trailing "~"), zero-padded, as AM_Dataset / collate_fn deliver them (dataset.py:498-503, 798-827)."""
    from kantts.datasets.batching import beta_binomial_prior_distribution

    b = sambert_batch(B=B, T_in=T_in, seed=seed, min_len=min_len, dur_hi=dur_hi, r=r, num_mels=num_mels)
    g = torch.Generator().manual_seed(seed + 1)
    T_mel = b["mel_targets"].shape[1]
    valid = torch.arange(T_mel)[None, :] < b["output_lengths"][:, None]
    pitch = torch.randn(B, T_mel, generator=g) * (torch.rand(B, T_mel, generator=g) > 0.3) * valid
    energy = torch.randn(B, T_mel, generator=g) * valid
    pri = torch.zeros(B, T_mel, T_in)
    for i in range(B):
        p = beta_binomial_prior_distribution(int(b["input_lengths"][i]) + 1, int(b["output_lengths"][i]))
        pri[i, :p.shape[0], :p.shape[1]] = p
    b.update(duration_targets=None, pitch_targets=pitch, energy_targets=energy, attn_priors=pri)
    return b


def inference_utterances(n_utt=128, seed=4321):
    """BASELINE config 5: ``n_utt`` synthetic utterances, T_in uniform 20..80, the id distributions of
    ``sambert_batch``.  Returns (lens, ling (n, T, 4), emo, spk) padded to the longest utterance; draw order on the generator:
    lens, the four linguistic streams, emotion ids."""
    g = torch.Generator().manual_seed(seed)
    vocab = (147, 10, 8, 8)
    lens = torch.randint(20, 81, (n_utt,), generator=g)
    T = int(lens.max())
    ling = torch.stack([torch.randint(0, vocab[k] - 3, (n_utt, T), generator=g) for k in range(4)], -1)
    emo = torch.randint(0, 33, (n_utt, T), generator=g)
    spk = torch.zeros(n_utt, T, dtype=torch.long)
    return lens, ling, emo, spk


def to_collate_format(b):
    """Model-argument names -> the keys of the reference's collate_fn (what the trainers consume)."""
    return {"input_lings": b["inputs_ling"], "input_emotions": b["inputs_emotion"], "input_speakers": b["inputs_speaker"],
            "valid_input_lengths": b["input_lengths"], "valid_output_lengths": b["output_lengths"],
            "mel_targets": b["mel_targets"], "durations": b["duration_targets"], "pitch_contours": b["pitch_targets"],
            "energy_contours": b["energy_targets"], "attn_priors": b.get("attn_priors")}
