"""Batch assembly either side of the hot path: the exact input layout the training steps consume.

``am_collate`` reproduces AM_Dataset.collate_fn (kantts/datasets/dataset.py:690-827) for the linguistic-symbol path
(sy | tone | syllable_flag | word_segment | emotion | speaker streams; durations given): per-stream pad ids, the
"~" end token kept in the padded inputs but excluded from ``valid_input_lengths``, mel targets padded to a multiple of
the reduction factor r, and ``Padder._pad_durations`` (:47-64) which parks the r-padding frames on the token right after
the last symbol.  ``voc_collate`` reproduces Voc_Dataset.collate_fn (:278-311): a random crop of ``batch_max_steps``
samples with the matching mel frames.  Both return pinned host tensors when a GPU is present (``pin=True``) so that the
trainers' ``.to(device, non_blocking=True)`` overlaps with compute.  Dataset discovery / feature files / the text
front-end stay with the reference package.
"""
import functools

import numpy as np
import torch


class Padder(object):
    def _pad1D(self, x, length, pad):
        return np.pad(x, (0, length - x.shape[0]), mode="constant", constant_values=pad)

    def _pad2D(self, x, length, pad):
        return np.pad(x, [(0, length - x.shape[0]), (0, 0)], mode="constant", constant_values=pad)

    def _pad_durations(self, duration, max_in_len, max_out_len):
        framenum, symbolnum = int(np.sum(duration)), duration.shape[0]
        out = np.zeros(max_in_len, dtype=duration.dtype)
        out[:symbolnum] = duration
        if framenum < max_out_len:
            out[symbolnum] = max_out_len - framenum  # the r-padding frames belong to the slot after the last symbol
        return out

    def _round_up(self, x, multiple):
        return x if x % multiple == 0 else x + multiple - x % multiple

    def _prepare_scalar_inputs(self, inputs, max_len, pad):
        return torch.from_numpy(np.stack([self._pad1D(x, max_len, pad) for x in inputs]))

    def _prepare_targets(self, targets, max_len, pad):
        return torch.from_numpy(np.stack([self._pad2D(t, max_len, pad) for t in targets])).float()

    def _prepare_durations(self, durations, max_in_len, max_out_len):
        return torch.from_numpy(np.stack([self._pad_durations(t, max_in_len, max_out_len) for t in durations])).long()


@functools.lru_cache(maxsize=256)
def beta_binomial_prior_distribution(phoneme_count, mel_count, scaling=1.0):
    """(mel_count, phoneme_count) float64 tensor: row i is the Beta-Binomial(P, s*i, s*(M+1-i)) pmf over phoneme
    positions 0..P-1 -- a diagonal-ish prior for the alignment attention (reference dataset.py:20-31, which
    evaluates scipy.stats.betabinom row by row).  pmf(k) = C(P,k) B(k+a, P-k+b) / B(a,b), evaluated in log space for
    all rows at once."""
    from scipy.special import betaln, gammaln

    P, M = int(phoneme_count), int(mel_count)
    k = np.arange(0, P, dtype=np.float64)[None, :]
    i = np.arange(1, M + 1, dtype=np.float64)[:, None]
    a, b = scaling * i, scaling * (M + 1 - i)
    log_comb = gammaln(P + 1) - gammaln(k + 1) - gammaln(P - k + 1)
    return torch.tensor(np.exp(log_comb + betaln(k + a, P - k + b) - betaln(a, b)))


def _pin(d, pin):
    if not (pin and torch.cuda.is_available()):
        return d
    return {k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in d.items()}


def am_collate(batch, r, pad_ids, pin=False, se=False):
    """batch: list of (ling_data, mel, dur, f0, energy, attn_prior, fp_label, se) as AM_Dataset.__getitem__ returns
    (ling_data = [sy, tone, syllable_flag, word_segment, emotion, speaker] integer arrays including the trailing "~");
    pad_ids: the six per-stream pad ids (ling_unit._sub_unit_pad in stream order)."""
    padder = Padder()
    with_duration = not any(item[2] is None for item in batch)
    max_in = max(len(x[0][0]) for x in batch)
    max_dur = max(x[2].shape[0] for x in batch) + 1 if with_duration else None
    streams = [padder._prepare_scalar_inputs([x[0][k] for x in batch], max_in, pad_ids[k]).long() for k in range(6)]
    out = {"input_lings": torch.stack(streams[:4], dim=2), "input_emotions": streams[4], "input_speakers": streams[5]}
    if se:  # speaker-embedding models: one (1, D) vector per utterance, repeated over its symbols (dataset.py:757-762)
        out["input_speakers"] = padder._prepare_targets([x[7].repeat(len(x[0][0]), axis=0) for x in batch], max_in, 0.0)
    out["valid_input_lengths"] = torch.as_tensor([len(x[0][0]) - 1 for x in batch], dtype=torch.long)  # minus "~"
    out["valid_output_lengths"] = torch.as_tensor([len(x[1]) for x in batch], dtype=torch.long)
    max_out = padder._round_up(int(out["valid_output_lengths"].max()), r)
    out["mel_targets"] = padder._prepare_targets([x[1] for x in batch], max_out, 0.0)
    out["durations"] = padder._prepare_durations([x[2] for x in batch], max_dur, max_out) if with_duration else None
    # duration-free (MAS) batches carry FRAME-level pitch / energy (averaged per phoneme inside the model) and the
    # beta-binomial alignment prior, zero-padded to (max mel, max text) (reference dataset.py:798-827)
    feat_len = max_in if with_duration else max_out
    out["pitch_contours"] = padder._prepare_scalar_inputs([x[3] for x in batch], feat_len, 0.0).float()
    out["energy_contours"] = padder._prepare_scalar_inputs([x[4] for x in batch], feat_len, 0.0).float()
    out["attn_priors"] = None
    if not with_duration:
        pri = torch.zeros(len(batch), max_out, max_in)
        for i, item in enumerate(batch):
            p = torch.as_tensor(item[5])
            pri[i, :p.shape[0], :p.shape[1]] = p
        out["attn_priors"] = pri
    return _pin(out, pin)


def voc_collate(batch, hop_length, batch_max_steps, rng=np.random, pin=False):
    """batch: list of (wav (T,), mel (frames, C)) with len(wav) == frames * hop_length -> (wav (B,1,S), mel (B,C,S/hop))."""
    frames = batch_max_steps // hop_length
    starts = np.array([rng.randint(0, len(mel) - frames) for _, mel in batch])
    wav = np.asarray([w[s * hop_length:s * hop_length + batch_max_steps] for (w, _), s in zip(batch, starts)])
    mel = np.asarray([m[s:s + frames] for (_, m), s in zip(batch, starts)])
    wav_t = torch.tensor(wav, dtype=torch.float32).unsqueeze(1)
    mel_t = torch.tensor(mel, dtype=torch.float32).transpose(2, 1)
    if pin and torch.cuda.is_available():
        wav_t, mel_t = wav_t.pin_memory(), mel_t.pin_memory()
    return wav_t, mel_t
