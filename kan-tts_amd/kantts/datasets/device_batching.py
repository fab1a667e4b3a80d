"""Batch assembly on the device -- the counterpart of batching.py for a corpus that is RESIDENT in HBM.

The reference assembles every batch on the host: ``np.pad`` per utterance, ``np.stack``, ``torch.tensor``, a (pageable)
H2D copy (kantts/datasets/dataset.py:34-85 Padder, :278-311 Voc_Dataset.collate_fn, :690-827 AM_Dataset.collate_fn).  On
eight GPUs that feeder is what a DataLoader worker pool has to keep up with.  An MI355X has 288 GB of HBM: a whole TTS
corpus (10 h of 22.05 kHz audio and its 80-bin mels is ~4 GB) fits next to the model, so

  * ``DeviceVocSet`` / ``DeviceAMSet`` upload the corpus ONCE as flat (rows, C) buffers + row offsets;
  * a batch is a list of utterance indices: the host draws the crop starts with the SAME generator calls as the host
    collate (so a seeded run sees the same batches), uploads B offsets / starts / lengths, and csrc/batching.hip gathers,
    pads, crops and transposes on the device (``kantts._hip.ragged_rows``);
  * the results are bit-identical to ``voc_collate`` / ``am_collate`` (tests/test_device_batching.py).

``PinnedPrefetcher`` is the other half of the row for corpora that stay on the host: it stages the NEXT host batch in
pinned memory and copies it on a side stream while the current step runs (double-buffered; what
``GraphedSambertStep.load_batch`` / ``GraphedGanStep.load_batch`` consume).
"""
import numpy as np
import torch

from kantts import _hip as hip
from kantts.datasets.batching import Padder


def _flat(arrays, dtype, device):
    """list of (rows_i, C) arrays -> ((sum rows, C) device tensor, row offsets (n + 1,) int64 numpy)."""
    arrays = [np.asarray(a, dtype=dtype) for a in arrays]
    arrays = [a.reshape(len(a), -1) for a in arrays]
    off = np.zeros(len(arrays) + 1, dtype=np.int64)
    np.cumsum([len(a) for a in arrays], out=off[1:])
    flat = torch.from_numpy(np.concatenate(arrays, axis=0)) if arrays else torch.zeros((0, 1))
    return flat.to(device), off


def _dev(a, dtype, device):
    return torch.as_tensor(np.asarray(a), dtype=dtype).to(device, non_blocking=True)


class DeviceVocSet:
    """Vocoder training set in HBM: ``items`` = [(wav (T,), mel (frames, C))] with len(wav) == frames * hop_length (what
    Voc_Dataset.__getitem__ returns).  ``batch(indices, rng)`` == ``voc_collate([items[i] for i in indices], ...)`` moved
    to the device: (wav (B, 1, S), mel (B, C, S / hop))."""

    def __init__(self, items, hop_length, batch_max_steps, device):
        self.hop, self.steps, self.frames = int(hop_length), int(batch_max_steps), int(batch_max_steps) // int(hop_length)
        self.device = torch.device(device)
        self.n_frames = np.array([len(m) for _, m in items], dtype=np.int64)
        for w, m in items:
            if len(w) != len(m) * self.hop:
                raise ValueError("wav / mel lengths disagree (len(wav) must be frames * hop_length)")
            if len(m) <= self.frames:
                # Voc_Dataset.__getitem__ pads every item to at least one frame more than a crop (dataset.py:124-130, as the
                # reference does); anything shorter would make the crop's randint(0, len - frames) raise mid-epoch
                raise ValueError("an item has %d frames, a crop needs more than %d: pad short utterances as "
                                 "Voc_Dataset.__getitem__ does" % (len(m), self.frames))
        self.wav, self.wav_off = _flat([np.asarray(w).reshape(-1, 1) for w, _ in items], np.float32, self.device)
        self.mel, self.mel_off = _flat([m for _, m in items], np.float32, self.device)

    def __len__(self):
        return len(self.n_frames)

    def batch(self, indices, rng=np.random):
        idx = np.asarray(indices, dtype=np.int64)
        # the same draws, in the same order, as voc_collate / the reference's collate_fn (:282-287)
        starts = np.array([rng.randint(0, int(self.n_frames[i]) - self.frames) for i in idx], dtype=np.int64)
        B = len(idx)
        wav = hip.ragged_rows(self.wav, _dev(self.wav_off[idx], torch.int64, self.device),
                              _dev(np.full(B, self.steps), torch.int32, self.device), self.steps,
                              start=_dev(starts * self.hop, torch.int32, self.device))
        mel = hip.ragged_rows(self.mel, _dev(self.mel_off[idx], torch.int64, self.device),
                              _dev(np.full(B, self.frames), torch.int32, self.device), self.frames,
                              start=_dev(starts, torch.int32, self.device), transpose=True)
        return wav.view(B, 1, self.steps), mel


class DeviceAMSet:
    """Acoustic-model training set in HBM (duration-supervised items of AM_Dataset.__getitem__: (ling_data, mel, dur, f0,
    energy, attn_prior, fp_label, se) with ling_data = six integer streams incl. the trailing "~").  ``batch(indices)``
    == ``am_collate([items[i] for i in indices], r, pad_ids)`` with device tensors."""

    KEYS = ("input_lings", "input_emotions", "input_speakers", "valid_input_lengths", "valid_output_lengths",
            "mel_targets", "durations", "pitch_contours", "energy_contours", "attn_priors")

    def __init__(self, items, r, pad_ids, device):
        if any(it[2] is None for it in items):
            raise NotImplementedError("duration-free (MAS) items carry a per-batch prior: use the host collate")
        self.r, self.device = int(r), torch.device(device)
        self.n_sym = np.array([len(it[0][0]) for it in items], dtype=np.int64)
        self.n_mel = np.array([len(it[1]) for it in items], dtype=np.int64)
        self.n_dur = np.array([it[2].shape[0] for it in items], dtype=np.int64)
        self.frames = np.array([int(np.sum(it[2])) for it in items], dtype=np.int64)
        self.ling, self.sym_off = _flat([np.stack([np.asarray(s) for s in it[0]], axis=1) for it in items], np.int64,
                                        self.device)
        self.mel, self.mel_off = _flat([it[1] for it in items], np.float32, self.device)
        self.dur, self.dur_off = _flat([np.asarray(it[2]).reshape(-1, 1) for it in items], np.int64, self.device)
        self.f0, self.f0_off = _flat([np.asarray(it[3]).reshape(-1, 1) for it in items], np.float32, self.device)
        self.energy, self.en_off = _flat([np.asarray(it[4]).reshape(-1, 1) for it in items], np.float32, self.device)
        self.n_f0 = np.array([len(it[3]) for it in items], dtype=np.int64)
        self.n_en = np.array([len(it[4]) for it in items], dtype=np.int64)
        self.pad_ids = torch.as_tensor(list(pad_ids), dtype=torch.int64).to(self.device)
        # longest duration among an utterance's real symbols (the trailing "~" excluded), on the host: a batch's attention
        # band width (kantts_sambert.py:981-985) without reading the device batch back
        self.max_dur = np.array([int(np.max(np.asarray(it[2])[: len(it[0][0]) - 1], initial=0)) for it in items], dtype=np.int64)

    def __len__(self):
        return len(self.n_sym)

    def band_width(self, indices):
        """x_band_width of the batch ``indices`` as the model computes it from the duration targets (host integer)."""
        return int(float(self.max_dur[np.asarray(indices, dtype=np.int64)].max()) / self.r + 0.5)

    def batch(self, indices):
        idx = np.asarray(indices, dtype=np.int64)
        d = self.device
        max_in = int(self.n_sym[idx].max())
        max_dur = int(self.n_dur[idx].max()) + 1
        max_out = Padder()._round_up(int(self.n_mel[idx].max()), self.r)
        i32 = lambda a: _dev(a, torch.int32, d)  # noqa: E731
        i64 = lambda a: _dev(a, torch.int64, d)  # noqa: E731
        ling = hip.ragged_rows(self.ling, i64(self.sym_off[idx]), i32(self.n_sym[idx]), max_in, pad=self.pad_ids)
        out = {"input_lings": ling[:, :, :4].contiguous(), "input_emotions": ling[:, :, 4].contiguous(),
               "input_speakers": ling[:, :, 5].contiguous()}
        out["valid_input_lengths"] = i64(self.n_sym[idx] - 1)  # minus "~"
        out["valid_output_lengths"] = i64(self.n_mel[idx])
        out["mel_targets"] = hip.ragged_rows(self.mel, i64(self.mel_off[idx]), i32(self.n_mel[idx]), max_out)
        dur = hip.ragged_rows(self.dur, i64(self.dur_off[idx]), i32(self.n_dur[idx]), max_dur).view(len(idx), max_dur)
        # Padder._pad_durations (:47-64): the frames that pad the mel to a multiple of r belong to the slot after the last
        # symbol -- B values, computed on the host from the cached frame counts
        extra = np.where(self.frames[idx] < max_out, max_out - self.frames[idx], 0)
        dur.scatter_add_(1, i64(self.n_dur[idx]).view(-1, 1), i64(extra).view(-1, 1))
        out["durations"] = dur
        out["pitch_contours"] = hip.ragged_rows(self.f0, i64(self.f0_off[idx]), i32(self.n_f0[idx]), max_in).view(len(idx), max_in)
        out["energy_contours"] = hip.ragged_rows(self.energy, i64(self.en_off[idx]), i32(self.n_en[idx]), max_in).view(len(idx), max_in)
        out["attn_priors"] = None
        out["band_width"] = self.band_width(idx)  # host integer (Sambert_Trainer: captured-step class of the batch)
        return out


class PinnedPrefetcher:
    """Iterate a loader of HOST batches (dicts / tuples / lists of tensors, as the collate functions return) and yield
    DEVICE batches: batch i + 1 is staged in pinned memory and copied on a side stream while the consumer runs step i
    (two staging sets, recycled).  The consumer's stream waits for the copy's event -- no host synchronisation."""

    def __init__(self, loader, device, r=None):
        """``r`` (outputs_per_step) for acoustic-model batches: the band width of a batch is then computed HERE, from the
        host copy, and travels in the dict as the host integer ``band_width`` (as DeviceAMSet.batch does) -- the trainer's
        captured step needs it per batch and would otherwise read it back from the device: a blocking synchronisation per
        step, the thing this prefetcher exists to remove."""
        self.loader, self.device, self.r = loader, torch.device(device), r
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self._staging = [{}, {}]
        self._copied = [None, None]  # event of the last copy out of each staging set

    def _stage(self, slot, path, t):
        if self.stream is None or not torch.is_tensor(t):
            return t.to(self.device) if torch.is_tensor(t) else t
        # one grow-only flat pinned buffer per (field, dtype) and slot: acoustic-model batches come in hundreds of distinct
        # (max_in, max_out) shapes, and a buffer per shape would page-lock host memory without bound
        key = (path, t.dtype)
        n = t.numel()
        buf = self._staging[slot].get(key)
        if buf is None or buf.numel() < n:
            buf = self._staging[slot][key] = torch.empty(max(n, 1) * 5 // 4 + 16, dtype=t.dtype).pin_memory()
        view = buf[:n].view(t.shape)
        view.copy_(t)
        return view.to(self.device, non_blocking=True)

    def _move(self, slot, batch, path=()):
        if isinstance(batch, dict):
            out = {k: self._move(slot, v, path + (k,)) for k, v in batch.items()}
            if (self.r and not path and "band_width" not in batch and torch.is_tensor(batch.get("durations"))
                    and torch.is_tensor(batch.get("valid_input_lengths")) and not batch["durations"].is_cuda):
                d, n = batch["durations"], batch["valid_input_lengths"]
                valid = torch.arange(d.size(1))[None, :] < n[:, None]
                out["band_width"] = int(float((d * valid).max()) / self.r + 0.5)  # == kantts_sambert.band_width_of
            return out
        if isinstance(batch, (tuple, list)):
            return type(batch)(self._move(slot, v, path + (i,)) for i, v in enumerate(batch))
        return self._stage(slot, path, batch)

    def _tensors(self, batch):
        if isinstance(batch, dict):
            batch = list(batch.values())
        if isinstance(batch, (tuple, list)):
            for v in batch:
                yield from self._tensors(v)
        elif torch.is_tensor(batch):
            yield batch

    def __iter__(self):
        it = iter(self.loader)
        pending, slot = None, 0

        def launch(host_batch, slot):
            if self.stream is None:
                return self._move(slot, host_batch), None
            if self._copied[slot] is not None:
                self._copied[slot].synchronize()  # the copy issued two batches ago has left this staging set
            with torch.cuda.stream(self.stream):
                dev = self._move(slot, host_batch)
                ev = torch.cuda.Event()
                ev.record(self.stream)
            self._copied[slot] = ev
            return dev, ev

        try:
            pending = launch(next(it), slot)
        except StopIteration:
            return
        while pending is not None:
            dev, ev = pending
            slot ^= 1
            try:
                pending = launch(next(it), slot)  # overlaps with the consumer's work on ``dev``
            except StopIteration:
                pending = None
            if ev is not None:
                torch.cuda.current_stream(self.device).wait_event(ev)
                for t in self._tensors(dev):
                    t.record_stream(torch.cuda.current_stream(self.device))
            yield dev


class DeviceCorpusLoader:
    """What the trainers iterate instead of a ``DataLoader`` when the corpus is resident in HBM: index batches (from a
    sampler -- e.g. the DistributedSampler of the data-parallel run --, a seeded shuffle, or an explicit list) turned into
    device batches by ``DeviceAMSet.batch`` / ``DeviceVocSet.batch``.  No worker processes, no host collate, no H2D copy of
    the payload: per batch the host uploads B offsets / starts / lengths (reference: kantts/bin/train_sambert.py:108-132,
    train_hifigan.py:96-121 build a DataLoader over the host collate)."""

    def __init__(self, device_set, batch_size, sampler=None, shuffle=True, drop_last=False, seed=0, batches=None, rng=None):
        self.set, self.batch_size, self.sampler = device_set, int(batch_size), sampler
        self.shuffle, self.drop_last, self.batches, self.rng = bool(shuffle), bool(drop_last), batches, rng
        self._gen = torch.Generator().manual_seed(int(seed))

    def _index_batches(self):
        if self.batches is not None:
            return [list(b) for b in self.batches]
        if self.sampler is not None:
            order = list(iter(self.sampler))
        elif self.shuffle:
            order = torch.randperm(len(self.set), generator=self._gen).tolist()
        else:
            order = list(range(len(self.set)))
        out = [order[i:i + self.batch_size] for i in range(0, len(order), self.batch_size)]
        if self.drop_last and out and len(out[-1]) < self.batch_size:
            out.pop()
        return out

    def __len__(self):
        if self.batches is not None:
            return len(self.batches)
        n = len(self.sampler) if self.sampler is not None else len(self.set)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        for idx in self._index_batches():
            yield self.set.batch(idx, self.rng) if self.rng is not None else self.set.batch(idx)


def corpus_bytes(items):
    """Payload bytes of a list of dataset items (nested tuples / lists of arrays)."""
    if items is None:
        return 0
    if isinstance(items, (tuple, list)):
        return sum(corpus_bytes(x) for x in items)
    return int(np.asarray(items).nbytes)


def make_train_loader(kind, dataset, host_loader, device, batch_size, sampler=None, mode="auto"):
    """``--device_corpus`` of the training entry points.  ``mode``: "hbm" = upload ``dataset`` once and assemble batches on
    the device (DeviceCorpusLoader); "pinned" = keep the host DataLoader and stage / copy its batches one step ahead
    (PinnedPrefetcher); "auto" = "hbm" when the payload takes less than half of the free device memory, else "pinned";
    "off" = ``host_loader`` as the reference builds it.  ``kind``: "am" (dataset items of AM_Dataset, duration-supervised)
    or "voc" (Voc_Dataset)."""
    device = torch.device(device)
    if mode == "off" or device.type != "cuda":
        return host_loader
    r = getattr(dataset, "r", None) if kind == "am" else None
    if kind == "am" and getattr(dataset, "mas_enable", False):
        return PinnedPrefetcher(host_loader, device)  # MAS items carry a per-batch prior: host collate
    if mode in ("auto", "hbm"):
        if mode == "auto":
            # decide from a SAMPLE before materialising the corpus: loading every item only to fall back to the prefetcher
            # paid the whole load time and held the corpus in host memory
            n = len(dataset)
            probe = [dataset[i] for i in sorted({int(j * (n - 1) / 15) for j in range(16)})] if n else []
            free, _ = torch.cuda.mem_get_info(device)
            if probe and corpus_bytes(probe) / len(probe) * n > free // 2:
                return PinnedPrefetcher(host_loader, device, r=r)
        items = [dataset[i] for i in range(len(dataset))]
        if kind == "am":
            pad_ids = [dataset.ling_unit._sub_unit_pad[t] for t in dataset.ling_unit._lfeat_type_list]
            dset = DeviceAMSet(items, dataset.r, pad_ids, device)
        else:
            dset = DeviceVocSet(items, dataset.hop_length, dataset.batch_max_steps, device)
        return DeviceCorpusLoader(dset, batch_size, sampler=sampler, shuffle=sampler is None,
                                  drop_last=bool(getattr(host_loader, "drop_last", False)))
    return PinnedPrefetcher(host_loader, device, r=r)
