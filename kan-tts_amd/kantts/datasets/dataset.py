"""Datasets over the on-disk feature layout.  Vocoder: the reader side of
``AudioProcessor.mel_extract``: ``<root>/wav/<utt>.wav`` + ``<root>/mel/<utt>.npy`` (+ ``frame_f0`` / ``frame_uv`` and
``f0/f0_{mean,std}.txt`` for NSF generators), utterance lists in ``train.lst`` / ``valid.lst``.

Same entry points as the reference (kantts/datasets/dataset.py:88-345): ``Voc_Dataset(metafile, root_dir, config)`` with
``__getitem__ -> (wav (T,), mel (frames, C))`` where ``T == frames * hop_length``, ``collate_fn`` (random crop of
``batch_max_steps`` samples, kantts.datasets.batching.voc_collate) and ``get_voc_datasets(config, root_dir)``.  Item
semantics follow the reference exactly (pinned by tests/golden/voc_dataset.pt): utterances not longer than a crop are
zero-padded to one frame more than a crop; longer ones get n_fft samples of reflect padding and are cut to
frames * hop_length; NSF items carry the de-normalised frame f0 and the voiced flag as two extra feature columns.
Two deliberate differences: ``gen_metafile`` only demands the f0 / uv files when the generator is an NSF one (the reference
always does), and ``load_meta_from_dir`` pairs ``<utt>.wav`` with ``<utt>.npy`` (the reference's version pairs it with
a ``.wav`` in the mel directory and returns tuples ``__getitem__`` cannot unpack).

Acoustic model: ``AM_Dataset`` / ``get_am_datasets`` (reference :391-869) over ``raw_metafile.txt`` + the ``mel/
duration/ f0/ energy/ frame_f0/ frame_uv/`` feature files, with the symbol tables of ``kantts.utils.ling_unit`` (native; the
language resource files are looked up as ``ling_unit.language_directory`` describes) and ``am_collate`` for batches --
items and batches pinned to the reference's by tests/golden/am_dataset.pt.  Waveforms must already be at the configured
rate.
"""
import glob
import logging
import os
import random

import numpy as np
import torch

from kantts.datasets.batching import Padder, am_collate, beta_binomial_prior_distribution, voc_collate
from kantts.preprocess.audio_processor.audio_processor import load_wav

DATASET_RANDOM_SEED = 1234


class Voc_Dataset(torch.utils.data.Dataset):
    """(wav, mel) pairs for HiFi-GAN training."""

    def __init__(self, metafile, root_dir, config):
        self.meta = []
        self.config = config
        self.sampling_rate = config["audio_config"]["sampling_rate"]
        self.n_fft = config["audio_config"]["n_fft"]
        self.hop_length = config["audio_config"]["hop_length"]
        self.batch_max_steps = config["batch_max_steps"]
        self.batch_max_frames = self.batch_max_steps // self.hop_length
        nsf = config["Model"]["Generator"]["params"].get("nsf_params", None)
        self.nsf_enable = nsf is not None
        metafile = metafile if isinstance(metafile, list) else [metafile]
        root_dir = root_dir if isinstance(root_dir, list) else [root_dir]
        for meta_file, data_dir in zip(metafile, root_dir):
            if not os.path.exists(meta_file):
                raise ValueError("[Voc_Dataset] meta file: {} not found".format(meta_file))
            if not os.path.exists(data_dir):
                raise ValueError("[Voc_Dataset] data dir: {} not found".format(data_dir))
            self.meta.extend(self.load_meta(meta_file, data_dir))
        if len(self.meta) == 0:
            for d in root_dir:
                self.meta.extend(self.load_meta_from_dir(os.path.join(d, "wav"), os.path.join(d, "mel")))
        self.allow_cache = config.get("allow_cache", False)
        self.caches = {}

    @staticmethod
    def gen_metafile(wav_dir, out_dir, split_ratio=0.98, need_f0=False):
        """train.lst / valid.lst: a seeded shuffle of the utterances whose features exist, split at split_ratio."""
        wav_files = sorted(glob.glob(os.path.join(wav_dir, "*.wav")))
        random.Random(DATASET_RANDOM_SEED).shuffle(wav_files)
        num_train = int(len(wav_files) * split_ratio) - 1
        needed = ["mel"] + (["frame_f0", "frame_uv"] if need_f0 else [])

        def write(path, files):
            with open(path, "w") as f:
                for wav_file in files:
                    index = os.path.splitext(os.path.basename(wav_file))[0]
                    if all(os.path.exists(os.path.join(out_dir, d, index + ".npy")) for d in needed):
                        f.write("{}\n".format(index))

        write(os.path.join(out_dir, "train.lst"), wav_files[:num_train])
        write(os.path.join(out_dir, "valid.lst"), wav_files[num_train:])

    def load_meta(self, metafile, data_dir):
        wav_dir, mel_dir = os.path.join(data_dir, "wav"), os.path.join(data_dir, "mel")
        if not os.path.exists(wav_dir) or not os.path.exists(mel_dir):
            raise ValueError("wav or mel directory not found")
        items = []
        with open(metafile, "r") as f:
            for name in f:
                name = name.strip()
                if name:
                    items.append((os.path.join(wav_dir, name + ".wav"), os.path.join(mel_dir, name + ".npy"),
                                  os.path.join(data_dir, "frame_f0", name + ".npy"),
                                  os.path.join(data_dir, "frame_uv", name + ".npy")))
        return items

    def load_meta_from_dir(self, wav_dir, mel_dir):
        if not os.path.exists(wav_dir) or not os.path.exists(mel_dir):
            raise ValueError("wav or mel directory not found")
        data_dir = os.path.dirname(os.path.normpath(wav_dir))
        items = []
        for wav_file in sorted(glob.glob(os.path.join(wav_dir, "*.wav"))):
            name = os.path.splitext(os.path.basename(wav_file))[0]
            mel_file = os.path.join(mel_dir, name + ".npy")
            if os.path.exists(mel_file):
                items.append((wav_file, mel_file, os.path.join(data_dir, "frame_f0", name + ".npy"),
                              os.path.join(data_dir, "frame_uv", name + ".npy")))
        return items

    def __len__(self):
        return len(self.meta)

    def __getitem__(self, idx):
        if self.allow_cache and idx in self.caches:
            return self.caches[idx]
        wav_file, mel_file, frame_f0_file, frame_uv_file = self.meta[idx]
        wav_data = load_wav(wav_file, self.sampling_rate)
        mel_data = np.load(mel_file)
        if self.nsf_enable:
            # frame f0 is stored mean / std normalised; the generator's source module wants Hz
            f0_dir = os.path.join(os.path.dirname(os.path.dirname(frame_f0_file)), "f0")
            f0_mean = np.loadtxt(os.path.join(f0_dir, "f0_mean.txt"))
            f0_std = np.loadtxt(os.path.join(f0_dir, "f0_std.txt"))
            frame_f0 = np.load(frame_f0_file).reshape(-1, 1) * f0_std + f0_mean
            frame_uv = np.load(frame_uv_file).reshape(-1, 1)
            mel_data = np.concatenate((mel_data, frame_f0, frame_uv), axis=1)
        if mel_data.shape[0] <= self.batch_max_frames:
            # at least one frame more than a crop, zeros after the utterance
            pad = np.zeros((self.batch_max_frames - mel_data.shape[0] + 1, mel_data.shape[1]))
            mel_data = np.concatenate((mel_data, pad), axis=0)
            wav_cache = np.zeros(mel_data.shape[0] * self.hop_length, dtype=np.float32)
            wav_cache[:len(wav_data)] = wav_data
            wav_data = wav_cache
        else:
            wav_data = np.pad(wav_data, (0, self.n_fft), mode="reflect")[:len(mel_data) * self.hop_length]
        assert len(mel_data) * self.hop_length == len(wav_data)
        if self.allow_cache:
            self.caches[idx] = (wav_data, mel_data)
        return wav_data, mel_data

    def collate_fn(self, batch):
        return voc_collate(batch, self.hop_length, self.batch_max_steps)


def get_voc_datasets(config, root_dir, split_ratio=0.98):
    root_dir = [root_dir] if isinstance(root_dir, str) else list(root_dir)
    need_f0 = config["Model"]["Generator"]["params"].get("nsf_params", None) is not None
    train_meta_lst, valid_meta_lst = [], []
    for data_dir in root_dir:
        train_meta, valid_meta = os.path.join(data_dir, "train.lst"), os.path.join(data_dir, "valid.lst")
        if not os.path.exists(train_meta) or not os.path.exists(valid_meta):
            Voc_Dataset.gen_metafile(os.path.join(data_dir, "wav"), data_dir, split_ratio, need_f0=need_f0)
        train_meta_lst.append(train_meta)
        valid_meta_lst.append(valid_meta)
    return Voc_Dataset(train_meta_lst, root_dir, config), Voc_Dataset(valid_meta_lst, root_dir, config)


class AM_Dataset(torch.utils.data.Dataset):
    """(linguistic ids, mel, durations, pitch, energy, alignment prior, fp label, speaker embedding) per utterance
    (reference dataset.py:391-545).  ``metafile``: path(s) of lines ``<id>\t{sy$tone$flag$ws$emo$spk} ...``; ``root_dir``:
    feature director(ies) with ``mel/ duration/ f0/ energy/ frame_f0/ frame_uv/ [se/se.npy]`` (``<id>.npy`` each; the layout
    AudioProcessor writes).  Behaviour kept from the reference: durations are used iff ``duration/`` exists and the model
    is not MAS (then a beta-binomial prior is attached instead); NSF models get frame-level f0 / voiced flag appended to
    the mel as two extra columns, f0 re-normalised to [global_min, global_max] when ``nsf_norm_type: global``; SE models
    load one speaker embedding per data directory.  Filled-pause (FP) metafiles are not supported (the FP model variant
    raises in this package).  Caching is a plain per-process dict (the reference shares a multiprocessing.Manager list)."""

    def __init__(self, config, metafile, root_dir, allow_cache=False, ling_unit=None):
        from kantts.utils.ling_unit import KanTtsLinguisticUnit

        params = config["Model"]["KanTtsSAMBERT"]["params"]
        self.config = config
        self.nsf_enable = bool(params.get("NSF", False))
        self.nsf_norm_type = params.get("nsf_norm_type", "mean_std")
        self.nsf_f0_global_minimum = params.get("nsf_f0_global_minimum", 30.0)
        self.nsf_f0_global_maximum = params.get("nsf_f0_global_maximum", 730.0)
        self.se_enable = bool(params.get("SE", False))
        self.mas_enable = bool(params.get("MAS", False))
        if params.get("FP", False):
            raise NotImplementedError("filled-pause (FP) metafiles are not supported")
        self.fp_enable = False
        self.r = params["outputs_per_step"]
        self.with_duration = True
        metafile = metafile if isinstance(metafile, list) else [metafile]
        root_dir = root_dir if isinstance(root_dir, list) else [root_dir]
        self.meta = []
        for meta_file, data_dir in zip(metafile, root_dir):
            if not os.path.exists(meta_file):
                raise ValueError("[AM_Dataset] meta file: {} not found".format(meta_file))
            if not os.path.exists(data_dir):
                raise ValueError("[AM_Dataset] data dir: {} not found".format(data_dir))
            self.meta.extend(self.load_meta(meta_file, data_dir))
        self.ling_unit = ling_unit if ling_unit is not None else KanTtsLinguisticUnit(config)
        self.padder = Padder()
        self.allow_cache = allow_cache
        self.caches = {}

    def __len__(self):
        return len(self.meta)

    def load_meta(self, metafile, data_dir):
        sub = {k: os.path.join(data_dir, k) for k in ("mel", "duration", "f0", "energy", "frame_f0", "frame_uv", "se")}
        # (the last directory decides for the whole dataset, as in the reference)
        self.with_duration = False if self.mas_enable else os.path.exists(sub["duration"])
        se_path = os.path.join(sub["se"], "se.npy")
        if self.se_enable and not os.path.exists(se_path):
            logging.warning("Missing se meta")
            return []
        items = []
        with open(metafile, "r") as f:
            for line in f:
                line = line.strip()
                if not line:
                    continue
                index, ling_txt = line.split("\t")
                npy = index + ".npy"
                items.append((ling_txt, os.path.join(sub["mel"], npy),
                              os.path.join(sub["duration"], npy) if self.with_duration else None,
                              os.path.join(sub["f0"], npy), os.path.join(sub["energy"], npy),
                              os.path.join(sub["frame_f0"], npy), os.path.join(sub["frame_uv"], npy), None, se_path))
        return items

    def __getitem__(self, idx):
        if self.allow_cache and idx in self.caches:
            return self.caches[idx]
        ling_txt, mel_file, dur_file, f0_file, energy_file, frame_f0_file, frame_uv_file, _, se_path = self.meta[idx]
        ling_data = self.ling_unit.encode_symbol_sequence(ling_txt)
        mel_data = np.load(mel_file)
        dur_data = np.load(dur_file) if dur_file is not None else None
        f0_data = np.load(f0_file)
        energy_data = np.load(energy_file)
        se_data = np.load(se_path) if self.se_enable else None
        attn_prior = None if self.with_duration else beta_binomial_prior_distribution(len(ling_data[0]), mel_data.shape[0])
        if self.nsf_enable:
            frame_f0 = np.load(frame_f0_file).reshape(-1, 1)  # stored mean / std normalised
            if self.nsf_norm_type == "global":
                f0_dir = os.path.join(os.path.dirname(os.path.dirname(frame_f0_file)), "f0")
                mean, std = np.loadtxt(os.path.join(f0_dir, "f0_mean.txt")), np.loadtxt(os.path.join(f0_dir, "f0_std.txt"))
                frame_f0 = ((frame_f0 * std + mean) - self.nsf_f0_global_minimum) / (
                    self.nsf_f0_global_maximum - self.nsf_f0_global_minimum)
            mel_data = np.concatenate([mel_data, frame_f0, np.load(frame_uv_file).reshape(-1, 1)], axis=1)
        item = (ling_data, mel_data, dur_data, f0_data, energy_data, attn_prior, None, se_data)
        if self.allow_cache:
            self.caches[idx] = item
        return item

    def collate_fn(self, batch):
        pad_ids = [self.ling_unit._sub_unit_pad[t] for t in self.ling_unit._lfeat_type_list]
        return am_collate(batch, self.r, pad_ids, se=self.se_enable)

    @staticmethod
    def gen_metafile(raw_meta_file, out_dir, train_meta_file, valid_meta_file, badlist=None, split_ratio=0.98,
                     se_enable=False):
        """Shuffle the raw metafile with the fixed dataset seed and split it; utterances without mel / frame_f0 /
        frame_uv (or, when ``duration/`` exists, without a duration file) are dropped (reference :626-688)."""
        with open(raw_meta_file, "r") as f:
            lines = f.readlines()
        random.Random(DATASET_RANDOM_SEED).shuffle(lines)  # == random.seed(1234); random.shuffle(lines)
        num_train = int(len(lines) * split_ratio) - 1
        have_dur = os.path.exists(os.path.join(out_dir, "duration"))
        se_missing = se_enable and os.path.exists(os.path.join(out_dir, "se")) and not os.path.exists(
            os.path.join(out_dir, "se", "se.npy"))

        def keep(index):
            if badlist is not None and index in badlist:
                return False
            if not all(os.path.exists(os.path.join(out_dir, d, index + ".npy")) for d in ("frame_f0", "frame_uv", "mel")):
                return False
            if have_dur and not os.path.exists(os.path.join(out_dir, "duration", index + ".npy")):
                return False
            return not se_missing

        for path, part in ((train_meta_file, lines[:num_train]), (valid_meta_file, lines[num_train:])):
            with open(path, "w") as f:
                f.writelines(ln for ln in part if keep(ln.split("\t")[0]))


def get_am_datasets(metafile, root_dir, config, allow_cache, split_ratio=0.98, se_enable=False):
    """(train, valid) AM_Datasets over ``am_train.lst`` / ``am_valid.lst`` of every data directory, generated from the raw
    metafile when missing (reference :831-869; its call passes ``split_ratio, se_enable`` into the positional slots of
    ``badlist, split_ratio`` and would raise a TypeError whenever it actually has to generate the lists -- here the
    arguments go where their names say)."""
    root_dir = root_dir if isinstance(root_dir, list) else [root_dir]
    metafile = metafile if isinstance(metafile, list) else [metafile]
    if config["Model"]["KanTtsSAMBERT"]["params"].get("FP", False):
        raise NotImplementedError("filled-pause (FP) metafiles are not supported")
    train_lst, valid_lst = [], []
    for raw_metafile, data_dir in zip(metafile, root_dir):
        train_meta, valid_meta = os.path.join(data_dir, "am_train.lst"), os.path.join(data_dir, "am_valid.lst")
        if not os.path.exists(train_meta) or not os.path.exists(valid_meta):
            AM_Dataset.gen_metafile(raw_metafile, data_dir, train_meta, valid_meta, split_ratio=split_ratio,
                                    se_enable=se_enable)
        train_lst.append(train_meta)
        valid_lst.append(valid_meta)
    return AM_Dataset(config, train_lst, root_dir, allow_cache), AM_Dataset(config, valid_lst, root_dir, allow_cache)


logging.getLogger(__name__).addHandler(logging.NullHandler())
