"""Vocoder dataset over the on-disk feature layout -- the reader side of
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

The acoustic-model dataset needs the text front-end's symbol tables (``ling_unit``), which live with the reference
package; ``get_am_datasets`` says so instead of half-working.  Waveforms must already be at the configured rate.
"""
import glob
import logging
import os
import random

import numpy as np
import torch

from kantts.datasets.batching import voc_collate
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


def reference_dataset_module():
    """The reference's own kantts/datasets/dataset.py loaded from the checkout named by KANTTS_REFERENCE_ROOT (its
    imports of kantts.utils.ling_unit etc. resolve through the same overlay); None without a checkout."""
    import importlib.util
    import sys

    import kantts

    name = "kantts.datasets._reference_dataset"
    if name in sys.modules:
        return sys.modules[name]
    if not kantts.REFERENCE_ROOT:
        return None
    path = os.path.join(kantts.REFERENCE_ROOT, "kantts", "datasets", "dataset.py")
    if not os.path.exists(path):
        return None
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    try:
        spec.loader.exec_module(mod)
    except Exception:
        del sys.modules[name]
        raise
    return mod


def get_am_datasets(*args, **kwargs):
    """SAM-BERT datasets: the reference's AM_Dataset (symbol tables, metafile parsing, feature files) from a checkout,
    when one is configured; its batches are what kantts.datasets.batching.am_collate reproduces."""
    ref = reference_dataset_module()
    if ref is None:
        raise ImportError("the acoustic-model dataset needs the text front-end's symbol tables, which live with the "
                          "reference package: set KANTTS_REFERENCE_ROOT to a KAN-TTS checkout, or pass --synthetic N")
    return ref.get_am_datasets(*args, **kwargs)


logging.getLogger(__name__).addHandler(logging.NullHandler())
