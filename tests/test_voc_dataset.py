"""Vocoder dataset over the on-disk feature layout (kantts.datasets.dataset.Voc_Dataset): items and a seeded batch must be
identical to the reference's Voc_Dataset on the same directory (tests/golden/voc_dataset.pt, which carries the raw file
contents), plain and NSF configuration; plus the full wav -> AudioProcessor.mel_extract -> Voc_Dataset -> batch chain."""
import os

import numpy as np
import pytest
import torch
from scipy.io import wavfile

from util import GOLDEN


def _rebuild(fix, d):
    for sub in ("wav", "mel", "frame_f0", "frame_uv", "f0"):
        os.makedirs(os.path.join(d, sub), exist_ok=True)
    for name, u in fix["utts"].items():
        wavfile.write(os.path.join(d, "wav", name + ".wav"), fix["sr"], u["wav"])
        np.save(os.path.join(d, "mel", name + ".npy"), u["mel"])
        np.save(os.path.join(d, "frame_f0", name + ".npy"), u["f0"])
        np.save(os.path.join(d, "frame_uv", name + ".npy"), u["uv"])
    np.savetxt(os.path.join(d, "f0", "f0_mean.txt"), np.array([fix["f0_mean"]]))
    np.savetxt(os.path.join(d, "f0", "f0_std.txt"), np.array([fix["f0_std"]]))
    with open(os.path.join(d, "train.lst"), "w") as f:
        f.write("\n".join(sorted(fix["utts"])) + "\n")


@pytest.mark.parametrize("tag", ["plain", "nsf"])
def test_voc_dataset_items_and_batch_match_reference(tmp_path, tag):
    from kantts.datasets.dataset import Voc_Dataset

    fix = torch.load(os.path.join(GOLDEN, "voc_dataset.pt"), weights_only=False)
    d = str(tmp_path)
    _rebuild(fix, d)
    nsf = None if tag == "plain" else {"nb_harmonics": 7, "sampling_rate": fix["sr"]}
    config = {"audio_config": {"sampling_rate": fix["sr"], "n_fft": fix["n_fft"], "hop_length": fix["hop"]},
              "batch_max_steps": fix["batch_max_steps"], "allow_cache": True,
              "Model": {"Generator": {"params": {"nsf_params": nsf}}}}
    ds = Voc_Dataset([os.path.join(d, "train.lst")], [d], config)
    exp = fix["expected"][tag]
    assert len(ds) == len(exp["items"])
    items = [ds[i] for i in range(len(ds))]
    for (w, m), (rw, rm) in zip(items, exp["items"]):
        assert w.dtype == rw.dtype and m.dtype == rm.dtype and w.shape == rw.shape and m.shape == rm.shape
        assert np.array_equal(w, rw) and np.array_equal(m, rm)
        assert len(w) == len(m) * fix["hop"]
    assert ds[1][0] is items[1][0]  # allow_cache
    np.random.seed(fix["collate_seed"])
    wav_b, mel_b = ds.collate_fn(items)
    assert torch.equal(wav_b, exp["batch"][0]) and torch.equal(mel_b, exp["batch"][1])


def test_metafile_generation_and_directory_fallback(tmp_path):
    from kantts.datasets.dataset import Voc_Dataset, get_voc_datasets

    fix = torch.load(os.path.join(GOLDEN, "voc_dataset.pt"), weights_only=False)
    d = str(tmp_path)
    _rebuild(fix, d)
    os.remove(os.path.join(d, "train.lst"))
    os.remove(os.path.join(d, "mel", "a03.npy"))  # an utterance without features is left out of the lists
    config = {"audio_config": {"sampling_rate": fix["sr"], "n_fft": fix["n_fft"], "hop_length": fix["hop"]},
              "batch_max_steps": fix["batch_max_steps"], "allow_cache": False,
              "Model": {"Generator": {"params": {}}}}
    train, valid = get_voc_datasets(config, d, split_ratio=0.98)
    names = [ln.strip() for f in ("train.lst", "valid.lst") for ln in open(os.path.join(d, f)) if ln.strip()]
    assert sorted(names) == ["a01", "a02", "a04"] and len(train) + len(valid) == 3
    # the split is a function of the seed only
    Voc_Dataset.gen_metafile(os.path.join(d, "wav"), d, 0.98)
    assert names == [ln.strip() for f in ("train.lst", "valid.lst") for ln in open(os.path.join(d, f)) if ln.strip()]
    # an empty list falls back to pairing wav/<utt>.wav with mel/<utt>.npy
    open(os.path.join(d, "empty.lst"), "w").close()
    ds = Voc_Dataset([os.path.join(d, "empty.lst")], [d], config)
    assert len(ds) == 3 and ds[0][1].shape[1] == 80
    with pytest.raises(ValueError):
        Voc_Dataset([os.path.join(d, "missing.lst")], [d], config)
    # (the acoustic-model counterpart, AM_Dataset / get_am_datasets, is covered by tests/test_am_dataset.py)


def _chain(tmp_path, device, train_steps=True):
    """wav files -> AudioProcessor.mel_extract -> get_voc_datasets -> DataLoader batches with matching wav / mel crops."""
    from torch.utils.data import DataLoader

    from kantts.datasets.dataset import get_voc_datasets
    from kantts.preprocess.audio_processor.audio_processor import AudioProcessor

    rng = np.random.RandomState(3)
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "wav"))
    for i, n in enumerate([9000, 20000, 12000, 15000, 11000]):
        x = 0.2 * np.sin(2 * np.pi * (150 + 30 * i) * np.arange(n) / 16000.0) + 0.02 * rng.randn(n)
        wavfile.write(os.path.join(d, "wav", "u%d.wav" % i), 16000, np.round(x * 32768).astype(np.int16))
    acfg = {"sampling_rate": 16000, "hop_length": 200, "win_length": 1000, "n_mels": 80, "n_fft": 2048, "fmin": 0.0,
            "fmax": 8000.0, "min_level_db": -100, "ref_level_db": 20, "max_norm": 1.0, "symmetric": False,
            "preemphasize": False, "num_workers": 1}
    AudioProcessor(acfg, batch_size=4, device=device).mel_extract(os.path.join(d, "wav"), os.path.join(d, "mel"))
    config = {"audio_config": acfg, "batch_max_steps": 4000, "allow_cache": False,
              "Model": {"Generator": {"params": {}}}}
    train, valid = get_voc_datasets(config, d, split_ratio=0.98)
    assert len(train) + len(valid) == 5
    for w, m in (train[i] for i in range(len(train))):
        assert len(w) == len(m) * 200 and m.shape[1] == 80
    np.random.seed(0)
    wav_b, mel_b = next(iter(DataLoader(train, batch_size=3, shuffle=False, collate_fn=train.collate_fn)))
    assert tuple(wav_b.shape) == (3, 1, 4000) and tuple(mel_b.shape) == (3, 80, 20)
    if not train_steps:
        return
    # ... and the training entry point on that directory: real files in, checkpoints out
    from kantts.bin.train_hifigan import train as train_voc

    opt = {"type": "Adam", "params": {"lr": 2e-4, "betas": [0.5, 0.9], "weight_decay": 0.0}}
    sch = {"type": "MultiStepLR", "params": {"gamma": 0.5, "milestones": [200000]}}
    voc = {"model_type": "hifigan", "audio_config": acfg, "Model": {
        "Generator": {"params": {"channels": 32, "upsample_scales": [5, 5, 4, 2], "upsample_kernal_sizes": [10, 10, 8, 4]},
                      "optimizer": opt, "scheduler": sch},
        "MultiPeriodDiscriminator": {"params": {"periods": [2, 3]}, "optimizer": opt, "scheduler": sch}},
        "Loss": {"generator_adv_loss": {"enable": True, "params": {}, "weights": 1.0},
                 "discriminator_adv_loss": {"enable": True, "params": {}, "weights": 1.0},
                 "mel_loss": {"enable": True, "params": {"fs": 16000, "fft_size": 2048, "hop_size": 200,
                                                         "win_length": 1000, "fmin": 0, "fmax": 8000}, "weights": 45.0},
                 "feat_match_loss": {"enable": True, "params": {}, "weights": 2.0}},
        "generator_grad_norm": -1, "discriminator_grad_norm": -1, "discriminator_train_start_steps": 0,
        "generator_train_start_steps": 0, "batch_size": 2, "batch_max_steps": 4000, "num_workers": 0,
        "pin_memory": False, "allow_cache": True, "log_interval_steps": 1, "save_interval_steps": 2,
        "train_max_steps": 3}
    tr = train_voc(voc, [d], os.path.join(d, "stage"))
    assert tr.steps >= 3 and os.path.exists(os.path.join(d, "stage", "ckpt", "checkpoint_2.pth"))


def test_wav_to_batches_chain_emulated(tmp_path, emulated_cabi):
    _chain(tmp_path, "cpu")


@pytest.mark.gpu
def test_wav_to_batches_chain_gpu(tmp_path):
    # extraction on the device + dataset + loader; the training entry point on real shapes is covered by
    # tests/test_entrypoints.py / test_trainer.py on the GPU and by the emulated variant of this test
    _chain(tmp_path, "cuda", train_steps=False)
