"""Offline mel extraction + on-disk feature formats: kantts.preprocess.audio_processor.AudioProcessor
.mel_extract against the float64 numpy restatement of the reference's dsp.melspectrogram (oracle/audio_oracle.py) and the
reference's statistics formulas (core/utils.py:404-434, :496-499).  librosa is not installed: its two ingredients are pinned by independent implementations
(tests/test_independent_pins.py: STFT conventions by scipy, mel basis by transformers)."""
import os

import numpy as np
import pytest

import audio_oracle as AO

CFG = {"sampling_rate": 16000, "hop_length": 200, "win_length": 1000, "n_mels": 80, "n_fft": 2048, "fmin": 0.0,
       "fmax": 8000.0, "min_level_db": -100, "ref_level_db": 20, "max_norm": 1.0, "symmetric": False,
       "preemphasize": False, "num_workers": 1}


def _write_corpus(tmp_path, lengths, seed=0):
    from scipy.io import wavfile

    rng = np.random.RandomState(seed)
    wav_dir = tmp_path / "wav"
    os.makedirs(wav_dir, exist_ok=True)
    pcm = {}
    for i, n in enumerate(lengths):
        t = np.arange(n) / 16000.0
        x = 0.3 * np.sin(2 * np.pi * (110 + 40 * i) * t) + 0.05 * rng.randn(n)
        q = np.clip(np.round(x * 32768), -32768, 32767).astype(np.int16)
        wavfile.write(str(wav_dir / ("utt%02d.wav" % i)), 16000, q)
        pcm["utt%02d" % i] = (q / 32768.0).astype(np.float32)
    return str(wav_dir), pcm


def _run(tmp_path, device):
    from kantts.preprocess.audio_processor.audio_processor import AudioProcessor

    wav_dir, pcm = _write_corpus(tmp_path, [9000, 12345, 8000, 16001, 4000])  # the last one is < 0.5 s: skipped
    out_dir = str(tmp_path / "mel")
    ap = AudioProcessor(dict(CFG), batch_size=3, device=device)
    assert ap.mel_extract(wav_dir, out_dir)
    assert ap.badcase_list == ["utt04"]
    names = sorted(n for n in pcm if n != "utt04")
    ref = {n: AO.dsp_melspectrogram(pcm[n].astype(np.float64), 16000, n_fft=2048, hop_length=200, win_length=1000, n_mels=80,
                                    max_norm=1.0, min_level_db=-100, ref_level_db=20, fmin=0.0, fmax=8000.0)
           for n in names}
    for n in names:
        assert ap.mel_dict[n].shape == (1 + len(pcm[n]) // 200, 80) and ap.mel_dict[n].dtype == np.float32
        assert np.abs(ap.mel_dict[n] - ref[n]).max() < 2e-4, n      # batching / zero padding does not leak between rows
    allf = np.concatenate([ap.mel_dict[n] for n in names], axis=0).astype(np.float64)
    mean, std = allf.mean(0, keepdims=True), allf.std(0, keepdims=True)
    got_mean = np.loadtxt(os.path.join(out_dir, "mel_mean.txt")).reshape(1, -1)
    got_std = np.loadtxt(os.path.join(out_dir, "mel_std.txt")).reshape(1, -1)
    assert got_mean.shape == (1, 80) and np.abs(got_mean - mean).max() < 2e-6 and np.abs(got_std - std).max() < 2e-6
    for n in names:
        normed = np.load(os.path.join(out_dir, n + ".npy"))
        assert normed.shape == ap.mel_dict[n].shape and normed.dtype == np.float64   # float32 mel - float64 statistics
        assert np.abs(normed - (ap.mel_dict[n] - mean) / std).max() < 1e-4   # mean / std recomputed in another order
    assert not os.path.exists(os.path.join(out_dir, "utt04.npy"))
    corpus = np.concatenate([np.load(os.path.join(out_dir, n + ".npy")) for n in names], axis=0)
    # per-utterance sums are taken in float32 (np.sum of float32 features), as in the reference -> ~1e-6, not 1e-16
    assert np.abs(corpus.mean(0)).max() < 1e-4 and np.abs(corpus.std(0) - 1).max() < 1e-4


def test_mel_extract_formats_emulated(tmp_path, emulated_cabi):
    _run(tmp_path, "cpu")


@pytest.mark.gpu
def test_mel_extract_formats_gpu(tmp_path):
    _run(tmp_path, "cuda")


def test_wrong_sampling_rate_is_refused(tmp_path):
    from scipy.io import wavfile

    from kantts.preprocess.audio_processor.audio_processor import load_wav

    p = str(tmp_path / "a.wav")
    wavfile.write(p, 22050, np.zeros(100, dtype=np.int16))
    with pytest.raises(ValueError):
        load_wav(p, 16000)
