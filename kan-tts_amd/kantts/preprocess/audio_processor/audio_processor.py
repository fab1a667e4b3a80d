"""Offline mel feature extraction and the on-disk formats the datasets read.

``AudioProcessor.mel_extract`` has the reference's contract (kantts/preprocess/audio_processor/audio_processor.py:317-387):
every ``*.wav`` of a directory -> global-normalised mel (frames, n_mels) -> corpus mean / std per mel bin ->
``mel_mean.txt`` / ``mel_std.txt`` (``%.6f`` rows of n_mels values) and one ``<name>.npy`` per utterance holding
``(mel - mean) / std``.  The reference fans the utterances out to 16 CPU workers, one librosa STFT each; here the waveforms
of a batch are zero-padded to a common length and go through ONE launch of the mel-STFT kernel (``dsp.melspectrogram_batch``)
-- zero padding on the right is what the centred STFT pads with anyway, so the first 1 + T // hop frames of the padded
signal are the utterance's own frames.  Statistics are accumulated in float64 on the host exactly as the reference's
``compute_mean`` / ``compute_std`` (core/utils.py:404-434) do, because their text files are part of the data contract.

Pitch / energy / duration extraction, loudness normalisation and silence trimming need pysptk / sox / librosa and stay
with the reference package; waveforms must already be at ``sampling_rate`` (the reference resamples through librosa).
"""
import logging
import os
from glob import glob

import numpy as np
import torch

from kantts.preprocess.audio_processor.core.dsp import melspectrogram_batch

default_audio_config = {
    "wav_normalize": True, "trim_silence": True, "trim_silence_threshold_db": 60, "preemphasize": False,
    "sampling_rate": 24000, "hop_length": 240, "win_length": 1024, "n_mels": 80, "n_fft": 1024, "fmin": 50.0,
    "fmax": 7600.0, "min_level_db": -100, "ref_level_db": 20, "phone_level_feature": True, "num_workers": 16,
    "norm_type": "mean_std", "max_norm": 1.0, "symmetric": False,
}


def load_wav(path, sr):
    """float32 samples in [-1, 1) of a PCM wav file that is already at ``sr``."""
    from scipy.io import wavfile

    rate, data = wavfile.read(path)
    if rate != sr:
        raise ValueError("%s is at %d Hz, expected %d Hz (resampling lives in the reference's librosa front-end)" %
                         (path, rate, sr))
    if data.ndim > 1:
        data = data.mean(axis=1)
    if data.dtype == np.int16:
        return (data / 32768.0).astype(np.float32)
    if data.dtype == np.int32:
        return (data / 2147483648.0).astype(np.float32)
    return data.astype(np.float32)


def compute_mean(data_list, dims=80):
    total = np.zeros((1, dims))
    frames = 0
    for data in data_list:
        if data is None:
            continue
        feats = data.reshape((-1, dims))
        total += np.sum(feats, axis=0)
        frames += feats.shape[0]
    return total / float(frames)


def compute_std(data_list, mean_vector, dims=80):
    total = np.zeros((1, dims))
    frames = 0
    for data in data_list:
        if data is None:
            continue
        feats = data.reshape((-1, dims))
        total += np.sum((feats - mean_vector) ** 2, axis=0)
        frames += feats.shape[0]
    return (total / float(frames)) ** 0.5


def norm_mean_std(x, mean, std):
    return (x - mean) / std


class AudioProcessor:
    def __init__(self, config=None, batch_size=64, device=None):
        if not isinstance(config, dict):
            logging.warning("[AudioProcessor] config is not a dict, fall into default config.")
            config = default_audio_config
        self.config = config
        for key in self.config:
            setattr(self, key, self.config[key])
        self.min_wav_length = int(self.config["sampling_rate"] * 0.5)
        self.batch_size = int(batch_size)
        self.device = device
        self.badcase_list = []
        self.pcm_dict = {}
        self.mel_dict = {}

    def _device(self):
        if self.device is not None:
            return torch.device(self.device)
        return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")

    def get_pcm_dict(self, src_wav_dir):
        if len(self.pcm_dict) > 0:
            return self.pcm_dict
        for wav_path in sorted(glob(os.path.join(src_wav_dir, "*.wav"))):
            name = os.path.splitext(os.path.basename(wav_path))[0]
            pcm = load_wav(wav_path, self.sampling_rate)
            if len(pcm) < self.min_wav_length:
                logging.warning("[AudioProcessor] %s is too short, skip", name)
                self.badcase_list.append(name)
                continue
            self.pcm_dict[name] = pcm
        return self.pcm_dict

    def melspec_dict(self, pcm_dict):
        """{name: (frames, n_mels) float32}: utterances sorted by length, ``batch_size`` of them per kernel launch."""
        dev = self._device()
        names = sorted(pcm_dict, key=lambda n: len(pcm_dict[n]))
        out = {}
        for i in range(0, len(names), self.batch_size):
            chunk = names[i:i + self.batch_size]
            T = max(len(pcm_dict[n]) for n in chunk)
            host = np.zeros((len(chunk), T), dtype=np.float32)
            for r, n in enumerate(chunk):
                host[r, :len(pcm_dict[n])] = pcm_dict[n]
            mel = melspectrogram_batch(torch.from_numpy(host).to(dev), self.sampling_rate, self.n_fft, self.hop_length,
                                       self.win_length, self.n_mels, self.max_norm, self.min_level_db, self.ref_level_db,
                                       self.fmin, self.fmax, self.symmetric, self.preemphasize).cpu().numpy()
            for r, n in enumerate(chunk):
                out[n] = mel[r, :1 + len(pcm_dict[n]) // self.hop_length].copy()
        return out

    def mel_extract(self, src_wav_dir, out_feature_dir):
        os.makedirs(out_feature_dir, exist_ok=True)
        pcm_dict = self.get_pcm_dict(src_wav_dir)
        logging.info("[AudioProcessor] Melspec extraction started")
        self.mel_dict.update(self.melspec_dict(pcm_dict))
        mels = list(self.mel_dict.values())
        mel_mean = compute_mean(mels, dims=self.n_mels)
        mel_std = compute_std(mels, mel_mean, dims=self.n_mels)
        np.savetxt(os.path.join(out_feature_dir, "mel_mean.txt"), mel_mean, fmt="%.6f")
        np.savetxt(os.path.join(out_feature_dir, "mel_std.txt"), mel_std, fmt="%.6f")
        for name, mel in self.mel_dict.items():
            np.save(os.path.join(out_feature_dir, name + ".npy"), norm_mean_std(mel, mel_mean, mel_std))
        logging.info("[AudioProcessor] Normed Melspec saved to %s", out_feature_dir)
        return True
