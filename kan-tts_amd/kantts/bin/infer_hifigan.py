"""HiFi-GAN inference entry point: mel (.npy, (T, C)) -> 16-bit wav, with the real-time factor the reference logs.

Mirrors kantts/bin/infer_hifigan.py:34-163 of the reference (same function names, arguments, checkpoint layout
``states["model"]["generator"]``, ``<ckpt>/../../config.yaml`` discovery, ``<utt>_gen.wav`` outputs).  The generator
runs on the MI355X kernels (kantts/models/hifigan); wav files are written with scipy (soundfile is not a
dependency here).  NSF generators take (T, C + 2) features whose last column (voiced flag) is re-binarised first, as in
the reference (:52-63, :112-113); multi-band (PQMF) generators are refused.
"""
import argparse
import glob
import logging
import os
import time

import numpy as np
import torch
import yaml
from scipy.io import wavfile

logging.basicConfig(format="%(asctime)s, %(levelname)-4s [%(filename)s:%(lineno)d] %(message)s",
                    datefmt="%Y-%m-%d:%H:%M:%S", level=logging.INFO)


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def _load_config(ckpt, config):
    if isinstance(config, dict):
        return config
    path = config if config is not None else os.path.join(os.path.dirname(os.path.dirname(ckpt)), "config.yaml")
    if not os.path.exists(path):
        raise ValueError("config file not found: {}".format(path))
    with open(path) as f:
        return yaml.load(f, Loader=yaml.Loader)


def load_model(ckpt, config=None):
    config = _load_config(ckpt, config)
    from kantts.models.hifigan.hifigan import Generator

    params = config["Model"]["Generator"]["params"]
    model = Generator(**params)
    states = torch.load(ckpt, map_location="cpu")
    model.load_state_dict(states["model"]["generator"])
    if params.get("out_channels", 1) > 1:  # multi-band generator: PQMF synthesis after it (reference :47-52, :120-121)
        from kantts.models.pqmf import PQMF

        model.pqmf = PQMF(subbands=params["out_channels"], **config.get("pqmf", {}))
    return model


def binarize(mel, threshold=0.6):
    """NSF features: the voiced / unvoiced column (last) predicted by the acoustic model back to {0, 1}."""
    out = np.array(mel, copy=True)
    out[:, -1] = (mel[:, -1] >= threshold).astype(out.dtype)
    return out


def _device():
    return torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")


def hifigan_infer(input_mel, ckpt_path, output_dir, config=None):
    device = _device()
    config = _load_config(ckpt_path, config)
    os.makedirs(output_dir, exist_ok=True)
    if os.path.isfile(input_mel):
        mel_lst = [input_mel]
    elif os.path.isdir(input_mel):
        mel_lst = sorted(glob.glob(os.path.join(input_mel, "*.npy")))
    else:
        raise ValueError("input_mel should be a file or a directory")
    model = load_model(ckpt_path, config)
    logging.info("Loaded model parameters from %s (%d parameters).", ckpt_path, count_parameters(model))
    model.remove_weight_norm()
    model = model.eval().to(device)
    sr = config["audio_config"]["sampling_rate"]
    pcm_len = 0
    with torch.no_grad():
        start = time.time()
        for mel in mel_lst:
            utt_id = os.path.splitext(os.path.basename(mel))[0]
            feats = np.load(mel)
            if model.nsf_enable:
                feats = binarize(feats)
            mel_data = torch.from_numpy(np.ascontiguousarray(feats)).float().to(device)
            y = model(mel_data.transpose(1, 0).unsqueeze(0))  # (T, C) -> (1, C, T)
            if hasattr(model, "pqmf"):
                y = model.pqmf.synthesis(y)
            y = y.reshape(-1).cpu().numpy()
            pcm_len += len(y)
            wavfile.write(os.path.join(output_dir, "%s_gen.wav" % utt_id), sr,
                          (np.clip(y, -1.0, 1.0) * 32767.0).astype(np.int16))
        rtf = (time.time() - start) / max(pcm_len / sr, 1e-9)
    logging.info("Finished generation of %d utterances (RTF = %.03f).", len(mel_lst), rtf)
    return rtf


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Infer hifigan model")
    parser.add_argument("--ckpt", type=str, required=True, help="Path to model checkpoint")
    parser.add_argument("--input_mel", type=str, required=True,
                        help="Path to input mel file or directory containing mel files")
    parser.add_argument("--output_dir", type=str, required=True, help="Path to output directory")
    parser.add_argument("--config", type=str, default=None, help="Path to config file")
    args = parser.parse_args()
    hifigan_infer(args.input_mel, args.ckpt, args.output_dir, args.config)
