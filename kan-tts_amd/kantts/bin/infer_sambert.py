"""SAM-BERT acoustic-model inference entry point: linguistic symbols -> mel (.npy) + duration / f0 / energy text files.

Mirrors kantts/bin/infer_sambert.py:58-239 of the reference: ``am_synthesis(symbol_seq, fsnet, ling_unit, device)`` and
``am_infer(sentence, ckpt, output_dir, se_file=None, config=None)`` with the same outputs (``feat/<id>_mel.npy`` holds
the post-net mel).  The text front-end (``KanTtsLinguisticUnit``) is not part of the hot path: ``am_infer`` imports it
from the reference package layout if it is installed, or accepts any object with ``encode_symbol_sequence`` /
``get_unit_size`` / ``using_byte`` through ``ling_unit=``.  The model itself runs free-running on the MI355X kernels
(AR duration predictor + AR decoder, kantts/models/sambert).
"""
import argparse
import logging
import os

import numpy as np
import torch
import yaml

logging.basicConfig(format="%(asctime)s, %(levelname)-4s [%(filename)s:%(lineno)d] %(message)s",
                    datefmt="%Y-%m-%d:%H:%M:%S", level=logging.INFO)


def am_synthesis(symbol_seq, fsnet, ling_unit, device, se=None):
    if ling_unit.using_byte():
        raise NotImplementedError("byte-index inputs (sambert_16k_MAS_byte.yaml) are outside the hot path")
    feats = ling_unit.encode_symbol_sequence(symbol_seq)
    sy, tone, syllable, ws, emo, spk = (torch.from_numpy(np.asarray(f)).long().to(device) for f in feats[:6])
    # the trailing "~" token is dropped from every stream (reference :117-122)
    inputs_ling = torch.stack([sy, tone, syllable, ws], dim=-1).unsqueeze(0)[:, :-1, :]
    inputs_emo = emo.unsqueeze(0)[:, :-1]
    if se is not None:
        # SE models (sambert_se_nsf_global_16k.yaml): the utterance-level speaker embedding of --se_file, (1, dim),
        # repeated over the symbols instead of speaker ids (reference :99-106)
        inputs_spk = torch.from_numpy(np.asarray(se).repeat(len(feats[5]), axis=0)).float().to(device).unsqueeze(0)[:, :-1, :]
    else:
        inputs_spk = spk.unsqueeze(0)[:, :-1]
    inputs_len = torch.full((1,), inputs_emo.size(1), dtype=torch.long, device=device)
    res = fsnet(inputs_ling, inputs_emo, inputs_spk, inputs_len)
    valid_length = int(res["LR_length_rounded"][0].item())
    dec_outputs = res["dec_outputs"][0, :valid_length, :].cpu().numpy()
    postnet_outputs = res["postnet_outputs"][0, :valid_length, :].cpu().numpy()
    duration_predictions = (torch.exp(res["log_duration_predictions"]) - 1 + 0.5).long().squeeze().cpu().numpy()
    pitch_predictions = res["pitch_predictions"].squeeze().cpu().numpy()
    energy_predictions = res["energy_predictions"].squeeze().cpu().numpy()
    logging.info("x_band_width:%s, h_band_width: %s", res["x_band_width"], res["h_band_width"])
    return dec_outputs, postnet_outputs, duration_predictions, pitch_predictions, energy_predictions


def denorm_f0(mel, scale, offset, f0_threshold=30.0, uv_threshold=0.6):
    """The last two channels of an NSF acoustic model's output (frames, n_mels + 2): voicing score -> {0, 1} at
    ``uv_threshold``, f0 -> ``f0 * scale + offset`` Hz floored at ``f0_threshold`` (reference :26-56; mean_std: scale = std,
    offset = mean; global: scale = max - min, offset = min).  In place, like the reference."""
    uv = mel[:, -1]
    mel[:, -1] = np.where(uv < uv_threshold, 0.0, 1.0)
    mel[:, -2] = np.maximum(mel[:, -2] * scale + offset, f0_threshold)
    return mel


def am_infer(sentence, ckpt, output_dir, se_file=None, config=None, ling_unit=None):
    device = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
    if not isinstance(config, dict):
        path = config if config is not None else os.path.join(os.path.dirname(os.path.dirname(ckpt)), "config.yaml")
        with open(path) as f:
            config = yaml.load(f, Loader=yaml.Loader)
    if ling_unit is None:  # symbol tables of this package; the sentences file holds symbol sequences, not raw text
        from kantts.utils.ling_unit import KanTtsLinguisticUnit

        ling_unit = KanTtsLinguisticUnit(config)
    config["Model"]["KanTtsSAMBERT"]["params"].update(ling_unit.get_unit_size())
    se_enable = config["Model"]["KanTtsSAMBERT"]["params"].get("SE", False)
    if se_enable and se_file is None:
        raise ValueError("this checkpoint takes a speaker embedding (SE: True): --se_file is required")
    se = np.load(se_file) if se_enable else None  # (the reference ignores --se_file for models without SE, :177-178)
    # NSF acoustic models predict two more channels behind the mel bins -- normalised f0 and a voicing score -- which the
    # vocoder's source module wants in Hz / as a 0-1 flag (reference :26-56, :180-193, :219-220)
    params = config["Model"]["KanTtsSAMBERT"]["params"]
    nsf = None
    if params.get("NSF", False):
        if params.get("nsf_norm_type", "mean_std") == "mean_std":
            mvn = np.load(os.path.join(os.path.dirname(os.path.dirname(ckpt)), "mvn.npy"))  # rows: mean, std of f0
            nsf = (float(np.asarray(mvn[1:]).reshape(-1)[0]), float(np.asarray(mvn[0:1]).reshape(-1)[0]))
        else:  # "global": f0 was mapped to [0, 1] between a global minimum and maximum
            lo, hi = params.get("nsf_f0_global_minimum", 30.0), params.get("nsf_f0_global_maximum", 730.0)
            nsf = (float(hi) - float(lo), float(lo))
    from kantts.models import model_builder

    model, _, _ = model_builder(config, device)
    fsnet = model["KanTtsSAMBERT"]
    logging.info("Loading checkpoint: %s", ckpt)
    fsnet.load_state_dict(torch.load(ckpt, map_location="cpu")["model"], strict=False)
    results_dir = os.path.join(output_dir, "feat")
    os.makedirs(results_dir, exist_ok=True)
    fsnet.eval()
    if device.type == "cuda":
        # bf16 mode: each autoregressive loop is one launch (kantts/models/sambert/ar_kernels.py); otherwise one decoder
        # step = one hipGraph replay (kantts/models/sambert/decode_graph.py)
        fsnet.mel_decoder.decode_mode = "kernel"
    with open(sentence, encoding="utf-8") as f:
        for line in f:
            line = line.strip().split("\t")
            if len(line) < 2:
                continue
            logging.info("Inference sentence: %s", line[0])
            with torch.no_grad():
                _, mel_post, dur, f0, energy = am_synthesis(line[1], fsnet, ling_unit, device, se=se)
            if nsf is not None:
                mel_post = denorm_f0(mel_post, scale=nsf[0], offset=nsf[1])
            np.save("%s/%s_mel.npy" % (results_dir, line[0]), mel_post)
            np.savetxt("%s/%s_dur.txt" % (results_dir, line[0]), dur)
            np.savetxt("%s/%s_f0.txt" % (results_dir, line[0]), f0)
            np.savetxt("%s/%s_energy.txt" % (results_dir, line[0]), energy)


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--sentence", type=str, required=True)
    parser.add_argument("--output_dir", type=str, required=True)
    parser.add_argument("--ckpt", type=str, required=True)
    parser.add_argument("--se_file", type=str, required=False)
    args = parser.parse_args()
    am_infer(args.sentence, args.ckpt, args.output_dir, args.se_file)
