"""Acoustic-model data side (SURVEY 8 row f2 / dataset row 19): the native symbol tables (kantts.utils.ling_unit) and
AM_Dataset / get_am_datasets (kantts.datasets.dataset) against items, batches and the seeded train / valid split recorded
from the reference (tests/golden/am_dataset.pt, generator: oracle/make_golden.py::am_dataset_case).  The fixture carries
the PinYin phone / tone inventories and every input file, so language and data directories are rebuilt in tmp_path and the
test runs without a reference checkout."""
import os

import numpy as np
import pytest
import torch

from util import ROOT

FIX = os.path.join(ROOT, "tests", "golden", "am_dataset.pt")


def _language_dir(tmp, phones, tone_lines):
    d = os.path.join(tmp, "languages", "PinYin")
    os.makedirs(d)
    with open(os.path.join(d, "PhoneSet.xml"), "w") as f:
        f.write('<?xml version="1.0" encoding="utf-8"?>\n<phoneSet xmlns="http://schemas.alibaba-inc.com/tts">\n')
        for i, p in enumerate(phones):
            f.write("  <phone>\n    <id>%d</id>\n    <name>%s</name>\n    <cv>vowel</cv>\n  </phone>\n" % (i, p))
        f.write("</phoneSet>\n")
    with open(os.path.join(d, "tonelist.txt"), "w") as f:
        f.write("\n".join(tone_lines) + "\n")
    return os.path.dirname(d)


def _data_dir(tmp, fx, only_clean=False):
    """``only_clean``: leave out the utterances whose line holds a symbol outside the inventory -- the sy stream drops it,
    the other streams do not, and a batch whose longest utterance is such a line cannot be padded (here as in the
    reference); real metafiles never contain one."""
    d = os.path.join(tmp, "data_clean" if only_clean else "data")
    for sub in ("mel", "duration", "f0", "energy", "frame_f0", "frame_uv"):
        os.makedirs(os.path.join(d, sub))
    utts = {n: u for n, u in fx["utts"].items() if not (only_clean and "not_a_phone" in u["ling"])}
    for name, u in utts.items():
        for sub, key in (("mel", "mel"), ("duration", "dur"), ("f0", "f0"), ("energy", "energy"),
                         ("frame_f0", "frame_f0"), ("frame_uv", "frame_uv")):
            if sub == "duration" and name == fx["missing_duration"] and not only_clean:
                continue
            np.save(os.path.join(d, sub, name + ".npy"), u[key])
    np.savetxt(os.path.join(d, "f0", "f0_mean.txt"), np.array([fx["f0_mean"]]))
    np.savetxt(os.path.join(d, "f0", "f0_std.txt"), np.array([fx["f0_std"]]))
    with open(os.path.join(d, "raw_metafile.txt"), "w") as f:
        for name in sorted(utts):
            f.write("%s\t%s\n" % (name, utts[name]["ling"]))
    return d


def _same(a, b, what):
    if a is None or b is None:
        assert a is None and b is None, what
    elif torch.is_tensor(a) or torch.is_tensor(b):
        assert torch.is_tensor(a) and torch.is_tensor(b) and a.dtype == b.dtype and a.shape == b.shape, what
        if a.dtype == torch.float64:  # the alignment prior: closed form in log space here, scipy's pmf row by row there
            assert torch.allclose(a, b, rtol=1e-10, atol=1e-300), what
        else:
            assert torch.equal(a, b), what
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), what
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, "%s[%d]" % (what, i))
    else:
        a, b = np.asarray(a), np.asarray(b)
        assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b), what


def test_ling_unit_and_am_dataset_match_the_reference(tmp_path, monkeypatch):
    from kantts.datasets.dataset import AM_Dataset, get_am_datasets
    from kantts.utils.ling_unit import KanTtsLinguisticUnit

    fx = torch.load(FIX, weights_only=False)
    monkeypatch.delenv("KANTTS_LANGUAGE_DIR", raising=False)
    lang = _language_dir(str(tmp_path), fx["phones"], fx["tones_file"])
    data = _data_dir(str(tmp_path), fx)
    unit = dict(fx["unit"], language_dir=lang)

    def config(extra):
        return {"linguistic_unit": dict(unit), "Model": {"KanTtsSAMBERT": {"params": dict(fx["base_params"], **extra)}}}

    lu = KanTtsLinguisticUnit(config({}))
    assert lu.get_unit_size() == fx["unit_size"] and lu._sub_unit_pad == fx["pad_ids"]
    assert not lu.using_byte()
    # the seeded shuffle / split, the bad list and the utterance without a duration file
    tr, va = os.path.join(data, "am_train.lst"), os.path.join(data, "am_valid.lst")
    AM_Dataset.gen_metafile(os.path.join(data, "raw_metafile.txt"), data, tr, va, badlist=fx["badlist"],
                            split_ratio=fx["split_ratio"])
    assert open(tr).read() == fx["split"]["train"] and open(va).read() == fx["split"]["valid"]
    for tag, extra in fx["variants"].items():
        exp = fx["expected"][tag]
        ds = AM_Dataset(config(extra), tr, data, allow_cache=(tag == "plain"))
        assert ds.with_duration == exp["with_duration"] and len(ds) == len(exp["items"])
        items = [ds[i] for i in range(len(ds))]
        for i, (got, ref) in enumerate(zip(items, exp["items"])):
            _same(list(got), list(ref), "%s item %d" % (tag, i))
        batch = ds.collate_fn(items[:5])
        assert set(batch) >= set(k for k in exp["batch"] if exp["batch"][k] is not None)
        for k, v in exp["batch"].items():
            _same(batch.get(k), v, "%s batch[%s]" % (tag, k))
        if tag == "plain":
            assert ds[0] is ds[0]  # cached
    # get_am_datasets finds the lists it needs (and would generate them with the arguments where their names say)
    train_set, valid_set = get_am_datasets(os.path.join(data, "raw_metafile.txt"), data, config({}), False)
    assert len(train_set) == len(fx["expected"]["plain"]["items"]) and len(valid_set) == fx["split"]["valid"].count("\n")


def test_ling_unit_semantics_on_a_toy_inventory(tmp_path):
    """Position-in-table ids, "~" terminator, prosodic marks after the phones, symbols outside the inventory dropped from
    the sy stream only (the other streams keep their length), unknown tone -> KeyError, decode round trip, byte mode."""
    from kantts.utils.ling_unit import KanTtsLinguisticUnit, load_language_symbols

    lang = _language_dir(str(tmp_path), ["aa", "bb", "cc"], ["1", "", "3"])
    phones, tones = load_language_symbols("PinYin", lang)
    assert phones == ["aa", "bb", "cc", "#1", "#2", "#3", "#4"] and tones == ["tone1", "tone_none", "tone3"]
    cfg = {"linguistic_unit": {"cleaners": "x", "speaker_list": "A,B", "language_dir": lang,
                               "lfeat_type_list": "sy,tone,syllable_flag,word_segment,emo_category,speaker_category"},
           "Model": {"KanTtsSAMBERT": {"params": {}}}}
    lu = KanTtsLinguisticUnit(cfg)
    assert lu.get_unit_size() == {"sy": 10, "tone": 6, "syllable_flag": 8, "word_segment": 8, "emotion": 36, "speaker": 5}
    seq = lu.encode_symbol_sequence("{bb$tone3$s_end$word_both$emotion_happy$B} {zz$tone_none$s_none$word_none$emotion_none$A} "
                                    "{#2$tone1$s_begin$word_begin$emotion_none$A}")
    assert [a.tolist() for a in seq] == [[1, 4, 8], [2, 1, 0, 4], [1, 2, 0, 6], [3, 4, 0, 6], [5, 0, 0, 34], [1, 0, 0, 3]]
    assert all(a.dtype == np.int32 for a in seq)
    assert lu.decode_symbol_sequence(seq)[0] == "sy:bb #2 ~"
    with pytest.raises(KeyError):
        lu.encode_symbol_sequence("{aa$tone9$s_end$word_both$emotion_happy$B}")
    byte = KanTtsLinguisticUnit({"linguistic_unit": {"cleaners": "x", "speaker_list": "A",
                                                     "lfeat_type_list": "byte_index,emo_category,speaker_category"}})
    assert byte.using_byte() and byte.get_unit_size() == {"byte_index": 259, "emotion": 36, "speaker": 4}
    assert byte.encode_symbol_sequence("{65$emotion_none$A} {255$emotion_none$A}")[0].tolist() == [65, 255, 257]
    with pytest.raises(FileNotFoundError):
        KanTtsLinguisticUnit({"linguistic_unit": dict(cfg["linguistic_unit"], language_dir=str(tmp_path / "nowhere"),
                                                      language="Klingon")})


def test_train_sambert_cli_on_a_data_directory_emulated(tmp_path, monkeypatch):
    """kantts.bin.train_sambert on a feature directory (raw_metafile.txt + mel / duration / f0 / energy files), no
    --synthetic: native symbol tables size the embeddings, get_am_datasets writes the train / valid lists, the DataLoader
    collates with am_collate, the trainer steps (emulated C ABI) and saves a checkpoint that infer_sambert.am_infer reads
    back with the same symbol tables."""
    from kantts.bin.infer_sambert import am_infer
    from kantts.bin.train_sambert import train as train_am
    from kantts.utils.synthetic import SAMBERT_VOCAB, sambert_16k_config
    from util import emulation

    fx = torch.load(FIX, weights_only=False)
    monkeypatch.delenv("KANTTS_LANGUAGE_DIR", raising=False)
    lang = _language_dir(str(tmp_path), fx["phones"], fx["tones_file"])
    data = _data_dir(str(tmp_path), fx, only_clean=True)
    params = {k: v for k, v in sambert_16k_config(tiny=True).items() if k not in SAMBERT_VOCAB}
    config = {"model_type": "sambert", "linguistic_unit": dict(fx["unit"], language_dir=lang),
              "Model": {"KanTtsSAMBERT": {
                  "params": params,
                  "optimizer": {"type": "Adam", "params": {"lr": 0.001, "betas": [0.9, 0.98], "eps": 1e-9,
                                                           "weight_decay": 0.0}},
                  "scheduler": {"type": "NoamLR", "params": {"warmup_steps": 4000}}}},
              "grad_norm": 1.0, "batch_size": 3, "num_workers": 0, "pin_memory": False, "allow_cache": False,
              "log_interval_steps": 1, "save_interval_steps": 2, "train_max_steps": 3}
    with emulation():
        tr = train_am(config, [data], str(tmp_path / "stage"))
        assert tr.steps >= 3 and os.path.exists(tmp_path / "stage" / "ckpt" / "checkpoint_2.pth")
        assert os.path.exists(os.path.join(data, "am_train.lst")) and os.path.exists(os.path.join(data, "am_valid.lst"))
        built = tr.config["Model"]["KanTtsSAMBERT"]["params"]
        assert built["sy"] == fx["unit_size"]["sy"] and built["speaker"] == fx["unit_size"]["speaker"]
        sent = tmp_path / "sentences.txt"
        name = sorted(n for n, u in fx["utts"].items() if "not_a_phone" not in u["ling"])[0]
        sent.write_text("%s\t%s\n" % (name, fx["utts"][name]["ling"]))
        ckpt = str(tmp_path / "stage" / "ckpt" / "checkpoint_2.pth")
        state = torch.load(ckpt, map_location="cpu", weights_only=False)
        state["model"]["variance_adaptor.duration_predictor.fc.bias"].fill_(1.2)  # three steps in: make it predict frames
        torch.save(state, ckpt)
        am_infer(str(sent), ckpt, str(tmp_path / "out"))
    mel = np.load(tmp_path / "out" / "feat" / (name + "_mel.npy"))
    assert mel.ndim == 2 and mel.shape[1] == 80 and np.isfinite(mel).all()
