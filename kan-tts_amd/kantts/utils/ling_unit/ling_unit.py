"""Symbol sequence -> id streams (reference kantts/utils/ling_unit/ling_unit.py:56-398, lang_symbols.py:30-88).

One metafile / inference line is a space-separated sequence of ``{sy$tone$syllable_flag$word_segment$emotion$speaker}``
groups; the model consumes one integer stream per field, each terminated by the id of ``~``.  Every stream is a table
``symbols + [_ , ~ , @[MASK]]`` (pad, end, mask); the ids are positions in that table, so they depend only on the order of
the language's phone / tone inventory:

    sy             "@" + every <name> of PhoneSet.xml in file order, then "@#1" .. "@#4"     (lang_symbols.py:30-47)
    tone           "tone" + every non-empty line of tonelist.txt, "tone_none" for an empty one (lang_symbols.py:50-67)
    syllable_flag  s_begin, s_end, s_none, s_both, s_middle
    word_segment   word_begin, word_end, word_middle, word_both, word_none
    emo_category   the 33 emotion names                                                        (emotion_types.py)
    speaker_category  ``linguistic_unit.speaker_list`` of the yaml
    byte_index     "@0" .. "@255"                                                              (byte-input models)

Differences in structure, not in results: the reference rebuilds a brace-wrapped string for the ``sy`` stream and runs it
through its text cleaners / a regular expression; every symbol of a metafile line sits inside braces, so that path
reduces to a table lookup that silently drops symbols missing from the inventory (``should_keep_sy``), which is what is
done here directly.  ``tests/test_am_dataset.py`` pins the ids to the reference's on recorded lines.
"""
import os
import xml.etree.ElementTree as ET

import numpy as np

SYLLABLE_FLAGS = ["s_begin", "s_end", "s_none", "s_both", "s_middle"]
WORD_SEGMENTS = ["word_begin", "word_end", "word_middle", "word_both", "word_none"]
EMOTION_TYPES = (["emotion_" + n for n in (
    "none neutral angry disgust fear happy sad surprise calm gentle relax lyrical serious disgruntled satisfied "
    "disappointed excited anxiety jealousy hate pity pleasure arousal dominance").split()]
    + ["emotion_placeholder%d" % i for i in range(1, 10)])
PAD, EOS, MASK = "_", "~", "@[MASK]"
_PHONESET, _TONELIST = "PhoneSet.xml", "tonelist.txt"


def language_directory(language, language_dir=None):
    """Directory holding PhoneSet.xml / tonelist.txt of ``language`` ("PinYin", "ZhHK", "Sichuan", "WuuShanghai", ...).
    Looked up, in order: the argument (``linguistic_unit.language_dir`` of the yaml), $KANTTS_LANGUAGE_DIR,
    ``kantts/preprocess/languages`` of this package.  Each candidate may be the language's own directory or the directory
    of all languages."""
    here = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "preprocess", "languages"))
    cands = [language_dir, os.environ.get("KANTTS_LANGUAGE_DIR"), here]
    tried = []
    for c in cands:
        if not c:
            continue
        for d in (c, os.path.join(c, language)):
            tried.append(d)
            if os.path.isfile(os.path.join(d, _PHONESET)) and os.path.isfile(os.path.join(d, _TONELIST)):
                return d
    raise FileNotFoundError(
        "no %s / %s for language %r (looked in %s): set linguistic_unit.language_dir in the yaml or "
        "KANTTS_LANGUAGE_DIR" % (_PHONESET, _TONELIST, language, ", ".join(tried) or "nowhere"))


def load_language_symbols(language="PinYin", language_dir=None):
    """(phones, tones): the phone names of PhoneSet.xml in file order plus the four prosodic boundary marks, and the tone
    names of tonelist.txt (one per line; an empty line is "tone_none")."""
    d = language_directory(language, language_dir)
    phones = []
    for node in ET.parse(os.path.join(d, _PHONESET)).getroot():
        if node.tag.rsplit("}", 1)[-1] != "phone":
            continue
        name = next((c.text for c in node if c.tag.rsplit("}", 1)[-1] == "name"), None)
        phones.append(name)
    phones += ["#%d" % i for i in range(1, 5)]
    with open(os.path.join(d, _TONELIST), "r") as f:
        tones = ["tone" + ln.strip() if ln.strip() else "tone_none" for ln in f.readlines()]
    return phones, tones


class _Stream:
    """One id table: position of a symbol in ``symbols + [pad, eos, mask]``."""

    def __init__(self, symbols, drop_unknown=False):
        self.symbols = list(symbols) + [PAD, EOS, MASK]
        self.ids = {s: i for i, s in enumerate(self.symbols)}
        self.pad, self.eos = self.ids[PAD], self.ids[EOS]
        self.drop_unknown = drop_unknown

    def __len__(self):
        return len(self.symbols)

    def encode(self, tokens):
        if self.drop_unknown:  # the sy stream: symbols outside the inventory (and the specials) vanish
            out = [self.ids[t] for t in tokens if t in self.ids and t != PAD and t != EOS]
        else:
            out = [self.ids[t] for t in tokens]  # KeyError on an unknown tone / flag / emotion / speaker, as in the reference
        out.append(self.eos)
        return np.asarray(out, dtype=np.int32)


class KanTtsLinguisticUnit:
    """config: the model yaml (``linguistic_unit`` section: ``lfeat_type_list``, ``speaker_list``, optional ``language``,
    ``language_dir``; ``cleaners`` is accepted and unused -- symbol sequences never reach a text cleaner)."""

    def __init__(self, config, language_dir=None):
        unit = config["linguistic_unit"]
        self.unit_config = unit
        self.lang_type = unit.get("language", "PinYin")
        self._pad, self._eos, self._mask = PAD, EOS, MASK
        self._lfeat_type_list = [t.strip() for t in unit["lfeat_type_list"].strip().split(",")]
        self.fp_enable = bool(config.get("Model", {}).get("KanTtsSAMBERT", {}).get("params", {}).get("FP", False))
        self._streams = {}
        if self.using_byte():
            self._streams["byte_index"] = _Stream(["@%d" % i for i in range(256)])
        else:
            phones, tones = load_language_symbols(self.lang_type, language_dir or unit.get("language_dir"))
            self.lang_phones, self.lang_tones = phones, tones
            self._streams["sy"] = _Stream(["@" + p for p in phones], drop_unknown=True)
            self._streams["tone"] = _Stream(tones)
            self._streams["syllable_flag"] = _Stream(SYLLABLE_FLAGS)
            self._streams["word_segment"] = _Stream(WORD_SEGMENTS)
        if "emo_category" in self._lfeat_type_list:
            self._streams["emo_category"] = _Stream(EMOTION_TYPES)
        if "speaker_category" in self._lfeat_type_list:
            self._streams["speaker_category"] = _Stream([s.strip() for s in unit["speaker_list"].strip().split(",")])
        unknown = [t for t in self._lfeat_type_list if t not in self._streams]
        if unknown:
            raise ValueError("unknown lfeat type(s): %s" % ", ".join(unknown))
        # the attribute names the reference's dataset / inference code reads
        self._sub_unit_dim = {k: len(v) for k, v in self._streams.items()}
        self._sub_unit_pad = {k: v.pad for k, v in self._streams.items()}

    def using_byte(self):
        return "byte_index" in self._lfeat_type_list

    def get_unit_size(self):
        """Embedding-table sizes as the model constructor expects them (kantts_sambert.py:260-275)."""
        names = {"emo_category": "emotion", "speaker_category": "speaker"}
        keys = ["byte_index"] if self.using_byte() else ["sy", "tone", "syllable_flag", "word_segment"]
        keys += [k for k in ("emo_category", "speaker_category") if k in self._streams]
        return {names.get(k, k): len(self._streams[k]) for k in keys}

    def encode_symbol_sequence(self, lfeat_symbol):
        """"{a$t$f$w$e$s} {..}" -> one int32 array per entry of ``lfeat_type_list`` (each ends with the id of "~")."""
        groups = [g.strip("{").strip("}").split("$") for g in lfeat_symbol.strip().split(" ")]
        out = []
        for k, kind in enumerate(self._lfeat_type_list):
            tokens = [g[k] for g in groups]
            if kind in ("sy", "byte_index"):
                tokens = ["@" + t for t in tokens]
            out.append(self._streams[kind].encode(tokens))
        return out

    def decode_symbol_sequence(self, sequence):
        """Inverse for display: one "type:symbols" string per stream."""
        res = []
        for kind, ids in zip(self._lfeat_type_list, sequence):
            syms = [self._streams[kind].symbols[int(i)] for i in np.asarray(ids).reshape(-1)]
            if kind in ("sy", "byte_index"):
                syms = [s[1:] if len(s) > 1 and s[0] == "@" else s for s in syms]
            res.append("%s:%s" % (kind, " ".join(syms)))
        return res
