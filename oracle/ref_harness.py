"""TEST INFRASTRUCTURE -- imports the *untouched* reference (``/root/reference``) on CPU.

Only usable inside the build container (the GPU box has no ``/root/reference``).  It is used by
``oracle/make_golden.py`` to dump golden vectors and by ``oracle/check_vs_reference.py`` (run by hand
in the build container) to pin the restatements in ``oracle/torch_oracle.py`` / ``hifigan_oracle.py`` live.

The reference imports six third-party packages that are not installed here.
We install minimal ``sys.modules`` stubs for them.  Three of the stubs carry arithmetic
(``librosa.filters.mel``, ``librosa.stft`` and ``pytorch_wavelets.DWT1DForward``); those are served by OUR
restatements in ``oracle/thirdparty.py`` of the packages' published algorithms -- the reference never pins them;
each is pinned by an independent implementation or definition (transformers' mel filterbank, scipy.signal.stft,
numpy.convolve + hand-computed DWT values: tests/test_independent_pins.py, tests/test_thirdparty_kat.py).
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("KANTTS_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "kantts", "models"))


def _install_stubs():
    import scipy.signal
    import scipy.signal.windows

    if not hasattr(scipy.signal, "kaiser"):  # kantts/models/pqmf.py:10
        scipy.signal.kaiser = scipy.signal.windows.kaiser

    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import thirdparty  # oracle/thirdparty.py

    if "librosa" not in sys.modules:  # kantts/utils/audio_torch.py:2
        librosa = types.ModuleType("librosa")
        filters = types.ModuleType("librosa.filters")
        filters.mel = thirdparty.librosa_mel
        librosa.filters = filters
        librosa.stft = thirdparty.librosa_stft  # kantts/preprocess/audio_processor/core/dsp.py:8-9 (scipy-pinned restatement)
        core = types.ModuleType("librosa.core")  # Voc_Dataset.__getitem__ (datasets/dataset.py:238): file IO only
        core.load = thirdparty.wav_load
        librosa.core = core
        librosa.load = thirdparty.wav_load
        sys.modules["librosa"] = librosa
        sys.modules["librosa.filters"] = filters
        sys.modules["librosa.core"] = core

    if "pytorch_wavelets" not in sys.modules:  # kantts/models/hifigan/hifigan.py:7
        pw = types.ModuleType("pytorch_wavelets")
        pw.DWT1DForward = thirdparty.DWT1DForward
        sys.modules["pytorch_wavelets"] = pw

    if "numba" not in sys.modules:  # kantts/models/sambert/alignment.py:2

        def jit(*args, **kwargs):
            if len(args) == 1 and callable(args[0]) and not kwargs:
                return args[0]
            return lambda f: f

        nb = types.ModuleType("numba")
        nb.jit = jit
        nb.prange = range
        sys.modules["numba"] = nb

    # logging / plotting / audio-file IO of the trainer shell (kantts/train/trainer.py:6-11): no arithmetic behind them
    # "sox" / "pysptk": imported at the top of preprocess/audio_processor/core/utils.py (volume normalisation, pitch
    # tracking); nothing on the mel-extraction path calls them
    for name in ("tensorboardX", "soundfile", "matplotlib", "matplotlib.pyplot", "sox", "pysptk"):
        try:
            __import__(name)
        except ImportError:
            m = types.ModuleType(name)
            if name == "tensorboardX":
                m.SummaryWriter = type("SummaryWriter", (), {"__init__": lambda self, *a, **k: None,
                                                             "add_scalar": lambda self, *a, **k: None,
                                                             "add_figure": lambda self, *a, **k: None})
            if name == "matplotlib":
                m.use = lambda *a, **k: None
            sys.modules[name] = m
    if "matplotlib.pyplot" in sys.modules and not hasattr(sys.modules["matplotlib"], "pyplot"):
        sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]

    if "ttsfrd" not in sys.modules:
        sys.modules["ttsfrd"] = types.ModuleType("ttsfrd")
    if "unidecode" not in sys.modules:
        u = types.ModuleType("unidecode")
        u.unidecode = lambda s: s
        sys.modules["unidecode"] = u
    if "inflect" not in sys.modules:
        inf = types.ModuleType("inflect")

        class _E:
            def number_to_words(self, n, **kw):
                return str(n)

        inf.engine = _E
        sys.modules["inflect"] = inf


def import_reference():
    """Return the reference ``kantts`` package (CPU).  Refuses if our drop-in ``kantts`` is loaded."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    mod = sys.modules.get("kantts")
    if mod is not None and not os.path.abspath(mod.__file__).startswith(REFERENCE_ROOT):
        raise RuntimeError(
            "a non-reference 'kantts' package is already imported in this process; "
            "run the reference harness in its own process"
        )
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import warnings

    warnings.filterwarnings("ignore")
    import kantts.models  # noqa: F401
    import kantts.train.loss  # noqa: F401
    import kantts.utils.audio_torch  # noqa: F401

    return sys.modules["kantts"]
