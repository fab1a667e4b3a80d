"""kantts -- MI355X-native drop-in for the data-parallel hot path of modelscope/KAN-TTS.

Same import names as the reference (kantts.models, kantts.train.loss, kantts.utils.audio_torch,
kantts.bin.*); the arithmetic runs in libkantts_hip.so (hand-written gfx950 HIP kernels, see
include/kantts_hip.h).  There is no CPU execution path: modules can be constructed and their
state_dict handled on the host, but forward/backward require a HIP device.
"""
__version__ = "0.1.0"

import os as _os

# Checkout of modelscope/KAN-TTS (the directory that holds its ``kantts/``), optional.  The text front-end of the
# reference (ttsfrd: raw text -> symbol sequences), its pitch / energy / duration extraction and its logging / plotting
# helpers are not re-implemented here (out of the hot path); with this variable set, sub-modules of
# kantts.utils / kantts.datasets / kantts.preprocess that this package does not ship resolve from the checkout, and the
# language resource files (PhoneSet.xml, tonelist.txt) of kantts.utils.ling_unit are found there.  Modules that exist here always win; models / train / bin never fall
# back to reference code.
REFERENCE_ROOT = _os.environ.get("KANTTS_REFERENCE_ROOT") or None


def _overlay(name, path):
    """Append the reference checkout's directory for sub-package ``name`` to its ``__path__`` (after this package's)."""
    if not REFERENCE_ROOT:
        return
    cand = _os.path.join(REFERENCE_ROOT, *name.split("."))
    if _os.path.isdir(cand) and cand not in path:
        path.append(cand)
