"""Offline feature extraction that runs on the device (audio_processor); the text side resolves from a reference checkout
when KANTTS_REFERENCE_ROOT is set (see kantts/__init__.py)."""
from kantts import _overlay

_overlay(__name__, __path__)
