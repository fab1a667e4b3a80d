"""Batch assembly (batching.py) and the vocoder dataset (dataset.py); the acoustic-model dataset comes from a reference
checkout when KANTTS_REFERENCE_ROOT is set (see kantts/__init__.py)."""
from kantts import _overlay

_overlay(__name__, __path__)
