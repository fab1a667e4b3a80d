"""audio_torch (mel-STFT on the HIP kernels), synthetic workloads; ling_unit / logging helpers resolve from a reference
checkout when KANTTS_REFERENCE_ROOT is set (see kantts/__init__.py)."""
from kantts import _overlay

_overlay(__name__, __path__)
