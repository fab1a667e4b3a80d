"""audio_torch (mel-STFT on the HIP kernels), ling_unit (symbol tables of the acoustic model), synthetic workloads; the
logging / plotting helpers of the reference resolve from a checkout when KANTTS_REFERENCE_ROOT is set (kantts/__init__.py)."""
from kantts import _overlay

_overlay(__name__, __path__)
