"""kantts -- MI355X-native drop-in for the data-parallel hot path of modelscope/KAN-TTS.

Same import names as the reference (kantts.models, kantts.train.loss, kantts.utils.audio_torch,
kantts.bin.*); the arithmetic runs in libkantts_hip.so (hand-written gfx950 HIP kernels, see
include/kantts_hip.h).  There is no CPU execution path: modules can be constructed and their
state_dict handled on the host, but forward/backward require a HIP device.
"""
__version__ = "0.1.0"
