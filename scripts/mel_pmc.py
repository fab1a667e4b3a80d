"""One mel-STFT launch at the saturating size (2048 x 8192 samples) for rocprofv3 --pmc runs.  Usage: python scripts/mel_pmc.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
import torch  # noqa: E402

from kantts.utils.audio_torch import MelSpectrogram  # noqa: E402

ms = MelSpectrogram().cuda()
x = torch.randn(2048, 8192, device="cuda") * 0.1
with torch.no_grad():
    for _ in range(3):
        ms(x)
torch.cuda.synchronize()
