"""mel-STFT forward at the benchmark size and at a saturating size; KANTTS_MELSPEC_V1=1 selects the round-1 kernel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch

import audio_oracle as A
from kantts.utils.audio_torch import MelSpectrogram


def ev(fn, n):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


ms = MelSpectrogram().cuda()
ms16 = MelSpectrogram(fs=16000, fft_size=2048, hop_size=200, win_length=1000, fmin=0, fmax=8000).cuda()
with torch.no_grad():
    for B in (32, 2048):
        x = torch.randn(B, 8192, device="cuda") * 0.1
        frames = B * 33
        t = ev(lambda: ms(x), 20)
        err = float((ms(x[:8]).cpu() - A.mel_spectrogram(x[:8].cpu())).abs().max())
        print("n_fft 1024  B %5d  %7d frames  %8.1f us  %7.1f GB/s algorithmic  frac %.4f  max err vs oracle %.2e" % (
            B, frames, t, frames * 1344.0 / t / 1e3, frames * 1344.0 / t / 1e3 / 8000, err))
    x = torch.randn(1024, 9600, device="cuda") * 0.1
    t = ev(lambda: ms16(x), 10)
    err = float((ms16(x[:4]).cpu() - A.mel_spectrogram(x[:4].cpu(), fs=16000, fft_size=2048, hop_size=200, win_length=1000,
                                                       fmin=0, fmax=8000)).abs().max())
    print("n_fft 2048  B  1024  %7d frames  %8.1f us  max err vs oracle %.2e" % (1024 * 49, t, err))
