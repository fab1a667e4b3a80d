"""mel-STFT forward: register-resident FFT kernel (n_fft 1024 / 2048) against the radix-2 LDS kernel (KANTTS_MEL_GENERIC=1),
same box, event-timed.  Usage (GPU box): python scripts/mel_bench.py   (MEL_SWEEP=1: persistent-grid sweep as well)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
import torch  # noqa: E402

from kantts.utils.audio_torch import MelSpectrogram, stft  # noqa: E402


def ev_ms(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    for name, kw, hop in (("n_fft 1024 hop 256", dict(), 256),
                          ("n_fft 2048 hop 200 (16 kHz)", dict(fs=16000, fft_size=2048, hop_size=200, win_length=1000, fmin=0, fmax=8000), 200)):
        ms = MelSpectrogram(**kw).cuda()
        for B in (32, 2048):
            x = torch.randn(B, 8192, device="cuda") * 0.1
            frames = B * (1 + 8192 // hop)
            row = []
            with torch.no_grad():
                for env in ({}, {"KANTTS_MEL_GENERIC": "1"}):
                    os.environ.update(env)
                    t = ev_ms(lambda: ms(x), 20 if B == 32 else 10)
                    for k in env:
                        os.environ.pop(k)
                    row.append(t)
                alg = frames * (hop * 4 + 80 * 4)
                print("%-28s B=%4d frames=%6d  register %.1f us (%.0f GB/s, frac %.3f)   generic %.1f us (%.0f GB/s)" % (
                    name, B, frames, row[0] * 1e3, alg / row[0] / 1e6, alg / row[0] / 1e6 / 8000, row[1] * 1e3, alg / row[1] / 1e6))
                if B == 2048:
                    a = ms(x[:64])
                    os.environ["KANTTS_MEL_GENERIC"] = "1"
                    b = ms(x[:64])
                    os.environ.pop("KANTTS_MEL_GENERIC")
                    print("    max |register - generic| = %.2e" % float((a - b).abs().max()))
    x = torch.randn(256, 8192, device="cuda") * 0.1
    for n_fft, hop, win in ((1024, 120, 600), (2048, 240, 1200), (512, 50, 240)):
        row = []
        with torch.no_grad():
            for env in ({}, {"KANTTS_MEL_GENERIC": "1"}):
                os.environ.update(env)
                row.append(ev_ms(lambda: stft(x, n_fft, hop, win, "hann"), 5))
                for k in env:
                    os.environ.pop(k)
        print("|STFT| n_fft %4d hop %3d, 256 x 8192: register %.1f us  generic %.1f us" % (n_fft, hop, row[0] * 1e3, row[1] * 1e3))
    if os.environ.get("MEL_SWEEP"):
        ms = MelSpectrogram().cuda()
        x = torch.randn(2048, 8192, device="cuda") * 0.1
        with torch.no_grad():
            for wgs in (256, 512, 768, 1024, 1536, 2048, 4608):
                os.environ["KANTTS_MEL_WGS"] = str(wgs)
                print("  workgroups %5d: %.1f us" % (wgs, ev_ms(lambda: ms(x), 10) * 1e3))


if __name__ == "__main__":
    main()
