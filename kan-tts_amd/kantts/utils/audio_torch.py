"""mel-STFT feature extractor on the fused HIP kernel (reference kantts/utils/audio_torch.py).

``MelSpectrogram`` / ``stft`` keep the reference signatures.  The Slaney mel basis (librosa 0.9.2
``filters.mel`` in the reference, an un-vendored dependency) is built here from its published
definition; the kernel consumes it in support form (start, length, packed weights per filter).
"""
import math

import numpy as np
import os

import torch

from kantts._hip import check, lib, ptr, stream


def slaney_mel_basis(sr, n_fft, n_mels=80, fmin=0.0, fmax=None):
    """(n_mels, 1 + n_fft//2) float32 triangular filters on the Slaney mel scale, area-normalised
    (librosa.filters.mel(htk=False, norm='slaney') -- call sites audio_torch.py:125-131, dsp.py:135-139)."""
    fmax = float(sr) / 2 if fmax is None else float(fmax)
    f_sp, brk = 200.0 / 3.0, 1000.0
    logstep = math.log(6.4) / 27.0

    def to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= brk, brk / f_sp + np.log(np.maximum(f, 1e-30) / brk) / logstep, f / f_sp)

    def to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= brk / f_sp, brk * np.exp(logstep * (m - brk / f_sp)), f_sp * m)

    bins = np.linspace(0.0, float(sr) / 2, 1 + n_fft // 2)
    edges = to_hz(np.linspace(to_mel(fmin), to_mel(fmax), n_mels + 2))
    width = np.diff(edges)
    rise = (bins[None, :] - edges[:-2, None]) / width[:-1, None]
    fall = (edges[2:, None] - bins[None, :]) / width[1:, None]
    tri = np.maximum(0.0, np.minimum(rise, fall)).astype(np.float32)
    tri *= (2.0 / (edges[2:] - edges[:-2]))[:, None].astype(np.float32)
    return tri


def _support_form(melmat_t):
    """melmat_t: (n_freq, n_mels) tensor -> int32 start/len/off + packed weights (host tensors)."""
    m = melmat_t.t().contiguous().cpu().numpy()
    starts, lens, offs, packed = [], [], [], []
    for row in m:
        nz = np.nonzero(row)[0]
        s, e = (int(nz[0]), int(nz[-1]) + 1) if nz.size else (0, 0)
        starts.append(s)
        lens.append(e - s)
        offs.append(len(packed))
        packed.extend(row[s:e].tolist())
    i32 = lambda v: torch.tensor(v, dtype=torch.int32)  # noqa: E731
    return i32(starts), i32(lens), i32(offs), torch.tensor(packed or [0.0], dtype=torch.float32)


_const_cache = {}


def _fft_consts(n_fft, win_length, window, device):
    key = (n_fft, win_length, window, str(device))
    c = _const_cache.get(key)
    if c is None:
        if window is None:
            w = torch.ones(win_length, dtype=torch.float32)
        else:
            w = getattr(torch, "%s_window" % window)(win_length, dtype=torch.float32)
        left = (n_fft - win_length) // 2
        wpad = torch.zeros(n_fft, dtype=torch.float32)
        wpad[left:left + win_length] = w
        t = np.arange(n_fft // 2, dtype=np.float64)
        tw = np.stack([np.cos(-2 * np.pi * t / n_fft), np.sin(-2 * np.pi * t / n_fft)], -1).astype(np.float32)
        c = (wpad.to(device), torch.from_numpy(tw).contiguous().to(device))
        _const_cache[key] = c
    return c


class _MelSpecFn(torch.autograd.Function):
    """Differentiable mel path (forward + kantts_melspec_bwd); the magnitude output has no backward yet."""

    @staticmethod
    def forward(ctx, x, cfg, ms, ml, mo, mw):
        n_fft, hop, win_length, window, pad_mode, eps_power, eps_mel = cfg
        out_mel, _ = _launch(x.detach(), n_fft, hop, win_length, window, pad_mode, eps_power, mel=(ms, ml, mo, mw),
                             eps_mel=eps_mel)
        ctx.cfg = cfg
        ctx.save_for_backward(x, ms, ml, mo, mw)
        return out_mel

    @staticmethod
    def backward(ctx, dmel):
        x, ms, ml, mo, mw = ctx.saved_tensors
        n_fft, hop, win_length, window, pad_mode, eps_power, eps_mel = ctx.cfg
        x = x.contiguous().float()
        B, T = x.shape
        frames = 1 + T // hop
        wpad, tw = _fft_consts(n_fft, win_length, window, x.device)
        dwav = torch.zeros_like(x)
        # the forward handed out a transposed view of its frame-major buffer: a gradient that keeps that layout (the
        # L1 / MSE criteria do) is read as it is, anything else is copied once
        dmel_fm = dmel.transpose(1, 2).contiguous()
        check(lib().kantts_melspec_bwd_fm(ptr(x, torch.float32), ptr(dmel_fm, torch.float32), B, T, n_fft, hop,
                                          frames, pad_mode, ptr(wpad), ptr(tw), float(eps_power), ptr(ms), ptr(ml), ptr(mo),
                                          ptr(mw), ms.numel(), float(eps_mel), 1, ptr(dwav), stream()), "melspec_bwd_fm")
        return dwav, None, None, None, None, None


class _StftMagFn(torch.autograd.Function):
    """Differentiable |STFT| (kantts_melspec_fwd's magnitude output + kantts_stft_mag_bwd): reference stft(),
    kantts/utils/audio_torch.py:8-31, as used by STFTLoss (kantts/train/loss.py:356-396)."""

    @staticmethod
    def forward(ctx, x, cfg):
        n_fft, hop, win_length, window, pad_mode, eps_power = cfg
        _, mag = _launch(x.detach(), n_fft, hop, win_length, window, pad_mode, eps_power, want_mag=True)
        ctx.cfg = cfg
        ctx.save_for_backward(x)
        return mag

    @staticmethod
    def backward(ctx, dmag):
        (x,) = ctx.saved_tensors
        n_fft, hop, win_length, window, pad_mode, eps_power = ctx.cfg
        x = x.contiguous().float()
        B, T = x.shape
        frames = 1 + T // hop
        wpad, tw = _fft_consts(n_fft, win_length, window, x.device)
        dwav = torch.zeros_like(x)
        check(lib().kantts_stft_mag_bwd(ptr(x, torch.float32), ptr(dmag.contiguous(), torch.float32), B, T, n_fft, hop,
                                        frames, pad_mode, ptr(wpad), ptr(tw), float(eps_power), ptr(dwav), stream()),
              "stft_mag_bwd")
        return dwav, None


_tuning = [None]


def _apply_tuning():
    """KANTTS_MEL_WGS (grid cap of the register-resident kernel) / KANTTS_MEL_GENERIC (radix-2 kernel only): sweep and test
    switches, read HERE (the host layer) and handed to the library through kantts_melspec_tuning when they change -- the
    C ABI itself consults no environment."""
    L = lib()
    want = (id(L), int(os.environ.get("KANTTS_MEL_WGS", "0") or 0), int(bool(os.environ.get("KANTTS_MEL_GENERIC"))))
    if want != _tuning[0]:
        check(L.kantts_melspec_tuning(want[1], want[2]), "melspec_tuning")
        _tuning[0] = want


def _launch(x, n_fft, hop, win_length, window, pad_mode, eps_power, mel=None, eps_mel=0.0, want_mag=False, norm=None):
    """norm: optional (ref_level_db, min_level_db, max_norm, symmetric) for the dB normalisation (forward only)."""
    if x.requires_grad:
        if want_mag and mel is None and norm is None:
            return None, _StftMagFn.apply(x, (n_fft, hop, win_length, window, pad_mode, eps_power))
        if want_mag or mel is None or norm is not None:
            raise NotImplementedError("only the mel path and the plain magnitude are differentiable")
        return _MelSpecFn.apply(x, (n_fft, hop, win_length, window, pad_mode, eps_power, eps_mel), *mel), None
    x = x.contiguous().float()
    B, T = x.shape
    frames = 1 + T // hop
    wpad, tw = _fft_consts(n_fft, win_length, window, x.device)
    out_mel = out_mag = None
    ms = ml = mo = mw = None
    n_mels = 0
    if mel is not None:
        ms, ml, mo, mw = mel
        n_mels = ms.numel()
        # frame-major buffer (a frame's channels are one contiguous store), handed out in the reference's
        # (B, n_mels, frames) shape as a transposed view
        out_mel = torch.empty((B, frames, n_mels), device=x.device, dtype=torch.float32)
    if want_mag:
        out_mag = torch.empty((B, frames, n_fft // 2 + 1), device=x.device, dtype=torch.float32)
    # MelSpectrogram.forward's fixed normalisation (ref 20 dB, floor -100 dB, symmetric +-4) unless the caller names one
    ref_db, min_db, max_norm, symmetric = norm if norm is not None else (20.0, -100.0, 4.0, True)
    _apply_tuning()
    check(lib().kantts_melspec_norm_fwd_fm(ptr(x, torch.float32), B, T, n_fft, hop, frames, pad_mode, ptr(wpad), ptr(tw),
                                           float(eps_power), ptr(ms), ptr(ml), ptr(mo), ptr(mw), n_mels, float(eps_mel),
                                           float(ref_db), float(min_db), float(max_norm), int(bool(symmetric)), 1,
                                           ptr(out_mel), ptr(out_mag), stream()), "melspec_norm_fwd_fm")
    return (None if out_mel is None else out_mel.transpose(1, 2)), out_mag


_dft_cache = {}


def _stft_dft_gemm(x, fft_size, hop_size, win_length, window_name):
    """|STFT| for FFT sizes that are not powers of two (the sub-band STFT loss resolutions 384 / 683 / 171 of
    multi-band configs): the DFT is a dense contraction frames (.., n_fft) @ [cos | -sin] (n_fft, 2 * bins) on the
    MFMA GEMM (differentiable through ops.linear); framing / windowing / magnitude are elementwise device ops."""
    import torch.nn.functional as F

    from kantts._hip import ops

    nb = fft_size // 2 + 1
    key = (fft_size, win_length, window_name, str(x.device))
    c = _dft_cache.get(key)
    if c is None:
        w = getattr(torch, "%s_window" % window_name)(win_length, dtype=torch.float32)
        left = (fft_size - win_length) // 2
        wpad = torch.zeros(fft_size, dtype=torch.float32)
        wpad[left:left + win_length] = w
        ang = 2 * np.pi * np.outer(np.arange(nb, dtype=np.float64), np.arange(fft_size, dtype=np.float64)) / fft_size
        basis = torch.from_numpy(np.concatenate([np.cos(ang), -np.sin(ang)], 0).astype(np.float32))  # (2 nb, n_fft)
        c = (wpad.to(x.device), basis.to(x.device))
        _dft_cache[key] = c
    wpad, basis = c
    xp = F.pad(x[:, None, :], (fft_size // 2, fft_size // 2), mode="reflect")[:, 0]
    fr = (xp.unfold(1, fft_size, hop_size) * wpad).contiguous()
    spec = ops.linear(fr, basis)
    re, im = spec[..., :nb], spec[..., nb:]
    return torch.sqrt(torch.clamp(re * re + im * im, min=1e-7))


_hann_checked = set()


def stft(x, fft_size, hop_size, win_length, window):
    """|STFT| (B, frames, fft_size//2+1) with clamp(re^2+im^2, 1e-7) (reference :8-31), differentiable.  ``window`` is a
    name like "hann" / "hann_window", or a window TENSOR as the reference's callers pass: the kernels build their window
    from the name, so a tensor is accepted only if it IS the periodic Hann window of ``win_length`` (every shipped config;
    anything else raises instead of being silently replaced)."""
    if isinstance(window, str):
        name = window.replace("_window", "")
    else:
        w = torch.as_tensor(window)
        # validated ONCE per (buffer, version, length): the comparison copies the window to the host -- a sync per call,
        # and an aborted capture when a caller passes its registered ``window`` buffer inside a captured step
        key = (w.data_ptr(), w._version, int(w.numel()), int(win_length), str(w.device))
        if key not in _hann_checked:
            if torch.cuda.is_available() and w.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("stft(): a window tensor must have been seen once outside hipGraph capture (or pass "
                                   "the window by name)")
            if w.numel() != win_length or not torch.allclose(
                    w.detach().float().cpu(), torch.hann_window(win_length, dtype=torch.float32), atol=1e-6):
                raise NotImplementedError("stft(): only the Hann window (by name, or the hann_window(win_length) tensor)")
            if len(_hann_checked) > 64:
                _hann_checked.clear()
            _hann_checked.add(key)
        name = "hann"
    if fft_size & (fft_size - 1):
        return _stft_dft_gemm(x, fft_size, hop_size, win_length, name)
    _, mag = _launch(x, fft_size, hop_size, win_length, name, 1, 1e-7, want_mag=True)
    return mag


class MelSpectrogram(torch.nn.Module):
    """Normalised log-mel spectrogram (B, n_mels, frames) -- reference :86-186 (log_base is ignored there too)."""

    def __init__(self, fs=22050, fft_size=1024, hop_size=256, win_length=None, window="hann", num_mels=80, fmin=80,
                 fmax=7600, center=True, normalized=False, onesided=True, eps=1e-10, log_base=10.0,
                 pad_mode="constant"):
        super().__init__()
        if not center or normalized or not onesided:
            raise NotImplementedError("only center=True, normalized=False, onesided=True (the shipped settings)")
        if window is not None and not hasattr(torch, f"{window}_window"):
            raise ValueError(f"{window} window is not implemented")
        self.fft_size = fft_size
        self.win_length = fft_size if win_length is None else win_length
        self.hop_size = hop_size
        self.center, self.normalized, self.onesided = center, normalized, onesided
        self.window = window
        self.eps = eps
        self.pad_mode = pad_mode
        self.log_base = log_base
        fmin = 0 if fmin is None else fmin
        fmax = fs / 2 if fmax is None else fmax
        melmat = slaney_mel_basis(sr=fs, n_fft=fft_size, n_mels=num_mels, fmin=fmin, fmax=fmax)
        self.register_buffer("melmat", torch.from_numpy(melmat.T.copy()).float())
        self._support = None

    def _mel_support(self, device):
        if self._support is None or self._support[0].device != device:
            self._support = tuple(t.to(device) for t in _support_form(self.melmat))
        return self._support

    def forward(self, x):
        if x.dim() == 3:
            x = x.reshape(-1, x.size(2))
        pad = {"constant": 0, "reflect": 1}[self.pad_mode]
        mel, _ = _launch(x, self.fft_size, self.hop_size, self.win_length, self.window, pad, self.eps,
                         mel=self._mel_support(x.device), eps_mel=self.eps)
        return mel
