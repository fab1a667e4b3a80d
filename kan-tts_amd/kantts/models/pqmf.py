"""Pseudo-QMF analysis / synthesis filter bank for multi-band generators (reference kantts/models/pqmf.py:14-148),
on the channels-last convolution kernels.

The reference runs analysis as conv1d(pad(x), h_analysis) followed by a one-hot strided conv (decimation), and synthesis
as a one-hot transposed conv (zero-stuffing x subbands) followed by conv1d(pad(.), h_synthesis).  Both pairs collapse:
  analysis   y[k, t] = sum_j h_k[j] x[t*S + j - taps/2]                       -> ONE strided convolution (1 -> S channels)
  synthesis  out[q*S + r] = S * sum_k sum_d g_k[d*S - r + taps/2] z[k, q + d]  -> ONE polyphase convolution over the
             low-rate signal (S -> S channels, output channel r = phase r), no zero-stuffed tensor, no wasted taps.
Buffers (analysis_filter, synthesis_filter, updown_filter) keep the reference's names / shapes for state_dict parity.
"""
import numpy as np
import torch

from kantts._hip import ops


def design_prototype_filter(taps=62, cutoff_ratio=0.142, beta=9.0):
    """Kaiser-window prototype low-pass of length taps + 1 (reference :14-46)."""
    assert taps % 2 == 0, "The number of taps mush be even number."
    assert 0.0 < cutoff_ratio < 1.0, "Cutoff ratio must be > 0.0 and < 1.0."
    n = np.arange(taps + 1) - 0.5 * taps
    omega_c = np.pi * cutoff_ratio
    with np.errstate(invalid="ignore", divide="ignore"):
        h_i = np.sin(omega_c * n) / (np.pi * n)
    h_i[taps // 2] = cutoff_ratio
    return h_i * np.kaiser(taps + 1, beta)


class PQMF(torch.nn.Module):
    def __init__(self, subbands=4, taps=62, cutoff_ratio=0.142, beta=9.0):
        super(PQMF, self).__init__()
        h_proto = design_prototype_filter(taps, cutoff_ratio, beta)
        n = np.arange(taps + 1) - (taps / 2)
        h_analysis = np.zeros((subbands, len(h_proto)))
        h_synthesis = np.zeros((subbands, len(h_proto)))
        for k in range(subbands):
            ph = (2 * k + 1) * (np.pi / (2 * subbands)) * n
            h_analysis[k] = 2 * h_proto * np.cos(ph + (-1) ** k * np.pi / 4)
            h_synthesis[k] = 2 * h_proto * np.cos(ph - (-1) ** k * np.pi / 4)
        self.register_buffer("analysis_filter", torch.from_numpy(h_analysis).float().unsqueeze(1))    # (S, 1, taps+1)
        self.register_buffer("synthesis_filter", torch.from_numpy(h_synthesis).float().unsqueeze(0))  # (1, S, taps+1)
        updown_filter = torch.zeros((subbands, subbands, subbands)).float()
        for k in range(subbands):
            updown_filter[k, k, 0] = 1.0
        self.register_buffer("updown_filter", updown_filter)
        self.subbands, self.taps = subbands, taps
        # polyphase synthesis weights W[r, k, d - d_min] = S * g_k[d*S - r + taps/2]
        S, half = subbands, taps // 2
        self.d_min = -((half + S - 1) // S)  # smallest d with d*S - r + half >= 0 for some r
        d_max = (half + S - 1) // S
        w = torch.zeros(S, S, d_max - self.d_min + 1)
        g = torch.from_numpy(h_synthesis).float()
        for r in range(S):
            for d in range(self.d_min, d_max + 1):
                j = d * S - r + half
                if 0 <= j <= taps:
                    w[r, :, d - self.d_min] = S * g[:, j]
        self.register_buffer("_poly_synthesis", w, persistent=False)

    def analysis(self, x):
        """(B, 1, T) -> (B, subbands, T // subbands)"""
        B, _, T = x.shape
        S = self.subbands
        Tout = (T - S) // S + 1  # conv1d 'same' length T, then stride-S decimation with a kernel of S
        y = ops.conv_cl(x.reshape(B, T, 1), self.analysis_filter, None, stride=S, pad=self.taps // 2, Tout=Tout)
        return y.transpose(1, 2)

    def synthesis(self, x):
        """(B, subbands, T // subbands) -> (B, 1, T)"""
        B, S, L = x.shape
        z = x.transpose(1, 2).contiguous()  # (B, L, S)
        y = ops.conv_cl(z, self._poly_synthesis, None, stride=1, pad=-self.d_min, Tout=L)  # (B, L, S): channel = phase
        return y.reshape(B, 1, L * S)
