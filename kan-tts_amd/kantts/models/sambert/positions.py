"""Position encoders (reference kantts/models/sambert/positions.py:8-98)."""
import numpy as np
import torch
import torch.nn as nn


class SinusoidalPositionEncoder(nn.Module):
    """Sinusoid table kept as a frozen Parameter (it is part of the reference state_dict).
    The table add itself is fused into the embedding gather kernel (TextFftEncoder); ``forward``
    is kept for API parity and only adds the table."""

    def __init__(self, max_len, depth):
        super(SinusoidalPositionEncoder, self).__init__()
        self.max_len = max_len
        self.depth = depth
        self.position_enc = nn.Parameter(
            self.get_sinusoid_encoding_table(max_len, depth).unsqueeze(0), requires_grad=False)

    def table_for(self, length, device):
        """(length, depth) table, regrown like the reference does when length > max_len (:21-27)."""
        if length > self.max_len:
            self.max_len = length
            self.position_enc.data = self.get_sinusoid_encoding_table(self.max_len, self.depth).unsqueeze(0).to(device)
        return self.position_enc[0]

    def forward(self, input):
        return input + self.table_for(input.size(1), input.device)[None, : input.size(1), :]

    @staticmethod
    def get_sinusoid_encoding_table(n_position, d_hid, padding_idx=None):
        """table[pos, j] = sin((pos+1) / 10000^(j/(d/2-1))) for j < d/2, cos(...) for the second half
        (reference :33-55: 1-based positions, divisor d_hid/2 - 1)."""
        pos = np.arange(1, n_position + 1, dtype=np.float64)[:, None]
        j = np.arange(d_hid // 2, dtype=np.float64)[None, :]
        angle = pos / np.power(10000, j / float(d_hid / 2 - 1))
        table = np.zeros((n_position, d_hid))
        table[:, : d_hid // 2] = np.sin(angle)
        table[:, d_hid // 2:] = np.cos(angle)
        if padding_idx is not None:
            table[padding_idx] = 0.0
        return torch.FloatTensor(table)


class DurSinusoidalPositionEncoder(nn.Module):
    """Duration-relative sinusoid: sin/cos of (frame index inside its phone) / 10000^(2*(j//2)/depth)
    on even / odd channels (reference :58-98).  The within-phone position comes from the
    length-regulator index kernel (kantts_lr_index), not from a dense one-hot matmul."""

    def __init__(self, depth, outputs_per_step):
        super(DurSinusoidalPositionEncoder, self).__init__()
        self.depth = depth
        self.outputs_per_step = outputs_per_step
        inv_timescales = [np.power(10000, 2 * (hid_idx // 2) / depth) for hid_idx in range(depth)]
        self.inv_timescales = nn.Parameter(torch.FloatTensor(inv_timescales), requires_grad=False)

    def from_positions(self, dur_pos):
        """dur_pos (B, Tp) already masked / padded -> (B, Tp, depth)."""
        e = dur_pos[:, :, None] / self.inv_timescales[None, None, :]
        even = torch.arange(self.depth, device=e.device) % 2 == 0
        return torch.where(even[None, None, :], torch.sin(e), torch.cos(e))

    def forward(self, durations, masks=None):
        from kantts._hip import ops

        reps_total = int((durations + 0.5).long().sum(dim=1).max().item())
        r = self.outputs_per_step
        Tp = reps_total + ((r - reps_total % r) % r)
        _, pos, _, _ = ops.lr_index(durations, Tp)
        t = torch.arange(Tp, device=pos.device)[None, :]
        pos = torch.where(t < reps_total, pos, torch.zeros_like(pos))
        if masks is not None:
            pos = pos.masked_fill(torch.nn.functional.pad(masks, (0, Tp - masks.size(1)), value=True), 0.0)
        return self.from_positions(pos)
