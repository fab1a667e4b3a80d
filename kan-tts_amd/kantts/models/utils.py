"""Mask / length helpers (reference: kantts/models/utils.py:7-23)."""
import torch


def init_weights(m, mean=0.0, std=0.01):
    """N(0, 0.01) init of every Conv* weight (reference kantts/models/utils.py:7-10)."""
    classname = m.__class__.__name__
    if classname.find("Conv") != -1:
        m.weight.data.normal_(mean, std)


def get_mask_from_lengths(lengths, max_len=None):
    """True = padded position (reference kantts/models/utils.py:13-23). Bit-exact (integer compare)."""
    if max_len is None:
        max_len = int(torch.max(lengths).item())
    ids = torch.arange(0, max_len, device=lengths.device)
    return ids.unsqueeze(0) >= lengths.unsqueeze(1)


class SeqInfo:
    """A padding mask together with the lengths it was built from, in the integer widths the HIP
    kernels take.  The reference passes only boolean masks between modules; the kernels index by
    length (no L x L or B x T masks are read on the device), so the top-level model builds this once
    and sub-modules accept it wherever the reference takes ``mask``."""

    __slots__ = ("mask", "lens64", "lens32")

    def __init__(self, lengths, max_len):
        self.lens64 = lengths.to(torch.int64).clamp(max=max_len).contiguous()
        self.lens32 = self.lens64.to(torch.int32)
        self.mask = get_mask_from_lengths(self.lens64, max_len)

    @classmethod
    def from_parts(cls, mask, lens64, lens32):
        """The three members as some launch already produced them (kantts_teacher_plan)."""
        o = cls.__new__(cls)
        o.mask, o.lens64, o.lens32 = mask, lens64, lens32
        return o

    @staticmethod
    def of(mask):
        """Accept None, a SeqInfo, or a prefix-valid boolean mask (B, T) as the reference passes."""
        if mask is None or isinstance(mask, SeqInfo):
            return mask
        lengths = (~mask).sum(dim=1)
        return SeqInfo(lengths, mask.shape[1])
