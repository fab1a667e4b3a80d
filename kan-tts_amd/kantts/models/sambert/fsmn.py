"""FSMN encoder on HIP kernels (reference kantts/models/sambert/fsmn.py:8-124), channels-last."""
import torch.nn as nn

from kantts._hip import ops
from kantts.models.utils import SeqInfo


class FeedForwardNet(nn.Module):
    """Conv1d(k=1)+ReLU+dropout, Conv1d(k=1, no bias) == two GEMM launches (reference :8-40)."""

    def __init__(self, d_in, d_hid, d_out, kernel_size=[1, 1], dropout=0.1):
        super().__init__()
        k_in, k_out = kernel_size
        # parameter holders (checkpoint keys w_1.{weight,bias}, w_2.weight); creation order fixes the seeded init
        self.w_1 = nn.Conv1d(d_in, d_hid, k_in, padding=(k_in - 1) // 2)
        self.w_2 = nn.Conv1d(d_hid, d_out, k_out, padding=(k_out - 1) // 2, bias=False)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        p = float(self.dropout.p) if self.training else 0.0
        # the hidden activation only feeds the second contraction: bf16 in bf16 mode (fp32 mode ignores the flag)
        h = ops.linear(x, self.w_1.weight, self.w_1.bias, relu=True, drop_p=p, out_bf16=True)
        return ops.linear(h, self.w_2.weight, None)


class MemoryBlockV2(nn.Module):
    """Depth-wise FIR memory with asymmetric padding (reference :43-72); mask, residual and the FIR
    are one kernel (kantts_fsmn_dwconv_fwd)."""

    def __init__(self, d, filter_size, shift, dropout=0.0):
        super(MemoryBlockV2, self).__init__()
        # taps before / after the current frame: round-half-even on the left (the reference's int(round(.))), floor on
        # the right; a positive shift moves `shift` taps from the future to the past (look-back only postnet)
        span = filter_size - 1
        look_back = max(shift, 0)
        self.lp = int(round(span / 2)) + look_back
        self.rp = span // 2 - look_back
        self.conv_dw = nn.Conv1d(d, d, filter_size, stride=1, padding=0, groups=d, bias=False)
        self.dropout = nn.Dropout(dropout)

    def forward(self, input, mask=None, res=None, outer_p=0.0):
        """``outer_p``: the encoder's own dropout on this block's output (reference :118), applied in the same pass as
        the block's dropout and the residual add (kantts_dropout2_add)."""
        info = SeqInfo.of(mask)
        lens = None if info is None else info.lens64
        p = float(self.dropout.p) if self.training else 0.0
        if p > 0.0 or outer_p > 0.0:
            out = ops.fsmn_memory(input, self.conv_dw.weight, lens, self.lp)
            return ops.dropout2_add(out, p, outer_p, res)
        return ops.fsmn_memory(input, self.conv_dw.weight, lens, self.lp, res=res)


class FsmnEncoderV2(nn.Module):
    """Stack of [FFN -> memory block -> (+x)] (reference :75-124)."""

    def __init__(self, filter_size, fsmn_num_layers, input_dim, num_memory_units, ffn_inner_dim, dropout=0.0,
                 shift=0):
        super(FsmnEncoderV2, self).__init__()
        self.filter_size, self.fsmn_num_layers = filter_size, fsmn_num_layers
        self.num_memory_units, self.ffn_inner_dim = num_memory_units, ffn_inner_dim
        self.dropout = dropout
        self.shift = list(shift) if isinstance(shift, list) else [shift] * fsmn_num_layers
        widths = [input_dim] + [num_memory_units] * (fsmn_num_layers - 1)  # only the first layer sees the input width
        self.ffn_lst = nn.ModuleList(
            FeedForwardNet(w, ffn_inner_dim, num_memory_units, dropout=dropout) for w in widths)
        self.memory_block_lst = nn.ModuleList(
            MemoryBlockV2(num_memory_units, filter_size, s, dropout) for s in self.shift[:fsmn_num_layers])

    def forward(self, input, mask=None):
        info = SeqInfo.of(mask)
        p = float(self.dropout) if self.training else 0.0
        x = ops.dropout2_add(input, p) if p > 0 else input
        for ffn, memory_block in zip(self.ffn_lst, self.memory_block_lst):
            context = ffn(x)
            same = self.num_memory_units == x.size(-1)
            x = memory_block(context, info, res=x if same else None, outer_p=p)
        return x
