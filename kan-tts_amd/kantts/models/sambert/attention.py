"""Alignment attention of the MAS path -- drop-in for kantts/models/sambert/attention.py of the reference
(same class names, constructor arguments and state_dict keys: key_proj.{0,2}.conv.*, query_proj.{0,2,4}.conv.*,
attn_proj.*).

The projections run on the SAM-BERT conv/GEMM kernels (channels-last, ReLU in the epilogue); everything after them --
pairwise isotropic-Gaussian scores, log_softmax + log prior, padding mask, softmax -- is one HIP kernel per direction
(csrc/mas.hip) instead of the reference's (B, C, T_mel, T_text) broadcast tensor (attention.py:103-125).
"""
import numpy as np
import torch
from torch import nn

from kantts._hip import ops


class ConvNorm(nn.Module):
    """Conv1d with xavier init (reference attention.py:6-39).  forward keeps the reference's (B, C, T) contract;
    forward_cl is the channels-last entry the model uses."""

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=None, dilation=1, bias=True,
                 w_init_gain="linear"):
        super(ConvNorm, self).__init__()
        if padding is None:
            assert kernel_size % 2 == 1
            padding = int(dilation * (kernel_size - 1) / 2)
        if stride != 1 or dilation != 1 or padding != (kernel_size - 1) // 2:
            raise NotImplementedError("ConvNorm: only the 'same' stride-1 convolutions ConvAttention uses")
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                              dilation=dilation, bias=bias)
        nn.init.xavier_uniform_(self.conv.weight, gain=nn.init.calculate_gain(w_init_gain))

    def forward_cl(self, x, relu=False):
        k = self.conv.kernel_size[0]
        return ops.linear(x, self.conv.weight, self.conv.bias, relu=relu, pad=(k - 1) // 2,
                          mode="conv" if k > 1 else None)

    def forward(self, signal):
        return self.forward_cl(signal.transpose(1, 2)).transpose(1, 2)


class ConvAttention(nn.Module):
    def __init__(self, n_mel_channels=80, n_text_channels=512, n_att_channels=80, temperature=1.0,
                 use_query_proj=True):
        super(ConvAttention, self).__init__()
        self.temperature = temperature
        self.att_scaling_factor = np.sqrt(n_att_channels)
        self.attn_proj = nn.Conv2d(n_att_channels, 1, kernel_size=1)  # unused by forward, kept for checkpoints
        self.use_query_proj = bool(use_query_proj)
        self.key_proj = nn.Sequential(
            ConvNorm(n_text_channels, n_text_channels * 2, kernel_size=3, bias=True, w_init_gain="relu"),
            nn.ReLU(),
            ConvNorm(n_text_channels * 2, n_att_channels, kernel_size=1, bias=True),
        )
        self.query_proj = nn.Sequential(
            ConvNorm(n_mel_channels, n_mel_channels * 2, kernel_size=3, bias=True, w_init_gain="relu"),
            nn.ReLU(),
            ConvNorm(n_mel_channels * 2, n_mel_channels, kernel_size=1, bias=True),
            nn.ReLU(),
            ConvNorm(n_mel_channels, n_att_channels, kernel_size=1, bias=True),
        )

    def forward_cl(self, queries, keys, in_lens, attn_prior=None):
        """queries (B, T_mel, n_mel), keys (B, T_text, n_text) channels-last; in_lens (B,) unpadded text lengths.
        Returns (attn, attn_logprob), both (B, 1, T_mel, T_text) as the reference."""
        k = self.key_proj[0].forward_cl(keys, relu=True)
        k = self.key_proj[2].forward_cl(k)
        q = queries
        if self.use_query_proj:
            q = self.query_proj[0].forward_cl(q, relu=True)
            q = self.query_proj[2].forward_cl(q, relu=True)
            q = self.query_proj[4].forward_cl(q)
        lens = in_lens.to(torch.int32).contiguous()
        return ops.align_attention(q, k, attn_prior, lens)

    def forward(self, queries, keys, mask=None, attn_prior=None):
        """Reference signature: queries B x C x T1, keys B x C2 x T2, mask B x T2 (True = padding)."""
        B, T2 = keys.shape[0], keys.shape[2]
        if mask is None:
            lens = torch.full((B,), T2, device=keys.device, dtype=torch.int32)
        else:
            lens = (~mask).sum(dim=1)  # padding is a suffix (get_mask_from_lengths)
        return self.forward_cl(queries.transpose(1, 2), keys.transpose(1, 2), lens, attn_prior)
