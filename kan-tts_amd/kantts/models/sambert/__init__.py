"""SAM-BERT transformer primitives on HIP kernels.

Same classes, constructor arguments and ``state_dict`` keys as the reference
(kantts/models/sambert/__init__.py:8-348); the ``nn.Linear`` / ``nn.Conv1d`` / ``nn.LayerNorm``
members are kept as parameter holders only -- their ``forward`` is never called.  The arithmetic
goes through kantts._hip.ops (segmented MFMA GEMM, fused LayerNorm, range-limited attention).
"""
import numpy as np
import torch
import torch.nn as nn

from kantts._hip import ops
from kantts.models.utils import SeqInfo


def _p(drop, training):
    """Effective dropout probability of an nn.Dropout holder."""
    return float(drop.p) if (training and drop.p > 0) else 0.0


class ScaledDotProductAttention(nn.Module):
    """Holder for temperature / attention-dropout (reference :8-29); the product attention runs in
    kantts_attn_fwd/bwd and never materialises q k^T."""

    def __init__(self, temperature, dropatt=0.0):
        super().__init__()
        self.temperature = temperature
        self.softmax = nn.Softmax(dim=2)
        self.dropatt = nn.Dropout(dropatt)


class Prenet(nn.Module):
    """Linear-ReLU-Dropout(0.5) stack (+ optional output Linear) -- reference :32-49.
    Each Linear+ReLU+Dropout is one GEMM launch (activation and dropout live in the epilogue)."""

    def __init__(self, in_units, prenet_units, out_units=0):
        super(Prenet, self).__init__()
        self.fcs = nn.ModuleList()
        for in_dim, out_dim in zip([in_units] + prenet_units[:-1], prenet_units):
            self.fcs.append(nn.Linear(in_dim, out_dim))
            self.fcs.append(nn.ReLU())
            self.fcs.append(nn.Dropout(0.5))
        if out_units:
            self.fcs.append(nn.Linear(prenet_units[-1], out_units))

    def forward(self, input):
        x = input
        mods = list(self.fcs)
        i = 0
        while i < len(mods):
            fc = mods[i]
            if i + 2 < len(mods) and isinstance(mods[i + 1], nn.ReLU):
                # hidden Prenet layers feed the next Linear only: bf16 storage in bf16 mode
                x = ops.linear(x, fc.weight, fc.bias, relu=True, drop_p=_p(mods[i + 2], self.training),
                               out_bf16=(i + 3 < len(mods)))
                i += 3
            else:
                x = ops.linear(x, fc.weight, fc.bias)
                i += 1
        return x


class MultiHeadSelfAttention(nn.Module):
    """Pre-LN multi-head self-attention with fused QKV (reference :52-106).

    LN -> [QKV GEMM] -> attention straight on the (B, L, 3*H*d) projection buffer (no head
    permutes) -> [output GEMM + dropout + residual (+ zeroing of padded rows, which FFTBlock
    applies right after, reference :177-178)] in one epilogue."""

    def __init__(self, n_head, d_in, d_model, d_head, dropout, dropatt=0.0):
        super().__init__()
        self.n_head = n_head
        self.d_head = d_head
        self.d_in = d_in
        self.d_model = d_model
        self.layer_norm = nn.LayerNorm(d_in, eps=1e-6)
        self.w_qkv = nn.Linear(d_in, 3 * n_head * d_head)
        self.attention = ScaledDotProductAttention(temperature=np.power(d_head, 0.5), dropatt=dropatt)
        self.fc = nn.Linear(n_head * d_head, d_model)
        self.dropout = nn.Dropout(dropout)
        # bf16 mode: the parameter arena also keeps the fragment-major images csrc/enc_attn.hip streams
        for lin in (self.w_qkv, self.fc):
            lin.weight._kantts_ffn_role = "lin"

    def forward(self, input, mask=None, zero_rows=None, return_attn=False, private_input=False, next_ln=None):
        """mask: key padding (SeqInfo or bool (B, L) / (B, L, L) as the reference builds it).
        ``private_input``: nothing but this sub-layer consumes ``input`` (see ops.layer_norm).
        ``next_ln``: the nn.LayerNorm of the sub-layer that consumes the output (computed in the output GEMM's epilogue)."""
        if torch.is_tensor(mask) and mask.dim() == 3:
            mask = mask[:, 0, :]
        info = SeqInfo.of(mask)
        # bf16 mode: the sub-layer's forward pass as ONE launch (csrc/enc_attn.hip) whose results the three ops below adopt
        # instead of launching; a no-op context whenever that launch does not apply
        with ops.enc_attn_fused(self, input, info, zero_rows, return_attn, next_ln, self.training):
            x, input = ops.layer_norm(input, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps,
                                      out_bf16=True, with_res=True, private_input=private_input)
            qkv = ops.linear(x, self.w_qkv.weight, self.w_qkv.bias)
            ctxv, attn = ops.self_attention(qkv, None if info is None else info.lens32, self.n_head,
                                            drop_p=_p(self.attention.dropatt, self.training), want_probs=return_attn)
            res = input if self.fc.out_features == input.size(-1) else None
            output = ops.linear(ctxv, self.fc.weight, self.fc.bias, res=res, rowmask=zero_rows,
                                drop_p=_p(self.dropout, self.training), ln_next=next_ln)
        return output, attn


class PositionwiseConvFeedForward(nn.Module):
    """LN -> Conv1d(k) -> ReLU -> mask -> Conv1d(1) -> +x (reference :109-149), channels-last:
    two GEMM launches (k=3 as three shifted row windows of the same activation, im2col-free)."""

    def __init__(self, d_in, d_hid, kernel_size=(3, 1), dropout_inner=0.1, dropout=0.1):
        super().__init__()
        self.w_1 = nn.Conv1d(d_in, d_hid, kernel_size=kernel_size[0], padding=(kernel_size[0] - 1) // 2)
        self.w_2 = nn.Conv1d(d_hid, d_in, kernel_size=kernel_size[1], padding=(kernel_size[1] - 1) // 2)
        self.layer_norm = nn.LayerNorm(d_in, eps=1e-6)
        self.dropout_inner = nn.Dropout(dropout_inner)
        self.dropout = nn.Dropout(dropout)
        # bf16 mode: the parameter arena also keeps the fragment-major images csrc/ffn_pair.hip streams (ops_bf16)
        self.w_1.weight._kantts_ffn_role = "w1"
        self.w_2.weight._kantts_ffn_role = "w2"

    def forward(self, x, mask=None, zero_rows=None, private_input=False, next_ln=None):
        info = SeqInfo.of(mask)
        pad_rows = None if info is None else info.mask
        h, x = ops.layer_norm(x, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps, out_bf16=True,
                              with_res=True, private_input=private_input)
        return ops.ffn(h, self.w_1.weight, self.w_1.bias, self.w_2.weight, self.w_2.bias, x, pad_rows=pad_rows,
                       zero_rows=zero_rows, p_inner=_p(self.dropout_inner, self.training),
                       p_out=_p(self.dropout, self.training), ln_next=next_ln)


class FFTBlock(nn.Module):
    """Feed-Forward-Transformer block (reference :152-184)."""

    def __init__(self, d_in, d_model, n_head, d_head, d_inner, kernel_size, dropout, dropout_attn=0.0,
                 dropout_relu=0.0):
        super(FFTBlock, self).__init__()
        self.slf_attn = MultiHeadSelfAttention(n_head, d_in, d_model, d_head, dropout=dropout, dropatt=dropout_attn)
        self.pos_ffn = PositionwiseConvFeedForward(d_model, d_inner, kernel_size, dropout_inner=dropout_relu,
                                                   dropout=dropout)

    def forward(self, input, mask=None, slf_attn_mask=None, return_attn=False, private_input=False, next_ln=None):
        """``private_input``: the caller hands ``input`` to this block only (a stack does, from its second block on): the
        producer's masking of the incoming gradient then happens inside this block's first LayerNorm backward.
        ``next_ln``: the nn.LayerNorm that consumes this block's output (the next block's, or the stack's final one)."""
        info = SeqInfo.of(mask)
        rows = None if info is None else info.mask
        key_info = info if info is not None else slf_attn_mask
        output, slf_attn = self.slf_attn(input, mask=key_info, zero_rows=rows, return_attn=return_attn,
                                         private_input=private_input, next_ln=self.pos_ffn.layer_norm)
        # the attention sub-layer's output goes nowhere but into the feed-forward sub-layer
        output = self.pos_ffn(output, mask=info, zero_rows=rows, private_input=True, next_ln=next_ln)
        return output, slf_attn


class MultiHeadPNCAAttention(nn.Module):
    """PNCA attention: causal-band self attention + look-ahead-band memory attention sharing Q
    (reference :187-306).  Training path: LN -> [x QKV GEMM], [h KV GEMM] -> both banded attentions
    in one op -> [fc_x + fc_h as one two-segment GEMM + dropout + residual + row zeroing]."""

    def __init__(self, n_head, d_model, d_mem, d_head, dropout, dropatt=0.0):
        super().__init__()
        self.n_head = n_head
        self.d_head = d_head
        self.d_model = d_model
        self.d_mem = d_mem
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-6)
        self.w_x_qkv = nn.Linear(d_model, 3 * n_head * d_head)
        self.fc_x = nn.Linear(n_head * d_head, d_model)
        self.w_h_kv = nn.Linear(d_mem, 2 * n_head * d_head)
        self.fc_h = nn.Linear(n_head * d_head, d_model)
        self.attention = ScaledDotProductAttention(temperature=np.power(d_head, 0.5), dropatt=dropatt)
        self.dropout = nn.Dropout(dropout)
        # bf16 mode: the parameter arena also keeps the fragment-major images csrc/pnca_block.hip streams
        for lin in (self.w_x_qkv, self.fc_x, self.fc_h):
            lin.weight._kantts_ffn_role = "lin"
        self.reset_state()

    def reset_state(self):
        self.h_k = None
        self.h_v = None
        self.h_state_size = 0
        self.x_k = None
        self.x_v = None
        self.x_state_size = 0

    def forward(self, x, h, info=None, x_band_width=0, h_band_width=0, zero_rows=None, return_attn=False,
                bw_dev=None, hkv=None, private_input=False, next_ln=None):
        """``hkv``: this block's memory K/V projection when the decoder computed all of them together
        (ops.shared_input_linears: one input-gradient launch for the twelve blocks)."""
        xn, x = ops.layer_norm(x, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps, out_bf16=True,
                               with_res=True, private_input=private_input)
        qkv = ops.linear(xn, self.w_x_qkv.weight, self.w_x_qkv.bias)
        if hkv is None:
            hkv = ops.linear(h, self.w_h_kv.weight, self.w_h_kv.bias)
        info = SeqInfo.of(info)
        ox, oh, attn_x, attn_h = ops.pnca_attention(
            qkv, hkv, None if info is None else info.lens32, x_band_width, h_band_width, self.n_head,
            drop_p=_p(self.attention.dropatt, self.training), want_probs=return_attn, bw_dev=bw_dev)
        output = ops.linear([ox, oh], [self.fc_x.weight, self.fc_h.weight], self.fc_x.bias, bias2=self.fc_h.bias,
                            mode="sum", res=x, rowmask=zero_rows, drop_p=_p(self.dropout, self.training), ln_next=next_ln)
        return output, attn_x, attn_h


    @torch.no_grad()
    def infer_step(self, step, x, h, x_band_width, h_band_width, zero_rows=None, lens32=None, bw_seq=None):
        """Decoder position ``step`` of the free-running decode (reference forward() under update_x_state /
        update_h_state).  State: ``qkv_buf`` (B, L, 3*H*d) receives row ``step`` from the QKV projection -- the K/V
        cache is that buffer, never re-concatenated -- and ``hkv`` (memory K/V) is projected once at step 0."""
        B, L = h.size(0), h.size(1)
        D = self.n_head * self.d_head
        if step == 0 or self.h_k is None:
            self.h_k = ops.linear(h, self.w_h_kv.weight, self.w_h_kv.bias)           # (B, L, 2D)
            self.x_k = torch.zeros((B, L, 3 * D), device=h.device, dtype=torch.float32)
            self.h_state_size, self.x_state_size = L, 0
        xr = x.reshape(B, -1)
        xn = ops.layer_norm(xr, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps)
        qkv = ops.linear(xn, self.w_x_qkv.weight, self.w_x_qkv.bias)              # (B, 3D)
        self.x_k[:, step, :] = qkv
        self.x_state_size = step + 1
        ox = torch.empty((B, D), device=h.device, dtype=torch.float32)
        oh = torch.empty((B, D), device=h.device, dtype=torch.float32)
        ops.attn_decode(qkv[:, :D], self.x_k[:, :, D:2 * D], self.x_k[:, :, 2 * D:], ox, lens32, self.n_head, step,
                        x_band_width, ops.MODE_BAND_X, bw_seq=bw_seq)
        ops.attn_decode(qkv[:, :D], self.h_k[:, :, :D], self.h_k[:, :, D:], oh, lens32, self.n_head, step, h_band_width,
                        ops.MODE_BAND_H, bw_seq=bw_seq)
        out = ops.linear([ox, oh], [self.fc_x.weight, self.fc_h.weight], self.fc_x.bias, bias2=self.fc_h.bias,
                         mode="sum", res=xr, rowmask=zero_rows)
        return out.view(B, 1, -1)


class PNCABlock(nn.Module):
    """PNCA block (reference :309-348)."""

    def __init__(self, d_model, d_mem, n_head, d_head, d_inner, kernel_size, dropout, dropout_attn=0.0,
                 dropout_relu=0.0):
        super(PNCABlock, self).__init__()
        self.pnca_attn = MultiHeadPNCAAttention(n_head, d_model, d_mem, d_head, dropout=dropout, dropatt=dropout_attn)
        self.pos_ffn = PositionwiseConvFeedForward(d_model, d_inner, kernel_size, dropout_inner=dropout_relu,
                                                   dropout=dropout)

    def forward(self, input, memory, mask=None, x_band_width=0, h_band_width=0, return_attn=False, bw_dev=None,
                hkv=None, private_input=False, next_ln=None):
        info = SeqInfo.of(mask)
        rows = None if info is None else info.mask
        # bf16 mode: the whole block's forward pass as ONE launch (csrc/pnca_block.hip) whose results the ops below adopt
        # instead of launching; a no-op context whenever that launch does not apply
        with ops.pnca_block_fused(self, input, hkv, info, x_band_width, h_band_width, bw_dev, return_attn, next_ln,
                                  self.training):
            output, ax, ah = self.pnca_attn(input, memory, info, x_band_width, h_band_width, zero_rows=rows,
                                            return_attn=return_attn, bw_dev=bw_dev, hkv=hkv, private_input=private_input,
                                            next_ln=self.pos_ffn.layer_norm)
            output = self.pos_ffn(output, mask=info, zero_rows=rows, private_input=True, next_ln=next_ln)
        return output, ax, ah

    @torch.no_grad()
    def infer_step(self, step, input, memory, x_band_width, h_band_width, zero_rows=None, lens32=None, bw_seq=None):
        output = self.pnca_attn.infer_step(step, input, memory, x_band_width, h_band_width, zero_rows=zero_rows,
                                           lens32=lens32, bw_seq=bw_seq)
        return self.pos_ffn(output, mask=None, zero_rows=zero_rows)

    def reset_state(self):
        self.pnca_attn.reset_state()
