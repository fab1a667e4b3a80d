"""Variance-adaptor building blocks on HIP kernels (reference kantts/models/sambert/adaptors.py:9-141)."""
import torch
import torch.nn as nn

from kantts._hip import ops
from kantts.models.sambert import Prenet
from kantts.models.sambert.fsmn import FsmnEncoderV2
from kantts.models.utils import SeqInfo


class LengthRegulator(nn.Module):
    """Duration -> frame expansion (reference :9-36).  The reference multiplies by a dense 0/1
    (B, T_mel, T_in) matrix; here kantts_lr_index builds the frame->token index once and
    kantts_lr_gather copies rows (bit-exact w.r.t. the one-hot matmul, deterministic backward)."""

    def __init__(self, r=1):
        super(LengthRegulator, self).__init__()
        self.r = r

    @staticmethod
    def padded_length(total, r):
        return total + ((r - total % r) % r)

    def index(self, durations, max_len=None):
        """Returns (idx, pos, cs, output_lens, Tp).  ``max_len`` avoids the host sync when the caller
        already knows max(output_lens) (training: mel_targets.size(1))."""
        if max_len is None:
            max_len = int((durations + 0.5).long().sum(dim=1).max().item())
        Tp = self.padded_length(max_len, self.r)
        idx, pos, cs, lens = ops.lr_index(durations, Tp)
        return idx, pos, cs, lens, Tp, max_len

    def forward(self, inputs, durations, masks=None, plan=None):
        if plan is None:
            plan = self.index(durations)
        idx, _, cs, output_lens, Tp, max_len = plan[:6]
        valid = None
        if len(plan) > 6:  # the clamp below, computed once with the plan (three gathers share it)
            valid = plan[6]
        elif masks is not None:
            info = SeqInfo.of(masks)
            valid = torch.clamp(info.lens64, max=max_len)
        elif Tp != max_len:
            valid = torch.full_like(output_lens, max_len)
        out = ops.lr_gather(inputs, idx, cs, valid)
        return out, output_lens


class VarRnnARPredictor(nn.Module):
    """Autoregressive duration predictor: Prenet || cond -> 2-layer LSTM -> Linear -> ReLU
    (reference :39-83).  Teacher-forced path: the concat is a two-segment GEMM feeding the hoisted
    LSTM input projection; the recurrence is the persistent kantts_lstm kernel."""

    def __init__(self, cond_units, prenet_units, rnn_units):
        super(VarRnnARPredictor, self).__init__()
        self.prenet = Prenet(1, prenet_units)
        self.lstm = nn.LSTM(prenet_units[-1] + cond_units, rnn_units, num_layers=2, batch_first=True,
                            bidirectional=False)
        self.fc = nn.Linear(rnn_units, 1)
        # free-running inference: None = the one-launch loop where it applies (HIP device, bf16 mode), False = always the
        # per-token launches, True = also on host tensors (the emulated C ABI of the CPU tests)
        self.ar_kernel = None
        self.ar_bf16 = False  # True: the bf16 MFMA loop of round 5 (flips ~0.2 % of the integer durations; bf16 mode only)
        self._ar = None

    def _layer(self, l):
        return [getattr(self.lstm, "%s_l%d" % (n, l)) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]

    def forward(self, inputs, cond, h=None, masks=None):
        if h is not None:
            raise NotImplementedError("stateful stepping is served by infer()")
        p = self.prenet(inputs)
        x = ops.lstm([p, cond], self._layer(0))
        x = ops.lstm(x, self._layer(1))
        info = SeqInfo.of(masks)
        x = ops.linear(x, self.fc.weight, self.fc.bias, relu=True,
                       rowmask=None if info is None else info.mask).squeeze(-1)
        return x, None

    @torch.no_grad()
    def infer(self, cond, masks=None):
        """Free-running prediction (reference :67-83): token i consumes the prediction of token i-1 through the
        prenet.  Per token: prenet (2 GEMMs), two LSTM cells -- each gate pre-activation is ONE multi-segment
        GEMM over [prenet | cond | h] -- and the output GEMM; no tensor is concatenated or re-allocated per
        step (the reference re-runs nn.LSTM on a length-1 sequence and torch.cat's the outputs)."""
        B, T = cond.size(0), cond.size(1)
        if self.ar_kernel is not False and (cond.is_cuda or self.ar_kernel is True):
            # the whole loop as ONE launch, a workgroup per sequence (csrc/ar_infer.hip); fp32 arithmetic in every mode
            from kantts.models.sambert.ar_kernels import DurationKernel

            if DurationKernel.eligible(self, cond, self.ar_bf16):
                if self._ar is None or self._ar.bf16 != bool(self.ar_bf16):
                    self._ar = DurationKernel(self, self.ar_bf16)
                info = SeqInfo.of(masks)
                return self._ar.run(cond, None if info is None else info.lens32)
        w_ih0, w_hh0, b_ih0, b_hh0 = self._layer(0)
        w_ih1, w_hh1, b_ih1, b_hh1 = self._layer(1)
        d_p = w_ih0.shape[1] - cond.size(2)
        # [x | cond | h] @ [W_ih | W_hh]^T : one weight with the segments side by side
        w0 = torch.cat([w_ih0, w_hh0], dim=1).contiguous()
        w1 = torch.cat([w_ih1, w_hh1], dim=1).contiguous()
        H = w_hh0.shape[1]
        dev = cond.device
        h0 = torch.zeros((B, H), device=dev)
        h1 = torch.zeros((B, H), device=dev)
        c0 = c1 = None
        x = torch.zeros((B, 1), device=dev)
        out = torch.empty((B, T), device=dev)
        assert d_p == self.prenet.fcs[-3].out_features
        for i in range(T):
            p = self.prenet(x)
            g0 = ops.linear([p, cond[:, i, :], h0], w0, b_ih0, bias2=b_hh0, mode="concat")
            h0, c0 = ops.lstm_cell(g0, c0)
            g1 = ops.linear([h0, h1], w1, b_ih1, bias2=b_hh1, mode="concat")
            h1, c1 = ops.lstm_cell(g1, c1)
            x = ops.linear(h1, self.fc.weight, self.fc.bias, relu=True)
            out[:, i] = x[:, 0]
        info = SeqInfo.of(masks)
        if info is not None:
            out = out.masked_fill(info.mask, 0.0)
        return out


class VarFsmnRnnNARPredictor(nn.Module):
    """FSMN -> packed BiLSTM -> Linear(->1) pitch / energy predictor (reference :86-141)."""

    def __init__(self, in_dim, filter_size, fsmn_num_layers, num_memory_units, ffn_inner_dim, dropout, shift,
                 lstm_units):
        super(VarFsmnRnnNARPredictor, self).__init__()
        self.fsmn = FsmnEncoderV2(filter_size, fsmn_num_layers, in_dim, num_memory_units, ffn_inner_dim, dropout,
                                  shift)
        self.blstm = nn.LSTM(num_memory_units, lstm_units, num_layers=1, batch_first=True, bidirectional=True)
        self.fc = nn.Linear(2 * lstm_units, 1)

    def forward(self, inputs, masks=None):
        info = SeqInfo.of(masks)
        x = self.fsmn(inputs, info)
        params = [self.blstm.weight_ih_l0, self.blstm.weight_hh_l0, self.blstm.bias_ih_l0, self.blstm.bias_hh_l0,
                  self.blstm.weight_ih_l0_reverse, self.blstm.weight_hh_l0_reverse, self.blstm.bias_ih_l0_reverse,
                  self.blstm.bias_hh_l0_reverse]
        x = ops.lstm(x, params, None if info is None else info.lens32)
        x = ops.linear(x, self.fc.weight, self.fc.bias, rowmask=None if info is None else info.mask).squeeze(-1)
        return x
