"""Free-running inference loops as ONE launch each (csrc/ar_infer.hip; bf16 mode).

The reference walks both autoregressive loops of SAM-BERT inference from Python: the mel decoder
(kantts/models/sambert/kantts_sambert.py:569-610 around HybridAttentionDecoder.infer :208-253) and the duration predictor
(kantts/models/sambert/adaptors.py:67-83).  ``decode_graph.py`` made a decoder step one hipGraph replay; here a whole loop
is one kernel launch -- a workgroup per sequence walks every step (kantts_pnca_decode_run / kantts_dur_ar_run in
include/kantts_hip.h).  This file only prepares what the kernels read: the weights of a loop packed into one bf16 blob
(fragment-major matrices, input width padded to 128) and one fp32 blob (biases, LayerNorm parameters), rebuilt when a parameter
changes.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

import kantts._hip as hip
from kantts._hip import ops


def _key(params):
    """Identity of the weights a packed blob was built from.  ``ops.weights_epoch`` counts the raw in-place parameter
    writes that do not bump ``Tensor._version`` (ArenaAdam's kantts_adam_step, optimizer restore, broadcast): without it
    inference between two training steps would keep the blob of the first call."""
    return (ops.weights_epoch[0],) + tuple((p.data_ptr(), p._version) for p in params)


def _kmajor4(w):
    """(out, in) fp32 matrix -> flat k-chunk-major: element (n, k) at ((k // 4) * out + n) * 4 + k % 4 (the layout of
    kantts_dur_ar_run_f32: thread n reads 16 bytes per chunk, a wave 1 KB of consecutive addresses)."""
    w = w.detach().float()
    n, k = w.shape
    assert k % 4 == 0
    return w.view(n, k // 4, 4).permute(1, 0, 2).reshape(-1)


def _mat(w):
    """(out, in) fp32 matrix -> flat, input width zero-padded to a multiple of 128, FRAGMENT-MAJOR: the 16 rows x 32 inputs
    one MFMA A operand takes are contiguous, lane (q, i) = (k-chunk, row) owning eight consecutive inputs (the layout of
    kantts_fragmajor_bf16).  ``out`` must be a multiple of 16."""
    w = w.detach().float()
    if w.dim() == 3:
        w = w.squeeze(-1)
    n, k = w.shape
    pitch = (k + 127) // 128 * 128
    if pitch != k:
        w = F.pad(w, (0, pitch - k))
    assert n % 16 == 0, n
    return w.view(n // 16, 16, pitch // 32, 4, 8).permute(0, 2, 3, 1, 4).reshape(-1)


def _pad_to(v, n):
    v = v.detach().float().reshape(-1)
    return v if v.numel() == n else F.pad(v, (0, n - v.numel()))


class DecoderKernel:
    """kantts_pnca_decode_run for one HybridAttentionDecoder."""

    def __init__(self, dec, d_mel):
        self.dec, self.d_mel = dec, d_mel
        self.key = None
        self.w = self.f = self.hkv_w = self.hkv_b = None

    @staticmethod
    def eligible(dec, d_mel, bw):
        """The shapes csrc/ar_infer.hip is compiled for: d_model 128, 8 heads x 16, feed-forward 1024 with 1-tap
        convolutions, prenet d_mel -> 256 -> 256 -> 128, band widths up to 127."""
        if hip.get_precision() != "bf16" or not dec.pnca:
            return False
        fcs = [m for m in dec.prenet.fcs if isinstance(m, nn.Linear)]
        att, ffn = dec.pnca[0].pnca_attn, dec.pnca[0].pos_ffn
        d_mem = att.d_mem
        return (dec.d_model == 128 and att.n_head == 8 and att.d_head == 16 and len(fcs) == 3
                and [m.out_features for m in fcs] == [256, 256, 128] and fcs[0].in_features == d_mel <= 128
                and tuple(ffn.w_1.weight.shape) == (1024, 128, 1) and tuple(ffn.w_2.weight.shape) == (128, 1024, 1)
                and dec.dec_in_proj.in_features == d_mem + 128 <= 512 and dec.dec_out_proj.out_features >= d_mel
                and 0 <= int(bw) <= 127)

    def _params(self):
        dec = self.dec
        ps = [p for m in dec.prenet.fcs if isinstance(m, nn.Linear) for p in (m.weight, m.bias)]
        ps += [dec.dec_in_proj.weight, dec.dec_in_proj.bias, dec.ln.weight, dec.ln.bias, dec.dec_out_proj.weight,
               dec.dec_out_proj.bias]
        for layer in dec.pnca:
            a, f = layer.pnca_attn, layer.pos_ffn
            ps += [a.layer_norm.weight, a.layer_norm.bias, a.w_x_qkv.weight, a.w_x_qkv.bias, a.fc_x.weight, a.fc_x.bias,
                   a.fc_h.weight, a.fc_h.bias, a.w_h_kv.weight, a.w_h_kv.bias, f.layer_norm.weight, f.layer_norm.bias,
                   f.w_1.weight, f.w_1.bias, f.w_2.weight, f.w_2.bias]
        return ps

    @torch.no_grad()
    def refresh(self):
        key = _key(self._params())
        if key == self.key:
            return
        dec = self.dec
        fcs = [m for m in dec.prenet.fcs if isinstance(m, nn.Linear)]
        d_out = dec.dec_out_proj.out_features
        n_out = (d_out + 15) // 16 * 16
        ws = [_mat(m.weight) for m in fcs] + [_mat(dec.dec_in_proj.weight)]
        fs = [m.bias.detach().float() for m in fcs] + [dec.dec_in_proj.bias.detach().float()]
        for layer in dec.pnca:
            a, f = layer.pnca_attn, layer.pos_ffn
            ws += [_mat(a.w_x_qkv.weight), _mat(torch.cat([a.fc_x.weight, a.fc_h.weight], dim=1)), _mat(f.w_1.weight),
                   _mat(f.w_2.weight)]
            fs += [a.layer_norm.weight, a.layer_norm.bias, a.w_x_qkv.bias, a.fc_x.bias + a.fc_h.bias, f.layer_norm.weight,
                   f.layer_norm.bias, f.w_1.bias, f.w_2.bias]
        ws.append(_mat(F.pad(dec.dec_out_proj.weight.detach().float(), (0, 0, 0, n_out - d_out))))
        fs += [dec.ln.weight, dec.ln.bias, _pad_to(dec.dec_out_proj.bias, n_out)]
        self.w = torch.cat(ws).to(torch.bfloat16).contiguous()
        self.f = torch.cat([t.detach().float().reshape(-1) for t in fs]).contiguous()
        sizes = hip.decode_blob_sizes(self.d_mel, dec.pnca[0].pnca_attn.d_mem, d_out, len(dec.pnca))
        assert sizes == (self.w.numel(), self.f.numel()), (sizes, self.w.numel(), self.f.numel())
        # the memory K | V projections of all layers as one contraction: layer i at columns [256 i, 256 i + 256)
        self.hkv_w = torch.cat([layer.pnca_attn.w_h_kv.weight.detach() for layer in dec.pnca], dim=0).contiguous()
        self.hkv_b = torch.cat([layer.pnca_attn.w_h_kv.bias.detach() for layer in dec.pnca], dim=0).contiguous()
        self.key = key

    @torch.no_grad()
    def run(self, memory, lens32, bw_seq, bw):
        """memory (B, L, d_mem) -> (B, L, d_out): every decoder step of every sequence in one launch."""
        self.refresh()
        dec = self.dec
        B, L = memory.size(0), memory.size(1)
        memory = memory.contiguous().float()
        hkv = ops.linear(memory, self.hkv_w, self.hkv_b).float().contiguous()
        nl = len(dec.pnca)
        xkv = torch.empty((nl, B, L, 256), device=memory.device, dtype=torch.float32)
        out = torch.empty((B, L, dec.dec_out_proj.out_features), device=memory.device, dtype=torch.float32)
        hip.pnca_decode_run(self.w, self.f, memory, hkv, xkv, out, lens32, bw_seq, int(bw), self.d_mel, nl,
                            dec.d_model ** 0.5, dec.ln.eps)
        return out


class DurationKernel:
    """kantts_dur_ar_run_f32 / kantts_dur_ar_run for one VarRnnARPredictor.  ``bf16=False`` (the default in every
    precision mode): the fp32 loop, whose outputs become index tensors downstream; ``bf16=True`` keeps the bf16 MFMA loop
    of round 5 reachable (``pred.ar_bf16 = True``; bf16 mode only)."""

    def __init__(self, pred, bf16=False):
        self.pred = pred
        self.bf16 = bool(bf16)
        self.key = None
        self.w = self.f = self.gc_w = self.gc_b = None

    @staticmethod
    def eligible(pred, cond, bf16=False):
        if (bf16 and hip.get_precision() != "bf16") or pred.lstm.num_layers != 2 or pred.lstm.hidden_size != 128:
            return False
        fcs = [m for m in pred.prenet.fcs if isinstance(m, nn.Linear)]
        return (len(fcs) == 2 and fcs[0].in_features == 1 and [m.out_features for m in fcs] == [128, 128]
                and pred.lstm.input_size == 128 + cond.size(2) and cond.size(2) % 8 == 0)

    def _params(self):
        p = self.pred
        return ([q for m in p.prenet.fcs if isinstance(m, nn.Linear) for q in (m.weight, m.bias)]
                + [t for l in (0, 1) for t in p._layer(l)] + [p.fc.weight, p.fc.bias])

    @torch.no_grad()
    def refresh(self):
        key = _key(self._params())
        if key == self.key:
            return
        p = self.pred
        fc1, fc2 = [m for m in p.prenet.fcs if isinstance(m, nn.Linear)]
        w_ih0, w_hh0, b_ih0, b_hh0 = p._layer(0)
        w_ih1, w_hh1, b_ih1, b_hh1 = p._layer(1)
        mats = (fc2.weight, torch.cat([w_ih0[:, :128], w_hh0], dim=1), torch.cat([w_ih1, w_hh1], dim=1))
        if self.bf16:
            self.w = torch.cat([_mat(m) for m in mats]).to(torch.bfloat16).contiguous()
        else:
            self.w = torch.cat([_kmajor4(m) for m in mats]).contiguous()
        self.f = torch.cat([t.detach().float().reshape(-1) for t in
                            (fc1.weight, fc1.bias, fc2.bias, b_ih1 + b_hh1, p.fc.weight, p.fc.bias,
                             torch.zeros(3, device=fc1.weight.device))]).contiguous()
        self.gc_w = w_ih0[:, 128:].detach().contiguous()
        self.gc_b = (b_ih0 + b_hh0).detach().contiguous()
        self.key = key

    @torch.no_grad()
    def run(self, cond, lens32):
        """cond (B, T, C) -> (B, T) predictions (0 at padded tokens)."""
        self.refresh()
        B, T = cond.size(0), cond.size(1)
        with hip.precision_scope(None if self.bf16 else "fp32"):
            gc = ops.linear(cond.contiguous(), self.gc_w, self.gc_b).float().contiguous()  # (B, T, 512)
        out = torch.empty((B, T), device=cond.device, dtype=torch.float32)
        hip.dur_ar_run(self.w, self.f, gc, out, lens32)
        return out
