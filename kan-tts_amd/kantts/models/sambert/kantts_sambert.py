"""KanTtsSAMBERT acoustic model on HIP kernels.

Drop-in for kantts/models/sambert/kantts_sambert.py of the reference: same class names, constructor
config keys, ``forward`` signature, result-dict keys and ``state_dict`` keys.  Differences that are
deliberate (documented in DESIGN.md):
  * attention probability tensors (8 + 24 dense (B*H, L, L) maps, ~1 GB at the training shape) are
    only materialised when ``model.return_attns`` is True; the dict keys are always present;
  * padding is described by lengths, never by materialised (B, L, L) masks;
  * the MAS alignment path (sambert_16k_MAS.yaml) and speaker-embedding input (SE: True) are implemented;
    FP / byte-input variants raise NotImplementedError.
"""
import contextlib

import torch
import torch.nn as nn
import torch.nn.functional as F

import os

import kantts._hip as hip_rt
from kantts._hip import ops
from kantts.models.sambert import FFTBlock, PNCABlock, Prenet
from kantts.models.sambert.alignment import b_mas
from kantts.models.sambert.attention import ConvAttention
from kantts.models.sambert.adaptors import LengthRegulator, VarFsmnRnnNARPredictor, VarRnnARPredictor
from kantts.models.sambert.fsmn import FsmnEncoderV2
from kantts.models.sambert.positions import DurSinusoidalPositionEncoder, SinusoidalPositionEncoder
from kantts.models.utils import SeqInfo, get_mask_from_lengths


# which gradient-carrying pieces run beside the encoder in a teacher-forced step (KanTtsSAMBERT._beside_encoder), measured
# one by one (profiles/r03_runAH_beside_parts.log, step ms): plan only 7.83-7.86; + decoder prenet 7.70; + pitch / energy
# embeddings 7.82 (within noise, and it re-associates text + pitch + energy: off, the reference's summation order stays);
# + emotion / speaker embeddings 8.33 (their table gradients are atomics-bound kernels that then share the chip with the
# encoder's backward: off)
_BESIDE_PARTS = tuple(x for x in os.environ.get("KANTTS_BESIDE_PARTS", "prenet").split(",") if x)
# extra flush points for deferred weight gradients inside the block stacks (ops.wgrad_flush_point): every N blocks; 0 = only
# the points at the postnet input and the encoder output.  Measured on the bench step: see DESIGN section 5.
# Round 3 re-measured on the final schedule (profiles/r03_runAJ_flush_points.log): encoder every 4 blocks 7.67 ms against 7.75
# (the encoder's own weight gradients start beside its second half instead of after it), every 2 blocks 8.10, decoder every
# 4 blocks 8.03 (smaller groups, more launches), no early flush at all 8.42.
# in a teacher-forced step the variance predictors (a side branch that only feeds its own losses) start beside the POSTNET
# (whose LSTM leaves 7/8 of the chip idle) instead of beside the decoder: 7.65 -> 7.60 ms (profiles/
# r03_runAK_predictors_late.log); KANTTS_PREDICTORS_EARLY restores the round-2 placement
_PREDICTORS_LATE = not os.environ.get("KANTTS_PREDICTORS_EARLY")
_PLAN_KERNEL = not os.environ.get("KANTTS_NO_PLAN_KERNEL")  # A/B switch: the target-only plan as one launch (csrc/seq.hip)
# [round 5] experiment switch KANTTS_PREDICTORS_AFTER_POSTNET=1: issue the predictors after the postnet's launches, so that
# autograd (which issues backward nodes newest first) issues their backward BEFORE the postnet's and it could run beside the
# postnet's 612-step LSTM backward (0.5 ms on 32 of 256 CUs with nothing beside it).  Measured (profiles/r05_runL_*): the
# replayed hipGraph runs the two branches one after the other in capture order either way -- 6.74-6.89 ms against 6.45 ms
# for the round-4 order on the same box -- so the round-4 order stays.
_PREDICTORS_AFTER_POSTNET = bool(os.environ.get("KANTTS_PREDICTORS_AFTER_POSTNET"))
# experiment switch KANTTS_EARLY_FORK=1: the predictor branch forks where its inputs became ready (beside the decoder) instead
# of where it is issued.  Measured 6.41-6.46 against 6.40-6.42 ms, forward 1.89 against 1.86 ms (profiles/r05_runM_*): off.
_EARLY_FORK = bool(os.environ.get("KANTTS_EARLY_FORK"))
# A/B switch: the predictor-only concatenations on the main chain, as until round 6 (VarianceAdaptor.forward)
_CATS_EARLY = bool(os.environ.get("KANTTS_CATS_EARLY"))
_FLUSH_EVERY = {"dec": int(os.environ.get("KANTTS_FLUSH_EVERY_DEC", "0")), "enc": int(os.environ.get("KANTTS_FLUSH_EVERY_ENC", "4"))}


class SelfAttentionEncoder(nn.Module):
    """Stack of FFT blocks + final LayerNorm (reference :22-87).  The sqrt(d_model) scaling and the
    sinusoid add are fused into the embedding gather of TextFftEncoder; when this module is called
    on its own they are applied here."""

    def __init__(self, n_layer, d_in, d_model, n_head, d_head, d_inner, dropout, dropout_att, dropout_relu,
                 position_encoder):
        super(SelfAttentionEncoder, self).__init__()
        self.d_in = d_in
        self.d_model = d_model
        self.dropout = dropout
        d_in_lst = [d_in] + [d_model] * (n_layer - 1)
        self.fft = nn.ModuleList([
            FFTBlock(d, d_model, n_head, d_head, d_inner, (3, 1), dropout, dropout_att, dropout_relu)
            for d in d_in_lst
        ])
        self.ln = nn.LayerNorm(d_model, eps=1e-6)
        self.ln._kantts_out_bf16 = False  # read by more than contractions: fp32 output (ops_bf16.PreNorm)
        self.position_enc = position_encoder

    def forward(self, input, mask=None, return_attns=False, prescaled=False):
        if not prescaled:
            input = self.position_enc(input * self.d_model ** 0.5)
        if self.training and self.dropout > 0:
            # counter-based masks regenerated in backward (ops.dropout2_add): no mask tensor, no torch generator state --
            # a captured step and an eager step given the same seeds draw the same masks
            input = ops.dropout2_add(input, p1=self.dropout)
        info = SeqInfo.of(mask)
        attns = []
        x = input
        every = _FLUSH_EVERY["enc"] if self.training else 0
        for i, layer in enumerate(self.fft):
            if every and i and i % every == 0:
                x = ops.wgrad_flush_point(x)  # backward: weight gradients of the later blocks start beside the earlier ones
            # from the second block on, x is the previous block's output and has no other reader
            nxt = self.fft[i + 1].slf_attn.layer_norm if i + 1 < len(self.fft) else self.ln
            x, a = layer(x, mask=info, return_attn=return_attns, private_input=i > 0, next_ln=nxt)
            if return_attns:
                attns.append(a)
        x = ops.layer_norm(x, self.ln.weight, self.ln.bias, self.ln.eps, private_input=len(self.fft) > 0)
        return x, attns


class HybridAttentionDecoder(nn.Module):
    """Prenet -> [memory | prenet] projection -> PNCA blocks -> LN -> output projection
    (reference :90-253).  The two band masks of get_pnca_attn_mask are never built: the attention
    kernel derives each query's key interval from (position, band width, length)."""

    def __init__(self, d_in, prenet_units, n_layer, d_model, d_mem, n_head, d_head, d_inner, dropout, dropout_att,
                 dropout_relu, d_out):
        super(HybridAttentionDecoder, self).__init__()
        self.d_model = d_model
        self.dropout = dropout
        self.prenet = Prenet(d_in, prenet_units, d_model)
        self.dec_in_proj = nn.Linear(d_model + d_mem, d_model)
        self.pnca = nn.ModuleList([
            PNCABlock(d_model, d_mem, n_head, d_head, d_inner, (1, 1), dropout, dropout_att, dropout_relu)
            for _ in range(n_layer)
        ])
        self.ln = nn.LayerNorm(d_model, eps=1e-6)
        self.ln._kantts_out_bf16 = False
        self.dec_out_proj = nn.Linear(d_model, d_out)

    def reset_state(self):
        for layer in self.pnca:
            layer.reset_state()

    def forward(self, input, memory, x_band_width, h_band_width, mask=None, return_attns=False, bw_dev=None,
                prenet_out=None):
        """``prenet_out``: self.prenet(input) when the caller has already formed it (teacher forcing: the prenet reads
        the target frames only, so it runs beside the text encoder)."""
        info = SeqInfo.of(mask)
        rows = None if info is None else info.mask
        x = prenet_out if prenet_out is not None else self.prenet(input)
        # cat([memory, prenet]) @ W^T as a two-segment GEMM; masked rows -> 0; * sqrt(d_model)
        x = ops.linear([memory, x], self.dec_in_proj.weight, self.dec_in_proj.bias, mode="concat", rowmask=rows,
                       alpha=self.d_model ** 0.5)
        if self.training and self.dropout > 0:
            x = ops.dropout2_add(x, p1=self.dropout)
        # the memory K/V projections of all blocks read the same tensor: one GEMM forward, one input-gradient launch
        # (running them beside this entry projection on a second stream changed nothing: profiles/r03_runAN_hkv_beside.log)
        hkvs = ops.shared_input_linears(memory, [layer.pnca_attn.w_h_kv for layer in self.pnca])
        ax_l, ah_l = [], []
        every = _FLUSH_EVERY["dec"] if self.training else 0
        for i, layer in enumerate(self.pnca):
            if every and i and i % every == 0:
                x = ops.wgrad_flush_point(x)
            x, ax, ah = layer(x, memory, mask=info, x_band_width=x_band_width, h_band_width=h_band_width,
                              return_attn=return_attns, bw_dev=bw_dev, hkv=hkvs[i], private_input=i > 0,
                              next_ln=self.pnca[i + 1].pnca_attn.layer_norm if i + 1 < len(self.pnca) else self.ln)
            if return_attns:
                ax_l.append(ax)
                ah_l.append(ah)
        x = ops.layer_norm(x, self.ln.weight, self.ln.bias, self.ln.eps, private_input=len(self.pnca) > 0)
        x = ops.linear(x, self.dec_out_proj.weight, self.dec_out_proj.bias)
        return x, ax_l, ah_l

    @torch.no_grad()
    def infer(self, step, input, memory, x_band_width, h_band_width, mask=None, return_attns=False, bw_seq=None):
        """One free-running decoder step (reference :208-253): input (B, 1, d_mel) = previous output frame.
        Per-step masks are intervals inside the attention kernel; attention maps are not materialised."""
        if return_attns:
            raise NotImplementedError("attention maps of the free-running decode are not materialised")
        info = SeqInfo.of(mask)
        rows = None if info is None else info.mask[:, step].contiguous()
        B = memory.size(0)
        x = self.prenet(input.reshape(B, -1))
        x = ops.linear([memory[:, step, :], x], self.dec_in_proj.weight, self.dec_in_proj.bias, mode="concat",
                       alpha=self.d_model ** 0.5)
        x = x.view(B, 1, -1)
        lens32 = None if info is None else info.lens32
        for layer in self.pnca:
            x = layer.infer_step(step, x, memory, x_band_width, h_band_width, zero_rows=rows, lens32=lens32,
                                 bw_seq=bw_seq)
        x = ops.layer_norm(x, self.ln.weight, self.ln.bias, self.ln.eps)
        x = ops.linear(x, self.dec_out_proj.weight, self.dec_out_proj.bias)
        return x, [], []


class TextFftEncoder(nn.Module):
    """Linguistic embeddings -> FFT encoder -> projection (reference :256-337)."""

    def __init__(self, config):
        super(TextFftEncoder, self).__init__()
        d_emb = config["embedding_dim"]
        self.using_byte = False
        self.ling_embedding_grad = False  # set by KanTtsSAMBERT when the MAS attention consumes ling_embedding
        if config.get("using_byte", False):
            raise NotImplementedError("byte-index inputs (sambert_16k_MAS_byte.yaml) are outside the hot path")
        self.sy_emb = nn.Embedding(config["sy"], d_emb)
        self.tone_emb = nn.Embedding(config["tone"], d_emb)
        self.syllable_flag_emb = nn.Embedding(config["syllable_flag"], d_emb)
        self.ws_emb = nn.Embedding(config["word_segment"], d_emb)
        max_len = config["max_len"]
        nb_layers = config["encoder_num_layers"]
        nb_heads = config["encoder_num_heads"]
        d_model = config["encoder_num_units"]
        d_head = d_model // nb_heads
        d_inner = config["encoder_ffn_inner_dim"]
        dropout = config["encoder_dropout"]
        dropout_attn = config["encoder_attention_dropout"]
        dropout_relu = config["encoder_relu_dropout"]
        d_proj = config["encoder_projection_units"]
        self.d_model = d_model
        position_enc = SinusoidalPositionEncoder(max_len, d_emb)
        self.ling_enc = SelfAttentionEncoder(nb_layers, d_emb, d_model, nb_heads, d_head, d_inner, dropout,
                                             dropout_attn, dropout_relu, position_enc)
        self.ling_proj = nn.Linear(d_model, d_proj, bias=False)

    def forward(self, inputs_ling, masks=None, return_attns=False):
        T = inputs_ling.size(1)
        pos = self.ling_enc.position_enc.table_for(T, inputs_ling.device)
        # one gather kernel: (sy + tone + syllable_flag + ws) * sqrt(d_model) (+ sinusoid table);
        # the scaled sum is also what the reference hands back as `ling_embedding` (in-place *=, :62)
        x, ling_embedding = ops.embed_sum(
            inputs_ling[:, :, :4],
            [self.sy_emb.weight, self.tone_emb.weight, self.syllable_flag_emb.weight, self.ws_emb.weight],
            pos=pos, scale=self.d_model ** 0.5, want_scaled="grad" if self.ling_embedding_grad else True)
        enc_output, attns = self.ling_enc(x, masks, return_attns, prescaled=True)
        if hasattr(self, "ling_proj"):
            enc_output = ops.linear(enc_output, self.ling_proj.weight, None)
        return enc_output, attns, ling_embedding


class VarianceAdaptor(nn.Module):
    """Pitch / energy / duration predictors + length regulation (reference :340-500)."""

    def __init__(self, config):
        super(VarianceAdaptor, self).__init__()
        input_dim = config["encoder_projection_units"] + config["emotion_units"] + config["speaker_units"]
        filter_size = config["predictor_filter_size"]
        fsmn_num_layers = config["predictor_fsmn_num_layers"]
        num_memory_units = config["predictor_num_memory_units"]
        ffn_inner_dim = config["predictor_ffn_inner_dim"]
        dropout = config["predictor_dropout"]
        shift = config["predictor_shift"]
        lstm_units = config["predictor_lstm_units"]
        dur_pred_prenet_units = config["dur_pred_prenet_units"]
        dur_pred_lstm_units = config["dur_pred_lstm_units"]
        self.pitch_predictor = VarFsmnRnnNARPredictor(input_dim, filter_size, fsmn_num_layers, num_memory_units,
                                                      ffn_inner_dim, dropout, shift, lstm_units)
        self.energy_predictor = VarFsmnRnnNARPredictor(input_dim, filter_size, fsmn_num_layers, num_memory_units,
                                                       ffn_inner_dim, dropout, shift, lstm_units)
        self.duration_predictor = VarRnnARPredictor(input_dim, dur_pred_prenet_units, dur_pred_lstm_units)
        self.length_regulator = LengthRegulator(config["outputs_per_step"])
        self.dur_position_encoder = DurSinusoidalPositionEncoder(config["encoder_projection_units"],
                                                                 config["outputs_per_step"])
        self.pitch_emb = nn.Conv1d(1, config["encoder_projection_units"], kernel_size=9, padding=4)
        self.energy_emb = nn.Conv1d(1, config["encoder_projection_units"], kernel_size=9, padding=4)

    def pitch_energy_embedding(self, pitch, energy):
        """Conv1d(1->32, k=9)(pitch) + Conv1d(1->32, k=9)(energy) (reference :420-443)."""
        return (ops.conv_cl(pitch.unsqueeze(-1).contiguous(), self.pitch_emb.weight, self.pitch_emb.bias, pad=4)
                + ops.conv_cl(energy.unsqueeze(-1).contiguous(), self.energy_emb.weight, self.energy_emb.bias, pad=4))

    def position_term(self, plan, out_info):
        """Duration-relative position encoding of the regulated frames: valid frames only, zero in the mask / r-padding
        (reference positions.py:83-90).  Depends on the durations alone."""
        idx, pos, cs, LR_length_rounded, Tp, max_len = plan[:6]
        t = torch.arange(Tp, device=pos.device)[None, :]
        # no output masks = free-running inference, which the reference runs one utterance at a time: a frame past a sequence's
        # own regulated length is then r-padding (position 0), also when a longer sequence of the batch makes it a frame
        # < max_len -- the frames of a sequence's LAST decoder step (length % r != 0) would otherwise differ between batched
        # and per-utterance inference (tests/test_config5_inference.py)
        limit = (LR_length_rounded if out_info is None else out_info.lens64).clamp(max=max_len)
        pos = torch.where(t < limit[:, None], pos, torch.zeros_like(pos))
        return self.dur_position_encoder.from_positions(pos)

    def forward(self, inputs_text_embedding, inputs_emo_embedding, inputs_spk_embedding, masks=None,
                output_masks=None, duration_targets=None, pitch_targets=None, energy_targets=None, max_out_len=None,
                teacher_plan=None):
        """``teacher_plan`` (teacher_forced_plan): everything below that depends on the targets only, computed by the
        caller beside the encoder."""
        info = SeqInfo.of(masks)
        out_info = SeqInfo.of(output_masks)
        # [text | spk | emo] is consumed by three GEMMs (two FSMN inputs + the duration LSTM); build it once

        def vp_inputs():
            return torch.cat([inputs_text_embedding, inputs_spk_embedding, inputs_emo_embedding], dim=-1)
        # Teacher-forced training: the predictors feed only their own losses (everything downstream uses the targets), so
        # they run as a side branch beside the length regulator / decoder / postnet (ops.side_branch; joined at the end of
        # KanTtsSAMBERT.forward).  Free-running inference needs their outputs at once: the branch is then a no-op.
        teacher = (self.training and duration_targets is not None and pitch_targets is not None
                   and energy_targets is not None)
        lens_keep = None if info is None else (info.lens64, info.mask)
        late = teacher and _PREDICTORS_LATE  # the caller runs self.deferred() later (beside the postnet)
        # [round 6] the two concatenations only the predictors read are formed INSIDE their side branch when that branch is
        # issued late: two launches fewer on the main chain in front of the length regulator (KANTTS_CATS_EARLY=1: as before)
        cats_late = late and duration_targets is not None and not _CATS_EARLY
        variance_predictor_inputs = None if cats_late else vp_inputs()
        self.deferred = None
        pitch_predictions = energy_predictions = None
        if not late:
            with ops.side_branch.fork(variance_predictor_inputs, *(lens_keep or ())) if teacher else contextlib.nullcontext():
                pitch_predictions = self.pitch_predictor(variance_predictor_inputs, info)
                energy_predictions = self.energy_predictor(variance_predictor_inputs, info)
        pitch_src = pitch_targets if pitch_targets is not None else pitch_predictions
        energy_src = energy_targets if energy_targets is not None else energy_predictions
        # text + Conv1d(1->32,k=9)(pitch) + Conv1d(1->32,k=9)(energy): two 9-tap GEMMs chained through the
        # residual port of the epilogue (reference :420-443)
        # single-input-channel convolutions: the streaming kernels of csrc/conv_c1.hip (one launch forward, one for the
        # weight gradient) -- as token-shifted GEMMs with K = 1 they took 9 launches per weight gradient on the generic
        # contraction kernel (0.4 ms of a 13 ms step, profiles/r02_runE_sambert_kernel_stats_top.csv)
        if teacher_plan is not None and "pe_emb" in teacher_plan:
            aug = inputs_text_embedding + teacher_plan["pe_emb"]  # the two target embeddings were formed beside the encoder
        else:  # the reference's order of the two adds: (text + pitch) + energy
            aug = (inputs_text_embedding
                   + ops.conv_cl(pitch_src.unsqueeze(-1).contiguous(), self.pitch_emb.weight, self.pitch_emb.bias, pad=4)
                   + ops.conv_cl(energy_src.unsqueeze(-1).contiguous(), self.energy_emb.weight, self.energy_emb.bias, pad=4))
        duration_predictor_cond = (None if cats_late
                                   else torch.cat([aug, inputs_spk_embedding, inputs_emo_embedding], dim=-1))
        if duration_targets is not None:
            prev = (teacher_plan["prev"] if teacher_plan is not None
                    else torch.log(F.pad(duration_targets[:, :-1].float(), (1, 0)) + 1).unsqueeze(-1))
            log_duration_predictions = None
            if late:
                ready = ops.side_branch.mark() if _EARLY_FORK else None  # the predictors' inputs exist from here on

                def deferred():
                    with ops.side_branch.fork(variance_predictor_inputs, duration_predictor_cond, prev, aug,
                                              inputs_text_embedding, inputs_spk_embedding, inputs_emo_embedding,
                                              *(lens_keep or ()), after=ready):
                        vpi = variance_predictor_inputs if variance_predictor_inputs is not None else vp_inputs()
                        cond = (duration_predictor_cond if duration_predictor_cond is not None
                                else torch.cat([aug, inputs_spk_embedding, inputs_emo_embedding], dim=-1))
                        p_ = self.pitch_predictor(vpi, info)
                        e_ = self.energy_predictor(vpi, info)
                        d_, _ = self.duration_predictor(prev, cond, masks=info)
                    return d_, p_, e_

                self.deferred = deferred
            else:
                with ops.side_branch.fork(duration_predictor_cond, prev) if teacher else contextlib.nullcontext():
                    log_duration_predictions, _ = self.duration_predictor(prev, duration_predictor_cond, masks=info)
            durations = duration_targets
        else:
            log_duration_predictions = self.duration_predictor.infer(duration_predictor_cond, masks=info)
            durations = torch.exp(log_duration_predictions) - 1
        if teacher_plan is not None:
            plan, pos_enc = teacher_plan["lr_plan"], teacher_plan["pos_enc"]
        else:
            plan = self.length_regulator.index(durations, max_len=max_out_len)
            pos_enc = self.position_term(plan, out_info)
        LR_length_rounded = plan[3]
        LR_text_outputs, _ = self.length_regulator(aug, durations, masks=out_info, plan=plan)
        if teacher_plan is not None and "LR_emo" in teacher_plan:
            LR_emo_outputs, LR_spk_outputs = teacher_plan["LR_emo"], teacher_plan["LR_spk"]
        else:
            LR_emo_outputs, _ = self.length_regulator(inputs_emo_embedding, durations, masks=out_info, plan=plan)
            LR_spk_outputs, _ = self.length_regulator(inputs_spk_embedding, durations, masks=out_info, plan=plan)
        LR_text_outputs = LR_text_outputs + pos_enc
        return (LR_text_outputs, LR_emo_outputs, LR_spk_outputs, LR_length_rounded, log_duration_predictions,
                pitch_predictions, energy_predictions)


class MelPNCADecoder(nn.Module):
    """Teacher-forced / free-running mel decoder (reference :503-612)."""

    def teacher_input(self, target, L):
        """go-frame followed by every r-th target frame, shifted by one decoder step (reference :556-559)."""
        B = target.size(0)
        input = torch.zeros((B, L, self.d_mel), device=target.device, dtype=target.dtype)
        input[:, 1:, :] = target[:, self.r - 1:: self.r, :][:, : L - 1, :]
        return input

    def __init__(self, config):
        super(MelPNCADecoder, self).__init__()
        prenet_units = config["decoder_prenet_units"]
        nb_layers = config["decoder_num_layers"]
        nb_heads = config["decoder_num_heads"]
        d_model = config["decoder_num_units"]
        d_head = d_model // nb_heads
        d_inner = config["decoder_ffn_inner_dim"]
        dropout = config["decoder_dropout"]
        dropout_attn = config["decoder_attention_dropout"]
        dropout_relu = config["decoder_relu_dropout"]
        outputs_per_step = config["outputs_per_step"]
        d_mem = (config["encoder_projection_units"] * outputs_per_step + config["emotion_units"]
                 + config["speaker_units"])
        d_mel = config["num_mels"]
        self.d_mel = d_mel
        self.r = outputs_per_step
        self.nb_layers = nb_layers
        self.mel_dec = HybridAttentionDecoder(d_mel, prenet_units, nb_layers, d_model, d_mem, nb_heads, d_head,
                                              d_inner, dropout, dropout_attn, dropout_relu, d_mel * outputs_per_step)
        # free-running decode: "loop" = one launch per op and step from Python (the round-1 path), "fused" = the
        # step function of decode_graph.py run eagerly, "graph" = that step captured in a hipGraph and replayed
        # "kernel" = the whole loop as one launch (ar_kernels.py; bf16 mode, falls back to "graph" / "loop")
        self.decode_mode = "loop"
        self._decode_cache = None
        self._decode_kernel = None

    def forward(self, memory, x_band_width, h_band_width, target=None, mask=None, return_attns=False, bw_dev=None,
                teacher_input=None, teacher_prenet=None):
        if target is None:
            return self._free_run(memory, x_band_width, h_band_width, mask, bw_dev)
        self.mel_dec.reset_state()
        input = teacher_input if teacher_input is not None else self.teacher_input(target, memory.size(1))
        return self.mel_dec(input, memory, x_band_width, h_band_width, mask=mask, return_attns=return_attns,
                            bw_dev=bw_dev, prenet_out=teacher_prenet)


    @torch.no_grad()
    def _free_run(self, memory, x_band_width, h_band_width, mask, bw_seq=None):
        """Free-running decode (reference :568-610): step t consumes the last mel frame of step t-1.  Outputs land
        in one preallocated (B, L, r*d_mel) buffer instead of a python list + torch.cat."""
        B, L = memory.size(0), memory.size(1)
        if self.decode_mode == "kernel":
            # the whole loop as ONE launch, a workgroup per sequence (csrc/ar_infer.hip); shapes / precisions it is not
            # compiled for fall through to the replayed graph (HIP device) or the per-op loop
            from kantts.models.sambert.ar_kernels import DecoderKernel
            from kantts.models.utils import SeqInfo as _SeqInfo

            host_bw = not torch.is_tensor(x_band_width) and not torch.is_tensor(h_band_width)
            bw_max = int(x_band_width) if host_bw else -1
            if host_bw and int(h_band_width) == bw_max and DecoderKernel.eligible(self.mel_dec, self.d_mel, bw_max):
                if self._decode_kernel is None:
                    self._decode_kernel = DecoderKernel(self.mel_dec, self.d_mel)
                info = _SeqInfo.of(mask)
                lens32 = None if info is None else info.lens32.clamp(max=L)
                return self._decode_kernel.run(memory, lens32, bw_seq, bw_max), [], []
        if (self.decode_mode in ("graph", "fused", "kernel") and memory.is_cuda) or self.decode_mode == "fused":
            # one decoder step = a static launch sequence with the step index in device memory (decode_graph.py);
            # "graph" replays it from a hipGraph captured per (B, L)
            from kantts.models.sambert.decode_graph import DecodeGraphCache
            from kantts.models.utils import SeqInfo as _SeqInfo

            if self._decode_cache is None:
                self._decode_cache = DecodeGraphCache()
            info = _SeqInfo.of(mask)
            lens32 = info.lens32 if info is not None else torch.full((B,), L, device=memory.device, dtype=torch.int32)
            bw = bw_seq if bw_seq is not None else torch.full((B,), int(x_band_width), device=memory.device,
                                                              dtype=torch.int32)
            # decoder lengths are bucketed to multiples of 16 steps so that a handful of captured graphs serve every
            # utterance length (steps past a sequence's length are masked rows, exactly like batch padding)
            Lp = (L + 15) // 16 * 16 if self.decode_mode != "fused" else L
            if Lp != L:
                memory = F.pad(memory, (0, 0, 0, Lp - L))
            fr = self._decode_cache.get(self.mel_dec, self.d_mel, self.r, B, Lp, memory.device)
            fr.load(memory, lens32.clamp(max=L), bw)
            return fr.run(graph=(self.decode_mode != "fused"))[:, :L].clone(), [], []
        self.mel_dec.reset_state()
        memory = memory.contiguous()
        out = torch.empty((B, L, self.d_mel * self.r), device=memory.device, dtype=torch.float32)
        frame = torch.zeros((B, 1, self.d_mel), device=memory.device, dtype=torch.float32)
        for step in range(L):
            o, _, _ = self.mel_dec.infer(step, frame, memory, x_band_width, h_band_width, mask=mask, bw_seq=bw_seq)
            out[:, step, :] = o[:, 0, :]
            frame = o[:, :, -self.d_mel:]
        return out, [], []


class PostNet(nn.Module):
    """FSMN -> LSTM -> Linear residual predictor (reference :615-649)."""

    def __init__(self, config):
        super(PostNet, self).__init__()
        self.filter_size = config["postnet_filter_size"]
        self.fsmn_num_layers = config["postnet_fsmn_num_layers"]
        self.num_memory_units = config["postnet_num_memory_units"]
        self.ffn_inner_dim = config["postnet_ffn_inner_dim"]
        self.dropout = config["postnet_dropout"]
        self.shift = config["postnet_shift"]
        self.lstm_units = config["postnet_lstm_units"]
        self.num_mels = config["num_mels"]
        self.fsmn = FsmnEncoderV2(self.filter_size, self.fsmn_num_layers, self.num_mels, self.num_memory_units,
                                  self.ffn_inner_dim, self.dropout, self.shift)
        self.lstm = nn.LSTM(self.num_memory_units, self.lstm_units, num_layers=1, batch_first=True)
        self.fc = nn.Linear(self.lstm_units, self.num_mels)

    def forward(self, x, mask=None, res=None, zero_rows=None):
        h = self.fsmn(x, mask)
        h = ops.lstm(h, [self.lstm.weight_ih_l0, self.lstm.weight_hh_l0, self.lstm.bias_ih_l0, self.lstm.bias_hh_l0])
        return ops.linear(h, self.fc.weight, self.fc.bias, res=res, rowmask=zero_rows)


def band_width_of(duration_targets, input_lengths, r):
    """x_band_width = h_band_width of a teacher-forced batch as a host integer (reference :981-985): int(max valid duration
    / r + 0.5).  On host tensors (a batch before its upload) this costs no device synchronisation."""
    T = duration_targets.size(1)
    valid = torch.arange(T, device=duration_targets.device)[None, :] < input_lengths[:, None]
    return int(float((duration_targets * valid).max()) / r + 0.5)


def average_frame_feat(pitch, durs):
    """Frame-level contour (B, F, T_mel) -> per-phoneme mean over each phoneme's NON-ZERO frames (B, F, N), 0 where a
    phoneme has none (reference :652-674).  Prefix sums + two gathers; durations are whole numbers stored as float."""
    ends = torch.cumsum(durs, dim=1).long()
    starts = F.pad(ends[:, :-1], (1, 0))
    n_form = pitch.size(1)
    e = ends[:, None, :].expand(-1, n_form, -1)
    s = starts[:, None, :].expand(-1, n_form, -1)
    count_ps = F.pad(torch.cumsum(pitch != 0.0, dim=2), (1, 0))
    value_ps = F.pad(torch.cumsum(pitch, dim=2), (1, 0))
    sums = (torch.gather(value_ps, 2, e) - torch.gather(value_ps, 2, s)).float()
    counts = (torch.gather(count_ps, 2, e) - torch.gather(count_ps, 2, s)).float()
    return torch.where(counts == 0.0, counts, sums / counts)


class KanTtsSAMBERT(nn.Module):
    """SAM-BERT acoustic model (reference :712-1044)."""

    def __init__(self, config):
        super(KanTtsSAMBERT, self).__init__()
        self.text_encoder = TextFftEncoder(config)
        self.se_enable = config.get("SE", False)
        if not self.se_enable:  # SE: the speaker stream arrives as per-token embedding vectors (reference :717-720, :928)
            self.spk_tokenizer = nn.Embedding(config["speaker"], config["speaker_units"])
        self.emo_tokenizer = nn.Embedding(config["emotion"], config["emotion_units"])
        self.variance_adaptor = VarianceAdaptor(config)
        self.mel_decoder = MelPNCADecoder(config)
        self.mel_postnet = PostNet(config)
        # True: the target-only part of a teacher-forced step (teacher_forced_plan) is computed inline, where the reference
        # computes it, instead of beside the encoder -- same arithmetic (tests/test_host_logic_emulated.py)
        self.inline_teacher_plan = False
        self.MAS = False
        if config.get("MAS", False):
            self.MAS = True
            self.text_encoder.ling_embedding_grad = True
            self.align_attention = ConvAttention(n_mel_channels=config["num_mels"],
                                                 n_text_channels=config["embedding_dim"],
                                                 n_att_channels=config["num_mels"])
        self.fp_enable = config.get("FP", False)
        if self.fp_enable:
            raise NotImplementedError("filled-pause predictor is outside the hot path")
        # the reference always returns 8 + 24 dense attention maps; here opt-in
        self.return_attns = False
        # True: the band width stays in device memory (no host sync) so that a whole training step can be
        # captured in a hipGraph; res["x_band_width"] is then a 1-element int32 tensor instead of an int
        self.device_band_width = False
        # with ``device_band_width``: an upper bound of the band width of the batches this module will see, promised by a
        # caller that knows them on the host (train/graph_step.py; ``band_width_of``); None = unknown.  It lets the decoder
        # choose the one-launch PNCA block (band <= 16) without reading the device value; that launch poisons its output
        # with NaN if the promise is broken.
        self.band_width_bound = None
        # inference: text encoder + variance adaptor (+ duration loop) in fp32 whatever the precision mode, so that the
        # index tensors derived from them are the reference's (forward(); False = round 5's all-bf16 front)
        self.infer_front_fp32 = True

    def get_lfr_mask_from_lengths(self, lengths, max_len):
        """ceil(len / r) valid decoder steps (reference :736-750, vectorised: no per-item .item())."""
        r = self.mel_decoder.r
        return get_mask_from_lengths((lengths + r - 1) // r, max_len=max_len // r)

    def binarize_attention_parallel(self, attn, in_lens, out_lens):
        """Hard alignment by monotonic alignment search; no gradient (reference :752-764).  The reference moves the
        map to the host, runs numba and copies it back; this stays on the device (csrc/mas.hip)."""
        with torch.no_grad():
            return b_mas(attn, in_lens, out_lens, width=1)

    @torch.no_grad()
    def teacher_forced_plan(self, in_info, output_lengths, mel_targets, duration_targets):
        """Everything of a teacher-forced step that depends on the TARGETS only -- output / LFR masks, the length-regulator
        index, duration positions and their sin / cos, the duration predictor's shifted input, the attention band width,
        the decoder's teacher-forcing frames: ~40 small launches that the reference (and rounds 1-2 here) issue between the
        encoder and the decoder.  ``forward`` runs them on a second stream BESIDE the encoder (ops.run_beside)."""
        r = self.mel_decoder.r
        va = self.variance_adaptor
        max_out_len = mel_targets.size(1)
        out_info = SeqInfo(output_lengths, max_out_len)
        lr_plan = va.length_regulator.index(duration_targets, max_len=max_out_len)
        Tp, max_len = lr_plan[4], lr_plan[5]
        lr_plan = tuple(lr_plan) + (torch.clamp(out_info.lens64, max=max_len),)
        plan = {"out_info": out_info, "lr_plan": lr_plan, "pos_enc": va.position_term(lr_plan, out_info),
                "prev": torch.log(F.pad(duration_targets[:, :-1].float(), (1, 0)) + 1).unsqueeze(-1),
                "lfr_info": SeqInfo((output_lengths + r - 1) // r, Tp // r),
                "bw_val": duration_targets.float().masked_fill(in_info.mask, 0).max() / r + 0.5,
                "dec_input": self.mel_decoder.teacher_input(mel_targets, Tp // r)}
        if self.device_band_width:
            plan["bw_dev"] = plan["bw_val"].to(torch.int32).reshape(1)  # trunc == int() for non-negative values
        return plan

    @torch.no_grad()
    def teacher_forced_plan_fused(self, input_lengths, T_in, output_lengths, mel_targets, duration_targets):
        """``SeqInfo(input_lengths, T_in)`` and ``teacher_forced_plan`` from TWO launches (kantts_lr_index, then
        kantts_teacher_plan) instead of ~50 stock elementwise ones: in a replayed hipGraph those ran one after the other in
        front of the text encoder, 0.37 ms of 2-5 us kernels and dependency gaps per step."""
        from kantts._hip import teacher_plan

        r = self.mel_decoder.r
        va = self.variance_adaptor
        max_out_len = mel_targets.size(1)
        idx, pos, cs, lens, Tp, max_len = va.length_regulator.index(duration_targets, max_len=max_out_len)
        o = teacher_plan(input_lengths.to(torch.int64).contiguous(), output_lengths.to(torch.int64).contiguous(),
                         duration_targets.contiguous(), mel_targets.contiguous(), pos,
                         va.dur_position_encoder.inv_timescales.detach(), Tp, max_len, r)
        in_info = SeqInfo.from_parts(o["in_mask"], o["in_l64"], o["in_l32"])
        plan = {"out_info": SeqInfo.from_parts(o["out_mask"], o["out_l64"], o["out_l32"]),
                "lr_plan": (idx, pos, cs, lens, Tp, max_len, o["valid"]), "pos_enc": o["pos_enc"], "prev": o["prev"],
                "lfr_info": SeqInfo.from_parts(o["lfr_mask"], o["lfr_l64"], o["lfr_l32"]), "bw_val": o["bw_val"],
                "dec_input": o["dec_input"]}
        if self.device_band_width:
            plan["bw_dev"] = o["bw_dev"]
        return in_info, plan

    def _embed_emo_spk(self, inputs_emotion, inputs_speaker):
        (emo_hid, _) = ops.embed_sum(inputs_emotion, [self.emo_tokenizer.weight])
        if self.se_enable:
            spk_hid = inputs_speaker.to(torch.float32)
        else:
            (spk_hid, _) = ops.embed_sum(inputs_speaker, [self.spk_tokenizer.weight])
        return emo_hid, spk_hid

    def _beside_encoder(self, in_info, output_lengths, mel_targets, duration_targets, inputs_emotion, inputs_speaker,
                        pitch_targets, energy_targets, base_plan=None):
        """What a teacher-forced step can do while the text encoder runs: the target-only plan (no gradient) and the
        emotion / speaker embeddings with their length regulation (they read the inputs and two embedding tables only).
        Autograd replays a node on the stream of its forward op, so the embedding-table gradients -- 31 us of atomics each,
        at the very end of the main stream's backward before -- also run beside the encoder's backward."""
        plan = base_plan if base_plan is not None else self.teacher_forced_plan(in_info, output_lengths, mel_targets,
                                                                                duration_targets)
        parts = _BESIDE_PARTS
        if "emb" in parts:
            emo_hid, spk_hid = self._embed_emo_spk(inputs_emotion, inputs_speaker)
            lr = self.variance_adaptor.length_regulator
            plan["emo_hid"], plan["spk_hid"] = emo_hid, spk_hid
            plan["LR_emo"], _ = lr(emo_hid, duration_targets, masks=plan["out_info"], plan=plan["lr_plan"])
            plan["LR_spk"], _ = lr(spk_hid, duration_targets, masks=plan["out_info"], plan=plan["lr_plan"])
        if "pe" in parts:
            plan["pe_emb"] = self.variance_adaptor.pitch_energy_embedding(pitch_targets, energy_targets)
        if "prenet" in parts:
            plan["dec_prenet"] = self.mel_decoder.mel_dec.prenet(plan["dec_input"])  # reads the target frames only
        return plan

    def forward(self, inputs_ling, inputs_emotion, inputs_speaker, input_lengths, output_lengths=None,
                mel_targets=None, duration_targets=None, pitch_targets=None, energy_targets=None, attn_priors=None,
                fp_label=None):
        batch_size = inputs_ling.size(0)
        r = self.mel_decoder.r
        T_in = inputs_ling.size(1)
        is_training = mel_targets is not None
        tplan = base_plan = in_info = None
        teacher = (is_training and not self.MAS and not getattr(self, "inline_teacher_plan", False)
                   and output_lengths is not None and duration_targets is not None and pitch_targets is not None
                   and energy_targets is not None)
        if (teacher and _PLAN_KERNEL and duration_targets.dtype == torch.int64 and mel_targets.dtype == torch.float32
                and mel_targets.dim() == 3 and inputs_ling.numel() > 0):
            in_info, base_plan = self.teacher_forced_plan_fused(input_lengths, T_in, output_lengths, mel_targets,
                                                                duration_targets)
        if in_info is None:
            in_info = SeqInfo(input_lengths, T_in)
        # Inference: the token-level front (text encoder, embeddings, variance adaptor incl. the duration loop) runs fp32
        # in EVERY precision mode.  Its outputs become index tensors -- durations = int(exp(log_dur) - 1 + 0.5), regulated
        # lengths, band widths (reference :455-460, :989-993) -- which north_star wants bit-exact; in bf16 it flipped 0.2 %
        # of the durations = 3 of 32 utterance lengths.  It is T_in <= ~80 tokens: a few % of an utterance's time.
        front = hip_rt.precision_scope("fp32" if (not is_training and self.infer_front_fp32) else None)
        with front:
            if teacher:
                (text_hid, enc_sla_attn_lst, ling_embedding), tplan = ops.run_beside(
                    lambda: self.text_encoder(inputs_ling, in_info, self.return_attns),
                    lambda: self._beside_encoder(in_info, output_lengths, mel_targets, duration_targets, inputs_emotion,
                                                 inputs_speaker, pitch_targets, energy_targets, base_plan=base_plan),
                    side_inputs=(output_lengths, mel_targets, duration_targets, in_info.mask, in_info.lens64, inputs_emotion,
                                 inputs_speaker, pitch_targets, energy_targets))
            else:
                text_hid, enc_sla_attn_lst, ling_embedding = self.text_encoder(inputs_ling, in_info, self.return_attns)
            # backward: once the gradient reaches the encoder output, the weight gradients of everything downstream start on
            # the side stream, beside the encoder's own backward (no-op unless weight gradients are deferred)
            text_hid = ops.wgrad_flush_point(text_hid)
            inter_lengths = input_lengths
            attn_soft = attn_hard = attn_logprob = None
            if self.MAS and is_training:
                # Monotonic-Alignment-Search (reference :901-925): soft attention mel <-> (scaled) linguistic embedding,
                # hard path by DP, durations = frames per phoneme, frame-level pitch / energy averaged per phoneme
                attn_soft, attn_logprob = self.align_attention.forward_cl(mel_targets, ling_embedding, input_lengths,
                                                                          attn_priors)
                attn_hard = self.binarize_attention_parallel(attn_soft, input_lengths, output_lengths)
                duration_targets = attn_hard.sum(2)[:, 0, :]
                pitch_targets = average_frame_feat(pitch_targets.unsqueeze(1), duration_targets).squeeze(1)
                energy_targets = average_frame_feat(energy_targets.unsqueeze(1), duration_targets).squeeze(1)
                # the slot after the last phoneme absorbs the r-padding so that durations sum to the padded mel length
                # (reference loop :921-924, vectorised: no per-item .item())
                pad = (mel_targets.size(1) - output_lengths).to(duration_targets.dtype)
                duration_targets.scatter_(1, input_lengths.view(-1, 1), pad.view(-1, 1))
            if tplan is not None and "emo_hid" in tplan:
                emo_hid, spk_hid = tplan["emo_hid"], tplan["spk_hid"]
            else:
                emo_hid, spk_hid = self._embed_emo_spk(inputs_emotion, inputs_speaker)
            out_info = None
            max_out_len = None
            if output_lengths is not None:
                max_out_len = mel_targets.size(1)
                out_info = tplan["out_info"] if tplan is not None else SeqInfo(output_lengths, max_out_len)
            (LR_text_outputs, LR_emo_outputs, LR_spk_outputs, LR_length_rounded, log_duration_predictions,
             pitch_predictions, energy_predictions) = self.variance_adaptor(
                text_hid, emo_hid, spk_hid, masks=in_info, output_masks=out_info, duration_targets=duration_targets,
                pitch_targets=pitch_targets, energy_targets=energy_targets, max_out_len=max_out_len, teacher_plan=tplan)
        Tp = LR_text_outputs.size(1)
        if tplan is not None:
            lfr_info = tplan["lfr_info"]
        elif output_lengths is not None:
            lfr_info = SeqInfo((output_lengths + r - 1) // r, Tp // r)
        else:
            out_info = SeqInfo(LR_length_rounded, Tp)
            # the reference infers one utterance at a time (mask None); in a batch every sequence keeps its own
            # decoder length and band width so that batched inference equals per-utterance inference
            lfr_info = SeqInfo((LR_length_rounded + r - 1) // r, Tp // r)
        # LFR: group r frames; memory = [text (r*d) | spk of the first frame | emo of the first frame]
        d_t, d_s, d_e = text_hid.shape[-1], spk_hid.shape[-1], emo_hid.shape[-1]
        memory = torch.cat([
            LR_text_outputs.reshape(batch_size, -1, r * d_t),
            LR_spk_outputs.reshape(batch_size, -1, r * d_s)[:, :, :d_s],
            LR_emo_outputs.reshape(batch_size, -1, r * d_e)[:, :, :d_e],
        ], dim=-1)
        if tplan is not None:
            bw_val = tplan["bw_val"]
        elif duration_targets is not None:
            bw_val = duration_targets.float().masked_fill(in_info.mask, 0).max() / r + 0.5
        else:
            bw_val = (torch.exp(log_duration_predictions) - 1).max() / r + 0.5
        bw_dev = None
        if duration_targets is None:
            pred = (torch.exp(log_duration_predictions) - 1)
            bw_dev = (pred.max(dim=1).values / r + 0.5).to(torch.int32).contiguous()  # per sequence (free-running)
        elif mel_targets is None:
            # inference with the durations given (no mel targets: the decoder free-runs): every sequence keeps its own band
            # width, as in the free-running case above, so that a batch equals its utterances inferred one at a time
            bw_dev = (duration_targets.float().masked_fill(in_info.mask, 0).max(dim=1).values / r + 0.5).to(
                torch.int32).contiguous()
        if self.device_band_width and duration_targets is not None and mel_targets is not None:
            bw_dev = tplan["bw_dev"] if (tplan is not None and "bw_dev" in tplan) else bw_val.to(torch.int32).reshape(1)
            x_band_width = h_band_width = bw_dev
            bw_int = 0
        else:
            x_band_width = h_band_width = bw_int = int(bw_val)  # host sync, as in the reference (:981-993)
        from kantts._hip import ops_bf16

        bound_before = ops_bf16.BAND_BOUND["max"]
        ops_bf16.BAND_BOUND["max"] = self.band_width_bound if (self.device_band_width and mel_targets is not None) else None
        try:
            dec_outputs, pnca_x_attn_lst, pnca_h_attn_lst = self.mel_decoder(
                memory, bw_int, bw_int, target=mel_targets, mask=lfr_info, return_attns=self.return_attns, bw_dev=bw_dev,
                teacher_input=None if tplan is None else tplan["dec_input"],
                teacher_prenet=None if tplan is None else tplan.get("dec_prenet"))
        finally:
            ops_bf16.BAND_BOUND["max"] = bound_before
        dec_outputs = dec_outputs.reshape(batch_size, -1, self.mel_decoder.d_mel)
        rows = out_info.mask
        if rows.size(1) != dec_outputs.size(1):
            rows = F.pad(rows, (0, dec_outputs.size(1) - rows.size(1)), value=True)
        dec_outputs = ops.wgrad_flush_point(dec_outputs.masked_fill(rows.unsqueeze(-1), 0))  # postnet weight gradients
        post_info = out_info if out_info.mask.size(1) == dec_outputs.size(1) else SeqInfo(out_info.lens64,
                                                                                         dec_outputs.size(1))
        deferred = getattr(self.variance_adaptor, "deferred", None)
        if deferred is not None and not _PREDICTORS_AFTER_POSTNET:
            log_duration_predictions, pitch_predictions, energy_predictions = deferred()
        # postnet residual add + final masking ride in the epilogue of the last GEMM
        postnet_outputs = self.mel_postnet(dec_outputs, post_info, res=dec_outputs, zero_rows=post_info.mask)
        if deferred is not None and _PREDICTORS_AFTER_POSTNET:
            log_duration_predictions, pitch_predictions, energy_predictions = deferred()
        self.variance_adaptor.deferred = None
        ops.side_branch.join()  # the predictors' outputs are read from here on (losses)
        res = {
            "x_band_width": x_band_width,
            "h_band_width": h_band_width,
            "enc_slf_attn_lst": enc_sla_attn_lst,
            "pnca_x_attn_lst": pnca_x_attn_lst,
            "pnca_h_attn_lst": pnca_h_attn_lst,
            "dec_outputs": dec_outputs,
            "postnet_outputs": postnet_outputs,
            "LR_length_rounded": LR_length_rounded,
            "log_duration_predictions": log_duration_predictions,
            "pitch_predictions": pitch_predictions,
            "energy_predictions": energy_predictions,
            "duration_targets": duration_targets,
            "pitch_targets": pitch_targets,
            "energy_targets": energy_targets,
            "fp_predictions": None,
            "valid_inter_lengths": inter_lengths,
        }
        if bw_dev is not None and bw_dev.numel() == batch_size and mel_targets is None:
            # batched inference: every sequence keeps the band width its own utterance would have had (an extra key; the
            # reference infers one utterance at a time, where x_band_width is this value)
            res["band_width_per_sequence"] = bw_dev
        res["LR_text_outputs"] = LR_text_outputs
        res["LR_emo_outputs"] = LR_emo_outputs
        res["LR_spk_outputs"] = LR_spk_outputs
        res["ling_embedding"] = ling_embedding
        if self.MAS and is_training:
            res["attn_soft"] = attn_soft
            res["attn_hard"] = attn_hard
            res["attn_logprob"] = attn_logprob
        return res


class KanTtsTextsyBERT(nn.Module):
    """Masked-LM pre-training wrapper.  Broken at the reference HEAD (SURVEY section 2 row 18:
    forward unpacks 2 of 3 encoder outputs); kept constructible for state_dict parity only."""

    def __init__(self, config):
        super(KanTtsTextsyBERT, self).__init__()
        self.text_encoder = TextFftEncoder(config)
        delattr(self.text_encoder, "ling_proj")
        self.fc = nn.Linear(self.text_encoder.d_model, config["sy"])

    def forward(self, inputs_ling, input_lengths):
        info = SeqInfo(input_lengths, inputs_ling.size(1))
        text_hid, attns, _ = self.text_encoder(inputs_ling, info, return_attns=False)
        return {"logits": ops.linear(text_hid, self.fc.weight, self.fc.bias), "enc_slf_attn_lst": attns}
