"""TEST INFRASTRUCTURE -- CPU restatement (plain PyTorch fp32, functional, no nn.Module tree) of the
reference's hot path.  NOT product code: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it.  The product (kan-tts_amd/kantts) never does.

Every function cites the reference file:line it restates (paths relative to /root/reference).
The restatement is pinned against the untouched reference executed in the build container:
``oracle/make_golden.py`` dumps reference outputs to tests/golden/*.pt and
tests/test_oracle_golden.py checks this file against them (and, when /root/reference is present,
``python oracle/check_vs_reference.py`` compares live).

All functions take ``P``: a dict name -> tensor holding a reference ``state_dict`` (same key names),
so autograd through this file yields reference-equivalent parameter gradients.
Dropout: parity runs force every dropout p to 0 (SURVEY.md section 7 "hard parts"); ``DROP["on"] = True`` switches
F.dropout on at the reference's sites (same p, torch's own generator) -- used only by bench.py's cpu_baseline leg to
time the "as shipped" training mode, never for parity.
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- small helpers
DROP = {"on": False}


def _drop(x, p):
    return F.dropout(x, p, True) if (DROP["on"] and p > 0) else x


def pad_mask(lengths, max_len):
    """True = padded.  kantts/models/utils.py:13-23"""
    return torch.arange(max_len, device=lengths.device)[None, :] >= lengths[:, None]


def _ln(x, P, pre, eps=1e-6):
    return F.layer_norm(x, (x.shape[-1],), P[pre + ".weight"], P[pre + ".bias"], eps)


def _linear(x, P, pre, bias=True):
    return F.linear(x, P[pre + ".weight"], P[pre + ".bias"] if bias else None)


def _split_heads(t, n_head):
    # (B, L, H*d) -> (H*B, L, d), head-major.  kantts/models/sambert/__init__.py:85-91
    B, L, HD = t.shape
    return t.view(B, L, n_head, HD // n_head).permute(2, 0, 1, 3).reshape(n_head * B, L, HD // n_head)


def _merge_heads(t, n_head):
    # (H*B, L, d) -> (B, L, H*d).  kantts/models/sambert/__init__.py:97-100
    HB, L, d = t.shape
    B = HB // n_head
    return t.view(n_head, B, L, d).permute(1, 2, 0, 3).reshape(B, L, n_head * d)


def _attend(q, k, v, mask, dropatt=0.0):
    """softmax(q k^T / sqrt(d) masked with -inf) v.  kantts/models/sambert/__init__.py:17-29"""
    s = torch.bmm(q, k.transpose(1, 2)) / math.sqrt(q.shape[-1])
    if mask is not None:
        s = s.masked_fill(mask, float("-inf"))
    p = _drop(torch.softmax(s, dim=2), dropatt)
    return torch.bmm(p, v), p


# ----------------------------------------------------------------------------- encoder
def self_attention(P, pre, x, key_pad, n_head, dropout=0.0, dropatt=0.0):
    """MultiHeadSelfAttention.forward, kantts/models/sambert/__init__.py:74-106"""
    B, L, d_in = x.shape
    h = _ln(x, P, pre + ".layer_norm")
    q, k, v = _linear(h, P, pre + ".w_qkv").chunk(3, -1)
    q, k, v = (_split_heads(t, n_head) for t in (q, k, v))
    mask = None
    if key_pad is not None:
        mask = key_pad[:, None, :].expand(-1, L, -1).repeat(n_head, 1, 1)
    o, p = _attend(q, k, v, mask, dropatt)
    o = _drop(_linear(_merge_heads(o, n_head), P, pre + ".fc"), dropout)
    if o.shape[-1] == d_in:
        o = o + x
    return o, p


def conv_ffn(P, pre, x, pad, dropout=0.0, dropout_inner=0.0):
    """PositionwiseConvFeedForward.forward, kantts/models/sambert/__init__.py:134-149"""
    h = _ln(x, P, pre + ".layer_norm").transpose(1, 2)
    w1 = P[pre + ".w_1.weight"]
    h = F.relu(F.conv1d(h, w1, P[pre + ".w_1.bias"], padding=(w1.shape[-1] - 1) // 2))
    if pad is not None:
        h = h.masked_fill(pad[:, None, :], 0)
    h = _drop(h, dropout_inner)
    w2 = P[pre + ".w_2.weight"]
    h = _drop(F.conv1d(h, w2, P[pre + ".w_2.bias"], padding=(w2.shape[-1] - 1) // 2), dropout)
    return h.transpose(1, 2) + x


def _zero_pad_rows(x, pad):
    return x if pad is None else x.masked_fill(pad[:, :, None], 0)


def sinusoid_table(n_position, d_hid):
    """SinusoidalPositionEncoder.get_sinusoid_encoding_table, kantts/models/sambert/positions.py:33-55
    (positions are 1-based, divisor is d_hid/2-1, first half sin / second half cos)."""
    import numpy as np

    pos = np.arange(1, n_position + 1, dtype=np.float64)[:, None]
    j = np.arange(d_hid // 2, dtype=np.float64)[None, :]
    ang = pos / np.power(10000.0, j / float(d_hid / 2 - 1))
    tab = np.concatenate([np.sin(ang), np.cos(ang)], axis=1)
    return torch.tensor(tab, dtype=torch.float32)


def text_encoder(P, cfg, inputs_ling, pad, pre="text_encoder"):
    """TextFftEncoder.forward + SelfAttentionEncoder.forward,
    kantts/models/sambert/kantts_sambert.py:308-337 and :61-87"""
    emb = (
        F.embedding(inputs_ling[:, :, 0], P[pre + ".sy_emb.weight"])
        + F.embedding(inputs_ling[:, :, 1], P[pre + ".tone_emb.weight"])
        + F.embedding(inputs_ling[:, :, 2], P[pre + ".syllable_flag_emb.weight"])
        + F.embedding(inputs_ling[:, :, 3], P[pre + ".ws_emb.weight"])
    )
    d_model = cfg["encoder_num_units"]
    emb = emb * d_model ** 0.5  # in-place in the reference: the returned ling_embedding is scaled
    L = emb.shape[1]
    x = _drop(emb + P[pre + ".ling_enc.position_enc.position_enc"][:, :L, :], cfg["encoder_dropout"])
    attns = []
    for i in range(cfg["encoder_num_layers"]):
        lp = "%s.ling_enc.fft.%d" % (pre, i)
        x, p = self_attention(P, lp + ".slf_attn", x, pad, cfg["encoder_num_heads"], cfg["encoder_dropout"],
                              cfg["encoder_attention_dropout"])
        x = _zero_pad_rows(x, pad)
        x = _zero_pad_rows(conv_ffn(P, lp + ".pos_ffn", x, pad, cfg["encoder_dropout"], cfg["encoder_relu_dropout"]),
                           pad)
        attns.append(p)
    x = _ln(x, P, pre + ".ling_enc.ln")
    return F.linear(x, P[pre + ".ling_proj.weight"]), attns, emb


# ----------------------------------------------------------------------------- FSMN / LSTM
def fsmn_encoder(P, pre, x, pad, n_layers, filter_size, shift, dropout=0.0):
    """FsmnEncoderV2 / FeedForwardNet / MemoryBlockV2, kantts/models/sambert/fsmn.py:8-124"""
    lp = int(round((filter_size - 1) / 2))
    rp = int((filter_size - 1) / 2)
    if shift > 0:
        lp, rp = lp + shift, rp - shift
    x = _drop(x, dropout)
    for i in range(n_layers):
        f = "%s.ffn_lst.%d" % (pre, i)
        ctx = _drop(F.relu(F.conv1d(x.transpose(1, 2), P[f + ".w_1.weight"], P[f + ".w_1.bias"])), dropout)
        ctx = F.conv1d(ctx, P[f + ".w_2.weight"]).transpose(1, 2)
        ctx = _zero_pad_rows(ctx, pad)
        w = P["%s.memory_block_lst.%d.conv_dw.weight" % (pre, i)]
        mem = F.conv1d(F.pad(ctx.transpose(1, 2), (lp, rp)), w, groups=w.shape[0]).transpose(1, 2)
        mem = _drop(_zero_pad_rows(_drop(mem + ctx, dropout), pad), dropout)
        x = mem + x if mem.shape[-1] == x.shape[-1] else mem
    return x


def lstm_layer(x, w_ih, w_hh, b_ih, b_hh, lengths=None, reverse=False):
    """One LSTM direction, batch_first, zero initial state; gate order i,f,g,o (torch.nn.LSTM).
    With ``lengths`` it reproduces pack_padded_sequence/pad_packed_sequence semantics
    (kantts/models/sambert/adaptors.py:126-134): row b only runs over t < lengths[b]
    (the reverse direction starts at t = lengths[b]-1) and padded outputs are zero."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    gx = F.linear(x, w_ih, b_ih + b_hh)
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    outs = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        g = gx[:, t] + F.linear(h, w_hh)
        i, f, gg, o = g.chunk(4, -1)
        c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h_new = torch.sigmoid(o) * torch.tanh(c_new)
        if lengths is not None:
            live = (t < lengths)[:, None]
            c = torch.where(live, c_new, c)
            h = torch.where(live, h_new, h)
            outs[t] = torch.where(live, h_new, torch.zeros_like(h_new))
        else:
            c, h = c_new, h_new
            outs[t] = h_new
    return torch.stack(outs, dim=1)


def _lstm_named(P, pre, x, layer=0, lengths=None, reverse=False):
    """One direction of one layer through the readable time loop above (kept as the DEFINITION: tests/test_oracle_golden.py
    checks ``lstm_aten`` against it)."""
    sfx = "_l%d%s" % (layer, "_reverse" if reverse else "")
    return lstm_layer(
        x, P[pre + ".weight_ih" + sfx], P[pre + ".weight_hh" + sfx],
        P[pre + ".bias_ih" + sfx], P[pre + ".bias_hh" + sfx], lengths, reverse,
    )


def lstm_aten(P, pre, x, num_layers=1, bidirectional=False, lengths=None):
    """nn.LSTM(batch_first=True) forward on the weights in ``P`` through the SAME ATen kernel the reference's modules call
    (``torch._VF.lstm``: nn.LSTM.forward, torch/nn/modules/rnn.py), zero initial state, no inter-layer dropout.  With
    ``lengths``: packed exactly where the reference packs (kantts/models/sambert/adaptors.py:126-134:
    pack_padded_sequence(enforce_sorted=False) -> LSTM -> pad_packed_sequence(total_length=T)).  Round 6: the time loop in
    ``lstm_layer`` made this oracle 2.2-2.4x slower than the reference it stands in for as ``cpu_baseline``; the values are
    the same (tests/test_oracle_golden.py::test_lstm_aten_equals_the_time_loop)."""
    dirs = 2 if bidirectional else 1
    flat = []
    for l in range(num_layers):
        for d in range(dirs):
            sfx = "_l%d%s" % (l, "_reverse" if d else "")
            flat += [P[pre + ".weight_ih" + sfx], P[pre + ".weight_hh" + sfx], P[pre + ".bias_ih" + sfx], P[pre + ".bias_hh" + sfx]]
    B, T, _ = x.shape
    H = flat[1].shape[1]
    z = x.new_zeros(num_layers * dirs, B, H)
    if lengths is None:
        return torch._VF.lstm(x, (z, z), flat, True, num_layers, 0.0, False, bidirectional, True)[0]
    packed = torch.nn.utils.rnn.pack_padded_sequence(x, lengths.tolist(), batch_first=True, enforce_sorted=False)
    # the packed sequence is in sorted order: so is the (zero) initial state -- nothing to permute
    y = torch._VF.lstm(packed.data, packed.batch_sizes, (z, z), flat, True, num_layers, 0.0, False, bidirectional)[0]
    y = torch.nn.utils.rnn.PackedSequence(y, packed.batch_sizes, packed.sorted_indices, packed.unsorted_indices)
    return torch.nn.utils.rnn.pad_packed_sequence(y, batch_first=True, total_length=T)[0]


# ----------------------------------------------------------------------------- variance adaptor
def nar_predictor(P, cfg, pre, x, pad):
    """VarFsmnRnnNARPredictor.forward, kantts/models/sambert/adaptors.py:118-141"""
    lengths = None if pad is None else (~pad).sum(1)
    h = fsmn_encoder(P, pre + ".fsmn", x, pad, cfg["predictor_fsmn_num_layers"],
                     cfg["predictor_filter_size"], cfg["predictor_shift"], cfg["predictor_dropout"])
    y = _linear(lstm_aten(P, pre + ".blstm", h, 1, True, lengths), P, pre + ".fc").squeeze(-1)
    return y if pad is None else y.masked_fill(pad, 0.0)


def prenet(P, pre, x, n_hidden, has_out):
    """Prenet.forward (dropout disabled), kantts/models/sambert/__init__.py:32-49.
    Linear layers sit at Sequential indices 0, 3, 6 ... because each is followed by ReLU, Dropout."""
    for i in range(n_hidden):
        x = _drop(F.relu(_linear(x, P, "%s.fcs.%d" % (pre, 3 * i))), 0.5)
    if has_out:
        x = _linear(x, P, "%s.fcs.%d" % (pre, 3 * n_hidden))
    return x


def ar_duration_predictor(P, cfg, pre, inputs, cond, pad):
    """VarRnnARPredictor.forward (teacher forced), kantts/models/sambert/adaptors.py:53-65"""
    x = torch.cat([prenet(P, pre + ".prenet", inputs, len(cfg["dur_pred_prenet_units"]), False), cond], -1)
    x = lstm_aten(P, pre + ".lstm", x, 2)
    x = F.relu(_linear(x, P, pre + ".fc").squeeze(-1))
    return x if pad is None else x.masked_fill(pad, 0.0)


def ar_duration_infer(P, cfg, pre, cond, pad):
    """VarRnnARPredictor.infer, kantts/models/sambert/adaptors.py:67-83 (scalar fed back per step)."""
    B, T, _ = cond.shape
    H = cfg["dur_pred_lstm_units"]
    hs = [cond.new_zeros(B, H) for _ in range(2)]
    cs = [cond.new_zeros(B, H) for _ in range(2)]
    x = cond.new_zeros(B, 1)
    outs = []
    for t in range(T):
        z = torch.cat([prenet(P, pre + ".prenet", x, len(cfg["dur_pred_prenet_units"]), False), cond[:, t]], -1)
        for l in range(2):
            g = (F.linear(z, P["%s.lstm.weight_ih_l%d" % (pre, l)], P["%s.lstm.bias_ih_l%d" % (pre, l)])
                 + F.linear(hs[l], P["%s.lstm.weight_hh_l%d" % (pre, l)], P["%s.lstm.bias_hh_l%d" % (pre, l)]))
            i, f, gg, o = g.chunk(4, -1)
            cs[l] = torch.sigmoid(f) * cs[l] + torch.sigmoid(i) * torch.tanh(gg)
            hs[l] = torch.sigmoid(o) * torch.tanh(cs[l])
            z = hs[l]
        x = F.relu(_linear(z, P, pre + ".fc"))  # (B,1)
        outs.append(x)
    y = torch.cat(outs, -1)
    return y if pad is None else y.masked_fill(pad, 0.0)


def expand_index(durations, r):
    """Index form of LengthRegulator / DurSinusoidalPositionEncoder
    (kantts/models/sambert/adaptors.py:15-36, positions.py:72-90).
    Returns token index per frame (B,T) (-1 where no token covers the frame), output lens,
    within-token 1-based position (float), T = max(lens) right-padded to a multiple of r.
    Bit-exact w.r.t. the reference's dense one-hot matmul (one-hot rows select a single input)."""
    reps = (durations + 0.5).long()
    lens = reps.sum(1)
    T = int(lens.max())
    cs = torch.cumsum(reps, 1)
    start = cs - reps
    t = torch.arange(T, device=durations.device)[None, :, None]
    hit = (start[:, None, :] <= t) & (cs[:, None, :] > t)  # (B,T,N)
    any_hit = hit.any(-1)
    idx = torch.where(any_hit, hit.float().argmax(-1), torch.full_like(any_hit, -1, dtype=torch.long))
    off = torch.gather(start, 1, idx.clamp_min(0))
    pos = torch.where(any_hit, (t[:, :, 0] - off + 1).float(), (t[:, :, 0] + 1).float().expand_as(off))
    Tp = T + ((r - T % r) % r)
    return idx, lens, pos, T, Tp


def length_regulate(x, idx, out_pad, Tp):
    """LengthRegulator.forward, kantts/models/sambert/adaptors.py:15-36"""
    B, T = idx.shape
    g = torch.gather(x, 1, idx.clamp_min(0)[:, :, None].expand(-1, -1, x.shape[-1]))
    g = g * (idx >= 0)[:, :, None].to(x.dtype)
    if out_pad is not None:
        g = g.masked_fill(out_pad[:, :T, None], 0.0)
    return F.pad(g, (0, 0, 0, Tp - T))


def dur_position_embedding(P, pre, pos, out_pad, Tp):
    """DurSinusoidalPositionEncoder.forward, kantts/models/sambert/positions.py:72-98"""
    T = pos.shape[1]
    if out_pad is not None:
        pos = pos.masked_fill(out_pad[:, :T], 0.0)
    pos = F.pad(pos, (0, Tp - T))
    e = pos[:, :, None] / P[pre + ".inv_timescales"][None, None, :]
    out = torch.empty_like(e)
    out[:, :, 0::2] = torch.sin(e[:, :, 0::2])
    out[:, :, 1::2] = torch.cos(e[:, :, 1::2])
    return out


def variance_adaptor(P, cfg, text, emo, spk, pad, out_pad, dur_t, pitch_t, energy_t, pre="variance_adaptor"):
    """VarianceAdaptor.forward, kantts/models/sambert/kantts_sambert.py:396-500"""
    r = cfg["outputs_per_step"]
    vin = torch.cat([text, spk, emo], -1)
    pitch_p = nar_predictor(P, cfg, pre + ".pitch_predictor", vin, pad)
    energy_p = nar_predictor(P, cfg, pre + ".energy_predictor", vin, pad)
    p_src = pitch_t if pitch_t is not None else pitch_p
    e_src = energy_t if energy_t is not None else energy_p
    p_emb = F.conv1d(p_src[:, None, :], P[pre + ".pitch_emb.weight"], P[pre + ".pitch_emb.bias"], padding=4)
    e_emb = F.conv1d(e_src[:, None, :], P[pre + ".energy_emb.weight"], P[pre + ".energy_emb.bias"], padding=4)
    aug = text + p_emb.transpose(1, 2) + e_emb.transpose(1, 2)
    cond = torch.cat([aug, spk, emo], -1)
    if dur_t is not None:
        prev = torch.cat([dur_t.new_zeros(dur_t.shape[0], 1), dur_t[:, :-1]], 1).float()
        log_dur = ar_duration_predictor(P, cfg, pre + ".duration_predictor",
                                        torch.log(prev + 1)[:, :, None], cond, pad)
        durs = dur_t
    else:
        log_dur = ar_duration_infer(P, cfg, pre + ".duration_predictor", cond, pad)
        durs = torch.exp(log_dur) - 1
    idx, lens, pos, T, Tp = expand_index(durs, r)
    lr_text = length_regulate(aug, idx, out_pad, Tp) + dur_position_embedding(
        P, pre + ".dur_position_encoder", pos, out_pad, Tp)
    lr_emo = length_regulate(emo, idx, out_pad, Tp)
    lr_spk = length_regulate(spk, idx, out_pad, Tp)
    return lr_text, lr_emo, lr_spk, lens, log_dur, pitch_p, energy_p


# ----------------------------------------------------------------------------- decoder
def pnca_masks(L, xbw, hbw, pad, device):
    """HybridAttentionDecoder.get_pnca_attn_mask, kantts/models/sambert/kantts_sambert.py:135-166.
    True = masked.  x: key j allowed iff i-xbw <= j <= i ; h: i <= j <= i+hbw ; OR key padding;
    rows that belong to padded queries are fully un-masked."""
    i = torch.arange(L, device=device)[:, None]
    j = torch.arange(L, device=device)[None, :]
    x_ok = (j >= (i - xbw).clamp_min(0)) & (j <= i)
    h_ok = (j >= i) & (j < (i + hbw + 1).clamp_max(L + 1))
    xm, hm = (~x_ok)[None], (~h_ok)[None]
    if pad is not None:
        key = pad[:, None, :].expand(-1, L, -1)
        qry = pad[:, :, None].expand(-1, -1, L)
        xm = (xm | key).masked_fill(qry, False)
        hm = (hm | key).masked_fill(qry, False)
    return xm, hm


def pnca_attention(P, pre, x, memory, xm, hm, n_head, dropout=0.0, dropatt=0.0):
    """MultiHeadPNCAAttention.forward, kantts/models/sambert/__init__.py:269-306"""
    q, k, v = _linear(_ln(x, P, pre + ".layer_norm"), P, pre + ".w_x_qkv").chunk(3, -1)
    hk, hv = _linear(memory, P, pre + ".w_h_kv").chunk(2, -1)
    q, k, v, hk, hv = (_split_heads(t, n_head) for t in (q, k, v, hk, hv))
    xm = None if xm is None else xm.expand(x.shape[0], -1, -1).repeat(n_head, 1, 1)
    hm = None if hm is None else hm.expand(x.shape[0], -1, -1).repeat(n_head, 1, 1)
    ox, px = _attend(q, k, v, xm, dropatt)
    oh, ph = _attend(q, hk, hv, hm, dropatt)
    o = _linear(_merge_heads(ox, n_head), P, pre + ".fc_x") + _linear(_merge_heads(oh, n_head), P, pre + ".fc_h")
    return _drop(o, dropout) + x, px, ph


def mel_decoder_train(P, cfg, memory, xbw, hbw, target, lfr_pad, pre="mel_decoder.mel_dec"):
    """MelPNCADecoder.forward (teacher forcing) + HybridAttentionDecoder.forward,
    kantts/models/sambert/kantts_sambert.py:544-568 and :169-205"""
    r = cfg["outputs_per_step"]
    B = memory.shape[0]
    go = memory.new_zeros(B, 1, cfg["num_mels"])
    x = torch.cat([go, target[:, r - 1 :: r, :]], 1)[:, :-1]
    x = prenet(P, pre + ".prenet", x, len(cfg["decoder_prenet_units"]), True)
    x = _linear(torch.cat([memory, x], -1), P, pre + ".dec_in_proj")
    x = _drop(_zero_pad_rows(x, lfr_pad) * cfg["decoder_num_units"] ** 0.5, cfg["decoder_dropout"])
    L = x.shape[1]
    xm, hm = pnca_masks(L, xbw, hbw, lfr_pad, x.device)
    px_l, ph_l = [], []
    for i in range(cfg["decoder_num_layers"]):
        lp = "%s.pnca.%d" % (pre, i)
        x, px, ph = pnca_attention(P, lp + ".pnca_attn", x, memory, xm, hm, cfg["decoder_num_heads"],
                                   cfg["decoder_dropout"], cfg["decoder_attention_dropout"])
        x = _zero_pad_rows(x, lfr_pad)
        x = _zero_pad_rows(conv_ffn(P, lp + ".pos_ffn", x, lfr_pad, cfg["decoder_dropout"],
                                    cfg["decoder_relu_dropout"]), lfr_pad)
        px_l.append(px)
        ph_l.append(ph)
    return _linear(_ln(x, P, pre + ".ln"), P, pre + ".dec_out_proj"), px_l, ph_l


def mel_decoder_infer(P, cfg, memory, xbw, hbw, pre="mel_decoder.mel_dec"):
    """MelPNCADecoder.forward (free running) + HybridAttentionDecoder.infer,
    kantts/models/sambert/kantts_sambert.py:569-612 and :208-253.
    Restated with a growing key/value list per layer; masks are taken from the same band rule."""
    n_head = cfg["decoder_num_heads"]
    B, L, _ = memory.shape
    nl = cfg["decoder_num_layers"]
    scale = cfg["decoder_num_units"] ** 0.5
    xm, hm = pnca_masks(L, xbw, hbw, None, memory.device)
    hk = [None] * nl
    hv = [None] * nl
    xk = [[] for _ in range(nl)]
    xv = [[] for _ in range(nl)]
    for i in range(nl):
        a, b = _linear(memory, P, "%s.pnca.%d.pnca_attn.w_h_kv" % (pre, i)).chunk(2, -1)
        hk[i], hv[i] = _split_heads(a, n_head), _split_heads(b, n_head)
    frame = memory.new_zeros(B, 1, cfg["num_mels"])
    outs = []
    for t in range(L):
        x = prenet(P, pre + ".prenet", frame, len(cfg["decoder_prenet_units"]), True)
        x = _linear(torch.cat([memory[:, t : t + 1], x], -1), P, pre + ".dec_in_proj") * scale
        for i in range(nl):
            ap = "%s.pnca.%d.pnca_attn" % (pre, i)
            q, k, v = _linear(_ln(x, P, ap + ".layer_norm"), P, ap + ".w_x_qkv").chunk(3, -1)
            q, k, v = (_split_heads(z, n_head) for z in (q, k, v))
            xk[i].append(k)
            xv[i].append(v)
            ox, _ = _attend(q, torch.cat(xk[i], 1), torch.cat(xv[i], 1),
                            xm[:, t : t + 1, : t + 1].expand(B * n_head, -1, -1))
            oh, _ = _attend(q, hk[i], hv[i], hm[:, t : t + 1, :].expand(B * n_head, -1, -1))
            x = _linear(_merge_heads(ox, n_head), P, ap + ".fc_x") + _linear(_merge_heads(oh, n_head), P, ap + ".fc_h") + x
            x = conv_ffn(P, "%s.pnca.%d.pos_ffn" % (pre, i), x, None)
        y = _linear(_ln(x, P, pre + ".ln"), P, pre + ".dec_out_proj")
        outs.append(y)
        frame = y[:, :, -cfg["num_mels"] :]
    return torch.cat(outs, 1)


def postnet(P, cfg, x, pad, pre="mel_postnet"):
    """PostNet.forward, kantts/models/sambert/kantts_sambert.py:642-649"""
    h = fsmn_encoder(P, pre + ".fsmn", x, pad, cfg["postnet_fsmn_num_layers"],
                     cfg["postnet_filter_size"], cfg["postnet_shift"], cfg["postnet_dropout"])
    return _linear(lstm_aten(P, pre + ".lstm", h, 1), P, pre + ".fc")


# ----------------------------------------------------------------------------- whole model
def sambert_forward(P, cfg, inputs_ling, inputs_emotion, inputs_speaker, input_lengths,
                    output_lengths=None, mel_targets=None, duration_targets=None,
                    pitch_targets=None, energy_targets=None):
    """KanTtsSAMBERT.forward (MAS/FP/SE disabled), kantts/models/sambert/kantts_sambert.py:862-1044"""
    r = cfg["outputs_per_step"]
    B, T_in = inputs_ling.shape[:2]
    pad = pad_mask(input_lengths, T_in)
    text, enc_attn, ling_emb = text_encoder(P, cfg, inputs_ling, pad)
    emo = F.embedding(inputs_emotion, P["emo_tokenizer.weight"])
    spk = F.embedding(inputs_speaker, P["spk_tokenizer.weight"])
    out_pad = None if output_lengths is None else pad_mask(output_lengths, mel_targets.shape[1])
    lr_text, lr_emo, lr_spk, lr_len, log_dur, pitch_p, energy_p = variance_adaptor(
        P, cfg, text, emo, spk, pad, out_pad, duration_targets, pitch_targets, energy_targets)
    Tp = lr_text.shape[1]
    if output_lengths is not None:
        # get_lfr_mask_from_lengths, kantts_sambert.py:736-750: ceil(len / r)
        lfr_pad = pad_mask((output_lengths + r - 1) // r, Tp // r)
    else:
        out_pad = pad_mask(lr_len, Tp)
        lfr_pad = None
    d = text.shape[-1]
    memory = torch.cat([
        lr_text.reshape(B, -1, r * d),
        lr_spk.reshape(B, -1, r * spk.shape[-1])[:, :, : spk.shape[-1]],
        lr_emo.reshape(B, -1, r * emo.shape[-1])[:, :, : emo.shape[-1]],
    ], -1)
    if duration_targets is not None:
        xbw = int(duration_targets.float().masked_fill(pad, 0).max() / r + 0.5)
    else:
        xbw = int((torch.exp(log_dur) - 1).max() / r + 0.5)
    hbw = xbw
    if mel_targets is not None:
        dec, px, ph = mel_decoder_train(P, cfg, memory, xbw, hbw, mel_targets, lfr_pad)
    else:
        dec, px, ph = mel_decoder_infer(P, cfg, memory, xbw, hbw), [], []
    dec = _zero_pad_rows(dec.reshape(B, -1, cfg["num_mels"]), out_pad)
    post = _zero_pad_rows(postnet(P, cfg, dec, out_pad) + dec, out_pad)
    return {
        "x_band_width": xbw, "h_band_width": hbw,
        "enc_slf_attn_lst": enc_attn, "pnca_x_attn_lst": px, "pnca_h_attn_lst": ph,
        "dec_outputs": dec, "postnet_outputs": post, "LR_length_rounded": lr_len,
        "log_duration_predictions": log_dur, "pitch_predictions": pitch_p,
        "energy_predictions": energy_p, "duration_targets": duration_targets,
        "pitch_targets": pitch_targets, "energy_targets": energy_targets,
        "fp_predictions": None, "valid_inter_lengths": input_lengths,
        "LR_text_outputs": lr_text, "LR_emo_outputs": lr_emo, "LR_spk_outputs": lr_spk,
        "ling_embedding": ling_emb, "text_hid": text, "memory": memory,
    }


def sambert_losses(res, input_lengths, output_lengths, mel_targets):
    """MelReconLoss + ProsodyReconLoss ('mae'), kantts/train/loss.py:18-37 and :51-85;
    total as in Sambert_Trainer.train_step, kantts/train/trainer.py:968."""
    om = (~pad_mask(output_lengths, mel_targets.shape[1])).float()
    n_out = om.sum() * mel_targets.shape[-1]
    mel_loss_ = ((mel_targets - res["dec_outputs"]).abs() * om[:, :, None]).sum() / n_out
    mel_loss = ((mel_targets - res["postnet_outputs"]).abs() * om[:, :, None]).sum() / n_out
    dt = res["duration_targets"]
    im = (~pad_mask(input_lengths, dt.shape[1])).float()
    n_in = im.sum()
    dur_loss = ((torch.log(dt.float() + 1) - res["log_duration_predictions"]).abs() * im).sum() / n_in
    pitch_loss = ((res["pitch_targets"] - res["pitch_predictions"]).abs() * im).sum() / n_in
    energy_loss = ((res["energy_targets"] - res["energy_predictions"]).abs() * im).sum() / n_in
    total = mel_loss_ + mel_loss + dur_loss + pitch_loss + energy_loss
    return {"mel_loss_": mel_loss_, "mel_loss": mel_loss, "dur_loss": dur_loss,
            "pitch_loss": pitch_loss, "energy_loss": energy_loss, "total": total}


# ----------------------------------------------------------------------------- synthetic batches
SAMBERT_VOCAB = dict(sy=147, tone=10, syllable_flag=8, word_segment=8, emotion=36, speaker=4)


def sambert_config(tiny=False):
    """kantts/configs/sambert_16k.yaml:6-52 + vocab sizes of SURVEY.md section 8 (PinYin, speaker F7)."""
    cfg = dict(
        max_len=800, embedding_dim=512, encoder_num_layers=8, encoder_num_heads=8,
        encoder_num_units=128, encoder_ffn_inner_dim=1024, encoder_dropout=0.1,
        encoder_attention_dropout=0.1, encoder_relu_dropout=0.1, encoder_projection_units=32,
        speaker_units=32, emotion_units=32, predictor_filter_size=41, predictor_fsmn_num_layers=3,
        predictor_num_memory_units=128, predictor_ffn_inner_dim=256, predictor_dropout=0.1,
        predictor_shift=0, predictor_lstm_units=128, dur_pred_prenet_units=[128, 128],
        dur_pred_lstm_units=128, decoder_prenet_units=[256, 256], decoder_num_layers=12,
        decoder_num_heads=8, decoder_num_units=128, decoder_ffn_inner_dim=1024,
        decoder_dropout=0.1, decoder_attention_dropout=0.1, decoder_relu_dropout=0.1,
        outputs_per_step=3, num_mels=80, postnet_filter_size=41, postnet_fsmn_num_layers=4,
        postnet_num_memory_units=256, postnet_ffn_inner_dim=512, postnet_dropout=0.1,
        postnet_shift=17, postnet_lstm_units=128, MAS=False,
    )
    cfg.update(SAMBERT_VOCAB)
    if tiny:
        cfg["encoder_num_layers"] = 2
        cfg["decoder_num_layers"] = 2
    return cfg


def synthetic_sambert_batch(B=32, T_in=64, seed=1234, min_len=32, dur_hi=17):
    """Seeded synthetic batch of SURVEY.md section 8(d) (draw order: lens, sy, tone, syllable_flag,
    word_segment, emo, dur, mel, pitch, energy).  Pad frames go to dur[b, lens[b]] as
    Padder._pad_durations does (kantts/datasets/dataset.py:47-64)."""
    g = torch.Generator().manual_seed(seed)
    V = (147, 10, 8, 8)
    lens = torch.randint(min_len, T_in, (B,), generator=g)
    lens[0] = T_in - 1
    ling = torch.stack([torch.randint(0, V[k] - 3, (B, T_in), generator=g) for k in range(4)], -1)
    emo = torch.randint(0, 33, (B, T_in), generator=g)
    spk = torch.zeros(B, T_in, dtype=torch.long)
    dur = torch.randint(2, dur_hi, (B, T_in), generator=g)
    dur = dur * (torch.arange(T_in)[None, :] < lens[:, None])
    out_lens = dur.sum(1)
    T_mel = int(math.ceil(int(out_lens.max()) / 3) * 3)
    for b in range(B):
        dur[b, lens[b]] = T_mel - out_lens[b]
    mel = torch.randn(B, T_mel, 80, generator=g)
    mel = mel * (torch.arange(T_mel)[None, :, None] < out_lens[:, None, None])
    pitch = torch.randn(B, T_in, generator=g)
    energy = torch.randn(B, T_in, generator=g)
    return dict(inputs_ling=ling, inputs_emotion=emo, inputs_speaker=spk, input_lengths=lens,
                output_lengths=out_lens, mel_targets=mel, duration_targets=dur,
                pitch_targets=pitch, energy_targets=energy)
