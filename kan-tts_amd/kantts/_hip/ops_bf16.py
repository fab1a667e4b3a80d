"""bf16-storage variants of the SAM-BERT contraction ops (numerics mode "bf16"; csrc/gemm_bf16.hip).

In bf16 mode the operands a contraction reads are bf16 IN HBM: weights come from a bf16 shadow of the fp32 master
parameters (the parameter arena refreshes it once per step; bare modules cast on demand), activations that only feed
contractions (LayerNorm outputs, the FFN hidden layer, Prenet layers) are written bf16 by the kernel that produces
them.  The residual stream, attention inputs / outputs, losses and every parameter gradient stay fp32.  The fp32 mode
(kantts._hip.ops._FusedLinear on the segmented GEMM) is untouched: it is the parity path.

Gradient dtypes follow autograd's rule (gradient dtype == forward tensor dtype): a bf16 activation receives a bf16
gradient; those tensors have exactly one consumer, so nothing is ever accumulated in bf16.
"""
import contextlib
import math
import os
import weakref

import torch

from . import bgemm_nt, bgemm_tn, check, ffn_pair, lib, pnca_block_bwd, pnca_block_fwd, ptr, rows_sum_accum, stream

BF16 = torch.bfloat16
_wcache = {}
# the feed-forward pair as one launch (csrc/ffn_pair.hip); KANTTS_NO_FFN_PAIR=1 keeps the two-launch form (A/B runs)
PAIR = {"on": os.environ.get("KANTTS_NO_FFN_PAIR", "") == ""}


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def to_bf16(x):
    """fp32 -> bf16 copy by the cast kernel (no autograd)."""
    x = _c(x.detach())
    if x.dtype == BF16:
        return x
    out = torch.empty(x.shape, device=x.device, dtype=BF16)
    if x.numel() % 8 == 0:
        check(lib().kantts_cast_f32_bf16(ptr(x, torch.float32), ptr(out, BF16), x.numel(), stream()), "cast_bf16")
    else:
        out.copy_(x)
    return out


def bf16_weight(w, tap_major=False):
    """bf16 copy of a weight tensor ((N, K), or tap-major (KT, N, Cin) of a Conv1d weight (N, Cin, KT)).
    Parameters that live in a ParamArena carry the arena's shadow views (refreshed once per step, also inside a captured
    hipGraph); anything else is cast on demand and cached until the tensor's version or storage changes."""
    sh = getattr(w, "_kantts_bf16_tap" if tap_major else "_kantts_bf16", None)
    if sh is not None:
        _fresh_shadow(w)
        return sh
    key = (id(w), bool(tap_major))
    hit = _wcache.get(key)
    # id() is only unique among LIVE objects: the entry remembers which tensor it was made from (weak reference), so a
    # new tensor that inherits the id, address, shape and version of a dead one does not inherit its image
    if (hit is not None and hit[4]() is w and hit[0] == w._version and hit[1] == w.data_ptr()
            and hit[2] == tuple(w.shape)):
        return hit[3]
    with torch.no_grad():
        src = w.detach()
        if tap_major:
            src = src.permute(2, 0, 1)
        t = to_bf16(src)
    if len(_wcache) > 4096:
        _wcache.clear()
    _wcache[key] = (w._version, w.data_ptr(), tuple(w.shape), t, weakref.ref(w))
    return t


def _fresh_shadow(w):
    """A parameter's arena images are rebuilt once per forward of the whole module; a direct sub-module call after an
    optimizer step / load_state_dict (or the first bf16-mode use of a model built in fp32 mode) finds them stale."""
    ref = getattr(w, "_kantts_arena", None)
    arena = ref() if ref is not None else None
    if arena is not None and arena.shadow_stale:
        arena.refresh_shadow()


def frag_major(mat):
    """(R, K) matrix -> flat bf16 image in the fragment-major layout of kantts_fragmajor_bf16: every 16 x 32 block is the
    1 KB one A-operand load of v_mfma_f32_16x16x32_bf16 reads (lane = ((k % 32) / 8) * 16 + r % 16, 8 consecutive k)."""
    R, K = mat.shape
    assert R % 16 == 0 and K % 32 == 0
    return to_bf16(mat.detach().reshape(R // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous()).view(-1)


def ffn_frag_weights(w1, w2):
    """The four weight images of csrc/ffn_pair.hip for Conv1d weights w1 (F, C, KT), w2 (N, F, 1):
    (forward phase 1: rows tap*F + f, forward phase 2: (N, F), backward phase 1: W2^T (F, N), backward phase 2: W1^T
    (C, F) per tap: rows tap*C + c).  Arena parameters carry them as attributes (refreshed once per step by
    one launch); anything else is converted on demand and cached until the tensors change."""
    a = getattr(w1, "_kantts_frag", None)
    b = getattr(w2, "_kantts_frag", None)
    if a is not None:
        _fresh_shadow(w1)
    if a is not None and b is not None:
        return a, b, getattr(w2, "_kantts_fragT", None), getattr(w1, "_kantts_fragT", None)
    key = (id(w1), id(w2), "frag")
    sig = (w1._version, w1.data_ptr(), tuple(w1.shape), w2._version, w2.data_ptr(), tuple(w2.shape))
    hit = _wcache.get(key)
    if hit is not None and hit[0] == sig and hit[2]() is w1 and hit[3]() is w2:
        return hit[1]
    F, C, KT = w1.shape
    N = w2.shape[0]
    with torch.no_grad():
        f1 = frag_major(w1.detach().permute(2, 0, 1).reshape(KT * F, C))
        f2 = frag_major(w2.detach().reshape(N, F))
        t2 = frag_major(w2.detach().reshape(N, F).t())
        t1 = frag_major(w1.detach().permute(2, 1, 0).reshape(KT * C, F))
    if len(_wcache) > 4096:
        _wcache.clear()
    _wcache[key] = (sig, (f1, f2, t2, t1), weakref.ref(w1), weakref.ref(w2))
    return f1, f2, t2, t1


def conv_weight_bf16(w):
    """(N, Cin, KT) Conv1d weight -> bf16 tap-major (KT, N, Cin) image; for KT = 1 that is the plain cast."""
    return bf16_weight(w, tap_major=(w.shape[2] > 1))


def eligible(xs, weights, mode, relu, res):
    """Can the bf16 kernels take this fused linear?  (16-byte vectors: every extent a multiple of 8.)"""
    if relu and res is not None:
        return False
    w0 = weights[0]
    if mode == "conv":
        if len(xs) != 1 or w0.shape[1] % 8 or w0.shape[0] % 8 or w0.shape[2] > 12 or xs[0].shape[-1] != w0.shape[1]:
            return False
        return xs[0].numel() > 0
    if w0.shape[0] % 8 or len(xs) > 12:
        return False
    return all(x.shape[-1] % 8 == 0 and x.numel() > 0 for x in xs)


def _splits(M):
    return 0  # library default


class RowMaskToken:
    """Hand-over of a sub-layer's output row mask to the LayerNorm that consumes the output.

    y = zero_rows(x + f(LN(x))) is followed, in a stack of pre-LN sub-layers, by the next sub-layer's LayerNorm, which (with
    ``with_res``) is the ONLY consumer of y.  The backward of the masking -- zero the same rows of the incoming gradient --
    then is a property of what that LayerNorm's backward writes: kantts_ln128_bwd_rows zeroes the rows as it stores dx and
    the producer skips its own pass over the gradient (one elementwise launch per sub-layer and step).  The producer
    attaches the token to its output (``y._kantts_rowmask``); a consumer that KNOWS it is the only one
    (``layer_norm(..., private_input=True)``) takes the mask and sets ``delegated``; any other flow leaves it unset and
    the producer masks the gradient itself."""
    __slots__ = ("mask", "delegated")

    def __init__(self, mask):
        self.mask, self.delegated = mask, False


PRENORM = {"on": not os.environ.get("KANTTS_NO_PRENORM")}  # A/B switch: LayerNorm in the producer's epilogue


class PreNorm:
    """LayerNorm(128) of a sub-layer's OUTPUT computed by the epilogue of the launch that produces it (kantts_bgemm_nt
    ``ln_*`` / kantts_ffn_pair ``ln_*``): the producer is told which nn.LayerNorm consumes its output (``ln_next``), leaves
    the normalised rows + row statistics on the output tensor (``y._kantts_prenorm``), and ``layer_norm128`` called with
    that very module's parameters adopts them instead of launching kantts_ln128_fwd.  Backward is the usual
    kantts_ln128_bwd_rows on the saved statistics."""
    __slots__ = ("gamma", "beta", "eps", "out_bf16", "xn", "mean", "rstd", "bwd")

    def __init__(self, ln):
        self.gamma, self.beta, self.eps = ln.weight, ln.bias, float(ln.eps)
        self.out_bf16 = bool(getattr(ln, "_kantts_out_bf16", True))
        self.xn = self.mean = self.rstd = self.bwd = None

    def matches(self, gamma, beta, eps, out_bf16):
        return (self.xn is not None and self.gamma is gamma and self.beta is beta and self.eps == float(eps)
                and self.out_bf16 == bool(out_bf16))


# A/B switch: LayerNorm BACKWARD in the epilogue of the launch that produces its output gradient (kantts_bgemm_nt_lnbwd: the
# input gradient of the QKV projection of every attention sub-layer).  [round 4] first device run
# (profiles/r04_runA_lnbwd_per_launch.log): 15.8 us against 17.4 us for the two launches at M = 6528, 11.9 against 14.7 us
# at M = 2048 -> ON by default.  The feed-forward pair's analogue measured 35.1 us against 23.0 us and was removed.
LNBWD = {"on": not os.environ.get("KANTTS_NO_LN_BWD_EPILOGUE")}


LNB_ROWS = {"on": not os.environ.get("KANTTS_LNB_ATOMICS")}  # A/B switch: partial rows (default) / atomics


class LnBwdToken:
    """Hand-over for the backward pass of a pre-LN sub-layer, the mirror image of PreNorm: the projection that consumes a
    LayerNorm's output computes, as the epilogue of its input-gradient launch, what kantts_ln128_bwd_rows would make of
    that gradient (dx, dgamma, dbeta) -- the normalised rows' gradient never goes through memory and the LayerNorm's own
    backward launch disappears.  ``layer_norm128`` leaves the token on both of its outputs; the consumer of the residual
    output (the sub-layer's output projection, whose backward runs first) deposits the residual branch's gradient in
    ``dres``; the consumer of the normalised output runs the fused launch and deposits dx / dgamma / dbeta, handing
    autograd its own saved bf16 input as a stand-in gradient (``placeholder``); the LayerNorm node's backward checks that
    what autograd delivers is exactly those two tensors -- anything else means another consumer contributed, and raises --
    and returns the deposited results."""
    __slots__ = ("x", "gamma", "mean", "rstd", "zero_rows", "with_res", "dres", "dx", "dg", "db", "placeholder", "by_block")

    def __init__(self, with_res):
        self.with_res = bool(with_res)
        self.x = self.gamma = self.mean = self.rstd = self.zero_rows = None
        self.dres = self.dx = self.dg = self.db = self.placeholder = None
        self.by_block = False  # dx / dg / db were produced by the fused block backward launch (ops._PncaAttention.backward)

    def ready(self):
        return self.x is not None and self.dx is None and (self.dres is not None or not self.with_res)


# A/B switch: the ReLU (+ dropout) gate of a hidden activation's gradient applied by the epilogue of the launch that
# PRODUCES that gradient (the consumer's input-gradient contraction; the kernel has had the ``gate`` epilogue since round 2)
# instead of a separate pass over it (kantts_relu_gate_bf16).  [round 4] first device runs
# (profiles/r04_runA_lnbwd_step_ab.log, r04_runB_step_ab.log: 40-step pairs on one box 7.386 -> 7.376, 7.404 -> 7.359 ms):
# a small gain and 12 launches fewer per step -> ON by default; KANTTS_NO_RELU_GATE_EPILOGUE switches it off.
RELUGATE = {"on": not os.environ.get("KANTTS_NO_RELU_GATE_EPILOGUE")}


class ReluGateToken:
    """h = dropout(relu(x W1 + b)) stored bf16 feeds exactly one contraction (kantts/models/sambert/fsmn.py:29-40,
    __init__.py:40-49).  Backward: that contraction's input gradient dh is gated by h > 0 and scaled by 1 / (1 - p) before
    anything else happens to it.  The producer of h leaves this token on it; the consumer's backward passes ``gate`` /
    ``scale`` to its input-gradient launch and records the tensor it wrote in ``dz``; the producer's backward takes that
    tensor as the already gated gradient -- after checking that what autograd delivers IS that tensor (a second consumer
    of h would have been summed in un-gated: refused)."""
    __slots__ = ("gate", "scale", "dz")

    def __init__(self):
        self.gate = self.dz = None
        self.scale = 1.0


def _same_tensor(a, b):
    return a is not None and b is not None and a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.dtype == b.dtype


def _attach_token(y, token):
    if token is not None:
        y._kantts_rowmask = token
    return y


def take_rowmask(x):
    """The row mask a private consumer of ``x`` applies to the gradient it sends back (None if there is none)."""
    token = getattr(x, "_kantts_rowmask", None)
    if token is None:
        return None
    token.delegated = True
    return token.mask


# ================================================================================================
# One PNCA decoder block forward in ONE launch (csrc/pnca_block.hip)
# ================================================================================================
# A/B switch: KANTTS_NO_PNCA_BLOCK=1 keeps the five-launch chain of a decoder block's forward pass
PNCA_BLOCK = {"on": not os.environ.get("KANTTS_NO_PNCA_BLOCK")}
# Upper bound of the band width of the batch in flight when the band width itself lives in device memory (captured training
# step: KanTtsSAMBERT.device_band_width); set by whoever knows the batch on the host (train/graph_step.py), None = unknown.
BAND_BOUND = {"max": None}
PB_MAX_BAND = 16  # csrc/pnca_block.hip: PB_HX / PB_HH


class _Adopt:
    """Results the fused block launch has ALREADY produced, in the order the block's ops will ask for them.

    The fused launch computes what the chain of autograd Functions of a PNCA block computes (QKV projection, both attention
    bands, output projection + LayerNorm, feed-forward pair + LayerNorm) and writes every tensor their backward passes
    save.  The block then runs its usual code: each Function's forward finds its entry here, adopts the tensors (and the
    dropout seeds the launch used) instead of launching, and saves for backward exactly what it always saves -- the backward
    pass, every hand-over between sub-layers (RowMaskToken, PreNorm, LnBwdToken) and the deferred weight gradients are
    untouched."""

    def __init__(self):
        self.q = None

    def take(self, kind):
        if not self.q:
            return None
        k, payload = self.q[0]
        if k != kind:
            raise RuntimeError("fused PNCA block: the block's ops ran in another order than the launch assumed (%s, "
                               "expected %s)" % (kind, k))
        self.q.pop(0)
        return payload


ADOPT = _Adopt()


@contextlib.contextmanager
def _adopting(entries):
    prev, ADOPT.q = ADOPT.q, list(entries)
    try:
        yield
        if ADOPT.q:
            raise RuntimeError("fused PNCA block: %d result(s) of the launch were never adopted" % len(ADOPT.q))
    finally:
        ADOPT.q = prev


def lin_frag(w):
    """Fragment-major bf16 image of an nn.Linear weight (N, K) (csrc/ffn_pair.hip layout): the arena's, refreshed once per
    step, for arena parameters; converted on demand and cached otherwise."""
    a = getattr(w, "_kantts_frag", None)
    if a is not None:
        _fresh_shadow(w)
        return a
    key = (id(w), "linfrag")
    sig = (w._version, w.data_ptr(), tuple(w.shape))
    hit = _wcache.get(key)
    if hit is not None and hit[0] == sig and hit[2]() is w:
        return hit[1]
    with torch.no_grad():
        f = frag_major(w.detach())
    if len(_wcache) > 4096:
        _wcache.clear()
    _wcache[key] = (sig, f, weakref.ref(w))
    return f


def lin_fragT(w):
    """Fragment-major image of the TRANSPOSE of an nn.Linear weight (N, K) -> (K, N): the A operand of its input gradient."""
    a = getattr(w, "_kantts_fragT", None)
    if a is not None:
        _fresh_shadow(w)
        return a
    key = (id(w), "linfragT")
    sig = (w._version, w.data_ptr(), tuple(w.shape))
    hit = _wcache.get(key)
    if hit is not None and hit[0] == sig and hit[2]() is w:
        return hit[1]
    with torch.no_grad():
        f = frag_major(w.detach().t())
    if len(_wcache) > 4096:
        _wcache.clear()
    _wcache[key] = (sig, f, weakref.ref(w))
    return f


# A/B switch: KANTTS_NO_PNCA_BLOCK_BWD=1 keeps the four launches of the row-local half of the block's backward
PNCA_BLOCK_BWD = {"on": not os.environ.get("KANTTS_NO_PNCA_BLOCK_BWD")}
# A/B switch: KANTTS_NO_PNCA_ATTN_BWD=1 keeps the two launches of the cross-row half (attention backward, QKV input gradient)
PNCA_ATTN_BWD = {"on": not os.environ.get("KANTTS_NO_PNCA_ATTN_BWD")}


class _BlockBwd:
    """Hand-over between the three backward nodes of a fused block (feed-forward pair -> its LayerNorm -> output projection).

    The feed-forward node runs first: with this plan on it, it launches kantts_pnca_block_bwd, which also computes what the
    LayerNorm node and the output projection's input-gradient launches would compute, deposits the results here and hands
    autograd stand-ins; the two later nodes check that autograd delivers exactly those stand-ins (anything else means a
    tensor of the chain had another consumer: refused) and return the deposited results.  Weight gradients stay with their
    nodes (deferred, grouped by shape as everywhere)."""
    __slots__ = ("y1", "mean1", "rstd1", "gamma1", "rows", "wfcxT", "wfchT", "fc_p", "fc_seed", "placeholder", "d_res", "g1",
                 "dg1", "db1", "d_ox", "d_oh", "tok0", "wqkvT")

    def __init__(self):
        for k in self.__slots__:
            setattr(self, k, None)

    def done(self):
        return self.g1 is not None


def pnca_block_fused(blk, x, hkv, info, bw_x, bw_h, bw_dev, return_attn, next_ln, training):
    """Context manager for kantts.models.sambert.PNCABlock.forward: launches the whole block's forward pass
    (csrc/pnca_block.hip) and lets the ops inside the ``with`` adopt its results; a no-op context when the launch does
    not apply (fp32 mode is decided by the caller; other shapes, attention maps requested, band width above 16 or unknown,
    KANTTS_NO_PNCA_BLOCK)."""
    at, ff = blk.pnca_attn, blk.pos_ffn
    if (not PNCA_BLOCK["on"] or not PAIR["on"] or not PRENORM["on"] or return_attn or hkv is None or next_ln is None
            or not torch.is_tensor(x) or x.dim() != 3 or x.dtype != torch.float32 or x.shape[-1] != 128 or x.numel() == 0
            or at.n_head != 8 or at.d_head != 16 or at.d_model != 128):
        return contextlib.nullcontext()
    w1, w2 = ff.w_1.weight, ff.w_2.weight
    if tuple(w1.shape) != (1024, 128, 1) or tuple(w2.shape) != (128, 1024, 1) or next_ln.weight.numel() != 128:
        return contextlib.nullcontext()
    B, L, _ = x.shape
    M = B * L
    if not (hkv.dim() == 3 and hkv.shape[-1] == 256 and hkv.dtype == torch.float32 and hkv.stride(2) == 1
            and hkv.stride(0) == L * hkv.stride(1) and hkv.stride(1) % 4 == 0 and hkv.data_ptr() % 16 == 0):
        return contextlib.nullcontext()
    if bw_dev is not None:
        bound = BAND_BOUND["max"]
        if bound is None or bound > PB_MAX_BAND:
            return contextlib.nullcontext()
    elif not (0 <= int(bw_x) <= PB_MAX_BAND and 0 <= int(bw_h) <= PB_MAX_BAND):
        return contextlib.nullcontext()
    from .ops import next_seed

    def p_of(drop):
        return float(drop.p) if (training and drop.p > 0) else 0.0

    att_p, fc_p, p_in, p_out = p_of(at.attention.dropatt), p_of(at.dropout), p_of(ff.dropout_inner), p_of(ff.dropout)
    x = _c(x)
    dev = x.device
    ln0 = at.layer_norm
    pre0 = getattr(x, "_kantts_prenorm", None)
    if pre0 is None or not pre0.matches(ln0.weight, ln0.bias, ln0.eps, True) or pre0.xn.numel() != x.numel():
        # the block's input has no producer that normalised it (the first block of the stack): one LayerNorm launch, left on
        # the tensor exactly as a producer's epilogue would have left it
        pre0 = PreNorm(ln0)
        pre0.out_bf16 = True
        pre0.xn = torch.empty((M, 128), device=dev, dtype=BF16)
        pre0.mean = torch.empty(M, device=dev, dtype=torch.float32)
        pre0.rstd = torch.empty(M, device=dev, dtype=torch.float32)
        check(lib().kantts_ln128_fwd(ptr(x.detach(), torch.float32), ptr(ln0.weight.detach(), torch.float32),
                                     ptr(ln0.bias.detach(), torch.float32), ptr(pre0.xn), 1, ptr(pre0.mean), ptr(pre0.rstd), M,
                                     float(ln0.eps), stream()), "ln128_fwd")
        x._kantts_prenorm = pre0
    # dropout seeds in the order the chain draws them (attention x, attention h, output projection, hidden, output)
    sx = next_seed() if att_p > 0 else 0
    sh = next_seed() if att_p > 0 else 0
    sf = next_seed() if fc_p > 0 else 0
    s1 = next_seed() if p_in > 0 else 0
    s2 = next_seed() if p_out > 0 else 0
    wf1, wf2, _, _ = ffn_frag_weights(w1, w2)
    f32 = dict(device=dev, dtype=torch.float32)
    qkv = torch.empty((B, L, 384), **f32)
    ox, oh = torch.empty((M, 128), **f32), torch.empty((M, 128), **f32)
    lsx, lsh = torch.empty((B, 8, L), **f32), torch.empty((B, 8, L), **f32)
    y1, out = torch.empty((M, 128), **f32), torch.empty((M, 128), **f32)
    xn1 = torch.empty((M, 128), device=dev, dtype=BF16)
    mean1, rstd1 = torch.empty(M, **f32), torch.empty(M, **f32)
    hid = torch.empty((M, 1024), device=dev, dtype=BF16)
    out_bf16 = bool(getattr(next_ln, "_kantts_out_bf16", True))
    xn2 = torch.empty((M, 128), device=dev, dtype=BF16 if out_bf16 else torch.float32)
    mean2, rstd2 = torch.empty(M, **f32), torch.empty(M, **f32)
    rows = None if info is None else _c(info.mask).view(M)
    ok = pnca_block_fwd(
        x.detach(), pre0.xn, hkv.detach(), hkv.stride(1), B, L, lens=None if info is None else info.lens32, bw_dev=bw_dev,
        bw_x=int(bw_x), bw_h=int(bw_h), rowmask=rows, wqkv=lin_frag(at.w_x_qkv.weight), bqkv=at.w_x_qkv.bias.detach(),
        wfcx=lin_frag(at.fc_x.weight), wfch=lin_frag(at.fc_h.weight), bfcx=at.fc_x.bias.detach(), bfch=at.fc_h.bias.detach(),
        ln1=(ff.layer_norm.weight.detach(), ff.layer_norm.bias.detach(), ff.layer_norm.eps), w1=wf1, w2=wf2,
        bias1=ff.w_1.bias.detach(), bias2=ff.w_2.bias.detach(), att_p=att_p, fc_p=fc_p, drop1_p=p_in, drop2_p=p_out,
        seeds=(sx, sh, sf, s1, s2), qkv=qkv, ox=ox, oh=oh, lse_x=lsx, lse_h=lsh, y1=y1, xn1=xn1, mean1=mean1, rstd1=rstd1,
        hid=hid, out=out, ln2=(next_ln.weight.detach(), next_ln.bias.detach(), next_ln.eps, xn2, mean2, rstd2))
    if not ok:
        raise RuntimeError("kantts_pnca_block_fwd declined a block pnca_block_fused() accepted")
    plan = None
    if PNCA_BLOCK_BWD["on"] and torch.is_grad_enabled() and x.requires_grad:
        plan = _BlockBwd()
        plan.y1, plan.mean1, plan.rstd1, plan.gamma1, plan.rows = y1, mean1, rstd1, ff.layer_norm.weight, rows
        plan.wfcxT, plan.wfchT, plan.fc_p, plan.fc_seed = lin_fragT(at.fc_x.weight), lin_fragT(at.fc_h.weight), fc_p, sf
        plan.wqkvT = lin_fragT(at.w_x_qkv.weight)  # the cross-row half (ops._PncaAttention.backward: kantts_pnca_attn_qkv_bwd)
    return _adopting([
        ("linear", dict(y=qkv.view(M, 384), seed=0, ln=None, qkv_of=plan)),
        ("attn", dict(ox=ox, oh=oh, lse_x=lsx, lse_h=lsh, sx=sx, sh=sh, bwd=plan)),
        ("linear", dict(y=y1, seed=sf, ln=(xn1, mean1, rstd1), bwd=plan)),
        ("ffn", dict(hid=hid, out=out, s1=s1, s2=s2, ln=(xn2, mean2, rstd2), bwd=plan)),
    ])


# A/B switch: KANTTS_NO_ENC_ATTN=1 keeps the three launches of an encoder block's attention sub-layer
ENC_ATTN = {"on": not os.environ.get("KANTTS_NO_ENC_ATTN")}


def enc_attn_fused(att, x, info, rows, return_attn, next_ln, training):
    """Context manager for kantts.models.sambert.MultiHeadSelfAttention.forward: launches the whole sub-layer's forward pass
    (csrc/enc_attn.hip: QKV projection, 8-head attention, output projection + dropout + residual + row mask, the consumer's
    LayerNorm) and lets the three ops inside the ``with`` adopt its results (the protocol of pnca_block_fused); a no-op context
    when the launch does not apply (other shapes, sequences of more than 128 tokens, attention maps requested, no consumer
    LayerNorm, KANTTS_NO_ENC_ATTN)."""
    if (not ENC_ATTN["on"] or not PRENORM["on"] or return_attn or next_ln is None or not torch.is_tensor(x) or x.dim() != 3
            or x.dtype != torch.float32 or x.shape[-1] != 128 or x.numel() == 0 or x.shape[1] > 128
            or att.n_head != 8 or att.d_head != 16 or att.d_model != 128 or att.d_in != 128 or att.fc.out_features != 128
            or next_ln.weight.numel() != 128 or att.w_qkv.bias is None or att.fc.bias is None):
        return contextlib.nullcontext()
    from .ops import next_seed

    def p_of(drop):
        return float(drop.p) if (training and drop.p > 0) else 0.0

    att_p, fc_p = p_of(att.attention.dropatt), p_of(att.dropout)
    x = _c(x)
    dev = x.device
    B, L, _ = x.shape
    M = B * L
    ln0 = att.layer_norm
    pre0 = getattr(x, "_kantts_prenorm", None)
    if pre0 is None or not pre0.matches(ln0.weight, ln0.bias, ln0.eps, True) or pre0.xn.numel() != x.numel():
        # the sub-layer's input has no producer that normalised it (the first block of the stack): one LayerNorm launch, left
        # on the tensor exactly as a producer's epilogue would have left it
        pre0 = PreNorm(ln0)
        pre0.out_bf16 = True
        pre0.xn = torch.empty((M, 128), device=dev, dtype=BF16)
        pre0.mean = torch.empty(M, device=dev, dtype=torch.float32)
        pre0.rstd = torch.empty(M, device=dev, dtype=torch.float32)
        check(lib().kantts_ln128_fwd(ptr(x.detach(), torch.float32), ptr(ln0.weight.detach(), torch.float32),
                                     ptr(ln0.bias.detach(), torch.float32), ptr(pre0.xn), 1, ptr(pre0.mean), ptr(pre0.rstd), M,
                                     float(ln0.eps), stream()), "ln128_fwd")
        x._kantts_prenorm = pre0
    # dropout seeds in the order the chain draws them (attention, output projection)
    sa = next_seed() if att_p > 0 else 0
    sf = next_seed() if fc_p > 0 else 0
    f32 = dict(device=dev, dtype=torch.float32)
    qkv = torch.empty((B, L, 384), **f32)
    o = torch.empty((M, 128), **f32)
    lse = torch.empty((B, 8, L), **f32)
    y1 = torch.empty((M, 128), **f32)
    out_bf16 = bool(getattr(next_ln, "_kantts_out_bf16", True))
    xn1 = torch.empty((M, 128), device=dev, dtype=BF16 if out_bf16 else torch.float32)
    mean1, rstd1 = torch.empty(M, **f32), torch.empty(M, **f32)
    rowmask = None if rows is None else _c(rows).view(M)
    from . import enc_attn_fwd

    ok = enc_attn_fwd(x.detach(), pre0.xn, B, L, lens=None if info is None else info.lens32, rowmask=rowmask,
                      wqkv=lin_frag(att.w_qkv.weight), bqkv=att.w_qkv.bias.detach(), wfc=lin_frag(att.fc.weight),
                      bfc=att.fc.bias.detach(), ln1=(next_ln.weight.detach(), next_ln.bias.detach(), next_ln.eps),
                      att_p=att_p, fc_p=fc_p, seeds=(sa, sf), qkv=qkv, o=o, lse=lse, y1=y1, xn1=xn1, mean1=mean1, rstd1=rstd1)
    if not ok:
        raise RuntimeError("kantts_enc_attn_fwd declined a sub-layer enc_attn_fused() accepted")
    return _adopting([
        ("linear", dict(y=qkv.view(M, 384), seed=0, ln=None)),
        ("attn", dict(o=o, lse=lse, seed=sa)),
        ("linear", dict(y=y1, seed=sf, ln=(xn1, mean1, rstd1))),
    ])


class _FusedLinearB(torch.autograd.Function):
    """y = rowmask( dropout( act( (sum_k x_k @ W_k^T + bias [+ bias2]) * alpha ) ) + res ) on kantts_bgemm_nt/tn.
    Same three modes as ops._FusedLinear (concat / sum / conv)."""

    @staticmethod
    def forward(ctx, opts, bias, bias2, res, rowmask, *t):
        nx, nw = opts["nx"], opts["nw"]
        xs_in = [_c(x) for x in t[:nx]]
        ws = list(t[nx:nx + nw])          # fp32 parameters (shapes for the gradients)
        wbs = list(t[nx + nw:])           # bf16 shadows
        mode, relu, alpha, drop_p = opts["mode"], opts["relu"], opts["alpha"], opts["drop_p"]
        lead = xs_in[0].shape[:-1]
        M = int(math.prod(lead))
        N = ws[0].shape[0]
        T = opts.get("T", 0)
        # the A operands of one launch share a dtype
        dts = {x.dtype for x in xs_in}
        xs = [to_bf16(x) for x in xs_in] if len(dts) > 1 else [x.detach() for x in xs_in]
        dev = xs[0].device
        y = torch.empty((M, N), device=dev, dtype=BF16 if opts["out_bf16"] else torch.float32)
        segs = []
        if mode == "conv":
            cin, kt = ws[0].shape[1], ws[0].shape[2]
            pad, dil = opts["pad"], opts.get("dilation", 1)
            for tap in range(kt):
                segs.append((xs[0], cin, (wbs[0], tap * N * cin), cin, cin, tap * dil - pad))
        elif mode == "concat":
            ldw = ws[0].shape[1]
            off = 0
            for x in xs:
                k = x.shape[-1]
                segs.append((x, k, (wbs[0], off), ldw, k, 0))
                off += k
            assert off == ldw, "concat widths do not match the weight"
        else:
            for x, wb in zip(xs, wbs):
                k = x.shape[-1]
                segs.append((x, k, wb, k, k, 0))
        from .ops import next_seed

        r = _c(res).view(M, N) if res is not None else None
        rm = _c(rowmask).view(M) if rowmask is not None else None
        ln = None
        pre = opts.get("ln_next")
        ad = ADOPT.take("linear") if ADOPT.q else None
        if ad is not None:
            # computed by the fused block launch (pnca_block_fused): same values, same dropout seed, nothing to launch
            y, seed = ad["y"], ad["seed"]
            assert tuple(y.shape) == (M, N) and y.dtype == (BF16 if opts["out_bf16"] else torch.float32)
            if pre is not None and ad["ln"] is not None:
                pre.xn, pre.mean, pre.rstd = ad["ln"]
                pre.bwd = ad.get("bwd")  # travels to the LayerNorm node that adopts these rows (_BlockBwd)
            opts["bwd"] = ad.get("bwd")
            if ad.get("qkv_of") is not None:  # the QKV projection of a fused block: its LayerNorm's token goes to the plan
                ad["qkv_of"].tok0 = opts.get("lnbwd")
        else:
            seed = next_seed() if drop_p > 0 else 0
            if pre is not None and N == 128 and not opts["out_bf16"]:
                # LayerNorm of the consuming sub-layer, computed by this launch's epilogue (PreNorm)
                pre.xn = torch.empty((M, N), device=dev, dtype=BF16 if pre.out_bf16 else torch.float32)
                pre.mean = torch.empty(M, device=dev, dtype=torch.float32)
                pre.rstd = torch.empty(M, device=dev, dtype=torch.float32)
                ln = (pre.gamma.detach(), pre.beta.detach(), pre.eps, pre.xn, pre.mean, pre.rstd)
            if not bgemm_nt(segs, M, N, y, N, T=T, bias=bias, bias2=bias2, alpha=alpha, relu=relu, drop_p=drop_p,
                            drop_seed=seed, res=r, ldr=N, rowmask=rm, ln=ln):
                raise RuntimeError("kantts_bgemm_nt declined a shape ops_bf16.eligible() accepted")
        ctx.opts, ctx.seed, ctx.M, ctx.N, ctx.lead = opts, seed, M, N, lead
        ctx.has = (bias is not None, bias2 is not None, res is not None)
        ctx.x_dtypes = [x.dtype for x in xs_in]
        ctx.w_shapes = [tuple(w.shape) for w in ws]
        ctx.save_for_backward(y if relu else None, rm, *xs, *wbs)
        if opts.get("relugate_self") is not None and y.dtype == BF16:
            tok = opts["relugate_self"]
            tok.gate, tok.scale = y, (alpha / (1.0 - drop_p) if drop_p > 0 else alpha)
        return y.view(*lead, N)

    @staticmethod
    def backward(ctx, dy):
        from .ops import gzeros, wgrad_overlap

        opts, M, N = ctx.opts, ctx.M, ctx.N
        nx, nw, mode, relu, alpha, drop_p = opts["nx"], opts["nw"], opts["mode"], opts["relu"], opts["alpha"], opts["drop_p"]
        T = opts.get("T", 0)
        sv = ctx.saved_tensors
        y_gate, rm = sv[0], sv[1]
        xs, wbs = list(sv[2:2 + nx]), list(sv[2 + nx:])
        has_bias, has_bias2, has_res = ctx.has
        dy = dy_in = _c(dy).view(M, N)
        token = opts.get("token")
        if rm is not None and not relu and not (token is not None and token.delegated):
            dy = dy.masked_fill(rm.bool().view(M, 1), 0.0)
        d_res = None
        if has_res:
            d_res = (dy if dy.dtype == torch.float32 else dy.float()).view(*ctx.lead, N)
            if opts.get("lnbwd_res") is not None:
                opts["lnbwd_res"].dres = d_res
        a_drop_p, a_seed, balpha = 0.0, 0, alpha
        rself = opts.get("relugate_self")
        if relu and rself is not None and rself.dz is not None:
            # gated and scaled by the launch that produced it (ReluGateToken)
            if not _same_tensor(dy, rself.dz.view(M, N)):
                raise RuntimeError("the ReLU gate of this gradient was applied by its producer's launch, but autograd "
                                   "delivers another tensor: the hidden activation has a second consumer")
            dz, balpha = dy, 1.0
            rself.gate = rself.dz = None
        elif relu:
            scale = alpha / (1.0 - drop_p) if drop_p > 0 else alpha
            dz = torch.empty((M, N), device=dy.device, dtype=BF16)
            check(lib().kantts_relu_gate_bf16(ptr(dy), int(dy.dtype == BF16), ptr(y_gate), int(y_gate.dtype == BF16),
                                              ptr(dz, BF16), float(scale), M * N, stream()), "relu_gate")
            balpha = 1.0
        else:
            dz = dy
            if drop_p > 0:
                if dz.dtype != torch.float32:
                    dz = dz.float()
                a_drop_p, a_seed = drop_p, ctx.seed
        needs = ctx.needs_input_grad  # (opts, bias, bias2, res, rowmask, *xs, *ws, *wbs)
        dxs, dws = [None] * nx, [None] * nw
        plan = opts.get("bwd")
        if plan is not None and plan.done():
            # fused PNCA block: the input gradients of the two context segments came out of the block's backward launch
            if not _same_tensor(dy_in, plan.g1) or nx != 2 or mode != "sum":
                raise RuntimeError("the input gradient of a fused PNCA block's output projection was computed by the block's "
                                   "backward launch, but autograd delivers another gradient than the one that launch produced")
            dxs = [plan.d_ox.view(xs[0].shape), plan.d_oh.view(xs[1].shape)]
            plan.g1 = plan.d_ox = plan.d_oh = None
        dbias = gzeros((N,), dy.device) if (has_bias or has_bias2) else None
        first = True
        kw = dict(alpha=balpha, a_drop_p=a_drop_p, a_drop_seed=a_seed)
        if mode == "conv":
            n_, cin, kt = ctx.w_shapes[0]
            pad, dil = opts["pad"], opts.get("dilation", 1)
            x = xs[0]
            if needs[5]:
                dx = torch.empty(x.shape, device=x.device, dtype=ctx.x_dtypes[0])
                segs = [(dz, N, (wbs[0], tap * N * cin), cin, N, pad - tap * dil) for tap in range(kt)]
                if not bgemm_nt(segs, M, cin, dx, cin, T=T, b_kn=True, a_drop_ld=N, **kw):
                    raise RuntimeError("bgemm_nt declined the conv input gradient")
                dxs[0] = dx
            if needs[5 + nx]:
                dw = gzeros(ctx.w_shapes[0], dy.device)  # the parameter's own (N, Cin, KT) layout
                with wgrad_overlap.side(dz, x):
                    if not bgemm_tn(dz, N, x, cin, M, N, cin, dw, cin * kt, kt, c_ts=1, T=T, ntaps=kt, shift0=-pad,
                                    shift_step=dil, db=dbias, **kw):
                        raise RuntimeError("bgemm_tn declined the conv weight gradient")
                first = False
                dws[0] = dw
        else:
            off = 0
            ldw = ctx.w_shapes[0][1]
            if mode == "concat" and needs[5 + nx]:
                dws[0] = gzeros(ctx.w_shapes[0], dy.device)
            for k, x in enumerate(xs):
                kk = x.shape[-1]
                wb = wbs[0] if mode == "concat" else wbs[k]
                woff = off if mode == "concat" else 0
                wld = ldw if mode == "concat" else kk
                tok = opts.get("lnbwd") if (needs[5 + k] and a_drop_p == 0 and balpha == 1.0) else None
                if tok is not None and tok.by_block and tok.dx is not None and tok.placeholder is None:
                    # the input gradient AND the LayerNorm backward came out of the block's backward launch
                    # (kantts_pnca_attn_qkv_bwd, issued by the attention node): only the stand-in is left to hand over
                    tok.placeholder = x
                    dxs[k] = x
                if tok is not None and tok.ready():
                    # the LayerNorm that produced x: its backward is this launch's epilogue (LnBwdToken)
                    from .ops import gzeros_like

                    ldx = torch.empty(tok.x.shape, device=x.device, dtype=torch.float32)
                    ldg, ldb = gzeros_like(tok.gamma), gzeros_like(tok.gamma)
                    # dgamma / dbeta leave the launch as one partial row per workgroup and are summed beside the critical path
                    # (256 atomics per workgroup onto the same 256 addresses cost more than the contraction itself)
                    part = torch.empty(((M + 31) // 32, 256), device=x.device, dtype=torch.float32) if LNB_ROWS["on"] else None
                    if bgemm_nt([(dz, N, (wb, woff), wld, N, 0)], M, kk, None, kk, b_kn=True, a_drop_ld=N, c_bf16=True,
                                lnb=(tok.x.view(M, 128), tok.gamma, tok.mean, tok.rstd,
                                     None if tok.dres is None else _c(tok.dres).view(M, 128), tok.zero_rows, ldx, ldg, ldb)
                                + ((part,) if part is not None else ())):
                        if part is not None:
                            rows_sum_accum(part, ldg, ldb)
                        tok.dx, tok.dg, tok.db, tok.placeholder = ldx, ldg, ldb, x
                        dxs[k] = x  # stand-in: the LayerNorm node returns tok.dx and never reads this
                rtok = opts.get("relugate") if (needs[5 + k] and dxs[k] is None and nx == 1) else None
                if rtok is not None and rtok.gate is not None and rtok.dz is None and ctx.x_dtypes[k] == BF16:
                    # x = dropout(relu(.)) of the producing layer: its gate and scale ride on this launch (ReluGateToken)
                    dx = torch.empty(x.shape, device=x.device, dtype=BF16)
                    gkw = dict(kw, alpha=kw["alpha"] * rtok.scale)
                    if bgemm_nt([(dz, N, (wb, woff), wld, N, 0)], M, kk, dx, kk, b_kn=True, a_drop_ld=N, gate=rtok.gate,
                                ldg=kk, **gkw):
                        rtok.dz = dx
                        dxs[k] = dx
                if needs[5 + k] and dxs[k] is None:
                    dx = torch.empty(x.shape, device=x.device, dtype=ctx.x_dtypes[k])
                    if not bgemm_nt([(dz, N, (wb, woff), wld, N, 0)], M, kk, dx, kk, b_kn=True, a_drop_ld=N, **kw):
                        raise RuntimeError("bgemm_nt declined an input gradient")
                    dxs[k] = dx
                need_w = needs[5 + nx] if mode == "concat" else needs[5 + nx + k]
                if need_w:
                    if mode != "concat":
                        dws[k] = gzeros(ctx.w_shapes[k], dy.device)
                    dwt = dws[0] if mode == "concat" else dws[k]
                    with wgrad_overlap.side(dz, x):
                        if not bgemm_tn(dz, N, x, kk, M, N, kk, (dwt, woff), wld, 1, db=dbias if first else None, **kw):
                            raise RuntimeError("bgemm_tn declined a weight gradient")
                    first = False
                off += kk
        if dbias is not None and first:
            raise RuntimeError("bias gradient without weight gradient is not supported")
        db2 = None
        if has_bias2:  # two biases share one gradient (see deferred_tn.shared_gradient)
            from . import deferred_tn

            db2 = deferred_tn.shared_gradient(dbias) if has_bias else dbias
        return (None, dbias if has_bias else None, db2, d_res, None, *dxs, *dws, *([None] * len(wbs)))


def linear(xs, weights, wbs, bias, *, mode, bias2, res, rowmask, relu, alpha, drop_p, pad, dilation, T, out_bf16,
           ln_next=None):
    token = RowMaskToken(rowmask) if (rowmask is not None and not relu and torch.is_grad_enabled()) else None
    pre = PreNorm(ln_next) if (ln_next is not None and PRENORM["on"] and not relu and not out_bf16
                               and weights[0].shape[0] == 128) else None
    opts = dict(nx=len(xs), nw=len(weights), mode=mode, relu=bool(relu), alpha=float(alpha), drop_p=float(drop_p),
                pad=int(pad), dilation=int(dilation), T=int(T), out_bf16=bool(out_bf16), token=token, ln_next=pre)
    if RELUGATE["on"] and torch.is_grad_enabled():
        if relu and out_bf16:
            opts["relugate_self"] = ReluGateToken()
        if len(xs) == 1 and mode != "conv" and xs[0].dtype == BF16:
            opts["relugate"] = getattr(xs[0], "_kantts_relugate", None)
    if LNBWD["on"] and torch.is_grad_enabled():
        if res is not None:  # this launch's backward sees the residual branch's gradient first
            opts["lnbwd_res"] = getattr(res, "_kantts_lnbwd_res", None)
        if (len(xs) == 1 and mode != "conv" and not relu and drop_p == 0 and alpha == 1.0 and xs[0].dtype == BF16
                and xs[0].shape[-1] == 128):
            opts["lnbwd"] = getattr(xs[0], "_kantts_lnbwd", None)
    y = _attach_token(_FusedLinearB.apply(opts, bias, bias2, res, rowmask, *xs, *weights, *wbs), token)
    if pre is not None and pre.xn is not None:
        y._kantts_prenorm = pre
    if opts.get("relugate_self") is not None and opts["relugate_self"].gate is not None:
        y._kantts_relugate = opts["relugate_self"]
    return y


# ================================================================================================
# LayerNorm(128)
# ================================================================================================
class _LayerNorm128(torch.autograd.Function):
    """y = LayerNorm(x) and, with ``with_res``, a second output that IS x: the residual branch of a pre-LN sub-layer takes
    that output instead of x itself, so both gradients of x arrive at this node and are summed inside the LayerNorm
    backward kernel -- autograd would otherwise add them with a separate elementwise kernel per sub-layer."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, out_bf16, with_res, zero_rows, xn, mean, rstd, token=None, bwd=None):
        x = _c(x)
        M = x.numel() // 128
        ctx.bwd = bwd
        if xn is not None:  # computed by the epilogue of the launch that produced x (PreNorm)
            y = xn.view(x.shape)
        else:
            y = torch.empty(x.shape, device=x.device, dtype=BF16 if out_bf16 else torch.float32)
            mean = torch.empty(M, device=x.device, dtype=torch.float32)
            rstd = torch.empty(M, device=x.device, dtype=torch.float32)
            check(lib().kantts_ln128_fwd(ptr(x, torch.float32), ptr(gamma, torch.float32), ptr(beta, torch.float32),
                                         ptr(y), int(out_bf16), ptr(mean), ptr(rstd), M, float(eps), stream()), "ln128_fwd")
        ctx.save_for_backward(x, gamma, mean, rstd, zero_rows)
        ctx.token = token
        if token is not None:
            token.x, token.gamma, token.mean, token.rstd, token.zero_rows = x, gamma.detach(), mean, rstd, zero_rows
        if with_res:
            return y, x.view_as(x)
        return y, None

    @staticmethod
    def backward(ctx, dy, dres):
        from .ops import gzeros_like

        x, gamma, mean, rstd, zero_rows = ctx.saved_tensors
        M = x.numel() // 128
        tok = ctx.token
        plan = ctx.bwd
        if plan is not None and plan.done():  # done by the fused block backward launch (_BlockBwd)
            if not _same_tensor(dy, plan.placeholder) or not _same_tensor(dres, plan.d_res):
                raise RuntimeError("the LayerNorm backward of a fused PNCA block was computed by the block's backward launch, "
                                   "but autograd delivers other gradients than the ones that launch saw: the normalised rows "
                                   "(or the residual output) have a second consumer")
            # (autograd adopts dg1 / db1 as p.grad only while nobody else references the tensor objects; a clone would be taken
            # before the deferred row sums have filled them)
            out = (plan.g1.view(x.shape), plan.dg1, plan.db1)
            plan.placeholder = plan.d_res = plan.dg1 = plan.db1 = None
            return (*out, None, None, None, None, None, None, None, None, None)
        if tok is not None and tok.dx is not None:  # done by the consumer's input-gradient launch (LnBwdToken)
            if not _same_tensor(dy, tok.placeholder) or (tok.with_res and not _same_tensor(dres, tok.dres)) or (
                    not tok.with_res and dres is not None):
                raise RuntimeError("LayerNorm backward was fused into its consumer's launch, but autograd delivers other "
                                   "gradients than the ones that launch saw: the normalised rows (or the residual output) "
                                   "have a second consumer")
            out = (tok.dx, tok.dg, tok.db)
            tok.x = tok.dres = tok.dx = tok.dg = tok.db = tok.placeholder = None
            return (*out, None, None, None, None, None, None, None, None, None)
        dx = torch.empty_like(x)
        dg, db = gzeros_like(gamma), gzeros_like(gamma)
        if dy is None:  # only the pass-through was used downstream
            if zero_rows is not None:
                dres = dres.masked_fill(zero_rows.bool().view(*dres.shape[:-1], 1), 0.0)
            return (dres, None, None, None, None, None, None, None, None, None, None, None)
        dy = _c(dy)
        dres = _c(dres) if dres is not None else None
        check(lib().kantts_ln128_bwd_rows(ptr(dy), int(dy.dtype == BF16), ptr(x), ptr(gamma), ptr(mean), ptr(rstd),
                                          ptr(dres, torch.float32), ptr(dx), ptr(dg), ptr(db), ptr(zero_rows, torch.uint8),
                                          M, stream()), "ln128_bwd")
        return dx, dg, db, None, None, None, None, None, None, None, None, None


def layer_norm128(x, gamma, beta, eps, out_bf16, with_res=False, private_input=False):
    """``private_input``: the caller guarantees that this LayerNorm (with its pass-through output, if ``with_res``) is the
    only consumer of ``x``; a row mask the producer of ``x`` left on it (RowMaskToken) is then applied to dx here."""
    zero_rows = take_rowmask(x) if (private_input and torch.is_grad_enabled()) else None
    if zero_rows is not None:
        zero_rows = _c(zero_rows).view(-1)
        if zero_rows.dtype == torch.bool:
            zero_rows = zero_rows.view(torch.uint8)
    token = LnBwdToken(with_res) if (LNBWD["on"] and out_bf16 and torch.is_grad_enabled() and x.requires_grad) else None
    pre = getattr(x, "_kantts_prenorm", None)
    if pre is not None and pre.matches(gamma, beta, eps, out_bf16) and pre.xn.numel() == x.numel():
        y, xr = _LayerNorm128.apply(x, gamma, beta, float(eps), bool(out_bf16), bool(with_res), zero_rows, pre.xn, pre.mean,
                                    pre.rstd, token, pre.bwd if with_res else None)
    else:
        y, xr = _LayerNorm128.apply(x, gamma, beta, float(eps), bool(out_bf16), bool(with_res), zero_rows, None, None, None,
                                    token)
    if token is not None:
        y._kantts_lnbwd = token
        if with_res:
            xr._kantts_lnbwd_res = token
    return (y, xr) if with_res else y


# ================================================================================================
# Position-wise feed-forward block: LN output -> Conv1d(k) + ReLU + dropout -> Conv1d(1) + dropout + residual
# ================================================================================================
class _FusedFFNB(torch.autograd.Function):
    """kantts/models/sambert/__init__.py:134-149 after the LayerNorm, as ONE autograd node: two contractions forward,
    four backward.  The hidden activation is bf16 and is never gated by a separate pass: its ReLU / dropout / padded-row
    gate is the epilogue of the input-gradient contraction through w_2."""

    @staticmethod
    def forward(ctx, h, w1, b1, w2, b2, res, pad_rows, zero_rows, wb1, wb2, wf1, wf2, wt2, wt1, cfg):
        from .ops import next_seed

        h = _c(h)
        lead = h.shape[:-1]
        M = int(math.prod(lead))
        F, C, kt = w1.shape
        N = w2.shape[0]
        T, pad, p_in, p_out = cfg["T"], cfg["pad"], cfg["p_inner"], cfg["p_out"]
        hb = h.detach() if h.dtype == BF16 else to_bf16(h)
        pr = _c(pad_rows).view(M) if pad_rows is not None else None
        zr = _c(zero_rows).view(M) if zero_rows is not None else None
        r = _c(res).view(M, N)
        ln = None
        pre = cfg.get("ln_next")
        ad = ADOPT.take("ffn") if ADOPT.q else None
        if ad is not None:  # computed by the fused block launch (pnca_block_fused)
            hid, out, s1, s2 = ad["hid"], ad["out"], ad["s1"], ad["s2"]
            assert cfg["pair"] and tuple(hid.shape) == (M, F) and tuple(out.shape) == (M, N)
            cfg["bwd"] = ad.get("bwd")
            if pre is not None:
                pre.xn, pre.mean, pre.rstd = ad["ln"]
            fused = True
        else:
            s1 = next_seed() if p_in > 0 else 0
            s2 = next_seed() if p_out > 0 else 0
            hid = torch.empty((M, F), device=h.device, dtype=BF16)
            out = torch.empty((M, N), device=h.device, dtype=torch.float32)
            if pre is not None and cfg["pair"] and N == 128:
                pre.xn = torch.empty((M, N), device=h.device, dtype=BF16 if pre.out_bf16 else torch.float32)
                pre.mean = torch.empty(M, device=h.device, dtype=torch.float32)
                pre.rstd = torch.empty(M, device=h.device, dtype=torch.float32)
                ln = (pre.gamma.detach(), pre.beta.detach(), pre.eps, pre.xn, pre.mean, pre.rstd)
            fused = cfg["pair"] and ffn_pair(hb.view(M, C), wf1, wf2, out, M=M, T=T, F=F, KT=kt, pad=pad, bias1=b1, bias2=b2,
                                             relu=True, drop1_p=p_in, drop1_seed=s1, drop2_p=p_out, drop2_seed=s2,
                                             rowmask1=pr, rowmask2=zr, t_out=hid, res=r, ln=ln)
        if not fused and pre is not None:
            pre.xn = pre.mean = pre.rstd = None
        if not fused:
            segs = [(hb, C, (wb1, tap * F * C), C, C, tap - pad) for tap in range(kt)]
            if not bgemm_nt(segs, M, F, hid, F, T=T, bias=b1, relu=True, drop_p=p_in, drop_seed=s1, rowmask=pr):
                raise RuntimeError("bgemm_nt declined the FFN up-projection")
            if not bgemm_nt([(hid, F, wb2, F, F, 0)], M, N, out, N, bias=b2, drop_p=p_out, drop_seed=s2, res=r, ldr=N,
                            rowmask=zr):
                raise RuntimeError("bgemm_nt declined the FFN down-projection")
        ctx.cfg, ctx.seeds, ctx.dims, ctx.lead = cfg, (s1, s2), (M, F, C, kt, N), lead
        ctx.h_dtype = h.dtype
        ctx.save_for_backward(hb, hid, zr, wb1, wb2, wt2, wt1)
        return out.view(*lead, N)

    @staticmethod
    def backward(ctx, dy):
        from .ops import gzeros, wgrad_overlap

        hb, hid, zr, wb1, wb2, wt2, wt1 = ctx.saved_tensors
        M, F, C, kt, N = ctx.dims
        cfg = ctx.cfg
        T, pad, p_in, p_out = cfg["T"], cfg["pad"], cfg["p_inner"], cfg["p_out"]
        s1, s2 = ctx.seeds
        dy = _c(dy).view(M, N)
        token = cfg.get("token")
        if zr is not None and not (token is not None and token.delegated):
            dy = dy.masked_fill(zr.bool().view(M, 1), 0.0)
        d_res = dy.view(*ctx.lead, N)
        dev = dy.device
        # gradient at the hidden pre-activation: (dropout(dy) @ w2) gated by hid > 0 (ReLU, inner dropout, padded rows)
        dz = torch.empty((M, F), device=dev, dtype=BF16)
        a1 = 1.0 / (1.0 - p_in) if p_in > 0 else 1.0
        # both input-gradient contractions in one launch (images of the TRANSPOSED weights)
        # (k = 3: the three taps are summed in phase 2 from a tile of dz with one halo row either side)
        can_pair = cfg["pair"] and kt in (1, 3) and wt1 is not None and wt2 is not None and (kt == 1 or M % T == 0)
        plan = cfg.get("bwd")
        if (plan is not None and can_pair and kt == 1 and ctx.h_dtype == BF16 and dy.dtype == torch.float32
                and ctx.needs_input_grad[0] and ctx.needs_input_grad[5]):
            # fused PNCA block: this launch also runs the LayerNorm backward of the sub-layer's input and the input gradient of
            # the attention's output projection (_BlockBwd); the two nodes that follow adopt its results
            from .ops import gzeros_like

            plan.g1 = torch.empty((M, N), device=dev, dtype=torch.float32)
            plan.d_ox = torch.empty((M, N), device=dev, dtype=torch.float32)
            plan.d_oh = torch.empty((M, N), device=dev, dtype=torch.float32)
            plan.dg1, plan.db1 = gzeros_like(plan.gamma1), gzeros_like(plan.gamma1)
            part = pnca_block_bwd(dy, hid, plan.y1, plan.mean1, plan.rstd1, plan.gamma1.detach(), plan.rows, wt2, wt1,
                                  plan.wfcxT, plan.wfchT, alpha1=a1, drop2_p=p_out, drop2_seed=s2, fc_p=plan.fc_p,
                                  fc_seed=plan.fc_seed, dz=dz, g1=plan.g1, d_ox=plan.d_ox, d_oh=plan.d_oh)
            # the LayerNorm's dgamma / dbeta: per-workgroup partial rows, summed like every other parameter gradient (deferred:
            # one launch for all recorded sums at the next flush of the weight gradients)
            rows_sum_accum(part, plan.dg1, plan.db1)
            plan.placeholder, plan.d_res = hb, d_res
            plan.y1 = plan.mean1 = plan.rstd1 = plan.wfcxT = plan.wfchT = None
            dh, fused = hb, True  # stand-in: the LayerNorm node returns plan.g1 and never reads this
        else:
            dh = torch.empty((M, C), device=dev, dtype=ctx.h_dtype)
            fused = can_pair and ffn_pair(dy, wt2, wt1, dh, M=M, T=T, F=F, alpha1=a1, xdrop_p=p_out, xdrop_seed=s2, gate=hid,
                                          t_out=dz, KT2=kt, s2_first=pad, s2_step=-1)
        if not fused and not bgemm_nt([(dy, N, wb2, F, N, 0)], M, F, dz, F, b_kn=True, gate=hid, ldg=F, alpha=a1,
                                      a_drop_p=p_out, a_drop_seed=s2, a_drop_ld=N):
            raise RuntimeError("bgemm_nt declined the FFN hidden gradient")
        dw2 = gzeros((N, F, 1), dev)
        db2 = gzeros((N,), dev)
        with wgrad_overlap.side(dy, hid):
            if not bgemm_tn(dy, N, hid, F, M, N, F, dw2, F, 1, db=db2, a_drop_p=p_out, a_drop_seed=s2):
                raise RuntimeError("bgemm_tn declined dW2")
        if not fused:
            segs = [(dz, F, (wb1, tap * F * C), C, F, pad - tap) for tap in range(kt)]
            if not bgemm_nt(segs, M, C, dh, C, T=T, b_kn=True):
                raise RuntimeError("bgemm_nt declined the FFN input gradient")
        dw1 = gzeros((F, C, kt), dev)
        db1 = gzeros((F,), dev)
        with wgrad_overlap.side(dz, hb):
            if not bgemm_tn(dz, F, hb, C, M, F, C, dw1, C * kt, kt, c_ts=1, T=T, ntaps=kt, shift0=-pad, shift_step=1,
                            db=db1):
                raise RuntimeError("bgemm_tn declined dW1")
        return (dh.view(*ctx.lead, C), dw1, db1, dw2, db2, d_res) + (None,) * 9


SHARED_ONE_GEMM = {"on": not os.environ.get("KANTTS_NO_SHARED_ONE_GEMM")}  # A/B switch


class _SharedInputLinearsB(torch.autograd.Function):
    """n Linear layers that read the SAME input (the memory K/V projections ``w_h_kv`` of every PNCA block,
    kantts/models/sambert/__init__.py:286-299: all twelve read the length-regulated memory tensor).  Forward: one launch
    over the stacked weights when the layers have one shape (else one per layer).  Backward: ONE launch for the input gradient -- sum_b dy_b . W_b is a contraction with
    one segment per layer -- instead of n launches plus the n - 1 accumulation adds autograd inserts for a tensor with n
    consumers; weight / bias gradients as everywhere (deferred, grouped by shape)."""

    @staticmethod
    def forward(ctx, x, n, *t):
        ws, bs, wbs = t[:n], t[n:2 * n], t[2 * n:3 * n]
        x = _c(x)
        lead, K = x.shape[:-1], x.shape[-1]
        M = int(math.prod(lead))
        outs = []
        same = all(w.shape == ws[0].shape for w in ws) and (all(b is not None for b in bs) or all(b is None for b in bs))
        if same and SHARED_ONE_GEMM["on"]:
            # ONE launch: the n weights stacked (n*N, K) (a 1 MB bf16 copy per step), the n outputs are column blocks of
            # one (M, n*N) buffer -- consumers take the row pitch (ops._PncaAttention)
            N = ws[0].shape[0]
            wcat = torch.cat([wb.view(N, K) for wb in wbs], 0)
            bcat = torch.cat(list(bs), 0) if bs[0] is not None else None
            y = torch.empty((M, n * N), device=x.device, dtype=torch.float32)
            if not bgemm_nt([(x.detach(), K, wcat, K, K, 0)], M, n * N, y, n * N, bias=bcat):
                raise RuntimeError("bgemm_nt declined the stacked shared-input projection")
            outs = list(y.view(*lead, n, N).unbind(-2))
        else:
            for w, b, wb in zip(ws, bs, wbs):
                N = w.shape[0]
                y = torch.empty((M, N), device=x.device, dtype=torch.float32)
                if not bgemm_nt([(x.detach(), K, wb, K, K, 0)], M, N, y, N, bias=b):
                    raise RuntimeError("bgemm_nt declined a shared-input projection")
                outs.append(y.view(*lead, N))
        ctx.dims = (n, M, K, lead, [tuple(w.shape) for w in ws], [b is not None for b in bs], x.dtype)
        ctx.save_for_backward(x.detach(), *wbs)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        from .ops import gzeros, wgrad_overlap

        n, M, K, lead, wshapes, has_b, x_dtype = ctx.dims
        x, wbs = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        dev = x.device
        dys = [torch.zeros((M, ws[0]), device=dev) if d is None else _c(d).view(M, ws[0]) for d, ws in zip(dys, wshapes)]
        needs = ctx.needs_input_grad  # (x, n, *ws, *bs, *wbs)
        dx = None
        if needs[0]:
            dx = torch.empty((M, K), device=dev, dtype=x_dtype)
            segs = [(d, ws[0], wb, K, ws[0], 0) for d, ws, wb in zip(dys, wshapes, wbs)]
            if not bgemm_nt(segs, M, K, dx, K, b_kn=True):
                raise RuntimeError("bgemm_nt declined the shared input gradient")
            dx = dx.view(*lead, K)
        dws, dbs = [None] * n, [None] * n
        for i, (d, ws) in enumerate(zip(dys, wshapes)):
            if not needs[2 + i]:
                continue
            N = ws[0]
            dw = gzeros(ws, dev)
            db = gzeros((N,), dev) if (has_b[i] and needs[2 + n + i]) else None
            with wgrad_overlap.side(d, x):
                if not bgemm_tn(d, N, x, K, M, N, K, dw, K, 1, db=db):
                    raise RuntimeError("bgemm_tn declined a shared-input weight gradient")
            dws[i], dbs[i] = dw, db
        return (dx, None) + tuple(dws) + tuple(dbs) + (None,) * n


def shared_input_linears(x, weights, biases):
    """[x @ W_i^T + b_i for i] for Linear weights W_i (N_i, K) that all read x; None when the bf16 kernels cannot take it."""
    n = len(weights)
    if n < 2 or n > 12 or x.shape[-1] % 8 or x.numel() == 0 or any(w.dim() != 2 or w.shape[0] % 8 or w.shape[1] != x.shape[-1]
                                                                   for w in weights):
        return None
    return list(_SharedInputLinearsB.apply(x, n, *weights, *biases, *[bf16_weight(w) for w in weights]))


def ffn_eligible(h, w1, w2):
    return (w1.dim() == 3 and w2.dim() == 3 and w2.shape[2] == 1 and w1.shape[2] <= 12 and w1.shape[0] % 8 == 0 and
            w1.shape[1] % 8 == 0 and w2.shape[0] % 8 == 0 and h.numel() > 0)


def ffn(h, w1, b1, w2, b2, res, *, pad_rows=None, zero_rows=None, p_inner=0.0, p_out=0.0, T=0, ln_next=None):
    """h: LayerNorm output (B, T, C) (bf16 or fp32); w1 (F, C, k), w2 (C_out, F, 1) Conv1d weights; res (B, T, C_out)."""
    F, C, kt = w1.shape
    # csrc/ffn_pair.hip: 128 channels either side, 1024 hidden units, odd kernel width
    pair = PAIR["on"] and C == 128 and w2.shape[0] == 128 and F == 1024 and kt % 2 == 1 and kt <= 9
    cfg = dict(T=int(T or h.shape[-2]), pad=(kt - 1) // 2, p_inner=float(p_inner), p_out=float(p_out), pair=pair)
    token = RowMaskToken(zero_rows) if (zero_rows is not None and torch.is_grad_enabled()) else None
    cfg["token"] = token
    pre = PreNorm(ln_next) if (ln_next is not None and PRENORM["on"] and pair) else None
    cfg["ln_next"] = pre
    wf1 = wf2 = wt2 = wt1 = None
    if pair:
        wf1, wf2, wt2, wt1 = ffn_frag_weights(w1, w2)
        if kt not in (1, 3):
            wt2 = wt1 = None  # backward stays on the two-launch form
    y = _attach_token(_FusedFFNB.apply(h, w1, b1, w2, b2, res, pad_rows, zero_rows, conv_weight_bf16(w1),
                                       conv_weight_bf16(w2), wf1, wf2, wt2, wt1, cfg), token)
    if pre is not None and pre.xn is not None:
        y._kantts_prenorm = pre
    return y
