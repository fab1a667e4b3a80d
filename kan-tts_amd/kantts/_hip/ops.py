"""torch.autograd.Function wrappers over the C ABI (one forward + one backward entry per fused op).

Activations are channels-last (rows = tokens) fp32; every Function saves only what its backward
kernels need (outputs for ReLU gating, LayerNorm statistics, attention log-sum-exp, LSTM gates).
Nothing here computes on the host or with ATen kernels except trivial views / allocations.
"""
import itertools
import os
import weakref
import math

import torch

from . import (E_UNSUPPORTED, act_cast_bf16, bgemm_nt, bgemm_tn, cconv, cconv_wgrad, check, get_precision, conv_c1, conv_n1, conv_wgrad,
               conv_win, gemm, lib, make_seg, ptr, rng_state, stream)
from . import ops_bf16

_seed_counter = itertools.count(1)


def next_seed():
    """Per-call dropout seed derived from torch's seed (deterministic under torch.manual_seed)."""
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + next(_seed_counter) * 0xD1B54A32D192ED03) & ((1 << 63) - 1)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _new_side_stream():
    """Streams of work that runs BESIDE the step's critical path (deferred weight gradients, the variance-predictor branch).
    KANTTS_SIDE_PRIORITY (experiment switch): HIP stream priority, larger = lower; default: the device default."""
    pr = os.environ.get("KANTTS_SIDE_PRIORITY")
    return torch.cuda.Stream(priority=int(pr)) if pr else torch.cuda.Stream()


class _ZeroPool:
    """Bump allocator over one pre-zeroed device buffer for the per-step gradient accumulators (weight / bias
    gradients are produced by split-K atomics and row sums, so they must start at zero).  One memset per step
    (reset) replaces ~400 tiny fill kernels.  Opt-in (ArenaAdam enables it); tensors handed out are only
    valid until the next reset, which ArenaAdam.zero_grad() performs."""

    def __init__(self):
        self.buf = None
        self.off = self.hw = 0
        # direct gradients (train/optim.py: ParamArena.enable_direct_grads): the n-th take() after a reset that turned out
        # to become a parameter's .grad is served from that parameter's range of the gradient arena from then on
        self.calls = 0
        self.plan = None  # {call index: arena gradient view}
        self.log = None   # while recording: [(call index, data pointer, numel)]

    def enable(self, numel, device):
        """Reserve room for ``numel`` more accumulators (every arena optimizer of the process adds its share:
        a GAN generator backward also fills the discriminators' gradients)."""
        have = 0 if (self.buf is None or self.buf.device != torch.device(device)) else self.buf.numel()
        self.buf = torch.zeros(int(have + numel), device=device, dtype=torch.float32)
        self.off = 0
        self.hw = 0  # highest offset ever handed out: everything beyond it is still zero

    def reset(self):
        """Re-zero what may have been written: the prefix up to the highest offset ever handed out, not the whole buffer
        (sized for every arena of the process: 0.5 GB for the three HiFi-GAN models, reset by each of their zero_grad()
        calls, three times per step).  ``hw`` only grows, so a captured reset covers at least what its own graph uses.
        (Skipping a reset that follows another one without a request in between was tried and is WRONG under capture: the
        reset in front of a capture is not part of the graph, the captured step then never re-zeroed its accumulators --
        the bench loss moved in the fourth digit, profiles/r04_runAD.)"""
        self.calls = 0
        if self.buf is not None:
            self.buf[:min(self.buf.numel(), max(64, (self.hw + 63) // 64 * 64))].zero_()
            self.off = 0

    def take(self, shape, device):
        n = 1
        for d in shape:
            n *= int(d)
        idx = self.calls
        self.calls += 1
        if self.plan is not None:
            v = self.plan.get(idx)
            # a changed call sequence (another code path, another module) only costs the copy that pack_grads() makes for
            # every gradient it does not find in its own slot
            if v is not None and v.numel() == n and v.device == device:
                return v.view(shape)
        t = self._take(shape, device, n)
        if self.log is not None:
            self.log.append((idx, t.data_ptr(), n))
        return t

    def _take(self, shape, device, n):
        if self.buf is None or self.buf.device != device or self.off + n > self.buf.numel():
            return torch.zeros(shape, device=device, dtype=torch.float32)
        a = (self.off + 63) // 64 * 64
        if a + n > self.buf.numel():
            return torch.zeros(shape, device=device, dtype=torch.float32)
        self.off = a + n
        self.hw = max(self.hw, self.off)
        return self.buf[a:a + n].view(shape)


zero_pool = _ZeroPool()


def gzeros(shape, device):
    """Zero-initialised fp32 accumulator (from the per-step pool when enabled)."""
    return zero_pool.take(tuple(shape), torch.device(device))


def gzeros_like(t):
    return zero_pool.take(tuple(t.shape), t.device)


def _splitk_for(n_out_rows, n_out_cols, k_total):
    tiles = ((n_out_rows + 63) // 64) * ((n_out_cols + 63) // 64)
    ktiles = max(1, (k_total + 31) // 32)
    want = max(1, 512 // max(1, tiles))
    return int(max(1, min(want, ktiles, 64)))


# ================================================================================================
# Weight-gradient side stream
# ================================================================================================
class _WgradOverlap:
    """Opt-in: weight-gradient contractions run on a second HIP stream.

    In backward the chain of input gradients is the critical path; every weight gradient is a leaf that only the
    optimizer reads.  The SAM-BERT contractions are small (1-6 workgroups per CU), so a leaf running beside the
    chain fills CUs that would idle.  ``with wgrad_overlap.side(*tensors)`` forks the side stream from the
    current one (event), runs the body there and keeps ``tensors`` alive until ``join()`` -- the caching
    allocator may otherwise hand their memory to a later main-stream allocation while the side kernel is still
    pending.  ``join()`` (ArenaAdam.step) makes the current stream wait for the side stream.  Under hipGraph
    capture the fork / join become parallel graph branches.  Off by default: code that reads ``p.grad`` right
    after ``backward()`` without an optimizer step (tests) must not enable it."""

    def __init__(self):
        self.enabled = False
        self._stream = None
        self._keep = []
        self._used = False

    def enable(self, on=True, group_wgrads=None):
        """``group_wgrads`` (default: same as ``on``): bf16-mode weight gradients are recorded and issued grouped by
        shape at ``join()`` instead of one launch each (kantts._hip.deferred_tn)."""
        from . import deferred_tn

        self.enabled = bool(on)
        side_branch.enabled = bool(on) and os.environ.get("KANTTS_NO_SIDE_BRANCH", "") == ""
        deferred_tn.enabled = bool(on if group_wgrads is None else group_wgrads)
        if not deferred_tn.enabled:
            deferred_tn.flush()

    class _Ctx:
        def __init__(self, owner, keep):
            self.owner, self.keep, self.cm = owner, keep, None

        def __enter__(self):
            o = self.owner
            if not o.enabled:
                return self
            if o._stream is None:
                o._stream = _new_side_stream()
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            o._stream.wait_event(ev)
            self.cm = torch.cuda.stream(o._stream)
            self.cm.__enter__()
            o._keep.extend(t for t in self.keep if t is not None)
            o._used = True
            return self

        def __exit__(self, *exc):
            if self.cm is not None:
                self.cm.__exit__(*exc)
            return False

    def side(self, *keep):
        return _WgradOverlap._Ctx(self, keep)

    def flush_early(self):
        """Issue the weight gradients recorded so far on the side stream NOW (called from ``wgrad_flush_point`` in the
        middle of backward): they then run beside the rest of the backward chain instead of after it."""
        from . import deferred_tn

        if not self.enabled or not (deferred_tn.groups or deferred_tn.copies or deferred_tn.rowsums):
            return
        # operands were allocated on the main stream; flush() drops its references once the launches are issued
        keep = [k for probs in deferred_tn.groups.values() for q in probs for k in q[6] if k is not None]
        keep += [r[0] for r in deferred_tn.rowsums]
        with self.side(*keep):
            deferred_tn.flush()

    def join(self):
        from . import deferred_tn

        deferred_tn.flush()  # recorded weight gradients: grouped launches on the current stream
        flush_weight_norm_backward()  # ... and the weight-norm backward of whole networks (one launch each)
        side_branch.join()
        side_branch.release()
        if self._used:
            ev = torch.cuda.Event()
            ev.record(self._stream)
            torch.cuda.current_stream().wait_event(ev)
            self._keep.clear()
            self._used = False


wgrad_overlap = _WgradOverlap()


_branch_streams = []


class _BranchExit(torch.autograd.Function):
    """Identity at the end of a branch (applied on the branch's stream).  Backward hands the branch a PRIVATE copy of the
    incoming gradient.  Why: the gradient of ``sum(branch outputs)`` is ONE tensor that autograd gives to every branch,
    each consuming it on its own stream.  Layers with a residual port pass their incoming gradient on as the residual's
    gradient (the same tensor object), and autograd adds the next contribution INTO such a tensor in place once it holds
    the last reference -- which it does as soon as the other branches' backward nodes have been *issued*, not when their
    kernels have *finished*.  The in-place add on one stream then races with reads still pending on the other two
    (batch 32 fp32 generator: 3-10 % error in some weight gradients, two runs out of five; profiles/r03_runAO-AR_*).
    One copy per branch and stage on the branch's own stream (~0.1 ms per generator backward) removes the sharing."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.clone()


def _private_grad(out):
    if torch.is_tensor(out):
        return _BranchExit.apply(out) if out.requires_grad else out
    if isinstance(out, (tuple, list)):
        return type(out)(_private_grad(o) for o in out)
    return out


def parallel_branches(thunks, inputs=(), private_grads=False):
    """Run independent sub-networks (``thunks``: callables without arguments) each on its own HIP stream and join them:
    the eight sub-discriminators of HiFi-GAN's MPD / MSD read the same waveform and share nothing, and a good half of
    their launches are tiny (weight-norm re-parametrisations, 1-channel first layers, strided layers over a few thousand
    tokens) -- back to back on one stream they leave most of the chip idle.  autograd replays every node on the stream of
    its forward op, so the backward passes are concurrent as well.  ``inputs``: tensors allocated on the current stream
    that the branches read (recorded on every branch stream for the caching allocator).  ``private_grads``: the branch
    outputs are summed by the caller and the branches contain residual ports -- see _BranchExit.  Sequential on the host,
    and when ``KANTTS_NO_BRANCH_STREAMS=1`` also on the device."""
    if (len(thunks) < 2 or not torch.cuda.is_available() or os.environ.get("KANTTS_NO_BRANCH_STREAMS", "") != ""
            ):
        return [t() for t in thunks]
    dev_inputs = [t for t in inputs if torch.is_tensor(t) and t.is_cuda]
    if not dev_inputs:
        return [t() for t in thunks]
    while len(_branch_streams) < len(thunks):
        _branch_streams.append(_new_side_stream())
    main = torch.cuda.current_stream()
    fork = torch.cuda.Event()
    fork.record(main)
    outs, done = [], []
    for t, st in zip(thunks, _branch_streams):
        st.wait_event(fork)
        for x in dev_inputs:
            x.record_stream(st)
            img = getattr(x, _IMG_ATTR, None)  # the bf16 operand image the branches read instead of x (act_image)
            if img is not None and torch.is_tensor(img[1]) and img[1].is_cuda:
                img[1].record_stream(st)
        with torch.cuda.stream(st):
            o = t()
            outs.append(_private_grad(o) if private_grads else o)
        ev = torch.cuda.Event()
        ev.record(st)
        done.append(ev)
    for ev in done:
        main.wait_event(ev)
    return outs


class _FlushPoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        wgrad_overlap.flush_early()
        return g


def wgrad_flush_point(x):
    """Identity.  When the gradient of ``x`` arrives in backward, every weight gradient recorded so far (all layers
    DOWNSTREAM of x in forward) is issued on the side stream.  Placed at the encoder output of SAM-BERT: the decoder's
    ~100 deferred weight gradients (1.2 ms of grouped launches) then run beside the encoder's backward -- 8 blocks over
    2048 tokens, a chain of launches with 64-128 workgroups on a 256-CU chip -- instead of after it; and at the postnet
    input (its 19 584-row weight gradients run beside the decoder's backward)."""
    from . import deferred_tn

    if not (wgrad_overlap.enabled and deferred_tn.enabled and torch.is_tensor(x) and x.requires_grad
            and os.environ.get("KANTTS_NO_EARLY_FLUSH", "") == ""):
        return x
    y = _FlushPoint.apply(x)
    for attr in ("_kantts_rowmask", "_kantts_prenorm"):  # hand-overs between sub-layers ride through the identity
        v = getattr(x, attr, None)
        if v is not None:
            setattr(y, attr, v)
    return y


class _SideBranch:
    """Opt-in: a sub-network whose results the main chain does not need until the end of forward runs on a second HIP
    stream -- in training the three variance predictors of SAM-BERT (pitch / energy / duration: FSMN + BiLSTM / LSTM
    stacks over the 64-symbol text axis, ~0.75 ms of small launches per step that occupy a quarter of the chip) only feed
    their own losses, while length regulation, decoder and postnet use the TARGET durations / pitch / energy
    (kantts_sambert.py:392-470).  ``with side_branch.fork(*tensors)`` makes the side stream wait for the current one and
    runs the body there; autograd replays every node on the stream of its forward op, so the branch's backward is
    concurrent with the decoder's backward as well, and under hipGraph capture both become parallel graph branches.
    ``tensors`` (allocated on the main stream, read by side-stream kernels) are kept alive until
    ``wgrad_overlap.join()`` (the optimizer step): the caching allocator would otherwise hand their memory to a later
    main-stream allocation while a side kernel is still pending.  ``join()`` makes the current stream wait for the branch.
    Enabled together with ``wgrad_overlap`` (GraphedSambertStep, bench.py); off by default."""

    def __init__(self):
        self.enabled = False
        self._stream = None
        self._keep = []
        self._open = False

    class _Ctx:
        def __init__(self, owner, keep, after=None):
            self.owner, self.keep, self.cm, self.after = owner, keep, None, after

        def __enter__(self):
            o = self.owner
            if not o.enabled or not torch.cuda.is_available():
                return self
            if o._stream is None:
                o._stream = _new_side_stream()
            ev = self.after
            if ev is None:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
            o._stream.wait_event(ev)
            self.cm = torch.cuda.stream(o._stream)
            self.cm.__enter__()
            o._keep.extend(t for t in self.keep if t is not None)
            o._open = True
            return self

        def __exit__(self, *exc):
            if self.cm is not None:
                self.cm.__exit__(*exc)
            return False

    def fork(self, *keep, after=None):
        """``after``: an event recorded (``mark()``) where the branch's inputs became ready; the branch then waits for THAT
        point of the issuing stream, not for everything issued since -- a branch that is issued late (so that autograd issues
        its backward early) is still a parallel branch from where its inputs were produced."""
        return _SideBranch._Ctx(self, keep, after)

    def mark(self):
        """An event on the current stream at this point (None when the side branch is off)."""
        if not self.enabled or not torch.cuda.is_available():
            return None
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        return ev

    def join(self):
        if self._open:
            ev = torch.cuda.Event()
            ev.record(self._stream)
            torch.cuda.current_stream().wait_event(ev)
            self._open = False

    def release(self):
        self._keep.clear()


side_branch = _SideBranch()


def helper_streams():
    """Every stream this module issues work on beside the caller's current one."""
    return [st for st in [wgrad_overlap._stream, side_branch._stream] + list(_attn_side_stream) + list(_branch_streams)
            if st is not None]


def join_capturing_side_streams():
    """Make the current (capturing) stream wait for every helper stream of this module that is part of the SAME capture.
    A hipGraph capture can only end when all streams forked from the origin stream have been joined back; autograd joins
    its streams at the end of backward(), so a capture that is cut IN THE MIDDLE of backward (train/segments.py: one graph
    per gradient bucket of a data-parallel step) has to do it itself.  Streams that are not capturing are left alone:
    waiting for an event recorded outside the capture would create a dependency across its boundary."""
    if not torch.cuda.is_available():
        return 0
    main = torch.cuda.current_stream()
    n = 0
    for st in helper_streams():
        if st == main:
            continue
        with torch.cuda.stream(st):
            capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            main.wait_stream(st)
            n += 1
    return n


def fork_helper_streams():
    """The counterpart at the START of a capture segment: every helper stream waits for the current (capturing) stream and
    thereby belongs to the new capture.  autograd replays a node on the stream of its forward op and synchronises that
    stream with the producer of the node's input when the input is PRODUCED -- possibly in the segment that has just
    ended; without this, the first backward node that runs on a helper stream after a cut would execute eagerly, outside
    any capture."""
    if not torch.cuda.is_available():
        return
    main = torch.cuda.current_stream()
    for st in helper_streams():
        if st != main:
            st.wait_stream(main)


# ================================================================================================
# Fused linear / token-convolution
# ================================================================================================
class _FusedLinear(torch.autograd.Function):
    """y = rowmask( dropout( act( (sum_k x_k @ W_k^T + bias [+ bias2]) * alpha ) ) + res )

    modes
      concat : one weight (N, sum K_k), inputs x_k side by side      (torch.cat + nn.Linear)
      sum    : separate weights W_k (N, K_k) and biases               (fc_x(.) + fc_h(.))
      conv   : one input, weight (N, Cin, KT), taps shift tokens      (nn.Conv1d over time, 'same' pad)
    """

    @staticmethod
    def forward(ctx, opts, bias, bias2, res, rowmask, *xw):
        nx = opts["nx"]
        xs = [_c(t) for t in xw[:nx]]
        ws = list(xw[nx:])
        mode = opts["mode"]
        relu, alpha, drop_p = opts["relu"], opts["alpha"], opts["drop_p"]
        lead = xs[0].shape[:-1]
        M = int(math.prod(lead))
        N = ws[0].shape[0]
        T = opts.get("T", 0)
        y = torch.empty((M, N), device=xs[0].device, dtype=torch.float32)
        segs = []
        if mode == "conv":
            # tap-major copy (KT, N, Cin) of the (N, Cin, KT) Conv1d weight: every tap becomes a k-contiguous
            # matrix, so forward / dgrad / wgrad all run on the float4-staged GEMM kernels
            cin, kt = ws[0].shape[1], ws[0].shape[2]
            w = ws[0].permute(2, 0, 1).contiguous()
            ws = [w]
            pad = opts["pad"]
            segs.append(make_seg(xs[0], cin, 1, w, cin, 1, cin, ntaps=kt, b_tap=N * cin, a_tok_axis=1,
                                 a_shift0=-pad, a_shift_step=opts.get("dilation", 1)))
        elif mode == "concat":
            w = _c(ws[0])
            ws = [w]
            ldw = w.shape[1]
            off = 0
            for x in xs:
                k = x.shape[-1]
                segs.append(make_seg(x, k, 1, (w, off), ldw, 1, k))
                off += k
            assert off == ldw, "concat widths do not match the weight"
        else:  # sum
            ws = [_c(w) for w in ws]
            for x, w in zip(xs, ws):
                k = x.shape[-1]
                assert w.shape[1] == k
                segs.append(make_seg(x, k, 1, w, k, 1, k))
        seed = next_seed() if drop_p > 0 else 0
        r = _c(res).view(M, N) if res is not None else None
        rm = _c(rowmask).view(M) if rowmask is not None else None
        gemm(segs, M, N, y, N, 1, bias=bias, bias2=bias2, res=r, r_is=N, r_js=1, rowmask=rm, alpha=alpha,
             relu=relu, T=T, drop_p=drop_p, drop_seed=seed)
        ctx.opts, ctx.seed, ctx.M, ctx.N, ctx.lead = opts, seed, M, N, lead
        ctx.has = (bias is not None, bias2 is not None, res is not None)
        # ReLU backward gates on the activation's own output; with a residual on top that is y - res (no shipped model
        # combines the two, the plain case keeps y itself), zeroed rows stay closed
        gate_src = None
        if relu:
            gate_src = y if r is None else (y - r)
            if r is not None and rm is not None:
                gate_src = gate_src.masked_fill(rm.bool().view(M, 1), 0.0)
        ctx.save_for_backward(gate_src, rm, *xs, *ws)
        return y.view(*lead, N)

    @staticmethod
    def backward(ctx, dy):
        opts, M, N = ctx.opts, ctx.M, ctx.N
        nx, mode, relu, alpha, drop_p = opts["nx"], opts["mode"], opts["relu"], opts["alpha"], opts["drop_p"]
        T = opts.get("T", 0)
        saved = ctx.saved_tensors
        y_gate, rm = saved[0], saved[1]
        xs, ws = list(saved[2:2 + nx]), list(saved[2 + nx:])
        has_bias, has_bias2, has_res = ctx.has
        dy = _c(dy).view(M, N)
        token = opts.get("token")  # ops_bf16.RowMaskToken: the consuming LayerNorm has zeroed these rows already
        if rm is not None and not relu and not (token is not None and token.delegated):
            dy = dy.masked_fill(rm.bool().view(M, 1), 0.0)
        d_res = dy.view(*ctx.lead, N) if has_res else None
        if has_res and rm is not None and relu:
            d_res = dy.masked_fill(rm.bool().view(M, 1), 0.0).view(*ctx.lead, N)
        # gradient that reaches the pre-activation: gate / dropout are applied inside the A loader
        gate = y_gate if relu else None
        a_drop_p, a_seed, balpha = 0.0, 0, alpha
        if drop_p > 0:
            if relu:
                balpha = alpha / (1.0 - drop_p)  # kept & active elements are exactly where y > 0
            else:
                a_drop_p, a_seed = drop_p, ctx.seed
        needs = ctx.needs_input_grad  # (opts, bias, bias2, res, rowmask, *xw)
        dxs, dws = [None] * nx, [None] * len(ws)
        dbias = gzeros((N,), dy.device) if (has_bias or has_bias2) else None
        first_tn = True
        if mode == "conv":
            w = ws[0]  # tap-major (KT, N, Cin)
            kt, cin = w.shape[0], w.shape[2]
            pad, dil = opts["pad"], opts.get("dilation", 1)
            x = xs[0]
            if needs[5]:
                dx = torch.empty_like(x)
                seg = make_seg(dy, N, 1, w, 1, cin, N, ntaps=kt, b_tap=N * cin, a_tok_axis=1, a_shift0=pad,
                               a_shift_step=-dil, a_gate=gate, a_drop_p=a_drop_p, a_drop_seed=a_seed)
                gemm([seg], M, cin, dx, cin, 1, alpha=balpha, T=T)
                dxs[0] = dx
            if needs[5 + nx]:
                dw = gzeros_like(w)
                sk = _splitk_for(N, cin, M)
                with wgrad_overlap.side(dy, x, gate):
                    for tap in range(kt):
                        seg = make_seg(dy, 1, N, x, 1, cin, M, a_gate=gate, a_drop_p=a_drop_p, a_drop_seed=a_seed,
                                       b_tok_axis=2, b_shift0=tap * dil - pad)
                        gemm([seg], N, cin, dw, cin, 1, c_off=tap * N * cin, alpha=balpha, accumulate=True, splitk=sk,
                             T=T, a_rowsum=dbias if (first_tn and dbias is not None) else None)
                        first_tn = False
                    # back to the parameter's (N, Cin, KT) layout.  Materialised HERE (same stream as the
                    # contractions): autograd clones a gradient whose strides differ from its parameter's, and
                    # that clone would run on the main stream before a side-stream weight gradient has landed.
                    dws[0] = dw.permute(1, 2, 0).contiguous()
        else:
            off = 0
            ldw = ws[0].shape[1]
            if mode == "concat" and needs[5 + nx]:
                dws[0] = gzeros_like(ws[0])
            for k, x in enumerate(xs):
                kk = x.shape[-1]
                w = ws[0] if mode == "concat" else ws[k]
                woff = off if mode == "concat" else 0
                wld = ldw if mode == "concat" else kk
                if needs[5 + k]:
                    dx = torch.empty_like(x)
                    seg = make_seg(dy, N, 1, (w, woff), 1, wld, N, a_gate=gate, a_drop_p=a_drop_p, a_drop_seed=a_seed)
                    gemm([seg], M, kk, dx, kk, 1, alpha=balpha)
                    dxs[k] = dx
                need_w = needs[5 + nx] if mode == "concat" else needs[5 + nx + k]
                if need_w:
                    if mode != "concat":
                        dws[k] = gzeros_like(w)
                    dwt = dws[0] if mode == "concat" else dws[k]
                    seg = make_seg(dy, 1, N, x, 1, kk, M, a_gate=gate, a_drop_p=a_drop_p, a_drop_seed=a_seed)
                    with wgrad_overlap.side(dy, x, gate):
                        gemm([seg], N, kk, dwt, wld, 1, c_off=woff, alpha=balpha, accumulate=True,
                             splitk=_splitk_for(N, kk, M),
                             a_rowsum=dbias if (first_tn and dbias is not None) else None)
                    first_tn = False
                off += kk
        if dbias is not None and first_tn:
            # no weight gradient was requested but a bias needs one: plain column sum through the GEMM
            raise RuntimeError("bias gradient without weight gradient is not supported")
        if dbias is not None and balpha != 1.0:
            with wgrad_overlap.side(dbias):
                dbias = dbias * balpha
        return (None, dbias if has_bias else None, dbias if has_bias2 else None, d_res, None, *dxs, *dws)


def linear(xs, weights, bias=None, *, mode=None, bias2=None, res=None, rowmask=None, relu=False, alpha=1.0,
           drop_p=0.0, pad=0, dilation=1, T=0, out_bf16=False, ln_next=None):
    """Functional entry: xs / weights are tensors or lists (see _FusedLinear).  In bf16 mode the contraction runs on
    the bf16-operand kernels (ops_bf16) whenever its extents allow; ``out_bf16`` then stores the result as bf16 (for
    outputs whose only consumers are contractions).  fp32 mode ignores it.  ``ln_next``: the nn.LayerNorm(128) of the
    pre-LN sub-layer that consumes the output -- bf16 mode computes it in this launch's epilogue (ops_bf16.PreNorm)."""
    xs = [xs] if torch.is_tensor(xs) else list(xs)
    weights = [weights] if torch.is_tensor(weights) else list(weights)
    if mode is None:
        mode = "conv" if weights[0].dim() == 3 and (weights[0].shape[2] > 1 or pad) else (
            "sum" if len(weights) > 1 else "concat")
    params = weights
    if mode != "conv" and weights[0].dim() == 3:  # Conv1d with kernel 1 == Linear
        weights = [w.squeeze(-1) if w.dim() == 3 else w for w in weights]
    if mode == "conv":
        T = T or xs[0].shape[-2]
    if get_precision() == "bf16" and ops_bf16.eligible(xs, weights, mode, relu, res):
        if mode == "conv":
            wbs = [ops_bf16.conv_weight_bf16(params[0])]
        else:
            wbs = [ops_bf16.bf16_weight(w) for w in params]
        return ops_bf16.linear(xs, weights, wbs, bias, mode=mode, bias2=bias2, res=res, rowmask=rowmask, relu=relu,
                               alpha=alpha, drop_p=drop_p, pad=pad, dilation=dilation, T=T, out_bf16=out_bf16,
                               ln_next=ln_next)
    xs = [x.float() if x.dtype != torch.float32 else x for x in xs]  # the segmented GEMM reads fp32 operands
    token = ops_bf16.RowMaskToken(rowmask) if (rowmask is not None and not relu and torch.is_grad_enabled()) else None
    opts = dict(nx=len(xs), mode=mode, relu=bool(relu), alpha=float(alpha), drop_p=float(drop_p), pad=int(pad),
                dilation=int(dilation), T=int(T), token=token)
    return ops_bf16._attach_token(_FusedLinear.apply(opts, bias, bias2, res, rowmask, *xs, *weights), token)


def pnca_block_fused(blk, x, hkv, info, bw_x, bw_h, bw_dev, return_attn, next_ln, training):
    """bf16 mode on a HIP device: ops_bf16.pnca_block_fused (one launch for the block's forward pass); otherwise a no-op
    context (the fp32 parity path keeps its launches)."""
    import contextlib

    if get_precision() != "bf16":
        return contextlib.nullcontext()
    return ops_bf16.pnca_block_fused(blk, x, hkv, info, bw_x, bw_h, bw_dev, return_attn, next_ln, training)


def enc_attn_fused(att, x, info, rows, return_attn, next_ln, training):
    """bf16 mode on a HIP device: ops_bf16.enc_attn_fused (one launch for the forward pass of an encoder block's attention
    sub-layer); otherwise a no-op context (the fp32 parity path keeps its launches)."""
    import contextlib

    if get_precision() != "bf16":
        return contextlib.nullcontext()
    return ops_bf16.enc_attn_fused(att, x, info, rows, return_attn, next_ln, training)


def shared_input_linears(x, linears):
    """``[lin(x) for lin in linears]`` for nn.Linear holders that all read the same tensor.  bf16 mode with gradients
    enabled: forward as usual, but ONE input-gradient launch for all of them (ops_bf16._SharedInputLinearsB)."""
    if get_precision() == "bf16" and torch.is_grad_enabled() and x.requires_grad:
        outs = ops_bf16.shared_input_linears(x, [m.weight for m in linears], [m.bias for m in linears])
        if outs is not None:
            return outs
    return [linear(x, m.weight, m.bias) for m in linears]


# ================================================================================================
# LayerNorm
# ================================================================================================
class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = _c(x)
        C = x.shape[-1]
        M = x.numel() // C
        y = torch.empty_like(x)
        mean = torch.empty(M, device=x.device, dtype=torch.float32)
        rstd = torch.empty(M, device=x.device, dtype=torch.float32)
        check(lib().kantts_layernorm_fwd(ptr(x, torch.float32), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd),
                                         M, C, float(eps), stream()), "layernorm_fwd")
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        dy = _c(dy)
        C = x.shape[-1]
        M = x.numel() // C
        dx = torch.empty_like(x)
        dg = gzeros_like(gamma)
        db = gzeros_like(gamma)
        check(lib().kantts_layernorm_bwd(ptr(dy, torch.float32), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx),
                                         ptr(dg), ptr(db), M, C, stream()), "layernorm_bwd")
        return dx, dg, db, None


def layer_norm(x, gamma, beta, eps=1e-6, out_bf16=False, with_res=False, private_input=False):
    """``out_bf16`` (bf16 mode, 128-wide rows only): the normalised activations are written bf16 -- they only feed
    contractions.  128-wide rows use the 16-lanes-per-row kernels in both modes.
    ``with_res``: returns (y, x_res) where x_res is x routed through this node -- use it as the sub-layer's residual input
    and the two gradients of x are summed inside the LayerNorm backward kernel (see ops_bf16._LayerNorm128).
    ``private_input``: promise that nothing else consumes ``x`` -- the row mask its producer applies to the incoming
    gradient moves into this node's backward kernel (ops_bf16.RowMaskToken)."""
    if x.shape[-1] == 128 and x.dtype == torch.float32 and x.numel() > 0:
        return ops_bf16.layer_norm128(x, gamma, beta, eps, out_bf16 and get_precision() == "bf16", with_res,
                                      private_input=private_input)
    y = _LayerNorm.apply(x, gamma, beta, eps)
    return (y, x) if with_res else y


def ffn(h, w1, b1, w2, b2, res, *, pad_rows=None, zero_rows=None, p_inner=0.0, p_out=0.0, ln_next=None):
    """Position-wise feed-forward after its LayerNorm (kantts/models/sambert/__init__.py:134-149):
    Conv1d(k) -> ReLU -> zero padded rows -> dropout -> Conv1d(1) -> dropout -> + res (-> zero rows).
    bf16 mode: one autograd node on the bf16-operand kernels; otherwise two fused linears."""
    k1, k2 = w1.shape[2], w2.shape[2]
    if get_precision() == "bf16" and ops_bf16.ffn_eligible(h, w1, w2):
        return ops_bf16.ffn(h, w1, b1, w2, b2, res, pad_rows=pad_rows, zero_rows=zero_rows, p_inner=p_inner, p_out=p_out,
                            ln_next=ln_next)
    hid = linear(h, w1, b1, relu=True, rowmask=pad_rows, drop_p=p_inner, pad=(k1 - 1) // 2,
                 mode="conv" if k1 > 1 else None)
    return linear(hid, w2, b2, res=res, rowmask=zero_rows, drop_p=p_out, pad=(k2 - 1) // 2,
                  mode="conv" if k2 > 1 else None, ln_next=ln_next)


# ================================================================================================
# Attention
# ================================================================================================
MODE_KEYPAD, MODE_BAND_X, MODE_BAND_H = 0, 1, 2


def _attn_fwd(q, qo, k, ko, v, vo, lens, bw_dev, bw, B, H, L, mode, drop_p, seed, want_probs):
    dev = q.device
    o = torch.empty((B * L, H * 16), device=dev, dtype=torch.float32)
    lse = torch.empty((B, H, L), device=dev, dtype=torch.float32)
    probs = torch.empty((H * B, L, L), device=dev, dtype=torch.float32) if want_probs else None
    check(lib().kantts_attn_fwd(ptr(q) + 4 * qo, ptr(k) + 4 * ko, ptr(v) + 4 * vo, q.shape[-1], k.shape[-1],
                                v.shape[-1], ptr(o), H * 16, ptr(lse), ptr(probs), ptr(lens), ptr(bw_dev), int(bw),
                                B, H, L, 16, mode, float(drop_p), int(seed),
                                ptr(rng_state(q.device)) if drop_p > 0 else None, stream()), "attn_fwd")
    return o, lse, probs


def _attn_bwd(q, qo, k, ko, v, vo, o, d_o, lse, dq, dqo, dk, dko, dv, dvo, acc_dq, lens, bw_dev, bw, B, H, L, mode,
              drop_p, seed):
    dvec = torch.empty_like(lse)
    check(lib().kantts_attn_bwd(ptr(q) + 4 * qo, ptr(k) + 4 * ko, ptr(v) + 4 * vo, q.shape[-1], k.shape[-1],
                                v.shape[-1], ptr(o), o.shape[-1], ptr(d_o), d_o.shape[-1], ptr(lse), ptr(dvec),
                                ptr(dq) + 4 * dqo, ptr(dk) + 4 * dko, ptr(dv) + 4 * dvo, dq.shape[-1], dk.shape[-1],
                                dv.shape[-1], int(acc_dq), ptr(lens), ptr(bw_dev), int(bw), B, H, L, 16, mode,
                                float(drop_p), int(seed), ptr(rng_state(q.device)) if drop_p > 0 else None,
                                stream()), "attn_bwd")


class _SelfAttention(torch.autograd.Function):
    """qkv: (B, L, 3*H*16) = [q | k | v] (fused projection output) -> ctx (B, L, H*16) [, probs]."""

    @staticmethod
    def forward(ctx, qkv, lens, H, drop_p, want_probs):
        qkv = _c(qkv)
        B, L, W = qkv.shape
        D = H * 16
        assert W == 3 * D
        q2 = qkv.view(B * L, W)
        ad = ops_bf16.ADOPT.take("attn") if ops_bf16.ADOPT.q else None
        if ad is not None:
            # computed by the fused sub-layer launch (ops_bf16.enc_attn_fused): same values, same dropout stream
            assert not want_probs
            o, lse, seed, probs = ad["o"], ad["lse"], ad["seed"], None
        else:
            seed = next_seed() if drop_p > 0 else 0
            o, lse, probs = _attn_fwd(q2, 0, q2, D, q2, 2 * D, lens, None, 0, B, H, L, MODE_KEYPAD, drop_p, seed,
                                      want_probs)
        ctx.save_for_backward(q2, o, lse, lens)
        ctx.cfg = (B, H, L, drop_p, seed)
        if want_probs:
            ctx.set_materialize_grads(False)  # no zero tensor for the gradient of a non-differentiable output
            ctx.mark_non_differentiable(probs)
            return o.view(B, L, D), probs
        return o.view(B, L, D), None

    @staticmethod
    def backward(ctx, d_o, _dp):
        q2, o, lse, lens = ctx.saved_tensors
        B, H, L, drop_p, seed = ctx.cfg
        D = H * 16
        d_o = _c(d_o).view(B * L, D)
        dqkv = torch.empty_like(q2)
        _attn_bwd(q2, 0, q2, D, q2, 2 * D, o, d_o, lse, dqkv, 0, dqkv, D, dqkv, 2 * D, 0, lens, None, 0, B, H, L,
                  MODE_KEYPAD, drop_p, seed)
        return dqkv.view(B, L, 3 * D), None, None, None, None


_attn_side_stream = []


def _pair_on_two_streams(fn_main, fn_side, side_inputs=()):
    """The x-band and the h-band attention of a PNCA block share only their query: two launches of 256 workgroups, each
    latency-bound (~15 us whatever else runs).  ``fn_side`` goes to a second stream, ``fn_main`` stays on the current
    one, both are joined before returning; under hipGraph capture the pair becomes two parallel branches.  Tensors the
    side function allocates are handed to the current stream (record_stream).  Sequential on the host / when
    KANTTS_NO_ATTN_STREAMS is set."""
    if (not torch.cuda.is_available() or os.environ.get("KANTTS_NO_ATTN_STREAMS")
            or not any(torch.is_tensor(t) and t.is_cuda for t in side_inputs)):
        return fn_main(), fn_side()
    if not _attn_side_stream:
        _attn_side_stream.append(torch.cuda.Stream())
    side, main = _attn_side_stream[0], torch.cuda.current_stream()
    side.wait_stream(main)
    for t in side_inputs:
        if torch.is_tensor(t) and t.is_cuda:
            t.record_stream(side)
    with torch.cuda.stream(side):
        rs = fn_side()
    rm = fn_main()
    main.wait_stream(side)
    for t in (rs if isinstance(rs, (tuple, list)) else (rs,)):
        if torch.is_tensor(t) and t.is_cuda:
            t.record_stream(main)
    return rm, rs


def _tensors_in(obj):
    """Every tensor inside nested tuples / lists / dicts / __slots__ objects (SeqInfo)."""
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, (tuple, list)):
        for o in obj:
            yield from _tensors_in(o)
    elif isinstance(obj, dict):
        for o in obj.values():
            yield from _tensors_in(o)
    elif hasattr(obj, "__slots__"):
        for name in obj.__slots__:
            yield from _tensors_in(getattr(obj, name, None))


BESIDE = {"on": not os.environ.get("KANTTS_NO_PLAN_BESIDE")}  # A/B switch of run_beside


def run_beside(fn_main, fn_side, side_inputs=()):
    """``fn_side`` (work that depends only on the batch, not on ``fn_main``'s result) on a second stream while ``fn_main``
    runs on the current one; joined before returning (two parallel branches under hipGraph capture).  Returns
    (fn_main(), fn_side()); every tensor in the side result (nested containers, SeqInfo) is handed to the current
    stream.  Sequential on the host or with KANTTS_NO_PLAN_BESIDE."""
    if (not BESIDE["on"] or not torch.cuda.is_available()
            or not any(torch.is_tensor(t) and t.is_cuda for t in side_inputs)):
        rs = fn_side()  # same host order as below: dropout seeds are drawn in issue order
        return fn_main(), rs
    if not _attn_side_stream:
        _attn_side_stream.append(torch.cuda.Stream())
    side, main = _attn_side_stream[0], torch.cuda.current_stream()
    side.wait_stream(main)
    for t in side_inputs:
        if torch.is_tensor(t) and t.is_cuda:
            t.record_stream(side)
    with torch.cuda.stream(side):
        rs = fn_side()
    rm = fn_main()
    main.wait_stream(side)
    for t in _tensors_in(rs):
        if t.is_cuda:
            t.record_stream(main)
    return rm, rs


class _PncaAttention(torch.autograd.Function):
    """PNCA dual attention sharing Q: x-band over the decoder's own K/V (from qkv) and h-band over
    the memory K/V (hkv = [k | v]).  Returns ctx_x, ctx_h (B, L, H*16) [, probs_x, probs_h]."""

    @staticmethod
    def forward(ctx, qkv, hkv, lens, bw_dev, bw_x, bw_h, H, drop_p, want_probs):
        qkv = _c(qkv)
        B, L, W = qkv.shape
        D = H * 16
        # the memory projection may be a column block of a wider buffer (all blocks' projections from one GEMM): rows at
        # a uniform pitch are all the fused kernels need
        if not (hkv.dim() == 3 and hkv.stride(2) == 1 and hkv.stride(0) == L * hkv.stride(1) and hkv.stride(1) % 4 == 0):
            hkv = hkv.contiguous()
        ldh = hkv.stride(1)
        q2 = qkv.view(B * L, W)
        h2 = hkv.as_strided((B * L, 2 * D), (ldh, 1), hkv.storage_offset())
        ad = ops_bf16.ADOPT.take("attn") if ops_bf16.ADOPT.q else None
        if ad is not None:
            # both bands were computed by the fused block launch (ops_bf16.pnca_block_fused): adopt contexts, log-sum-exps and
            # the seeds it drew; backward is the usual one
            assert not want_probs
            sx, sh = ad["sx"], ad["sh"]
            ox, oh, lsex, lseh = ad["ox"], ad["oh"], ad["lse_x"], ad["lse_h"]
            ctx.save_for_backward(q2, h2, ox, oh, lsex, lseh, lens, bw_dev)
            ctx.cfg = (B, H, L, bw_x, bw_h, drop_p, sx, sh)
            ctx.plan = ad.get("bwd")
            return ox.view(B, L, D), oh.view(B, L, D), None, None
        ctx.plan = None
        sx = next_seed() if drop_p > 0 else 0
        sh = next_seed() if drop_p > 0 else 0
        ctx.save_cfg = (B, H, L, bw_x, bw_h, drop_p, sx, sh)
        if not want_probs and not os.environ.get("KANTTS_NO_PNCA_FUSED"):
            # both bands as ONE launch (csrc/attn.hip: attn_multi_lds_kernel); declined when a head does not fit in LDS
            dev = q2.device
            ox = torch.empty((B * L, D), device=dev, dtype=torch.float32)
            oh = torch.empty((B * L, D), device=dev, dtype=torch.float32)
            lsex = torch.empty((B, H, L), device=dev, dtype=torch.float32)
            lseh = torch.empty((B, H, L), device=dev, dtype=torch.float32)
            rc = lib().kantts_pnca_attn_fwd(ptr(q2), ptr(h2), ldh, ptr(ox), ptr(oh), ptr(lsex), ptr(lseh), ptr(lens), ptr(bw_dev),
                                            int(bw_x), int(bw_h), B, H, L, 16, float(drop_p), int(sx), int(sh),
                                            ptr(rng_state(dev)) if drop_p > 0 else None, stream())
            if rc == 0:
                ctx.save_for_backward(q2, h2, ox, oh, lsex, lseh, lens, bw_dev)
                ctx.cfg = ctx.save_cfg
                return ox.view(B, L, D), oh.view(B, L, D), None, None
            if rc != E_UNSUPPORTED:
                check(rc, "pnca_attn_fwd")
        h2 = _c(h2)  # the per-band launches take the row pitch from the shape
        (ox, lsex, px), (oh, lseh, ph) = _pair_on_two_streams(
            lambda: _attn_fwd(q2, 0, q2, D, q2, 2 * D, lens, bw_dev, bw_x, B, H, L, MODE_BAND_X, drop_p, sx, want_probs),
            lambda: _attn_fwd(q2, 0, h2, 0, h2, D, lens, bw_dev, bw_h, B, H, L, MODE_BAND_H, drop_p, sh, want_probs),
            side_inputs=(q2, h2, lens, bw_dev))
        ctx.save_for_backward(q2, h2, ox, oh, lsex, lseh, lens, bw_dev)
        ctx.cfg = (B, H, L, bw_x, bw_h, drop_p, sx, sh)
        if want_probs:
            ctx.set_materialize_grads(False)  # no zero tensor for the gradient of a non-differentiable output
            ctx.mark_non_differentiable(px, ph)
        return ox.view(B, L, D), oh.view(B, L, D), px, ph

    @staticmethod
    def backward(ctx, d_ox, d_oh, _a, _b):
        q2, h2, ox, oh, lsex, lseh, lens, bw_dev = ctx.saved_tensors
        B, H, L, bw_x, bw_h, drop_p, sx, sh = ctx.cfg
        D = H * 16
        d_ox, d_oh = _c(d_ox).view(B * L, D), _c(d_oh).view(B * L, D)
        dqkv = torch.empty_like(q2)
        dhkv = torch.empty((B * L, 2 * D), device=q2.device, dtype=torch.float32)
        plan = getattr(ctx, "plan", None)
        tok = plan.tok0 if plan is not None else None
        if (tok is not None and ops_bf16.PNCA_ATTN_BWD["on"] and tok.ready() and plan.wqkvT is not None and H == 8
                and tok.x.numel() == B * L * D and (bw_dev is not None or (bw_x <= 16 and bw_h <= 16))):
            # fused block: this launch is also the input gradient of the QKV projection and the backward of the LayerNorm in
            # front of it (ops_bf16.LnBwdToken: the projection's backward node only hands the stand-in over)
            from kantts._hip import pnca_attn_qkv_bwd, rows_sum_accum

            dx = torch.empty(tok.x.shape, device=q2.device, dtype=torch.float32)
            part = pnca_attn_qkv_bwd(q2, h2, h2.stride(0), ox, oh, d_ox, d_oh, lsex, lseh, B, L, lens=lens, bw_dev=bw_dev,
                                     bw_x=bw_x, bw_h=bw_h, att_p=drop_p, seed_x=sx, seed_h=sh, wqkvT=plan.wqkvT, x=tok.x,
                                     mean0=tok.mean, rstd0=tok.rstd, gamma0=tok.gamma,
                                     dres=None if tok.dres is None else _c(tok.dres).view(B * L, D),
                                     zero_rows=tok.zero_rows, dqkv=dqkv, dhkv=dhkv, dx=dx)
            if part is not None:
                dg, db = gzeros_like(tok.gamma), gzeros_like(tok.gamma)
                rows_sum_accum(part, dg, db)
                tok.dx, tok.dg, tok.db, tok.by_block = dx, dg, db, True
                plan.tok0 = plan.wqkvT = None
                return dqkv.view(B, L, 3 * D), dhkv.view(B, L, 2 * D), None, None, None, None, None, None, None
        if not os.environ.get("KANTTS_NO_PNCA_FUSED"):
            dqh = torch.empty((B * L, D), device=q2.device, dtype=torch.float32)
            rc = lib().kantts_pnca_attn_bwd(ptr(q2), ptr(h2), h2.stride(0), ptr(ox), ptr(oh), ptr(d_ox), ptr(d_oh), ptr(lsex),
                                            ptr(lseh),
                                            ptr(dqkv), ptr(dqh), ptr(dhkv), ptr(lens), ptr(bw_dev), int(bw_x), int(bw_h), B,
                                            H, L, 16, float(drop_p), int(sx), int(sh),
                                            ptr(rng_state(q2.device)) if drop_p > 0 else None, stream())
            if rc in (0, 1):
                if rc == 1:  # long sequences: the two bands' query gradients come back separately
                    dqkv[:, :D].add_(dqh)
                return dqkv.view(B, L, 3 * D), dhkv.view(B, L, 2 * D), None, None, None, None, None, None, None
            if rc != E_UNSUPPORTED:
                check(rc, "pnca_attn_bwd")
        h2 = _c(h2)

        def bwd_h():
            # the h band's query gradient goes to its own buffer: the two bands then share nothing they write
            dqh = torch.empty((B * L, D), device=q2.device, dtype=torch.float32)
            _attn_bwd(q2, 0, h2, 0, h2, D, oh, d_oh, lseh, dqh, 0, dhkv, 0, dhkv, D, 0, lens, bw_dev, bw_h, B, H, L,
                      MODE_BAND_H, drop_p, sh)
            return dqh

        _, dqh = _pair_on_two_streams(
            lambda: _attn_bwd(q2, 0, q2, D, q2, 2 * D, ox, d_ox, lsex, dqkv, 0, dqkv, D, dqkv, 2 * D, 0, lens, bw_dev, bw_x,
                              B, H, L, MODE_BAND_X, drop_p, sx),
            bwd_h, side_inputs=(q2, h2, oh, d_oh, lseh, dhkv, lens, bw_dev))
        dqkv[:, :D].add_(dqh)  # one launch (``+=`` on a slice is add + copy-back)
        return dqkv.view(B, L, 3 * D), dhkv.view(B, L, 2 * D), None, None, None, None, None, None, None


def self_attention(qkv, lens_i32, n_head, drop_p=0.0, want_probs=False):
    return _SelfAttention.apply(qkv, lens_i32, n_head, float(drop_p), bool(want_probs))


def pnca_attention(qkv, hkv, lens_i32, bw_x, bw_h, n_head, drop_p=0.0, want_probs=False, bw_dev=None):
    return _PncaAttention.apply(qkv, hkv, lens_i32, bw_dev, int(bw_x), int(bw_h), n_head, float(drop_p),
                                bool(want_probs))


def attn_decode(q, k, v, out, lens_i32, n_head, step, bw, mode, bw_seq=None):
    """One decoder position against a (B, L, .) K/V buffer (inference only, no autograd).  q / out: (B, .) row
    views (any leading stride), k / v: column slices of contiguous (B, L, W) buffers; bw_seq: optional per-sequence
    band widths (int32, B)."""
    B, L = k.shape[0], k.shape[1]
    check(lib().kantts_attn_decode(ptr(q, torch.float32), ptr(k, torch.float32), ptr(v, torch.float32), q.stride(0),
                                   k.stride(1), v.stride(1), ptr(out, torch.float32), out.stride(0), ptr(lens_i32),
                                   ptr(bw_seq), B, n_head, L, 16, int(mode), int(step), int(bw), stream()), "attn_decode")
    return out


def lstm_cell(gates, c_prev):
    """gates (B, 4H) [i|f|g|o] -> (h, c); inference only."""
    gates = _c(gates)
    B, H = gates.shape[0], gates.shape[1] // 4
    h = torch.empty((B, H), device=gates.device, dtype=torch.float32)
    c = torch.empty_like(h)
    check(lib().kantts_lstm_cell(ptr(gates, torch.float32), ptr(c_prev, torch.float32), ptr(h), ptr(c), B, H, stream()),
          "lstm_cell")
    return h, c


# ================================================================================================
# LSTM
# ================================================================================================
class _LSTM(torch.autograd.Function):
    """One LSTM layer (1 or 2 directions), zero initial state, batch_first.
    xs: inputs concatenated along features (list); per direction (w_ih, w_hh, b_ih, b_hh)."""

    @staticmethod
    def forward(ctx, nx, ndir, lens, *t):
        xs = [_c(x) for x in t[:nx]]
        params = list(t[nx:])  # ndir * 4
        B, T = xs[0].shape[0], xs[0].shape[1]
        H = params[1].shape[1]
        G = 4 * H
        dev = xs[0].device
        M = B * T
        gx = torch.empty((M, ndir * G), device=dev, dtype=torch.float32)
        # bf16 mode: the input projection and its gradients run on the bf16-operand kernels (fp32 activations are
        # rounded while staged, W_ih comes from the bf16 shadow); the recurrence itself is unchanged
        use_b = (get_precision() == "bf16" and M > 0 and all(x.shape[-1] % 8 == 0 and x.dtype == torch.float32 for x in xs)
                 and len(xs) <= 12)
        wbs = [ops_bf16.bf16_weight(params[4 * d]) for d in range(ndir)] if use_b else []
        for d in range(ndir):
            w_ih, b_ih = _c(params[4 * d]), params[4 * d + 2]
            ld = w_ih.shape[1]
            off, segs = 0, []
            for x in xs:
                k = x.shape[-1]
                segs.append((x, k, (wbs[d], off), ld, k, 0) if use_b else make_seg(x, k, 1, (w_ih, off), ld, 1, k))
                off += k
            if use_b:
                if not bgemm_nt(segs, M, G, (gx, d * G), ndir * G, bias=b_ih):
                    raise RuntimeError("bgemm_nt declined the LSTM input projection")
            else:
                gemm(segs, M, G, gx, ndir * G, 1, c_off=d * G, bias=b_ih)
        whh = torch.stack([_c(params[4 * d + 1]) for d in range(ndir)], 0) if ndir > 1 else _c(params[1]).unsqueeze(0)
        bhh = torch.stack([params[4 * d + 3] for d in range(ndir)], 0) if ndir > 1 else params[3].unsqueeze(0)
        whh, bhh = _c(whh), _c(bhh)
        out = torch.empty((B, T, ndir * H), device=dev, dtype=torch.float32)
        gates = torch.empty((ndir, B, T, G), device=dev, dtype=torch.float32)
        cst = torch.empty((ndir, B, T, H), device=dev, dtype=torch.float32)
        prec = 1 if get_precision() == "bf16" else 0
        check(lib().kantts_lstm_fwd(ptr(gx), ptr(whh), ptr(bhh), ptr(lens), ptr(out), ptr(gates), ptr(cst), B, T, H,
                                    ndir, 0, prec, stream()), "lstm_fwd")
        ctx.cfg = (nx, ndir, B, T, H, prec)
        ctx.use_b = use_b
        ctx.save_for_backward(lens, whh, out, gates, cst, *xs, *params, *wbs)
        return out

    @staticmethod
    def backward(ctx, dout):
        nx, ndir, B, T, H, prec = ctx.cfg
        G, M = 4 * H, B * T
        sv = ctx.saved_tensors
        lens, whh, out, gates, cst = sv[:5]
        xs, params = list(sv[5:5 + nx]), list(sv[5 + nx:5 + nx + 4 * ndir])
        wbs = list(sv[5 + nx + 4 * ndir:])
        dout = _c(dout)
        dg = torch.empty((ndir, B, T, G), device=dout.device, dtype=torch.float32)
        check(lib().kantts_lstm_bwd(ptr(dout, torch.float32), ptr(whh), ptr(lens), ptr(gates), ptr(cst), ptr(dg), B, T,
                                    H, ndir, 0, prec, stream()), "lstm_bwd")
        needs = ctx.needs_input_grad  # (nx, ndir, lens, *xs, *params)
        dxs = [None] * nx
        dparams = [None] * (4 * ndir)
        for d in range(ndir):
            dgd = dg[d].view(M, G)
            w_ih = _c(params[4 * d])
            ld = w_ih.shape[1]
            off = 0
            dw_ih = gzeros_like(w_ih)
            db = gzeros((G,), dout.device)
            first = True
            for k, x in enumerate(xs):
                kk = x.shape[-1]
                if ctx.use_b:
                    if needs[3 + k]:
                        first_dir = dxs[k] is None
                        if first_dir:
                            dxs[k] = torch.empty_like(x)
                        # the second direction adds onto the first one's result (read as the fp32 residual)
                        if not bgemm_nt([(dgd, G, (wbs[d], off), ld, G, 0)], M, kk, dxs[k], kk, b_kn=True,
                                        res=None if first_dir else dxs[k], ldr=kk):
                            raise RuntimeError("bgemm_nt declined the LSTM input gradient")
                    with wgrad_overlap.side(dgd, x):
                        if not bgemm_tn(dgd, G, x, kk, M, G, kk, (dw_ih, off), ld, 1, db=db if first else None):
                            raise RuntimeError("bgemm_tn declined the LSTM input-weight gradient")
                else:
                    if needs[3 + k]:
                        if dxs[k] is None:
                            dxs[k] = torch.zeros_like(x) if ndir > 1 else torch.empty_like(x)
                        gemm([make_seg(dgd, G, 1, (w_ih, off), 1, ld, G)], M, kk, dxs[k], kk, 1, accumulate=(ndir > 1))
                    gemm([make_seg(dgd, 1, G, x, 1, kk, M)], G, kk, dw_ih, ld, 1, c_off=off, accumulate=True,
                         splitk=_splitk_for(G, kk, M), a_rowsum=db if first else None)
                first = False
                off += kk
            dw_hh = gzeros((G, H), dout.device)
            shift = 1 if d == 1 else -1  # h_{prev}: previous step in this direction's time order
            if ctx.use_b:
                with wgrad_overlap.side(dgd, out):
                    if not bgemm_tn(dgd, G, (out, d * H), ndir * H, M, G, H, dw_hh, H, 1, T=T, shift0=shift):
                        raise RuntimeError("bgemm_tn declined the LSTM recurrent-weight gradient")
            else:
                seg = make_seg(dgd, 1, G, (out, d * H), 1, ndir * H, M, b_tok_axis=2, b_shift0=shift)
                gemm([seg], G, H, dw_hh, H, 1, accumulate=True, splitk=_splitk_for(G, H, M), T=T)
            # b_ih and b_hh share one gradient (see deferred_tn.shared_gradient)
            from . import deferred_tn

            dparams[4 * d], dparams[4 * d + 1] = dw_ih, dw_hh
            dparams[4 * d + 2], dparams[4 * d + 3] = db, (deferred_tn.shared_gradient(db) if ctx.use_b else db)
        return (None, None, None, *dxs, *dparams, *([None] * len(wbs)))


def lstm(xs, params, lens_i32=None):
    """params: [w_ih, w_hh, b_ih, b_hh] (+ the 4 reverse-direction tensors for a BiLSTM)."""
    xs = [xs] if torch.is_tensor(xs) else list(xs)
    ndir = len(params) // 4
    return _LSTM.apply(len(xs), ndir, lens_i32, *xs, *params)


# ================================================================================================
# Embedding gather-sum
# ================================================================================================
def _ptr_array(tensors):
    import ctypes

    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = ptr(t, torch.float32) if t is not None else None
    return arr


class _EmbedSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, pos, scale, want_scaled, *tables):
        ids = _c(ids)
        if ids.dim() == 2:
            ids = ids.unsqueeze(-1)
        B, T, n = ids.shape
        assert n == len(tables) and ids.dtype == torch.int64
        D = tables[0].shape[1]
        out = torch.empty((B, T, D), device=ids.device, dtype=torch.float32)
        scaled = torch.empty_like(out) if want_scaled else None
        tabs = [_c(t) for t in tables]
        check(lib().kantts_embed_sum_fwd(_ptr_array(tabs), n, ptr(ids), ptr(pos), ptr(out), ptr(scaled), B * T, T, D,
                                         float(scale), stream()), "embed_sum_fwd")
        ctx.save_for_backward(ids)
        ctx.cfg = (scale, [tuple(t.shape) for t in tables])
        if want_scaled and want_scaled != "grad":
            ctx.mark_non_differentiable(scaled)
        return out, scaled

    @staticmethod
    def backward(ctx, dout, d_scaled):
        (ids,) = ctx.saved_tensors
        scale, shapes = ctx.cfg
        if d_scaled is not None:  # out = scaled + pos: both outputs feed the tables with the same factor
            dout = d_scaled if dout is None else dout + d_scaled
        dout = _c(dout)
        B, T, n = ids.shape
        D = shapes[0][1]
        dt = [gzeros(s, dout.device) for s in shapes]
        check(lib().kantts_embed_sum_bwd(_ptr_array(dt), n, ptr(ids), ptr(dout, torch.float32), B * T, D, float(scale),
                                         stream()), "embed_sum_bwd")
        return (None, None, None, None, *dt)


def embed_sum(ids, tables, pos=None, scale=1.0, want_scaled=False):
    """want_scaled: False | True (second output without gradient) | "grad" (second output differentiable: the MAS path
    feeds the scaled embedding to the alignment attention)."""
    return _EmbedSum.apply(ids, pos, float(scale), want_scaled if want_scaled == "grad" else bool(want_scaled), *tables)


# ================================================================================================
# Length regulator
# ================================================================================================
def lr_index(durations, Tp):
    """durations (B,N) int64 or fp32 -> idx (B,Tp) i32, pos (B,Tp) f32, cs (B,N+1) i32, lens (B) i64."""
    d = _c(durations)
    B, N = d.shape
    dev = d.device
    idx = torch.empty((B, Tp), device=dev, dtype=torch.int32)
    pos = torch.empty((B, Tp), device=dev, dtype=torch.float32)
    cs = torch.empty((B, N + 1), device=dev, dtype=torch.int32)
    lens = torch.empty((B,), device=dev, dtype=torch.int64)
    if d.dtype == torch.int64:
        di, df = ptr(d), None
    else:
        di, df = None, ptr(d, torch.float32)
    check(lib().kantts_lr_index(di, df, ptr(idx), ptr(pos), ptr(cs), ptr(lens), B, N, Tp, stream()), "lr_index")
    return idx, pos, cs, lens


class _LRGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx, cs, valid):
        x = _c(x)
        B, N, C = x.shape
        Tp = idx.shape[1]
        out = torch.empty((B, Tp, C), device=x.device, dtype=torch.float32)
        check(lib().kantts_lr_gather_fwd(ptr(x, torch.float32), ptr(idx), ptr(valid), ptr(out), B, N, Tp, C, C, 0,
                                         stream()), "lr_gather_fwd")
        ctx.save_for_backward(cs, valid)
        ctx.cfg = (B, N, Tp, C)
        return out

    @staticmethod
    def backward(ctx, dout):
        cs, valid = ctx.saved_tensors
        B, N, Tp, C = ctx.cfg
        dout = _c(dout)
        dx = torch.empty((B, N, C), device=dout.device, dtype=torch.float32)
        check(lib().kantts_lr_gather_bwd(ptr(dout, torch.float32), ptr(cs), ptr(valid), ptr(dx), B, N, Tp, C, C, 0, 0,
                                         stream()), "lr_gather_bwd")
        return dx, None, None, None


def lr_gather(x, idx, cs, valid_lens):
    return _LRGather.apply(x, idx, cs, valid_lens)


# ================================================================================================
# FSMN memory block
# ================================================================================================
class _FsmnMemory(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, res, lens, lp):
        x, w = _c(x), _c(w)
        B, T, C = x.shape
        K = w.shape[-1]
        y = torch.empty_like(x)
        r = _c(res) if res is not None else None
        check(lib().kantts_fsmn_dwconv_fwd(ptr(x, torch.float32), ptr(w, torch.float32), ptr(r), ptr(lens), ptr(y), B,
                                           T, C, K, int(lp), stream()), "fsmn_dwconv_fwd")
        ctx.save_for_backward(x, w, lens)
        ctx.cfg = (B, T, C, K, int(lp), res is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, lens = ctx.saved_tensors
        B, T, C, K, lp, has_res = ctx.cfg
        dy = _c(dy)
        dx = torch.empty_like(x)
        dw = gzeros_like(w)
        ws_n = int(lib().kantts_fsmn_dwconv_bwd_ws(B, T, C, K))
        ws = torch.empty(max(ws_n, 1), device=dy.device, dtype=torch.float32)
        check(lib().kantts_fsmn_dwconv_bwd(ptr(dy, torch.float32), ptr(x), ptr(w), ptr(lens), ptr(dx), None, None, 0,
                                           B, T, C, K, lp, stream()), "fsmn_dwconv_bwd (dx)")
        # the filter gradient is a leaf: on the weight-gradient stream when the step overlaps them (partials + reduce are
        # 41 us per layer, ten layers per step)
        # (dw itself must NOT be in the keep-alive list: a second reference to the returned gradient makes AccumulateGrad
        # CLONE it on the main stream instead of adopting it -- before the side stream has written it.  The parameter's
        # .grad keeps it alive.)
        with wgrad_overlap.side(dy, x, ws):
            check(lib().kantts_fsmn_dwconv_bwd(ptr(dy, torch.float32), ptr(x), ptr(w), ptr(lens), None, ptr(dw), ptr(ws),
                                               ws_n, B, T, C, K, lp, stream()), "fsmn_dwconv_bwd (dw)")
        return dx, dw, (dy if has_res else None), None, None


def fsmn_memory(x, w, lens_i64, left_pad, res=None):
    return _FsmnMemory.apply(x, w, res, lens_i64, left_pad)


# ================================================================================================
# Masked L1
# ================================================================================================
class _MaskedL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, lens):
        pred, target = _c(pred), _c(target)
        if pred.dim() == 2:
            B, T = pred.shape
            C = 1
        else:
            B, T, C = pred.shape
        loss = torch.zeros((), device=pred.device, dtype=torch.float32)
        need_grad = ctx.needs_input_grad[0]
        grad = torch.empty_like(pred) if need_grad else None
        check(lib().kantts_masked_l1(ptr(pred, torch.float32), ptr(target, torch.float32), ptr(lens, torch.int64),
                                     ptr(loss), ptr(grad), B, T, C, stream()), "masked_l1")
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None


def masked_l1(pred, target, lens_i64):
    return _MaskedL1.apply(pred, target, lens_i64)


class _MaskedL1Many(torch.autograd.Function):
    """Up to five masked-L1 terms and their sum from ONE launch (kantts_masked_l1_many), the gradients of all terms written
    in the same pass; backward is one launch that scales them by the upstream gradient of the sum (a device scalar).
    Returns (total, components (n,)); the components are detached values for logging."""

    @staticmethod
    def forward(ctx, spec, *preds):
        from . import LOSS_MAX_TERMS, LossTerm

        n = len(preds)
        assert 1 <= n <= LOSS_MAX_TERMS and len(spec) == n
        terms = (LossTerm * n)()
        grads, keep = [], []
        for k, (p, (target, lens, log1p)) in enumerate(zip(preds, spec)):
            p, target = _c(p), _c(target)
            B, T = p.shape[0], p.shape[1]
            C = 1 if p.dim() == 2 else p.shape[2]
            assert tuple(target.shape) == tuple(p.shape) and p.dtype == torch.float32
            g = torch.empty_like(p) if ctx.needs_input_grad[1 + k] else None
            q = terms[k]
            q.pred, q.lens, q.grad = ptr(p, torch.float32), ptr(lens, torch.int64), ptr(g)
            q.target = ptr(target, torch.int64) if log1p else ptr(target, torch.float32)
            q.B, q.T, q.C, q.target_log1p = int(B), int(T), int(C), int(bool(log1p))
            grads.append(g)
            keep.extend((p, target, lens))
        # not from the per-step zero pool: the loss outlives optimizer.zero_grad() (the reference's trainer, and ours, clear
        # the gradients between the forward pass and backward()), which re-zeroes the pool
        losses = torch.zeros((LOSS_MAX_TERMS + 1,), device=preds[0].device, dtype=torch.float32)
        check(lib().kantts_masked_l1_many(terms, n, ptr(losses, torch.float32), stream()), "masked_l1_many")
        ctx.grads = grads
        total = losses[LOSS_MAX_TERMS]
        comps = losses[:n]
        ctx.mark_non_differentiable(comps)
        return total, comps

    @staticmethod
    def backward(ctx, g, _gc):
        import ctypes

        live = [t for t in ctx.grads if t is not None and t.numel() > 0]
        if live:
            g = _c(g).reshape(1)
            xs = (ctypes.c_void_p * len(live))(*[ptr(t, torch.float32) for t in live])
            ns = (ctypes.c_longlong * len(live))(*[t.numel() for t in live])
            check(lib().kantts_scale_many(xs, ns, len(live), ptr(g, torch.float32), stream()), "scale_many")
        return (None, *ctx.grads)


def masked_l1_many(terms):
    """terms: list of (pred, target, lens_i64, target_is_int64_log1p).  -> (sum of the terms, (n,) detached components)."""
    spec = tuple((t, lens, bool(log1p)) for _, t, lens, log1p in terms)
    return _MaskedL1Many.apply(spec, *[p for p, _, _, _ in terms])


class _ElemLoss(torch.autograd.Function):
    """scale * sum |a - b|  (mode 0)  or  scale * sum (a - target)^2  (mode 1); b is treated as a constant."""

    @staticmethod
    def forward(ctx, a, b, target, mode, scale):
        a = _c(a)
        b = _c(b) if b is not None else None
        loss = torch.zeros((), device=a.device, dtype=torch.float32)
        grad = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        check(lib().kantts_elem_loss(ptr(a, torch.float32), ptr(b), float(target), int(mode), float(scale), ptr(loss),
                                     ptr(grad), a.numel(), stream()), "elem_loss")
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g if grad is not None else None), None, None, None, None


class _ElemLossMany(torch.autograd.Function):
    """Many _ElemLoss terms from ONE launch per <= 64 terms (kantts_elem_loss_many): spec[k] = (b or None, target, mode,
    scale, out); returns the (n_out,) sums.  Backward: the stored gradients times the upstream gradient of their sum, one
    launch per output."""

    @staticmethod
    def forward(ctx, n_out, spec, *a_list):
        from . import ELOSS_MAX_TERMS, ElossTerm

        dev = a_list[0].device
        losses = torch.zeros((n_out,), device=dev, dtype=torch.float32)
        grads, keep = [], []
        for s0 in range(0, len(a_list), ELOSS_MAX_TERMS):
            chunk = list(zip(a_list[s0:s0 + ELOSS_MAX_TERMS], spec[s0:s0 + ELOSS_MAX_TERMS]))
            terms = (ElossTerm * len(chunk))()
            for k, (a, (b, target, mode, scale, out)) in enumerate(chunk):
                a = _c(a)
                g = torch.empty_like(a) if ctx.needs_input_grad[2 + s0 + k] else None
                q = terms[k]
                q.a, q.grad, q.n = ptr(a, torch.float32), ptr(g), a.numel()
                if mode == 0:
                    b = _c(b)
                    assert b.numel() == a.numel()
                    q.b = ptr(b, torch.float32)
                    keep.append(b)
                q.target, q.scale, q.mode, q.out = float(target), float(scale), int(mode), int(out)
                grads.append(g)
                keep.append(a)
            check(lib().kantts_elem_loss_many(terms, len(chunk), ptr(losses, torch.float32), stream()), "elem_loss_many")
        ctx.grads = grads
        ctx.outs = [int(sp[4]) for sp in spec]
        ctx.shapes = [tuple(a.shape) for a in a_list]
        return losses

    @staticmethod
    def backward(ctx, g_losses):
        import ctypes

        from . import ELOSS_MAX_TERMS

        g_losses = _c(g_losses)
        for out in sorted(set(ctx.outs)):
            live = [t for t, o in zip(ctx.grads, ctx.outs) if t is not None and o == out and t.numel() > 0]
            for s0 in range(0, len(live), ELOSS_MAX_TERMS):
                part = live[s0:s0 + ELOSS_MAX_TERMS]
                xs = (ctypes.c_void_p * len(part))(*[ptr(t, torch.float32) for t in part])
                ns = (ctypes.c_longlong * len(part))(*[t.numel() for t in part])
                check(lib().kantts_scale_many(xs, ns, len(part), ptr(g_losses[out:out + 1], torch.float32), stream()),
                      "scale_many")
        return (None, None, *[None if t is None else t.view(sh) for t, sh in zip(ctx.grads, ctx.shapes)])


def elem_loss_many(terms, n_out=1):
    """terms: list of (a, b_or_None, target, mode, scale, out) -- mode 0: scale * sum |a - b| (b constant), mode 1:
    scale * sum (a - target)^2 -- summed into out < n_out.  Returns the (n_out,) tensor of sums."""
    a_list, spec = [], []
    for a, b, target, mode, scale, out in terms:
        if mode == 0:
            a, b = _dense_order(a, b.detach())
            b = b.reshape(a.shape)
        a_list.append(a)
        spec.append((b, target, mode, scale, out))
    return _ElemLossMany.apply(int(n_out), tuple(spec), *a_list)


def _dense_order(a, b):
    """A mean over all elements does not care about their order: when a and b are the same permuted view of dense
    buffers (the discriminators hand their channels-last feature maps out as (B, C, T[, p]) views), undo the permutation
    on both instead of materialising two contiguous copies per feature map (~200 copy launches per GAN step)."""
    if a.shape == b.shape and a.stride() == b.stride() and not a.is_contiguous():
        order = sorted(range(a.dim()), key=lambda d: (-a.stride(d), d))
        ap = a.permute(order)
        if ap.is_contiguous():
            return ap, b.permute(order)
    return a, b


def l1_mean(a, b):
    """F.l1_loss(a, b.detach()) in one pass (loss + gradient)."""
    a, b = _dense_order(a, b.detach())
    return _ElemLoss.apply(a, b.reshape(a.shape), 0.0, 0, 1.0 / a.numel())


def mse_to_const(a, target):
    """F.mse_loss(a, full_like(a, target))."""
    return _ElemLoss.apply(a, None, float(target), 1, 1.0 / a.numel())


# ================================================================================================
# Optimiser kernels on flat arenas
# ================================================================================================
def sumsq_into(x_flat, out_scalar, workspace=None):
    """out += sum x^2 (atomics), or with a zero-initialised ``workspace`` (>= 1025 floats) out = sum x^2 in a fixed
    summation order (bit-reproducible across replicas and runs)."""
    if workspace is not None:
        check(lib().kantts_sumsq_det(ptr(x_flat, torch.float32), ptr(out_scalar, torch.float32),
                                     ptr(workspace, torch.float32), workspace.numel(), x_flat.numel(), stream()),
              "sumsq_det")
        return
    check(lib().kantts_sumsq(ptr(x_flat, torch.float32), ptr(out_scalar, torch.float32), x_flat.numel(), stream()),
          "sumsq")


def advance_rng(device):
    """Advance the device-resident dropout offset (call once per training step)."""
    rng_state(device).add_(0x632BE59BD9B4E019)


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, gnorm_sq=None, max_norm=0.0, dyn=None):
    """dyn: optional device tensor [lr, step] (fp32) overriding the host lr / step (graph replay)."""
    bc1 = 1.0 - beta1 ** max(step, 1)
    bc2 = 1.0 - beta2 ** max(step, 1)
    check(lib().kantts_adam_step(ptr(p, torch.float32), ptr(g, torch.float32), ptr(m, torch.float32),
                                 ptr(v, torch.float32), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
                                 float(weight_decay), float(bc1), float(bc2), ptr(gnorm_sq),
                                 float(max_norm if max_norm else 0.0), ptr(dyn), stream()), "adam_step")


# ================================================================================================
# HiFi-GAN: channels-last convolutions on the segmented GEMM
# ================================================================================================
_tap_matrix_cache = {}

# Packed re-layouts of a weight (block-diagonal merges of small groups) are valid until ANY parameter arena's master weights
# change (ParamArena.shadow_stale bumps the epoch): the discriminators run twice per phase of a GAN step on the same weights
# (generated and real audio), and every re-layout is a zero-fill + copies + a cast of its own.
weights_epoch = [0]
_pack_cache = {}


def _cached_pack(w, tag, make, stable):
    """``stable``: ``w`` is a view of a network's weight-norm image buffer (ops.weight_norm_image) -- the same address for
    the same layer all step long.  Anything else (a per-layer re-parametrisation: a fresh tensor per forward pass whose
    address the allocator may hand to ANOTHER layer's weight next) is never cached."""
    if not stable or os.environ.get("KANTTS_NO_PACK_CACHE"):
        return make()
    key = (w.data_ptr(), tuple(w.shape), tuple(w.stride()), tag, str(w.device))
    hit = _pack_cache.get(key)
    if hit is not None and hit[0] == weights_epoch[0] and hit[2] == w._version:
        return hit[1]
    if len(_pack_cache) > 512:
        for k in [k for k, v in _pack_cache.items() if v[0] != weights_epoch[0]]:
            del _pack_cache[k]
    out = make()
    _pack_cache[key] = (weights_epoch[0], out, w._version)
    return out


def _upsample_tap_matrix(K, up, pad, dil, device):
    """S[j - jmin, k] = #{r in [0, up): r + pad - k*dil == j}: combines the K taps of a convolution over a
    nearest-upsampled input into the (K-1)*dil + up taps of the equivalent strided convolution over its output
    gradient (see _ConvCL.backward)."""
    key = (K, up, pad, dil, str(device))
    hit = _tap_matrix_cache.get(key)
    if hit is None:
        jmin, jmax = pad - (K - 1) * dil, up - 1 + pad
        S = torch.zeros(jmax - jmin + 1, K)
        for r in range(up):
            for k in range(K):
                S[r + pad - k * dil - jmin, k] += 1.0
        hit = (S.to(device), jmin)
        _tap_matrix_cache[key] = hit
    return hit


_EYE_CACHE = {}


def _group_pack(groups, cr, ng):
    """Number P of ADJACENT groups that are merged into one dense block-diagonal contraction.  A group with 8 input and
    16 output channels (the scale discriminator's k=41 layers, hifigan.py:332-407) fills 1/4 x 1/4 of a 32-deep x
    64-wide MFMA tile; P groups side by side fill it, at P times the arithmetic -- 10-27 TFLOP/s effective became
    the per-shape table's worst entries (profiles/r01_hifigan_conv_shapes.log).  Channels-last keeps the P*cr input
    channels of adjacent groups contiguous, so only the (tiny) weight tensor is re-laid."""
    if groups <= 1 or (cr >= 32 and ng >= 32):
        return 1
    P = 1
    while P < groups and (P * cr < 32 or P * ng < 32) and groups % (2 * P) == 0:
        P *= 2
    return P


def _eye(P, device):
    key = (P, str(device))
    if key not in _EYE_CACHE:
        _EYE_CACHE[key] = torch.eye(P, device=device, dtype=torch.float32)
    return _EYE_CACHE[key]


def _blockdiag_pack(w, groups, P):
    """(K, groups*NG, CR) -> (K, groups*NG, P*CR): output channel n of group g keeps its CR weights at channel block
    g % P of its merged group, zeros elsewhere."""
    K, N, CR = w.shape
    NG = N // groups
    v = w.view(K, groups // P, P, NG, 1, CR) * _eye(P, w.device).view(1, 1, P, 1, P, 1)
    return v.reshape(K, N, P * CR)


def _blockdiag_unpack(wp, groups, P):
    """Inverse selection for gradients: (K, groups*NG, P*CR) -> (K, groups*NG, CR) (the diagonal blocks)."""
    K, N, PCR = wp.shape
    NG, CR = N // groups, PCR // P
    d = torch.diagonal(wp.view(K, groups // P, P, NG, P, CR), dim1=2, dim2=4)  # (K, G', NG, CR, P)
    return d.permute(0, 1, 4, 2, 3).reshape(K, N, CR)


class _ConvCL(torch.autograd.Function):
    """Channels-last Conv1d / (k,1)-Conv2d:  x (B, Tin, inner, Cin) -> y (B, Tout, inner, Cout)

        y[b,q,p,co] = act_out( bias[co] + sum_k sum_ci act_in(x[b, (q*stride + k*dil - pad) // up, p, ci]) w[co,ci,k] ) (+ res)

    ``up`` > 1 reads a nearest-neighbour-upsampled view of x (the x``up`` tensor of the reference's
    repeat_upsamples is never materialised); ``inner`` folds the period axis of the MPD; ``groups``
    batches grouped convolutions over gridDim.z.  LeakyReLU on the way in / out is fused in the
    loader / epilogue.  Weights are re-laid tap-major ((K, Cout, Cin_g) for forward / weight gradient,
    (K, groups, Cin_g, Cout_g) for the input gradient) so that every operand has a unit-stride reduction
    axis and the launches qualify for the vector-staged kernels of csrc/gemm_fast.hip.
    Reference: kantts/models/hifigan/layers.py:15-91, hifigan.py:82-97,217-267,332-407.
    """

    @staticmethod
    def forward(ctx, x, w, bias, res, cfg):
        x, w = _c(x), _c(w)
        stride, dil, pad, up, groups = cfg["stride"], cfg["dilation"], cfg["pad"], cfg["up"], cfg["groups"]
        inner, Tout = cfg["inner"], cfg["Tout"]
        B, Tin = x.shape[0], x.shape[1]
        Cin = x.shape[-1]
        tap_major = cfg.get("tap_major", False)
        if tap_major:
            K, Cout, Cin_g = w.shape
        else:
            Cout, Cin_g, K = w.shape
        Cout_g = Cout // groups
        assert Cin_g * groups == Cin
        M = B * Tout * inner
        y = torch.empty((B, Tout, inner, Cout) if x.dim() == 4 else (B, Tout, Cout), device=x.device, dtype=torch.float32)
        r = _c(res) if res is not None else None
        ctx.cfg = cfg
        ctx.has = (bias is not None, res is not None)
        ctx.save_for_backward(x, w, y if cfg["out_leaky"] is not None else None)
        ctx.res_for_gate = r if cfg["out_leaky"] is not None else None  # y = act(conv) + res: the gate is sign(y - res)
        # one input channel (first discriminator layers): streaming kernels, weights stay (Cout, K)
        ctx.c1 = (Cin == 1 and groups == 1 and up == 1 and res is None and cfg["in_leaky"] is None and not tap_major)
        # bf16 mode: the 1-channel first layer also writes the bf16 image its consumer reads (cfg["image"] = None: no further
        # activation), handed to conv_cl() through the cfg dict -- one more output would change the node's signature
        y16 = None
        if ctx.c1 and cfg.get("image", False) is None and get_precision() == "bf16" and y.numel() % 8 == 0:
            y16 = torch.empty(y.shape, device=x.device, dtype=torch.bfloat16)
        if ctx.c1 and conv_c1(0, x=x, y=y, w=w, bias=bias, B=B, Tsrc=Tin, Tdst=Tout, Cout=Cout, K=K, stride=stride,
                              dil=dil, pad=pad, inner=inner, out_leaky=cfg["out_leaky"], y_bf16=y16):
            cfg["c1_image"] = y16
            return y
        ctx.c1 = False
        # one OUTPUT channel (conv_post of the generator and of every sub-discriminator): a dot product per position on
        # csrc/conv_n1.hip -- as a windowed GEMM such a layer is a single output tile, i.e. one workgroup
        ctx.n1 = (Cout == 1 and groups == 1 and up == 1 and res is None and cfg["out_leaky"] is None and Cin % 4 == 0)
        if ctx.n1:
            ks, cs = (Cin, 1) if tap_major else (1, K)
            ctx.n1 = conv_n1(0, x=x, y=y, w=w, w_ks=ks, w_cs=cs, bias=bias, B=B, Tsrc=Tin, Tdst=Tout, Cin=Cin, K=K,
                             stride=stride, dil=dil, pad=pad, inner=inner, in_leaky=cfg["in_leaky"])
            if ctx.n1:
                return y
        wt = w if tap_major else (w.permute(2, 0, 1).contiguous() if K > 1 else w)  # (K, Cout, Cin_g)
        P = _group_pack(groups, Cin_g, Cout_g) if up == 1 else 1
        if P > 1 and conv_win(
                x, _blockdiag_pack(wt.reshape(K, Cout, Cin_g), groups, P), y, B=B, Tsrc=Tin, Tdst=Tout, groups=groups // P,
                CR=P * Cin_g, NG=P * Cout_g, K=K, in_mul=stride, in_add=-pad, in_kstep=dil, in_div=1, phases=1,
                inner=inner, up=up, bias=bias, res=r, in_leaky=cfg["in_leaky"], out_leaky=cfg["out_leaky"]):
            return y
        if conv_win(
                x, wt, y, B=B, Tsrc=Tin, Tdst=Tout, groups=groups, CR=Cin_g, NG=Cout_g, K=K, in_mul=stride,
                in_add=-pad, in_kstep=dil, in_div=1, phases=1, inner=inner, up=up, bias=bias, res=r, in_leaky=cfg["in_leaky"],
                out_leaky=cfg["out_leaky"]):
            return y
        seg = make_seg(x, Cin, 1, wt, Cin_g, 1, Cin_g, ntaps=K, b_tap=Cout * Cin_g, a_tok_axis=1, a_shift0=-pad,
                       a_shift_step=dil, a_map=dict(inner=inner, Tq=Tout, Tsrc=Tin, mul=stride, up=up),
                       a_leaky=cfg["in_leaky"])
        gemm([seg], M, Cout_g, y, Cout, 1, bias=bias, res=r, r_is=Cout, r_js=1, groups=groups, a_gs=Cin_g,
             b_gs=Cout_g * Cin_g, c_gs=Cout_g, bias_gs=Cout_g, r_gs=Cout_g, out_leaky=cfg["out_leaky"])
        return y

    @staticmethod
    def backward(ctx, dy):
        cfg = ctx.cfg
        x, w, y = ctx.saved_tensors
        has_bias, has_res = ctx.has
        stride, dil, pad, up, groups = cfg["stride"], cfg["dilation"], cfg["pad"], cfg["up"], cfg["groups"]
        inner, Tout = cfg["inner"], cfg["Tout"]
        dy = _c(dy)
        B, Tin, Cin = x.shape[0], x.shape[1], x.shape[-1]
        tap_major = cfg.get("tap_major", False)
        if tap_major:
            K, Cout, Cin_g = w.shape
        else:
            Cout, Cin_g, K = w.shape
        Cout_g = Cout // groups
        gate, gslope = (y, cfg["out_leaky"]) if cfg["out_leaky"] is not None else (None, 0.0)
        if gate is not None and ctx.res_for_gate is not None:
            gate = y - ctx.res_for_gate  # (no shipped model combines an output activation with a residual)
        dx = dw = db = None
        if ctx.c1:
            kw = dict(B=B, Tsrc=Tin, Tdst=Tout, Cout=Cout, K=K, stride=stride, dil=dil, pad=pad, inner=inner, gate=gate,
                      gate_slope=gslope)
            if ctx.needs_input_grad[0]:
                dx = torch.zeros_like(x)
                if not conv_c1(1, dx=dx, y=dy, w=w, **kw):
                    raise RuntimeError("conv_c1 dgrad refused a shape its forward accepted")
            if ctx.needs_input_grad[1]:
                dw = gzeros(tuple(w.shape), dy.device)
                db = gzeros((Cout,), dy.device) if has_bias else None
                if not conv_c1(2, x=x, y=dy, w=w, dw=dw, db=db, **kw):
                    raise RuntimeError("conv_c1 wgrad refused a shape its forward accepted")
            return dx, dw, db, None, None
        if getattr(ctx, "n1", False):
            ks, cs = (Cin, 1) if tap_major else (1, K)
            kw = dict(w=w, w_ks=ks, w_cs=cs, B=B, Tsrc=Tin, Tdst=Tout, Cin=Cin, K=K, stride=stride, dil=dil, pad=pad,
                      inner=inner, in_leaky=cfg["in_leaky"])
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                if not conv_n1(1, x=x, y=dy, dx=dx, **kw):
                    raise RuntimeError("conv_n1 dgrad refused a shape its forward accepted")
            if ctx.needs_input_grad[1]:
                dw = gzeros(tuple(w.shape), dy.device)
                db = gzeros((1,), dy.device) if has_bias else None
                if not conv_n1(2, x=x, y=dy, dw=dw, db=db, **kw):
                    raise RuntimeError("conv_n1 wgrad refused a shape its forward accepted")
            return dx, dw, db, None, None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            Mx = B * Tin * inner
            if tap_major:
                wd = w.view(K, groups, Cout_g, Cin_g).transpose(2, 3).contiguous()
            else:
                wd = w.view(groups, Cout_g, Cin_g, K).permute(3, 0, 2, 1).contiguous()  # (K, groups, Cin_g, Cout_g)
            if up > 1 and stride == 1:
                # x-token t feeds the `up` virtual tokens t*up + r:  dx[t] = sum_j dy[t*up + j] W'[j] with
                # W'[j] = sum_{(r,k): r + pad - k*dil = j} w[k] -- ONE strided window pass over dy instead of `up`
                # accumulating passes (the tap-combination matrix is a tiny cached constant)
                S, jmin = _upsample_tap_matrix(K, up, pad, dil, dy.device)
                wj = torch.matmul(S, wd.reshape(K, -1)).view(S.shape[0], groups, Cin_g, Cout_g)
                done = conv_win(dy, wj, dx, B=B, Tsrc=Tout, Tdst=Tin, groups=groups, CR=Cout_g, NG=Cin_g, K=S.shape[0],
                                in_mul=up, in_add=jmin, in_kstep=1, in_div=1, phases=1, inner=inner, in_gate=gate,
                                in_gate_slope=gslope, out_gate=x if cfg["in_leaky"] is not None else None,
                                out_gate_slope=cfg["in_leaky"] or 0.0)
            else:
                Pd = _group_pack(groups, Cout_g, Cin_g) if up == 1 else 1
                done = Pd > 1 and conv_win(
                    dy, _blockdiag_pack(wd.reshape(K, Cin, Cout_g), groups, Pd), dx, B=B, Tsrc=Tout, Tdst=Tin,
                    groups=groups // Pd, CR=Pd * Cout_g, NG=Pd * Cin_g, K=K, in_mul=1, in_add=pad, in_kstep=-dil,
                    in_div=stride, phases=stride, inner=inner, in_gate=gate, in_gate_slope=gslope,
                    out_gate=x if cfg["in_leaky"] is not None else None, out_gate_slope=cfg["in_leaky"] or 0.0)
                done = done or (up == 1 and conv_win(
                    dy, wd, dx, B=B, Tsrc=Tout, Tdst=Tin, groups=groups, CR=Cout_g, NG=Cin_g, K=K, in_mul=1, in_add=pad,
                    in_kstep=-dil, in_div=stride, phases=stride, inner=inner, in_gate=gate, in_gate_slope=gslope,
                    out_gate=x if cfg["in_leaky"] is not None else None, out_gate_slope=cfg["in_leaky"] or 0.0))
            first = True
            for r in range(0 if done else up):
                # x-domain token t receives dy[(t*up + r + pad - k*dil) / stride] (exact division only)
                seg = make_seg(dy, Cout, 1, wd, Cout_g, 1, Cout_g, ntaps=K, b_tap=Cin_g * Cout, a_tok_axis=1,
                               a_shift0=r + pad, a_shift_step=-dil,
                               a_map=dict(inner=inner, Tq=Tin, Tsrc=Tout, mul=up, div=stride), a_gate=gate,
                               a_gate_slope=gslope)
                gemm([seg], Mx, Cin_g, dx, Cin, 1, groups=groups, a_gs=Cout_g, b_gs=Cin_g * Cout_g, c_gs=Cin_g,
                     accumulate=not first, gate=x if cfg["in_leaky"] is not None else None,
                     gate_slope=cfg["in_leaky"] or 0.0)
                first = False
        if ctx.needs_input_grad[1]:
            if has_bias:
                db = gzeros((Cout,), dy.device)
            Mtok = B * Tout * inner
            Pw = _group_pack(groups, Cin_g, Cout_g) if up == 1 else 1
            if Pw > 1:
                dwp = gzeros((K, Cout, Pw * Cin_g), dy.device)
                if conv_wgrad(x, dy, dwp, db, B=B, Tsrc=Tin, Tdst=Tout, groups=groups // Pw, CR=Pw * Cin_g,
                              NG=Pw * Cout_g, K=K, stride=stride, dil=dil, pad=pad, inner=inner, up=up, dy_gate=gate,
                              dy_gate_slope=gslope, x_leaky=cfg["in_leaky"]):
                    dwt = _blockdiag_unpack(dwp, groups, Pw)
                    return dx, (dwt if tap_major else dwt.permute(1, 2, 0)), db, (dy if has_res else None), None
                if has_bias:
                    db.zero_()
            dwt = gzeros((K, Cout, Cin_g), dy.device)
            if conv_wgrad(x, dy, dwt, db, B=B, Tsrc=Tin, Tdst=Tout, groups=groups, CR=Cin_g, NG=Cout_g, K=K,
                          stride=stride, dil=dil, pad=pad, inner=inner, up=up, dy_gate=gate, dy_gate_slope=gslope,
                          x_leaky=cfg["in_leaky"]):
                return dx, (dwt if tap_major else dwt.permute(1, 2, 0)), db, (dy if has_res else None), None
            seg = make_seg(dy, 1, Cout, x, 1, Cin, Mtok, ntaps=K, a_gate=gate, a_gate_slope=gslope, b_tok_axis=2,
                           b_shift0=-pad, b_shift_step=dil,
                           b_map=dict(inner=inner, Tq=Tout, Tsrc=Tin, mul=stride, up=up), b_leaky=cfg["in_leaky"])
            gemm([seg], Cout_g, Cin_g, dwt, Cin_g, 1, groups=groups, a_gs=Cout_g, b_gs=Cin_g,
                 c_gs=Cout_g * Cin_g, bias_gs=Cout_g, accumulate=True,
                 splitk=_splitk_for(Cout_g * groups * K, Cin_g, Mtok), z_taps=K, c_tap=Cout * Cin_g, a_rowsum=db)
            dw = dwt if tap_major else dwt.permute(1, 2, 0)
        elif has_bias and ctx.needs_input_grad[2]:
            raise RuntimeError("bias gradient without weight gradient is not supported")
        return dx, dw, db, (dy if has_res else None), None


# bf16 mode: convolutions above this many multiply-adds x 2 run on csrc/cconv.hip (bf16 operand images + global_load_lds
# tiles); smaller ones keep the fp32-operand kernels (two extra cast launches would cost more than they save)
CCONV_MIN_FLOPS = float(os.environ.get("KANTTS_CCONV_MIN_FLOPS", 2e8))


def _cconv_ok(x, Cin_g, Cout_g, K, M, groups):
    return (get_precision() == "bf16" and not os.environ.get("KANTTS_NO_CCONV") and Cin_g % 8 == 0 and Cout_g % 8 == 0
            and K <= 64 and x.dtype == torch.float32
            and 2.0 * M * groups * Cout_g * Cin_g * K >= CCONV_MIN_FLOPS)


class _CConvCL(torch.autograd.Function):
    """The contract of _ConvCL on csrc/cconv.hip (bf16 mode, channel counts that are multiples of 8): the activated
    input is rounded to bf16 ONCE (act_cast_bf16) and that image serves the forward contraction, the weight gradient
    and the LeakyReLU' gate of the input gradient; the incoming gradient is gated and rounded once and serves both
    gradient contractions.  Reference: kantts/models/hifigan/layers.py:15-91, hifigan.py:82-97,217-267,332-407."""

    @staticmethod
    def forward(ctx, x, w, bias, res, cfg, x_img, w_imgs=(None, None)):
        x, w = _c(x), _c(w)
        stride, dil, pad, up, groups = cfg["stride"], cfg["dilation"], cfg["pad"], cfg["up"], cfg["groups"]
        inner, Tout = cfg["inner"], cfg["Tout"]
        B, Tin, Cin = x.shape[0], x.shape[1], x.shape[-1]
        tap_major = cfg.get("tap_major", False)
        if tap_major:
            K, Cout, Cin_g = w.shape
            wt = w
        else:
            Cout, Cin_g, K = w.shape
            wt = w.permute(2, 0, 1)
        Cout_g = Cout // groups
        # the bf16 image of the activated input: handed over by the producer (an epilogue wrote it) or made here
        xa = x_img if x_img is not None else act_cast_bf16(x, act_slope=cfg["in_leaky"])
        # small groups (the scale discriminators' 8 -> 16 / 16 -> 32 channel groups) are merged P at a time into dense
        # block-diagonal groups, as in _ConvCL: a 32-deep x 32-wide MFMA tile is the least the kernel can fill
        P = _group_pack(groups, Cin_g, Cout_g) if up == 1 else 1
        ctx.wd_img = w_imgs[1] if P == 1 else None
        if P > 1:
            wb = _cached_pack(w, ("fwd", groups, P, tap_major),
                              lambda: _blockdiag_pack(wt.reshape(K, Cout, Cin_g), groups, P).to(torch.bfloat16),
                              cfg.get("w_stable", False))
        elif w_imgs[0] is not None:
            wb = w_imgs[0]
        else:
            wb = torch.empty((K, Cout, Cin_g), device=x.device, dtype=torch.bfloat16).copy_(wt)
        y = torch.empty((B, Tout, inner, Cout) if x.dim() == 4 else (B, Tout, Cout), device=x.device, dtype=torch.float32)
        # cfg["image"]: also write bf16(LeakyReLU(y, slope)) (slope None: bf16(y)) for the convolution that consumes y
        want = cfg.get("image", False)
        y_img = torch.empty(y.shape, device=x.device, dtype=torch.bfloat16) if want is not False else None
        r = _c(res) if res is not None else None
        if not cconv(xa, wb, out=y, out_bf=y_img, bf_leaky=want if want is not False else None, B=B, Tsrc=Tin, Tdst=Tout,
                     groups=groups // P, CR=P * Cin_g, NG=P * Cout_g, K=K, in_mul=stride, in_add=-pad, in_kstep=dil, in_div=1,
                     phases=1, inner=inner, up=up, bias=bias, res=r, out_leaky=cfg["out_leaky"]):
            raise RuntimeError("cconv refused a shape _cconv_ok accepted")
        ctx.cfg = cfg
        ctx.has = (bias is not None, res is not None)
        ctx.xshape = tuple(x.shape)
        ctx.save_for_backward(xa, w, y if cfg["out_leaky"] is not None else None)
        ctx.res_for_gate = r if cfg["out_leaky"] is not None else None
        if y_img is None:
            return y
        ctx.set_materialize_grads(False)  # no zero tensor for the gradient of a non-differentiable output
        ctx.mark_non_differentiable(y_img)
        return y, y_img

    @staticmethod
    def backward(ctx, dy, _dimg=None):
        cfg = ctx.cfg
        xa, w, y = ctx.saved_tensors
        has_bias, has_res = ctx.has
        stride, dil, pad, up, groups = cfg["stride"], cfg["dilation"], cfg["pad"], cfg["up"], cfg["groups"]
        inner, Tout = cfg["inner"], cfg["Tout"]
        dy = _c(dy)
        B, Tin, Cin = xa.shape[0], xa.shape[1], xa.shape[-1]
        tap_major = cfg.get("tap_major", False)
        if tap_major:
            K, Cout, Cin_g = w.shape
        else:
            Cout, Cin_g, K = w.shape
        Cout_g = Cout // groups
        gate, gslope = (y, cfg["out_leaky"]) if cfg["out_leaky"] is not None else (None, 0.0)
        if gate is not None and ctx.res_for_gate is not None:
            gate = y - ctx.res_for_gate
        dyb = act_cast_bf16(dy, gate=gate, gate_slope=gslope)
        dx = dw = db = None
        in_gate = xa if cfg["in_leaky"] is not None else None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(ctx.xshape, device=dy.device, dtype=torch.float32)
            if tap_major:
                wd = w.view(K, groups, Cout_g, Cin_g).transpose(2, 3)
            else:
                wd = w.view(groups, Cout_g, Cin_g, K).permute(3, 0, 2, 1)  # (K, groups, Cin_g, Cout_g)
            if up > 1 and stride == 1:
                # one strided pass over dy with the combined taps W'[j] (see _ConvCL.backward)
                S, jmin = _upsample_tap_matrix(K, up, pad, dil, dy.device)
                wj = torch.matmul(S, wd.reshape(K, -1)).to(torch.bfloat16)
                ok = cconv(dyb, wj, out=dx, B=B, Tsrc=Tout, Tdst=Tin, groups=groups, CR=Cout_g, NG=Cin_g, K=S.shape[0],
                           in_mul=up, in_add=jmin, in_kstep=1, in_div=1, phases=1, inner=inner, out_gate=in_gate,
                           out_gate_slope=cfg["in_leaky"] or 0.0)
            elif up == 1:
                Pd = _group_pack(groups, Cout_g, Cin_g)
                if Pd > 1:
                    wdb = _cached_pack(w, ("bwd", groups, Pd, tap_major),
                                       lambda: _blockdiag_pack(wd.reshape(K, Cin, Cout_g), groups, Pd).to(torch.bfloat16),
                                       cfg.get("w_stable", False))
                elif ctx.wd_img is not None:
                    wdb = ctx.wd_img.view(K, Cin, Cout_g)
                else:
                    wdb = torch.empty((K, Cin, Cout_g), device=dy.device, dtype=torch.bfloat16).copy_(wd.reshape(K, Cin, Cout_g))
                ok = cconv(dyb, wdb, out=dx, B=B, Tsrc=Tout, Tdst=Tin, groups=groups // Pd, CR=Pd * Cout_g, NG=Pd * Cin_g,
                           K=K, in_mul=1, in_add=pad, in_kstep=-dil, in_div=stride, phases=stride, inner=inner,
                           out_gate=in_gate, out_gate_slope=cfg["in_leaky"] or 0.0)
            else:
                ok = False
            if not ok:
                raise RuntimeError("cconv input gradient refused a shape its forward accepted")
        if ctx.needs_input_grad[1]:
            Pw = _group_pack(groups, Cin_g, Cout_g) if up == 1 else 1
            dwt = gzeros((K, Cout, Pw * Cin_g), dy.device)
            db = gzeros((Cout,), dy.device) if has_bias else None
            if not cconv_wgrad(xa, dyb, dwt, db, B=B, Tsrc=Tin, Tdst=Tout, groups=groups // Pw, CR=Pw * Cin_g,
                               NG=Pw * Cout_g, K=K, stride=stride, dil=dil, pad=pad, inner=inner, up=up):
                raise RuntimeError("cconv weight gradient refused a shape its forward accepted")
            if Pw > 1:
                dwt = _blockdiag_unpack(dwt, groups, Pw)
            dw = dwt if tap_major else dwt.permute(1, 2, 0)
        elif has_bias and ctx.needs_input_grad[2]:
            raise RuntimeError("bias gradient without weight gradient is not supported")
        return dx, dw, db, (dy if has_res else None), None, None, None


class _ResStackBF16(torch.autograd.Function):
    """A HiFi-GAN residual block -- n x [LeakyReLU -> conv(k, d_i) -> LeakyReLU -> conv(k, 1) -> + x] (reference
    kantts/models/hifigan/layers.py:168-226) -- as ONE autograd node on csrc/cconv.hip (bf16 mode).  Only the residual
    stream x_i and its gradient g_i exist in fp32; everything between two convolutions is a bf16 image written by the
    producing epilogue and read by the consuming loader:
      forward   a_i = bf16(LReLU(x_i))                          (epilogue of the previous conv, or one cast pass for i = 0)
                ta_i = bf16(LReLU(conv1_i(a_i) + b1_i))         (the dilated conv writes ONLY this image)
                x_{i+1} = conv2_i(ta_i) + b2_i + x_i  (fp32)  and  a_{i+1}
      backward  gb = bf16(g_{i+1})                              (epilogue of the previous input gradient, or one cast pass)
                dW2_i, db2_i from (ta_i, gb);   dt_i = bf16(dgrad2(gb) * LReLU'(ta_i))
                dW1_i, db1_i from (a_i, dt_i);  g_i = g_{i+1} + dgrad1(dt_i) * LReLU'(a_i)  (fp32) and bf16(g_i)
    against twelve separate nodes this drops the fp32 write of every dilated conv's output, every gate / cast pass of the
    backward (two reads of 4 bytes and a write per element and conv) and half of the input-gradient writes.
    Arguments: x, x_img (bf16(LReLU(x)) or None), slope, cfg = [(K, dil, pad), ...] per conv1 (conv2: dilation 1, pad2),
    then w1_0, b1_0, w2_0, b2_0, w1_1, ... (tap-major (K, C, C) weights)."""

    @staticmethod
    def forward(ctx, x, x_img, slope, cfgs, imgs, *wb):
        x = _c(x)
        B, T, C = x.shape
        n = len(cfgs)
        a = x_img if x_img is not None else act_cast_bf16(x, act_slope=slope)
        saved, wimgs = [], []
        xi = x
        ctx.wd_imgs = [(im1[1], im2[1]) for im1, im2 in imgs]
        for i, (K, dil, pad1, pad2) in enumerate(cfgs):
            w1, b1, w2, b2 = wb[4 * i:4 * i + 4]
            (f1, _), (f2, _) = imgs[i]
            w1b = f1 if f1 is not None else torch.empty((K, C, C), device=x.device, dtype=torch.bfloat16).copy_(w1)
            w2b = f2 if f2 is not None else torch.empty((K, C, C), device=x.device, dtype=torch.bfloat16).copy_(w2)
            ta = torch.empty((B, T, C), device=x.device, dtype=torch.bfloat16)
            if not cconv(a, w1b, out_bf=ta, bf_leaky=slope, B=B, Tsrc=T, Tdst=T, groups=1, CR=C, NG=C, K=K, in_mul=1,
                         in_add=-pad1, in_kstep=dil, in_div=1, phases=1, bias=b1):
                raise RuntimeError("cconv refused a residual-block convolution")
            xn = torch.empty((B, T, C), device=x.device, dtype=torch.float32)
            an = torch.empty((B, T, C), device=x.device, dtype=torch.bfloat16) if i + 1 < n else None
            if not cconv(ta, w2b, out=xn, out_bf=an, bf_leaky=slope, B=B, Tsrc=T, Tdst=T, groups=1, CR=C, NG=C, K=K,
                         in_mul=1, in_add=-pad2, in_kstep=1, in_div=1, phases=1, bias=b2, res=xi):
                raise RuntimeError("cconv refused a residual-block convolution")
            saved += [a, ta]
            wimgs += [w1, w2]
            a, xi = an, xn
        ctx.cfgs, ctx.slope, ctx.shape = cfgs, slope, (B, T, C)
        ctx.has_bias = [(wb[4 * i + 1] is not None, wb[4 * i + 3] is not None) for i in range(n)]
        ctx.save_for_backward(*saved, *wimgs)
        return xi

    @staticmethod
    def backward(ctx, g):
        cfgs, slope = ctx.cfgs, ctx.slope
        B, T, C = ctx.shape
        n = len(cfgs)
        t = ctx.saved_tensors
        imgs, ws = t[:2 * n], t[2 * n:]
        g = _c(g)
        gb = act_cast_bf16(g)
        grads = [None] * (4 * n)
        need_x = ctx.needs_input_grad[0]
        for i in range(n - 1, -1, -1):
            K, dil, pad1, pad2 = cfgs[i]
            a, ta = imgs[2 * i], imgs[2 * i + 1]
            w1, w2 = ws[2 * i], ws[2 * i + 1]
            # conv2 (dilation 1): weight / bias gradient, then the gradient of its input image gated by LReLU'(t_i)
            b1, b2 = ctx.has_bias[i]
            dw2 = gzeros((K, C, C), g.device)
            db2 = gzeros((C,), g.device) if b2 else None
            if not cconv_wgrad(ta, gb, dw2, db2, B=B, Tsrc=T, Tdst=T, groups=1, CR=C, NG=C, K=K, stride=1, dil=1, pad=pad2):
                raise RuntimeError("cconv weight gradient refused a residual-block convolution")
            d1, d2 = ctx.wd_imgs[i]
            w2d = (d2.view(K, C, C) if d2 is not None
                   else torch.empty((K, C, C), device=g.device, dtype=torch.bfloat16).copy_(w2.transpose(1, 2)))
            dt = torch.empty((B, T, C), device=g.device, dtype=torch.bfloat16)
            if not cconv(gb, w2d, out_bf=dt, B=B, Tsrc=T, Tdst=T, groups=1, CR=C, NG=C, K=K, in_mul=1, in_add=pad2,
                         in_kstep=-1, in_div=1, phases=1, out_gate=ta, out_gate_slope=slope):
                raise RuntimeError("cconv input gradient refused a residual-block convolution")
            dw1 = gzeros((K, C, C), g.device)
            db1 = gzeros((C,), g.device) if b1 else None
            if not cconv_wgrad(a, dt, dw1, db1, B=B, Tsrc=T, Tdst=T, groups=1, CR=C, NG=C, K=K, stride=1, dil=dil, pad=pad1):
                raise RuntimeError("cconv weight gradient refused a residual-block convolution")
            grads[4 * i:4 * i + 4] = [dw1, db1, dw2, db2]
            if i > 0 or need_x:
                w1d = (d1.view(K, C, C) if d1 is not None
                       else torch.empty((K, C, C), device=g.device, dtype=torch.bfloat16).copy_(w1.transpose(1, 2)))
                gn = torch.empty((B, T, C), device=g.device, dtype=torch.float32)
                gnb = torch.empty((B, T, C), device=g.device, dtype=torch.bfloat16) if i > 0 else None
                # the identity path g joins AFTER the LeakyReLU' gate of the convolution branch (res_after_gate)
                if not cconv(dt, w1d, out=gn, B=B, Tsrc=T, Tdst=T, groups=1, CR=C, NG=C, K=K, in_mul=1, in_add=pad1,
                             in_kstep=-dil, in_div=1, phases=1, out_gate=a, out_gate_slope=slope, res=g,
                             res_after_gate=True, out_bf=gnb):
                    raise RuntimeError("cconv input gradient refused a residual-block convolution")
                g, gb = gn, gnb
        return (g if need_x else None, None, None, None, None) + tuple(grads)


def res_stack_ok(x, K):
    """Does the fused residual-stack node apply to input x (B, T, C) with kernel size K?  (bf16 mode, MFMA-sized)"""
    return (x.dim() == 3 and not os.environ.get("KANTTS_NO_RES_STACK")
            and _cconv_ok(x, x.shape[2], x.shape[2], K, x.shape[0] * x.shape[1], 1))


def res_stack(x, slope, convs):
    """ResidualBlock forward on the fused bf16 node.  ``convs``: [(w1, b1, K, dil, pad1, w2, b2, pad2), ...] with
    tap-major (K, C, C) weights; returns None when a weight has another shape."""
    B, T, C = x.shape
    for w1, b1, K, dil, pad1, w2, b2, pad2 in convs:
        if tuple(w1.shape) != (K, C, C) or tuple(w2.shape) != (K, C, C):
            return None
    cfgs = [(int(K), int(dil), int(pad1), int(pad2)) for _, _, K, dil, pad1, _, _, pad2 in convs]
    flat = []
    for w1, b1, K, dil, pad1, w2, b2, pad2 in convs:
        flat += [w1, b1, w2, b2]
    x_img = get_image(x, slope) if x.is_contiguous() else None
    imgs = [(weight_images(w1, 1), weight_images(w2, 1)) for w1, _, _, _, _, w2, _, _ in convs]
    return _ResStackBF16.apply(x, x_img, float(slope), cfgs, imgs, *flat)


_IMG_ATTR = "_kantts_bf16_image"


def set_image(t, slope, img):
    """Attach bf16(LeakyReLU(t, slope)) (slope None: bf16(t)) to the tensor object ``t``: the next convolution that reads
    ``t`` with that activation takes it as its operand image instead of making one (bf16 mode).  The attribute lives on
    the Python object only -- views and copies do not carry it, which is the safe direction.  The tensor's version
    counter is recorded: an in-place change of ``t`` after this call (masked_fill_, add_, a user hook) makes the image
    stale, and get_image then ignores it instead of handing the convolution yesterday's activations."""
    setattr(t, _IMG_ATTR, (slope, img, t._version))
    return t


def get_image(t, slope):
    hit = getattr(t, _IMG_ATTR, None)
    if (hit is not None and hit[0] == slope and hit[1].shape == t.shape and hit[1].device == t.device
            and hit[2] == t._version):
        return hit[1]
    return None


class _MeanMany(torch.autograd.Function):
    """scale * sum of n same-shaped fp32 tensors in ONE pass (kantts_mean_many), optionally with the bf16 image of
    LeakyReLU(result, image_slope) for the convolution that consumes it; backward = one launch that writes every input
    its OWN gradient buffer (the inputs are outputs of branches that ran on their own streams: see _BranchExit)."""

    @staticmethod
    def forward(ctx, scale, image_slope, *xs):
        import ctypes

        xs = [_c(x) for x in xs]
        n, numel = len(xs), xs[0].numel()
        out = torch.empty_like(xs[0])
        img = None
        if image_slope is not None and get_precision() == "bf16" and numel % 8 == 0:
            img = torch.empty(xs[0].shape, device=xs[0].device, dtype=torch.bfloat16)
        arr = (ctypes.c_void_p * n)(*[ptr(x, torch.float32) for x in xs])
        check(lib().kantts_mean_many(arr, n, float(scale), ptr(out), ptr(img), float(image_slope or 0.0), numel, stream()),
              "mean_many")
        ctx.cfg = (float(scale), n)
        if img is not None:
            ctx.set_materialize_grads(False)  # no zero tensor for the gradient of a non-differentiable output
            ctx.mark_non_differentiable(img)
        return out, img

    @staticmethod
    def backward(ctx, g, _g_img):
        import ctypes

        scale, n = ctx.cfg
        g = _c(g)
        outs = [torch.empty_like(g) for _ in range(n)]
        arr = (ctypes.c_void_p * n)(*[ptr(o, torch.float32) for o in outs])
        check(lib().kantts_scale_to_many(ptr(g, torch.float32), scale, arr, n, g.numel(), stream()), "scale_to_many")
        return (None, None, *outs)


def mean_many_applies(n, like):
    """Whether mean_many() will take its one-launch path for ``n`` tensors shaped like ``like`` (the caller then skips the
    per-branch gradient clones of parallel_branches)."""
    return (2 <= n <= 8 and like.dtype == torch.float32 and like.numel() % 4 == 0
            and not os.environ.get("KANTTS_NO_MEAN_MANY"))


def mean_many(xs, image_slope=None):
    """mean of the tensors in ``xs`` (fp32, same shape, numel % 4 == 0; anything else: the ATen chain).  image_slope: also
    attach the bf16 image of LeakyReLU(mean, image_slope) for the next convolution (bf16 mode)."""
    xs = list(xs)
    ok = (len(xs) >= 2 and len(xs) <= 8 and all(x.dtype == torch.float32 and x.shape == xs[0].shape and x.is_cuda == xs[0].is_cuda
                                                 for x in xs) and xs[0].numel() % 4 == 0
          and not os.environ.get("KANTTS_NO_MEAN_MANY"))
    if not ok:
        acc = xs[0]
        for x in xs[1:]:
            acc = acc + x
        return acc / len(xs)
    out, img = _MeanMany.apply(1.0 / len(xs), image_slope, *xs)
    if img is not None:
        set_image(out, image_slope, img)
    return out


def act_image(t, slope):
    """Make (once) and attach the bf16 image of LeakyReLU(t, slope) in bf16 mode; a no-op otherwise.  Call it before
    forking parallel branches that all read ``t``: the image is then produced once, on the forking stream."""
    if get_precision() != "bf16" or not t.is_floating_point() or t.dtype != torch.float32 or t.numel() % 8:
        return t
    if get_image(t, slope) is None:
        set_image(t, slope, act_cast_bf16(_c(t.detach()), act_slope=slope))
    return t


def conv_cl(x, w, bias=None, *, stride=1, dilation=1, pad=0, Tout=None, up=1, groups=1, inner=1, in_leaky=None,
            out_leaky=None, res=None, tap_major=False, image=False):
    """pad = left padding in (upsampled) input samples; Tout defaults to the 'same'/causal length.
    tap_major: w is (K, Cout, Cin_g) (ops.weight_norm_tap) instead of the parameter layout (Cout, Cin_g, K).
    image: a slope (or None for no activation) asks the bf16-mode kernel to also write the bf16 image of
    LeakyReLU(result, slope) and attach it to the result (set_image) for the convolution that consumes it; ignored by the
    fp32-operand kernels."""
    K = w.shape[0] if tap_major else w.shape[-1]
    Tin = x.shape[1]
    if Tout is None:
        Tout = Tin * up if stride == 1 else (Tin + 2 * pad - dilation * (K - 1) - 1) // stride + 1
    cfg = dict(stride=int(stride), dilation=int(dilation), pad=int(pad), Tout=int(Tout), up=int(up), groups=int(groups),
               inner=int(inner), in_leaky=in_leaky, out_leaky=out_leaky, tap_major=bool(tap_major),
               w_stable=bool(getattr(w, "_kantts_stable", False)))
    Cout, Cin_g = (w.shape[1], w.shape[2]) if tap_major else (w.shape[0], w.shape[1])
    if (up == 1 or stride == 1) and _cconv_ok(x, Cin_g, Cout // int(groups), K, x.shape[0] * int(Tout) * int(inner),
                                                int(groups)):
        x_img = get_image(x, in_leaky) if x.is_contiguous() else None
        w_imgs = weight_images(w, groups) if tap_major else (None, None)
        if image is False:
            return _CConvCL.apply(x, w, bias, res, cfg, x_img, w_imgs)
        cfg["image"] = image
        y, y_img = _CConvCL.apply(x, w, bias, res, cfg, x_img, w_imgs)
        return set_image(y, image, y_img)
    if image is None:
        cfg["image"] = None
    y = _ConvCL.apply(x, w, bias, res, cfg)
    img = cfg.pop("c1_image", None)
    return y if img is None else set_image(y, None, img)


class _ConvTransposeCL(torch.autograd.Function):
    """Causal polyphase ConvTranspose1d (kernel K = taps*stride, output trimmed to Tin*stride):
        y[b, q*s + r, co] = bias[co] + sum_j sum_ci act_in(x[b, q - j, ci]) w[ci, co, r + j*s]   (+ res)
    ONE contraction for all s output phases: the output (B, Tin*s, Cout) is the row-major matrix
    (B*Tin, s*Cout), the weight is re-laid as W2[j][(r,co)][ci] and the K/s taps are token shifts of x --
    nothing is zero-stuffed, x is read once per tap from L2 and y is written once, fully coalesced.
    Reference: CausalConvTranspose1d, kantts/models/hifigan/layers.py:125-165, hifigan.py:67-80,160."""

    @staticmethod
    def forward(ctx, x, w, bias, res, s, in_leaky, act=None):
        x, w = _c(x), _c(w)
        B, Tin, Cin = x.shape
        _, Cout, K = w.shape
        assert K % s == 0
        taps = K // s
        ctx.bf = False
        if _cconv_ok(x, Cin, s * Cout, taps, B * Tin, 1) and Tin >= 16:
            # bf16 mode: the activated bf16 image of x (handed over by the producer, or made here in one pass) feeds the
            # forward contraction, the weight gradient and the LeakyReLU' gate of the input gradient
            xa = act if act is not None else act_cast_bf16(x, act_slope=in_leaky)
            y = upsample_forward(xa, w, bias, s, res=res)
            if y is None:
                y = torch.empty((B, Tin * s, Cout), device=x.device, dtype=torch.float32)
                w2b = w.view(Cin, Cout, taps, s).permute(2, 3, 1, 0).to(torch.bfloat16).contiguous()  # (taps, s, Cout, Cin)
                if not cconv(xa, w2b, out=y, B=B, Tsrc=Tin, Tdst=Tin, groups=1, CR=Cin, NG=s * Cout, K=taps, in_mul=1,
                             in_add=0, in_kstep=-1, in_div=1, phases=1, bias=bias.repeat(s) if bias is not None else None,
                             res=_c(res) if res is not None else None):
                    raise RuntimeError("cconv refused a transposed convolution _cconv_ok accepted")
            ctx.bf = True
            ctx.cfg = (s, in_leaky, bias is not None, res is not None)
            ctx.xshape = tuple(x.shape)
            ctx.save_for_backward(xa, w)
            return y
        y = torch.empty((B, Tin * s, Cout), device=x.device, dtype=torch.float32)
        r_t = _c(res) if res is not None else None
        w2 = w.view(Cin, Cout, taps, s).permute(2, 3, 1, 0).contiguous()  # (taps, s, Cout, Cin)
        brep = bias.repeat(s) if bias is not None else None
        ctx.cfg = (s, in_leaky, bias is not None, res is not None)
        ctx.save_for_backward(x, w)
        # the polyphase form IS a stride-1 convolution with K/s taps onto s*Cout channels: the window kernel reads x
        # once per 32-channel chunk (and runs 128x128 tiles on the wide early layers)
        # (sequences shorter than a 64-row tile stay on the GEMM, whose rows run across batch items)
        if Tin >= 64 and conv_win(x, w2, y, B=B, Tsrc=Tin, Tdst=Tin, groups=1, CR=Cin, NG=s * Cout, K=taps, in_mul=1, in_add=0,
                    in_kstep=-1, in_div=1, phases=1, bias=brep, res=r_t, in_leaky=in_leaky):
            return y
        seg = make_seg(x, Cin, 1, w2, Cin, 1, Cin, ntaps=taps, b_tap=s * Cout * Cin, a_tok_axis=1, a_shift0=0,
                       a_shift_step=-1, a_map=dict(Tq=Tin, Tsrc=Tin), a_leaky=in_leaky)
        gemm([seg], B * Tin, s * Cout, y, s * Cout, 1, bias=brep, res=r_t, r_is=s * Cout, r_js=1)
        return y

    @staticmethod
    def backward(ctx, dy):
        s, in_leaky, has_bias, has_res = ctx.cfg
        x, w = ctx.saved_tensors
        dy = _c(dy)
        B, Tin, Cin = x.shape
        _, Cout, K = w.shape
        taps = K // s
        N2 = s * Cout
        M = B * Tin
        dx = dw = db = None
        if ctx.bf:  # x is the bf16 activated image
            dyb = act_cast_bf16(dy)
            if ctx.needs_input_grad[0]:
                dx = torch.empty(ctx.xshape, device=dy.device, dtype=torch.float32)
                w3b = w.view(Cin, Cout, taps, s).permute(2, 0, 3, 1).to(torch.bfloat16).contiguous()  # (taps, Cin, s, Cout)
                if not cconv(dyb, w3b, out=dx, B=B, Tsrc=Tin, Tdst=Tin, groups=1, CR=N2, NG=Cin, K=taps, in_mul=1, in_add=0,
                             in_kstep=1, in_div=1, phases=1, out_gate=x if in_leaky is not None else None,
                             out_gate_slope=in_leaky or 0.0):
                    raise RuntimeError("cconv input gradient refused a transposed convolution")
            if ctx.needs_input_grad[1]:
                dw2 = gzeros((taps, N2, Cin), dy.device)
                db2 = gzeros((N2,), dy.device) if (has_bias and ctx.needs_input_grad[2]) else None
                if not cconv_wgrad(x, dyb, dw2, db2, B=B, Tsrc=Tin, Tdst=Tin, groups=1, CR=Cin, NG=N2, K=taps, stride=1,
                                   dil=1, pad=taps - 1):
                    raise RuntimeError("cconv weight gradient refused a transposed convolution")
                dw = dw2.flip(0).view(taps, s, Cout, Cin).permute(3, 2, 0, 1).reshape(Cin, Cout, K)
                if db2 is not None:
                    db = db2.view(s, Cout).sum(0)
            elif has_bias and ctx.needs_input_grad[2]:
                db = dy.sum(dim=(0, 1))
            return dx, dw, db, (dy if has_res else None), None, None, None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            w3 = w.view(Cin, Cout, taps, s).permute(2, 0, 3, 1).contiguous()  # (taps, Cin, s, Cout)
            if Tin < 64 or not conv_win(dy, w3, dx, B=B, Tsrc=Tin, Tdst=Tin, groups=1, CR=N2, NG=Cin, K=taps, in_mul=1, in_add=0,
                            in_kstep=1, in_div=1, phases=1, out_gate=x if in_leaky is not None else None,
                            out_gate_slope=in_leaky or 0.0):
                seg = make_seg(dy, N2, 1, w3, N2, 1, N2, ntaps=taps, b_tap=Cin * N2, a_tok_axis=1, a_shift0=0,
                               a_shift_step=1, a_map=dict(Tq=Tin, Tsrc=Tin))
                gemm([seg], M, Cin, dx, Cin, 1, gate=x if in_leaky is not None else None, gate_slope=in_leaky or 0.0)
        if ctx.needs_input_grad[1]:
            dw2 = gzeros((taps, N2, Cin), dy.device)
            db2 = gzeros((N2,), dy.device) if (has_bias and ctx.needs_input_grad[2]) else None
            # tap j reads x[q - j]: with k' = taps-1-j this is x[q + k' - (taps-1)], a stride-1 weight gradient
            if conv_wgrad(x, dy.view(B, Tin, N2), dw2, db2, B=B, Tsrc=Tin, Tdst=Tin, groups=1, CR=Cin, NG=N2, K=taps,
                          stride=1, dil=1, pad=taps - 1, x_leaky=in_leaky):
                dw2 = dw2.flip(0)
            else:
                seg = make_seg(dy, 1, N2, x, 1, Cin, M, ntaps=taps, b_tok_axis=2, b_shift0=0, b_shift_step=-1,
                               b_map=dict(Tq=Tin, Tsrc=Tin), b_leaky=in_leaky)
                gemm([seg], N2, Cin, dw2, Cin, 1, accumulate=True, splitk=_splitk_for(N2 * taps, Cin, M), z_taps=taps,
                     c_tap=N2 * Cin, a_rowsum=db2)
            dw = dw2.view(taps, s, Cout, Cin).permute(3, 2, 0, 1).reshape(Cin, Cout, K)
            if db2 is not None:
                db = db2.view(s, Cout).sum(0)
        elif has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(dim=(0, 1))
        return dx, dw, db, (dy if has_res else None), None, None, None


def conv_transpose_cl(x, w, bias, stride, in_leaky=None, res=None, act=None):
    """``act``: optional bf16(LeakyReLU(x)) (ops.sin_add(..., act_slope=)): bf16 mode then runs the forward pass on the
    streaming upsampling kernels."""
    return _ConvTransposeCL.apply(x, w, bias, res, int(stride), in_leaky, act)


class _WeightNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, g):
        v, g = _c(v), _c(g)
        rows = v.shape[0]
        cols = v.numel() // rows
        w = torch.empty_like(v)
        check(lib().kantts_weight_norm_fwd(ptr(v, torch.float32), ptr(g, torch.float32), ptr(w), rows, cols, stream()),
              "weight_norm_fwd")
        ctx.save_for_backward(v, g)
        return w

    @staticmethod
    def backward(ctx, dw):
        v, g = ctx.saved_tensors
        dw = _c(dw)
        rows = v.shape[0]
        cols = v.numel() // rows
        dv, dg = torch.empty_like(v), torch.empty_like(g)
        check(lib().kantts_weight_norm_bwd(ptr(dw, torch.float32), ptr(v), ptr(g), ptr(dv), ptr(dg), rows, cols,
                                           stream()), "weight_norm_bwd")
        return dv, dg


class _WeightNormTap(torch.autograd.Function):
    """w_tap[k][co][ci] = g[co] * v[co][ci][k] / ||v[co]||: the weight-norm reparametrisation written straight into
    the tap-major layout the convolution kernels read; backward takes the tap-major weight gradient they produce."""

    @staticmethod
    def forward(ctx, v, g, groups):
        v, g = _c(v), _c(g)
        Cout, cin, K = v.shape
        w = torch.empty((K, Cout, cin), device=v.device, dtype=torch.float32)
        ctx.save_for_backward(v, g)
        if groups:  # bf16 mode: the operand images of csrc/cconv.hip in the same launch
            wf = torch.empty((K, Cout, cin), device=v.device, dtype=torch.bfloat16)
            wd = torch.empty((K, groups, cin, Cout // groups), device=v.device, dtype=torch.bfloat16)
            check(lib().kantts_weight_norm_tap_images(ptr(v, torch.float32), ptr(g, torch.float32), ptr(w), ptr(wf), ptr(wd),
                                                      Cout, cin, K, int(groups), stream()), "weight_norm_tap_images")
            ctx.set_materialize_grads(False)  # no zero tensor for the gradient of a non-differentiable output
            ctx.mark_non_differentiable(wf, wd)
            return w, wf, wd
        check(lib().kantts_weight_norm_strided_fwd(ptr(v, torch.float32), ptr(g, torch.float32), ptr(w), Cout, cin, K, cin,
                                                   1, Cout * cin, stream()), "weight_norm_strided_fwd")
        return w

    @staticmethod
    def backward(ctx, dw, *_unused):
        v, g = ctx.saved_tensors
        dw = _c(dw)
        Cout, cin, K = v.shape
        dv, dg = torch.empty_like(v), torch.empty_like(g)
        check(lib().kantts_weight_norm_strided_bwd(ptr(dw, torch.float32), ptr(v), ptr(g), ptr(dv), ptr(dg), Cout, cin, K,
                                                   cin, 1, Cout * cin, stream()), "weight_norm_strided_bwd")
        return dv, dg, None


class _WeightNormImage(torch.autograd.Function):
    """The tap-major weight-normed weight of one layer, ALREADY computed for the whole network by the arena's table-driven
    launch (ParamArena.build_weight_norm_images / kantts_weight_norm_table): forward hands out an alias of the layer's
    slice of that buffer, backward is the per-layer reparametrisation backward of _WeightNormTap."""

    @staticmethod
    def forward(ctx, v, g, holder):
        ctx.save_for_backward(v, g)
        ctx.holder = holder
        return holder[1].view(holder[1].shape)  # (holder: not a tensor argument -- the result is no view of an input)

    @staticmethod
    def backward(ctx, dw):
        v, g = ctx.saved_tensors
        dw = _c(dw)
        arena = ctx.holder[0]()
        how = arena.defer_weight_norm_backward(ctx.holder[5], dw) if arena is not None else False
        if how == "extra":  # a further application of the layer in this step: added to its slot at the flush
            return None, None, None
        if how:
            # one table-driven launch per network before the optimizer step (ParamArena.flush_weight_norm_backward, called
            # from wgrad_overlap.join()): autograd receives views of the gradient arena that are filled then
            return arena.grad_slot(v), arena.grad_slot(g), None
        v = _c(v)
        if v.dim() == 4:
            v = v.squeeze(-1)
        Cout, cin, K = v.shape
        dv, dg = torch.empty_like(v), torch.empty_like(g)
        check(lib().kantts_weight_norm_strided_bwd(ptr(dw, torch.float32), ptr(v), ptr(g), ptr(dv), ptr(dg), Cout, cin, K,
                                                   cin, 1, Cout * cin, stream()), "weight_norm_strided_bwd")
        return dv.view(ctx.saved_tensors[0].shape), dg, None


wn_pending_arenas = []  # arenas with deferred weight-norm backward work (flushed by wgrad_overlap.join)


def flush_weight_norm_backward():
    while wn_pending_arenas:
        wn_pending_arenas.pop().flush_weight_norm_backward()


_WIMG_ATTR = "_kantts_bf16_weight_images"


def weight_norm_image(m):
    """(K, Cout, Cin_g) weight of the weight-normed convolution module ``m`` from its network's image buffers, or None
    when there are none / they do not reflect the current parameters (the caller then re-parametrises this layer itself)."""
    wn = getattr(m, "_kantts_wn", None)
    if wn is None or os.environ.get("KANTTS_NO_WEIGHT_NORM_TABLE"):
        return None
    arena = wn[0]()
    if arena is None or not arena.weight_norm_images_fresh():
        return None
    w = _WeightNormImage.apply(m.weight_v, m.weight_g, wn)
    w._kantts_stable = True  # a view of the network's image buffer: same address all step long (_cached_pack)
    if get_precision() == "bf16" and wn[2] is not None and not os.environ.get("KANTTS_NO_WEIGHT_IMAGES"):
        setattr(w, _WIMG_ATTR, (wn[2], wn[3], wn[4]))
    return w


def weight_norm_tap(v, g, groups=1):
    """Weight-normed conv weight in tap-major layout (K, Cout, Cin_g) for conv_cl(..., tap_major=True).  In bf16 mode the
    same launch also writes the bf16 images the MFMA convolution kernels read -- (K, Cout, Cin_g) for the forward and
    (K, groups, Cin_g, Cout_g) for the input-gradient contraction -- and attaches them to the returned tensor object."""
    if get_precision() == "bf16" and v.shape[1] % 8 == 0 and (v.shape[0] // int(groups)) % 8 == 0 and not os.environ.get(
            "KANTTS_NO_WEIGHT_IMAGES"):
        w, wf, wd = _WeightNormTap.apply(v, g, int(groups))
        setattr(w, _WIMG_ATTR, (wf, wd, int(groups)))
        return w
    return _WeightNormTap.apply(v, g, 0)


def weight_images(w, groups):
    """(forward image, input-gradient image) attached by weight_norm_tap for this group count, or (None, None)."""
    hit = getattr(w, _WIMG_ATTR, None)
    if hit is not None and hit[2] == int(groups):
        return hit[0], hit[1]
    return None, None


def weight_norm(v, g):
    """w = g * v / ||v|| over all dims but 0 (torch.nn.utils.weight_norm, dim=0)."""
    return _WeightNorm.apply(v, g)


class _SinAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        y = torch.empty_like(x)
        check(lib().kantts_sinadd_fwd(ptr(x, torch.float32), ptr(y), x.numel(), stream()), "sinadd_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(x)
        check(lib().kantts_sinadd_bwd(ptr(dy, torch.float32), ptr(x), ptr(dx), x.numel(), stream()), "sinadd_bwd")
        return dx


class _SinAddAct(torch.autograd.Function):
    """sin(x) + x plus a non-differentiable bf16 LeakyReLU image of the result (kantts_sinadd_lrelu_fwd)."""

    @staticmethod
    def forward(ctx, x, slope):
        x = _c(x)
        y = torch.empty_like(x)
        act = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
        check(lib().kantts_sinadd_lrelu_fwd(ptr(x, torch.float32), ptr(y), ptr(act), float(slope), x.numel(), stream()),
              "sinadd_lrelu_fwd")
        ctx.save_for_backward(x)
        ctx.set_materialize_grads(False)  # no zero tensor for the gradient of a non-differentiable output
        ctx.mark_non_differentiable(act)
        return y, act

    @staticmethod
    def backward(ctx, dy, _da):
        (x,) = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(x)
        check(lib().kantts_sinadd_bwd(ptr(dy, torch.float32), ptr(x), ptr(dx), x.numel(), stream()), "sinadd_bwd")
        return dx, None


class _Dropout2Add(torch.autograd.Function):
    """x * keep1 * keep2 (+ res) with regenerated masks (kantts_dropout2_add); backward = the same kernel on dy."""

    @staticmethod
    def forward(ctx, x, res, p1, p2):
        from . import rng_ptr

        x = _c(x)
        y = torch.empty_like(x)
        s1 = next_seed() if p1 > 0 else 0
        s2 = next_seed() if p2 > 0 else 0
        r = None if res is None else _c(res)
        check(lib().kantts_dropout2_add(ptr(x, torch.float32), ptr(r, torch.float32), ptr(y), x.numel(), float(p1), s1,
                                        float(p2), s2, rng_ptr(x.device), stream()), "dropout2_add")
        ctx.cfg = (float(p1), s1, float(p2), s2, res is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import rng_ptr

        p1, s1, p2, s2, has_res = ctx.cfg
        dy = _c(dy)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(dy)
            check(lib().kantts_dropout2_add(ptr(dy, torch.float32), None, ptr(dx), dy.numel(), p1, s1, p2, s2,
                                            rng_ptr(dy.device), stream()), "dropout2_add")
        return dx, (dy if has_res and ctx.needs_input_grad[1] else None), None, None


def dropout2_add(x, p1=0.0, p2=0.0, res=None):
    """dropout_{p2}(dropout_{p1}(x)) (+ res) in one pass (fp32, numel % 4 == 0); plain add / identity when both p are 0."""
    if p1 <= 0.0 and p2 <= 0.0:
        return x if res is None else x + res
    if x.dtype != torch.float32 or x.numel() % 4 or (res is not None and (res.shape != x.shape or res.dtype != x.dtype)):
        import torch.nn.functional as F

        y = F.dropout(F.dropout(x, p1, True), p2, True)
        return y if res is None else y + res
    return _Dropout2Add.apply(x, res, float(p1), float(p2))


def sin_add(x, act_slope=None):
    """sin(x) + x; with ``act_slope`` also returns bf16(LeakyReLU(result)) -- the operand of the streaming transposed
    convolution (bf16 mode)."""
    if act_slope is not None:
        return _SinAddAct.apply(x, float(act_slope))
    return _SinAdd.apply(x)


def upsample_weights(w, s, bias=None):
    """(Cin, Cout, 2s) transposed-conv weight -> bf16 polyphase images; with ``bias`` a third element: the bias repeated
    over the s phases, as the wide layers' contraction takes it (upsample_forward otherwise repeats it per call -- one
    more launch per stage).  Narrow layers (csrc/upsample.hip): (linear,
    permuted) matrices (s*Cout, 2*Cin), row (r, co), column (j, ci) = w[ci, co, r + j*s], the permuted copy ordered for
    16-byte stores.  Wide layers (csrc/cconv.hip): (tap-major (2, s*Cout, Cin), None)."""
    Cin, Cout, K = w.shape
    assert K % s == 0
    taps = K // s
    wb = ops_bf16.to_bf16(w) if w.numel() % 8 == 0 else w.detach().to(torch.bfloat16)  # one cast, then bf16 re-layouts
    wp = None
    extra = () if bias is None else (bias.detach().repeat(s),)
    if taps != 2:  # (the fused dual-path stages of the generator: J = 4 taps at stride 2) tap-major image only
        return (wb.view(Cin, Cout, taps, s).permute(2, 3, 1, 0).reshape(taps, s * Cout, Cin).contiguous(), None) + extra
    if Cout % 32 == 0 and (Cin, Cout, s) in ((128, 64, 2), (64, 32, 2)):
        wl = wb.view(Cin, Cout, 2, s).permute(3, 1, 2, 0).reshape(s * Cout, 2 * Cin)
        wp = wl.view(s, Cout // 32, 4, 2, 4, 2 * Cin).permute(0, 1, 3, 2, 4, 5).reshape(s * Cout, 2 * Cin)
        return (wl, wp) + extra
    # wide layers: the tap-major image (2, s*Cout, Cin) of csrc/cconv.hip
    return (wb.view(Cin, Cout, 2, s).permute(2, 3, 1, 0).reshape(2, s * Cout, Cin).contiguous(), None) + extra


_up_wcache = {}


def upsample_forward(act, w, bias, s, res=None, out_bf16=False, in_slope=1.0, prepared=None):
    """Polyphase CausalConvTranspose1d (kernel 2s, stride s) on a bf16 (B, T, Cin) input: the HBM-streaming kernel for the
    narrow layers, the two-segment bf16 contraction for the wide ones.  Returns (B, T*s, Cout) or None if unsupported.
    ``prepared``: the result of upsample_weights(w, s) when the caller keeps it (inference: weights are constants);
    parameters are cached by version, freshly computed weights (weight norm in training) are re-laid per call."""
    B, T, Cin = act.shape
    Cout = w.shape[1]
    if w.shape[2] % s or w.shape[2] < 2 * s or act.dtype != torch.bfloat16 or Cin % 8 or (s * Cout) % 8:
        return None
    taps = w.shape[2] // s
    if prepared is None and isinstance(w, torch.nn.Parameter):
        hit = _up_wcache.get(id(w))
        if hit is not None and hit[2]() is w and hit[0] == (w._version, w.data_ptr(), s):  # see ops_bf16.bf16_weight
            prepared = hit[1]
        else:
            prepared = upsample_weights(w, s)
            _up_wcache[id(w)] = ((w._version, w.data_ptr(), s), prepared, weakref.ref(w))
    prep = prepared if prepared is not None else upsample_weights(w, s)
    wl, wp = prep[0], prep[1]
    out = torch.empty((B, T * s, Cout), device=act.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    r = _c(res) if res is not None else None
    if wp is not None and (Cin, Cout, s) in ((128, 64, 2), (64, 32, 2)) and (r is None or r.dtype == out.dtype):
        rc = lib().kantts_upsample_stream(ptr(act), ptr(wp), ptr(bias, torch.float32), ptr(r), ptr(out), B, T, Cin, Cout, s,
                                          float(in_slope), int(out_bf16), stream())
        if rc == 0:
            return out
        if rc != -2:
            check(rc, "upsample_stream")
    if in_slope != 1.0 or (r is not None and r.dtype != torch.float32) or wp is not None:
        return None
    brep = prep[2] if len(prep) > 2 else (bias.repeat(s) if bias is not None else None)
    # polyphase form = a ``taps``-tap convolution onto s*Cout channels (tap j reads token t - j)
    if not cconv(act, wl, out=None if out_bf16 else out, out_bf=out if out_bf16 else None, B=B, Tsrc=T, Tdst=T, groups=1,
                 CR=Cin, NG=s * Cout, K=taps, in_mul=1, in_add=0, in_kstep=-1, in_div=1, phases=1, bias=brep, res=r):
        return None
    return out


# ================================================================================================
# MAS alignment path
# ================================================================================================
def mas_width1(attn, in_lens, out_lens):
    """Hard monotonic alignment of a soft attention map, entirely on the device.
    attn: (B, 1, T_mel, T_text) or (B, T_mel, T_text) float32; in_lens / out_lens: (B,) integer tensors.
    Returns the 0/1 map with attn's shape (zeros outside each utterance's (out_len x in_len) corner)."""
    a = _c(attn.detach())
    B, To, Ti = a.shape[0], a.shape[-2], a.shape[-1]
    opt = torch.empty_like(a)
    ws = torch.empty((B, To, Ti), device=a.device, dtype=torch.float32)
    il = in_lens.to(device=a.device, dtype=torch.int32).contiguous()
    ol = out_lens.to(device=a.device, dtype=torch.int32).contiguous()
    check(lib().kantts_mas_width1(ptr(a, torch.float32), ptr(il), ptr(ol), ptr(opt), ptr(ws), B, To, Ti, stream()),
          "mas_width1")
    return opt


class _AlignAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, prior, in_lens_i32):
        q, k = _c(q), _c(k)
        B, T1, C = q.shape
        T2 = k.shape[1]
        prior = None if prior is None else _c(prior.to(torch.float32))
        logprob = torch.empty((B, 1, T1, T2), device=q.device, dtype=torch.float32)
        soft = torch.empty_like(logprob)
        check(lib().kantts_align_attn_fwd(ptr(q, torch.float32), ptr(k, torch.float32), ptr(prior), ptr(in_lens_i32),
                                          ptr(logprob), ptr(soft), B, T1, T2, C, stream()), "align_attn_fwd")
        ctx.save_for_backward(q, k, logprob, soft)
        ctx.prior = prior
        ctx.mark_non_differentiable(in_lens_i32)
        return soft, logprob

    @staticmethod
    def backward(ctx, d_soft, d_logprob):
        q, k, logprob, soft = ctx.saved_tensors
        B, T1, C = q.shape
        T2 = k.shape[1]
        d_soft = None if d_soft is None else _c(d_soft)
        d_logprob = None if d_logprob is None else _c(d_logprob)
        g = torch.empty((B, T1, T2), device=q.device, dtype=torch.float32)
        dq, dk = torch.empty_like(q), torch.empty_like(k)
        check(lib().kantts_align_attn_bwd(ptr(q), ptr(k), ptr(ctx.prior), ptr(logprob), ptr(soft), ptr(d_logprob),
                                          ptr(d_soft), ptr(g), ptr(dq), ptr(dk), B, T1, T2, C, stream()), "align_attn_bwd")
        return dq, dk, None, None


def align_attention(q, k, prior, in_lens_i32):
    """q (B, T_mel, C), k (B, T_text, C) -> (attn_soft, attn_logprob), each (B, 1, T_mel, T_text)."""
    return _AlignAttention.apply(q, k, prior, in_lens_i32)
