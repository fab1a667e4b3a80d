"""Flat parameter arena + fused Adam for the MI355X training step.

The reference steps ``torch.optim.Adam`` over ~400 small tensors after
``torch.nn.utils.clip_grad_norm_`` (kantts/train/trainer.py:997-1004) and lets DDP all-reduce 25 MB
buckets (kantts/models/__init__.py:118-121).  Here every trainable parameter of a module is a view
into ONE contiguous fp32 buffer; gradients are packed into a mirror buffer after backward, so that
  * the global gradient norm is one reduction kernel whose result stays in device memory,
  * clip + Adam is one elementwise kernel over the arena (kantts_adam_step),
  * data-parallel training exchanges the gradient arena with a handful of large RCCL all-reduces
    (xGMI is per-link bound: few large messages, not hundreds of small ones).
``ArenaAdam`` keeps ``torch.optim.Optimizer`` semantics (param_groups / state_dict layout of Adam:
step, exp_avg, exp_avg_sq) so LR schedulers and reference checkpoints keep working.
"""
import torch
import torch.distributed as dist

from kantts._hip import ops


class ParamArena:
    def __init__(self, module, bf16_shadow=False):
        self.module = module
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("module has no trainable parameters")
        dev = self.params[0].device
        # every parameter starts on a 256-byte boundary: the GEMM kernels stage weights with 16-byte vector
        # loads, which an unpadded pack would break after the first odd-sized tensor (e.g. a (1,) bias)
        self.align = 64
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + self.align - 1) // self.align * self.align
        self.numel = off
        self.flat = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
        self.grad_views = [self.grad[o:o + p.numel()].view(p.shape) for p, o in zip(self.params, self.offsets)]
        self._zero_cache = {}
        self.world_size = 1
        self.force_exchange = False
        self.bucket_ready = None  # set while a data-parallel step is being captured in segments (train/graph_step.py)
        self.capture_stream = None
        self.n_buckets = 4
        self.flat_bf16 = None
        self._weights_version = 0  # bumped whenever the fp32 master weights change (the shadow_stale setter)
        self._shadow_stale = False
        self._wn_table = None
        self._direct, self._plan = None, None  # None / "record" / "on": enable_direct_grads()
        if bf16_shadow:
            self._build_shadow()

    @property
    def shadow_stale(self):
        return self._shadow_stale

    @shadow_stale.setter
    def shadow_stale(self, value):
        """Every writer of the fp32 master (optimizer step, restore, broadcast, load_state_dict) sets this to True: the
        version count also tells the weight-norm images (build_weight_norm_images) that they are out of date."""
        self._shadow_stale = bool(value)
        if value:
            self._weights_version += 1
            ops.weights_epoch[0] += 1  # cached re-layouts of any weight (ops._cached_pack) are out of date

    # ------------------------------------------------------------------------------------------ direct gradients
    def enable_direct_grads(self):
        """Let the layers' backward kernels write parameter gradients straight into this arena's gradient buffer.

        The gradient accumulators of a backward pass come from ops.zero_pool in a fixed order (same model, same mode).  The
        next backward pass is RECORDED: which of those requests ended up as which parameter's ``.grad``.  From the pass
        after it, exactly those requests are served from the parameter's own range of ``self.grad`` (zeroed per step like
        the pool), autograd adopts the view as ``p.grad``, and pack_grads() finds every gradient in place: the ~7
        multi-tensor copy launches (130 us, the serial tail of a captured SAM-BERT step: profiles/r04_runH) disappear.
        Nothing depends on the order staying the same: pack_grads() copies whatever is not where it belongs.  Only for a
        process with ONE arena optimizer (the pool's call counter restarts at every zero_grad of any of them)."""
        if self._direct is None and not __import__("os").environ.get("KANTTS_NO_DIRECT_GRADS"):
            self._direct = "record"

    def _direct_begin(self):
        """The plan is only in force between this arena's zero_grad and its step: another arena's backward pass in between
        (a second model in the process) must not be handed ranges of this gradient buffer."""
        if self._direct == "record":
            ops.zero_pool.plan, ops.zero_pool.log = None, []
        elif self._direct == "on":
            self.grad.zero_()
            ops.zero_pool.plan, ops.zero_pool.log = self._plan, None

    def _direct_end(self):
        if self._direct == "record":
            self._direct_finish_recording()
        elif self._direct == "on" and ops.zero_pool.plan is self._plan:
            ops.zero_pool.plan = None

    def _direct_finish_recording(self):
        log, ops.zero_pool.log = ops.zero_pool.log, None
        if not log:
            return
        where = {ptr_: (idx, n) for idx, ptr_, n in log}
        plan = {}
        for i, p in enumerate(self.params):
            g = p.grad
            if g is None or not g.is_contiguous() or g.dtype != torch.float32:
                continue
            hit = where.get(g.data_ptr())
            if hit is not None and hit[1] == p.numel():
                plan[hit[0]] = self.grad_views[i]
        self._plan = plan
        self._direct = "on"

    # ------------------------------------------------------------------------------------------ weight-norm images
    def build_weight_norm_images(self):
        """HiFi-GAN: the effective weight w = g * v / ||v|| of EVERY weight-normed convolution of the module
        (kantts/models/hifigan/layers.py:29,67; hifigan.py:224,332 of the reference re-parametrise per layer and per
        forward pass through torch's weight_norm hook) in the convolution kernels' tap-major layout -- fp32 (K, Cout, Cin_g)
        plus the two bf16 operand images of csrc/cconv.hip -- rebuilt by ONE table-driven launch per optimizer step
        (kantts_weight_norm_table) from the flat parameter arena, instead of one launch per layer and forward pass
        (round 3: 6 700 launches of 15 us = 7.4 % of the kernel time of the benchmark).  The per-layer autograd node
        (ops._WeightNormImage) hands out views of these buffers and keeps the per-layer backward.  Layers whose input
        channel count is not a multiple of 4 (1-3 channel first layers) and transposed convolutions keep the per-layer
        path."""
        import torch.nn as nn

        index = {id(p): i for i, p in enumerate(self.params)}
        rows_tab, layers = [], []
        w_off = wf_off = wd_off = row0 = 0
        al = self.align
        for m in self.module.modules():
            g, v = getattr(m, "weight_g", None), getattr(m, "weight_v", None)
            if g is None or v is None or id(g) not in index or id(v) not in index:
                continue
            if isinstance(m, (nn.ConvTranspose1d, nn.ConvTranspose2d)):
                continue
            shape = tuple(v.shape[:3]) if (v.dim() == 4 and v.shape[3] == 1) else tuple(v.shape)
            if len(shape) != 3 or shape[1] % 4:
                continue
            cout, cin, k = shape
            groups = int(getattr(m, "groups", 1))
            n = cout * cin * k
            img = cin % 8 == 0 and (cout // groups) % 8 == 0
            rows_tab.append([self.offsets[index[id(v)]], self.offsets[index[id(g)]], w_off, wf_off if img else -1,
                             wd_off if img else -1, cout | (cin << 32), k | (groups << 32), row0])
            layers.append((m, w_off, wf_off if img else -1, wd_off if img else -1, (k, cout, cin), groups))
            pad = (n + al - 1) // al * al
            w_off += pad
            if img:
                wf_off += pad
                wd_off += pad
            row0 += (cout + 7) // 8  # the launch has one workgroup per tile of 8 output rows
        if not layers:
            return 0
        dev = self.flat.device
        self._wn_tiles = []    # 8-row tiles per table entry
        self._wn_pending = []  # (table entry, weight gradient) of this backward pass, see defer_weight_norm_backward
        self._wn_extra = []    # the same for a layer's second, third ... application since zero_grad (accumulated)
        self._wn_seen = set()  # table entries whose slot the table launch fills / has filled since zero_grad
        self._wn_defer = False
        self._index_of = index
        self._wn_w = torch.zeros(max(w_off, 8), device=dev, dtype=torch.float32)
        self._wn_wf = torch.zeros(max(wf_off, 8), device=dev, dtype=torch.bfloat16)
        self._wn_wd = torch.zeros(max(wd_off, 8), device=dev, dtype=torch.bfloat16)
        self._wn_table = torch.tensor(rows_tab, dtype=torch.int64, device=dev)
        self._wn_rows = row0
        self._wn_version = -1
        self._wn_params = [q for m, *_ in layers for q in (m.weight_v, m.weight_g)]
        self._wn_sig = None
        import weakref

        ref = weakref.ref(self)
        for m, wo, fo, do, (k, cout, cin), groups in layers:
            n = k * cout * cin
            m._kantts_wn = (ref, self._wn_w[wo:wo + n].view(k, cout, cin),
                            None if fo < 0 else self._wn_wf[fo:fo + n].view(k, cout, cin),
                            None if do < 0 else self._wn_wd[do:do + n].view(k, groups, cin, cout // groups), groups,
                            len(self._wn_tiles))
            self._wn_tiles.append((cout + 7) // 8)
        # once per forward of the WHOLE module, before it forks its branch streams (every branch reads the images)
        def _pre(m, a):  # (a pre-hook's return value replaces the module's input: return None)
            self.refresh_weight_norm_images()

        self.module.register_forward_pre_hook(_pre)
        self.module.register_load_state_dict_post_hook(lambda m, keys: self.mark_shadow_stale())
        return len(layers)

    # ---- backward of the reparametrisation, deferred to ONE launch per network (kantts_weight_norm_table_bwd) ----------
    def grad_slot(self, p):
        """A fresh view of parameter ``p``'s range of the gradient arena (a new tensor object: autograd keeps it as
        ``p.grad`` without copying; packing it into the arena later is a copy onto itself)."""
        i = self._index_of[id(p)]
        return self.grad[self.offsets[i]:self.offsets[i] + p.numel()].view(p.shape)

    def defer_weight_norm_backward(self, entry, dw):
        """Record (table entry, tap-major weight gradient) if an optimizer step of this arena will follow (zero_grad armed
        it) -- else the caller runs the per-layer kernel at once (code that reads .grad right after backward())."""
        if not self._wn_defer or self._wn_table is None:
            return False
        entry = int(entry)
        if not self._wn_pending and not self._wn_extra:
            ops.wn_pending_arenas.append(self)
        if entry in self._wn_seen:
            # the layer was applied more than once between zero_grad and step (the two-pass discriminator loss, gradient
            # accumulation over several backward passes, a shared module).  The table kernel ASSIGNS a layer's slot, so a
            # second entry for the same layer cannot ride on it: its reparametrisation backward runs per layer at the flush
            # and is ADDED to the slot the first application filled (the caller hands autograd no gradient for this one)
            self._wn_extra.append((entry, dw))
            return "extra"
        self._wn_seen.add(entry)
        self._wn_pending.append((entry, dw))
        return True

    def flush_weight_norm_backward(self):
        import ctypes

        from kantts._hip import WN_BWD_MAX, WnBwdArgs, check, lib, ptr, stream

        pend, self._wn_pending = self._wn_pending, []
        extra, self._wn_extra = self._wn_extra, []
        for s0 in range(0, len(pend), WN_BWD_MAX):
            chunk = pend[s0:s0 + WN_BWD_MAX]
            a = WnBwdArgs()
            t = 0
            for l, (entry, dw) in enumerate(chunk):
                a.dw[l], a.desc[l], a.tile0[l] = ptr(dw, torch.float32), entry, t
                t += self._wn_tiles[entry]
            a.tile0[len(chunk)] = t
            a.nl = len(chunk)
            check(lib().kantts_weight_norm_table_bwd(ptr(self.flat, torch.float32), ptr(self.grad, torch.float32),
                                                     ptr(self._wn_table), ctypes.byref(a), stream()), "weight_norm_table_bwd")
        for entry, _ in pend:
            # autograd was handed VIEWS of the slots the launch above has just filled.  AccumulateGrad normally adopts such a
            # view as ``p.grad``; if it copied instead (grad mode on, another reference to the view), ``p.grad`` is a copy of
            # the slot from BEFORE it was filled and pack_grads() would write that copy over the result: re-point it
            for q in self._wn_params[2 * entry:2 * entry + 2]:
                slot = self.grad_slot(q)
                if q.grad is not None and q.grad.data_ptr() != slot.data_ptr():
                    q.grad = slot
        for entry, dw in extra:
            v, g = self._wn_params[2 * entry:2 * entry + 2]
            v3 = v.detach().squeeze(-1) if v.dim() == 4 else v.detach()
            v3 = v3 if v3.is_contiguous() else v3.contiguous()
            cout, cin, k = v3.shape
            dv, dg = torch.empty_like(v3), torch.empty_like(g)
            check(lib().kantts_weight_norm_strided_bwd(ptr(dw, torch.float32), ptr(v3), ptr(g.detach()), ptr(dv), ptr(dg), cout,
                                                       cin, k, cin, 1, cout * cin, stream()), "weight_norm_strided_bwd")
            for q, d in ((v, dv), (g, dg)):
                slot = self.grad_slot(q)
                if q.grad is not None and q.grad.data_ptr() != slot.data_ptr():
                    q.grad = slot
                slot.add_(d.view(slot.shape))

    def weight_norm_images_fresh(self):
        return self._wn_table is not None and self._wn_version == self._weights_version

    def refresh_weight_norm_images(self, force=False):
        import os

        if self._wn_table is None or (os.environ.get("KANTTS_NO_WEIGHT_NORM_TABLE") and not force):  # (A/B switch)
            return False
        # in-place writes that did not go through the optimizer / load_state_dict (``with torch.no_grad(): p.copy_(..)``,
        # an initialiser applied after the arena was built) bump the tensors' version counters: once per forward of the
        # whole module that is cheap to look at
        sig = sum(q._version for q in self._wn_params)
        if sig != self._wn_sig:
            self._wn_sig = sig
            self._weights_version += 1
            # the images are rebuilt at the same addresses by a raw kernel: anything cached per weight image
            # (ops._cached_pack: block-diagonal packs of the grouped convolutions) must not outlive them
            ops.weights_epoch[0] += 1
        if not force and self._wn_version == self._weights_version:
            return False
        from kantts._hip import check, lib, ptr, stream

        check(lib().kantts_weight_norm_table(ptr(self.flat, torch.float32), ptr(self._wn_w, torch.float32),
                                             ptr(self._wn_wf, torch.bfloat16), ptr(self._wn_wd, torch.bfloat16),
                                             ptr(self._wn_table), int(self._wn_table.shape[0]), int(self._wn_rows),
                                             stream()), "weight_norm_table")
        self._wn_version = self._weights_version
        return True

    # ------------------------------------------------------------------------------------------ bf16 shadow
    def _build_shadow(self):
        """bf16 copy of every parameter at the same arena offset, plus tap-major (KT, N, Cin) copies of the Conv1d
        weights with KT > 1 -- the operand images of csrc/gemm_bf16.hip.  Refreshed by ``refresh_shadow`` (two launches)
        before every forward of the module (a forward pre-hook: also part of a captured training step), so the copies can
        never be stale whatever changed the fp32 master (Adam, load_state_dict, a broadcast, a test poking a weight)."""
        dev = self.flat.device
        self.flat_bf16 = torch.zeros(self.numel, device=dev, dtype=torch.bfloat16)
        rows, off = [], 0
        for p, o in zip(self.params, self.offsets):
            p._kantts_bf16 = self.flat_bf16[o:o + p.numel()].view(p.shape)
            if p.dim() == 3 and p.shape[2] > 1:
                n, cin, kt = p.shape
                rows.append((o, off, n, cin, kt, p, "_kantts_bf16_tap"))
                off += (p.numel() + self.align - 1) // self.align * self.align
        self.tap_bf16 = torch.zeros(max(off, 8), device=dev, dtype=torch.bfloat16)
        self._tap_blocks = 1
        tab = []
        for o, to, n, cin, kt, p, attr in rows:
            setattr(p, attr, self.tap_bf16[to:to + p.numel()].view(kt, n, cin))
            tab.append([o, to, n | (cin << 32), kt])  # kantts_tapmajor_desc: two int64 offsets + N, Cin, KT, pad (int32)
            self._tap_blocks = max(self._tap_blocks, min(64, (n * cin * kt + 1023) // 1024))
        self._tap_table = torch.tensor(tab, dtype=torch.int64, device=dev) if tab else None
        # fragment-major images of the feed-forward weights (csrc/ffn_pair.hip; kantts_fragmajor_desc = src_off, dst_off,
        # sr, sk (int64), R, K (int32)): forward images for every flagged pair of the supported shape and images of the
        # transposed weights (the backward operands), one per tap
        ftab, foff = [], 0
        self._frag_blocks = 1

        def image(p, src_off, R, K, sr, sk, attr, n_img=1):
            nonlocal foff
            base = foff
            for t in range(n_img):
                ftab.append([src_off + t, base + t * R * K, sr, sk, R | (K << 32)])
            foff += (n_img * R * K + self.align - 1) // self.align * self.align
            self._frag_blocks = max(self._frag_blocks, min(64, (R * K + 2047) // 2048))
            return (attr, p, base, n_img * R * K)

        views = []
        for p, o in zip(self.params, self.offsets):
            role = getattr(p, "_kantts_ffn_role", None)
            if role == "w1" and p.dim() == 3 and p.shape[0] == 1024 and p.shape[1] == 128 and p.shape[2] % 2 == 1:
                F_, C_, KT_ = p.shape
                views.append(image(p, o, F_, C_, C_ * KT_, KT_, "_kantts_frag", n_img=KT_))
                views.append(image(p, o, C_, F_, KT_, C_ * KT_, "_kantts_fragT", n_img=KT_))  # W1[tap]^T: rows tap*C + c
            elif role == "w2" and p.dim() == 3 and p.shape[0] == 128 and p.shape[1] == 1024 and p.shape[2] == 1:
                N_, F_, _ = p.shape
                views.append(image(p, o, N_, F_, F_, 1, "_kantts_frag"))
                views.append(image(p, o, F_, N_, 1, F_, "_kantts_fragT"))
            elif role == "lin" and p.dim() == 2 and p.shape[0] % 16 == 0 and p.shape[1] % 32 == 0:
                # nn.Linear weights the fused PNCA block launch streams (csrc/pnca_block.hip: w_x_qkv, fc_x, fc_h)
                N_, K_ = p.shape
                views.append(image(p, o, N_, K_, K_, 1, "_kantts_frag"))
                if N_ % 32 == 0 and K_ % 16 == 0:  # W^T: the A operand of the input gradient (pnca_block_bwd_kernel)
                    views.append(image(p, o, K_, N_, 1, K_, "_kantts_fragT"))
        self.frag_bf16 = torch.zeros(max(foff, 8), device=dev, dtype=torch.bfloat16)
        for attr, p, base, n in views:
            setattr(p, attr, self.frag_bf16[base:base + n])
        self._frag_table = torch.tensor(ftab, dtype=torch.int64, device=dev) if ftab else None
        import weakref

        ref = weakref.ref(self)
        for p in self.params:
            p._kantts_arena = ref  # ops_bf16.bf16_weight re-builds a stale shadow before handing it out
        self.shadow_stale = True
        self.refresh_shadow()
        # once per forward of the WHOLE module (also inside a captured step).  A sub-module called on its own after an
        # optimizer step / load_state_dict is covered by the staleness check in ops_bf16.bf16_weight; fp32 mode never reads
        # the images, so it does not pay for the three launches either (they are built on the first bf16-mode use).
        self.module.register_forward_pre_hook(lambda m, a: self.refresh_shadow(only_if_bf16=True))
        self.module.register_load_state_dict_post_hook(lambda m, keys: self.mark_shadow_stale())

    def mark_shadow_stale(self):
        """The fp32 master weights changed (optimizer step, load_state_dict, broadcast, restore)."""
        self.shadow_stale = True

    def refresh_shadow(self, only_if_bf16=False):
        if self.flat_bf16 is None:
            return
        from kantts._hip import check, get_precision, lib, ptr, stream

        if only_if_bf16 and get_precision() != "bf16":
            return
        self.shadow_stale = False

        # (three launches one after the other, ~49 us at the head of a step.  [round 6] Forked onto two side streams -- same
        # masters in, disjoint images out, joined before the first consumer -- they should cost the longest of them, ~21 us;
        # measured, the captured step got SLOWER: 5.69 / 5.67 / 5.66 ms side by side against 5.43 / 5.60 / 5.61 ms serial,
        # interleaved on one box (profiles/r06_runIM_image_refresh_side_streams_ab.log).  Serial stays.)
        check(lib().kantts_cast_f32_bf16(ptr(self.flat, torch.float32), ptr(self.flat_bf16, torch.bfloat16), self.numel,
                                         stream()), "cast_f32_bf16")
        if self._tap_table is not None:
            check(lib().kantts_tapmajor_bf16(ptr(self.flat, torch.float32), ptr(self.tap_bf16, torch.bfloat16),
                                             ptr(self._tap_table), int(self._tap_table.shape[0]), int(self._tap_blocks),
                                             stream()), "tapmajor_bf16")
        if self._frag_table is not None:
            check(lib().kantts_fragmajor_bf16(ptr(self.flat, torch.float32), ptr(self.frag_bf16, torch.bfloat16),
                                              ptr(self._frag_table), int(self._frag_table.shape[0]),
                                              int(self._frag_blocks), stream()), "fragmajor_bf16")

    def view_of(self, flat, i):
        p = self.params[i]
        return flat[self.offsets[i]:self.offsets[i] + p.numel()].view(p.shape)

    def enable_data_parallel(self, n_buckets=4, overlap=True):
        """Data-parallel training over torch.distributed (RCCL over xGMI on MI355X, gloo in the CPU tests).
        ``overlap``: the gradient arena is cut into ``n_buckets`` contiguous ranges; parameters were registered in
        forward order, so the LAST range is the first whose gradients are complete in backward.  A post-accumulate hook
        on every parameter counts its bucket down; when a bucket is complete its gradients are packed (one multi-tensor
        copy) and its all-reduce is issued asynchronously -- it runs on the communication stream while backward
        continues with the earlier layers (what the reference gets from DDP's 25 MB buckets,
        kantts/models/__init__.py:118-121).  ``ArenaAdam.step`` waits for the handles.  xGMI rings are per-link bound:
        few large messages (default 4 x ~12 MB for SAM-BERT), not hundreds of small ones."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.world_size = dist.get_world_size()
        self.n_buckets = n_buckets
        # replicas must start from identical weights (DDP broadcasts rank 0's copy at wrap time)
        dist.broadcast(self.flat, src=0)
        self.shadow_stale = True
        self.refresh_shadow(only_if_bf16=True)
        # KANTTS_DP_FORCE=1 (bench.py --rccl-world1): a world of ONE rank still runs the whole exchange path -- buckets,
        # hooks, graph segments, the collective itself -- so that a one-GPU box executes RCCL where the step does
        self.force_exchange = bool(__import__("os").environ.get("KANTTS_DP_FORCE")) and self.world_size == 1
        self.overlap = bool(overlap) and (self.world_size > 1 or self.force_exchange)
        if self.overlap:
            self._build_buckets()

    def _build_buckets(self):
        """Greedy cut of the arena into at most ``n_buckets`` contiguous parameter runs of about numel / n_buckets
        elements: a bucket is closed once its (aligned) extent reaches the target, so the ranges are disjoint and cover
        the arena by construction -- also when one parameter alone is larger than the target."""
        n = self.numel
        step = (n + self.n_buckets - 1) // self.n_buckets
        self.buckets = []
        self.bucket_of = [0] * len(self.params)
        cur = None
        for i, o in enumerate(self.offsets):
            if cur is None:
                cur = {"lo": o, "hi": o, "params": [], "pending": 0, "handle": None}
            cur["params"].append(i)
            self.bucket_of[i] = len(self.buckets)
            end = self.offsets[i + 1] if i + 1 < len(self.offsets) else n  # aligned start of the next parameter
            cur["hi"] = end
            if cur["hi"] - cur["lo"] >= step and len(self.buckets) + 1 < self.n_buckets:
                self.buckets.append(cur)
                cur = None
        if cur is not None:
            self.buckets.append(cur)
        if self.buckets:
            self.buckets[0]["lo"], self.buckets[-1]["hi"] = 0, n
        # the ranges partition [0, numel): every element is exchanged exactly once
        assert all(x["hi"] == y["lo"] for x, y in zip(self.buckets, self.buckets[1:])) and (
            not self.buckets or (self.buckets[0]["lo"] == 0 and self.buckets[-1]["hi"] == n))
        assert all(0 <= self.bucket_of[i] < len(self.buckets) for i in range(len(self.params)))
        self._active = False
        if not getattr(self, "_hooked", False):
            for i, p in enumerate(self.params):
                p.register_post_accumulate_grad_hook(lambda _p, i=i: self._on_grad(i))
            self._hooked = True

    def begin_step(self):
        """Arm the bucket counters for one backward pass (ArenaAdam.zero_grad calls this); weight-norm backward passes of
        this arena are deferred from here until the step (flush_weight_norm_backward).  An un-armed arena ignores
        gradient arrivals: the discriminators' parameters also receive gradients during the generator's backward, but
        those are discarded by the next zero_grad and must not be exchanged."""
        if self._wn_table is not None and not __import__("os").environ.get("KANTTS_NO_WEIGHT_NORM_TABLE"):
            self._wn_defer = True
            self._wn_seen.clear()
        if self._direct is not None:
            self._direct_begin()
        if not getattr(self, "overlap", False):
            return
        for b in self.buckets:
            b["pending"], b["handle"] = len(b["params"]), None
        self._active = True

    def _on_grad(self, i):
        if not self._active:
            return
        b = self.buckets[self.bucket_of[i]]
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch_bucket(b)

    def _launch_bucket(self, b):
        cap = self.capture_stream
        if cap is not None and torch.cuda.current_stream() != cap:
            # segmented capture: the hook may run with a foreign current stream (an AccumulateGrad node created before
            # the capture stream existed); pack and cut on the capture stream, after whatever that stream captured
            if torch.cuda.is_current_stream_capturing():
                cap.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cap):
                return self._launch_bucket_on_current_stream(b)
        return self._launch_bucket_on_current_stream(b)

    def _launch_bucket_on_current_stream(self, b):
        # the bucket's gradients were accumulated on whatever streams their layers ran on (the sub-discriminators and the
        # residual stacks of HiFi-GAN each have their own): the packing below must come after ALL of them, not only after
        # the stream whose hook completed the bucket.  Under capture that is a graph edge (a missing one lets the replayed
        # copy read half-written gradients); eagerly it is a stream wait.
        if self.bucket_ready is not None:
            ops.join_capturing_side_streams()
        elif torch.cuda.is_available() and self.grad.is_cuda:
            cur = torch.cuda.current_stream()
            for st in ops.helper_streams():
                if st != cur:
                    cur.wait_stream(st)
        ops.wgrad_overlap.join()  # weight gradients produced on the side stream must have landed before they are packed
        dst, src = [], []
        for i in b["params"]:
            p = self.params[i]
            g = p.grad
            if g is None:
                self.grad_views[i].zero_()
            elif g.data_ptr() != self.grad_views[i].data_ptr():  # (in place already: grad_slot / direct gradients)
                dst.append(self.grad_views[i])
                src.append(self._out_of_arena(g))
        if dst:
            torch._foreach_copy_(dst, src)
        if self.bucket_ready is not None:
            # hipGraph capture of a data-parallel step (GraphedSambertStep): the exchange is issued by the host between
            # the replays of two graph segments -- the segment that ends here has just packed this bucket
            b["handle"] = "captured"
            self.bucket_ready(b)
            return
        b["handle"] = self.exchange(b)

    def exchange(self, b):
        """Asynchronous all-reduce (sum) of one bucket's range of the gradient arena; returns the work handle."""
        return dist.all_reduce(self.grad[b["lo"]:b["hi"]], op=dist.ReduceOp.SUM, async_op=True)

    def finish_reduce(self):
        """Issue whatever was not triggered during backward (parameters without a gradient this step), wait for every
        bucket and average."""
        for b in self.buckets:
            if b["handle"] is None:
                self._launch_bucket(b)
        for b in self.buckets:
            if self.bucket_ready is None:  # (captured step: train/segments.py waits between the replays)
                b["handle"].wait()
            b["handle"] = None
        self._active = False
        self.grad.mul_(1.0 / self.world_size)
        return self.grad

    def _out_of_arena(self, g):
        """A gradient that lives in this arena's gradient buffer but NOT in its own slot (a direct-gradient plan applied
        to a changed call sequence) is moved out before anything is copied into the arena."""
        base = self.grad.data_ptr()
        if base <= g.data_ptr() < base + 4 * self.numel:
            return g.clone()
        return g

    def pack_grads(self):
        """Every ``p.grad`` into the (padded) gradient arena with one multi-tensor copy; gradients that are already views of
        their own slot (grad_slot, enable_direct_grads) are left alone."""
        if self._direct is not None:
            self._direct_end()
        dst, src = [], []
        for i, p in enumerate(self.params):
            g = p.grad
            if g is None:
                z = self._zero_cache.get(tuple(p.shape))
                if z is None:
                    z = torch.zeros(p.shape, device=self.grad.device, dtype=torch.float32)
                    self._zero_cache[tuple(p.shape)] = z
                g = z
            elif g.data_ptr() == self.grad_views[i].data_ptr():
                continue
            dst.append(self.grad_views[i])
            src.append(g)
        if dst:
            src = [self._out_of_arena(g) for g in src]  # (all clones before the first copy)
            torch._foreach_copy_(dst, src)
        return self.grad

    def all_reduce_grads(self):
        """Average the (already packed) gradient arena over the data-parallel group: ``n_buckets`` large asynchronous
        all-reduces (the non-overlapped form: between the two halves of a captured training step)."""
        if self.world_size <= 1 and not getattr(self, "force_exchange", False):
            return
        n = self.numel
        step = (n + self.n_buckets - 1) // self.n_buckets
        handles = []
        for s in range(0, n, step):
            handles.append(dist.all_reduce(self.grad[s:min(n, s + step)], op=dist.ReduceOp.SUM, async_op=True))
        for h in handles:
            h.wait()
        self.grad.mul_(1.0 / self.world_size)


class ArenaAdam(torch.optim.Optimizer):
    """torch.optim.Adam(amsgrad=False) over a ParamArena, with optional fused global-norm clipping."""

    def __init__(self, arena, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, **unused):
        if amsgrad:
            raise NotImplementedError("amsgrad")
        self.arena = arena
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False)
        super().__init__(arena.params, defaults)
        dev = arena.flat.device
        self.exp_avg = torch.zeros_like(arena.flat)
        self.exp_avg_sq = torch.zeros_like(arena.flat)
        self.gnorm_sq = torch.zeros((), device=dev, dtype=torch.float32)
        self._norm_ws = torch.zeros(1032, device=dev, dtype=torch.float32)  # fixed-order norm reduction (replica parity)
        self.max_grad_norm = 0.0
        self._step = 0
        self.dyn = None  # optional device tensor [lr, step] (hipGraph replay), see enable_device_state()
        if arena.flat.is_cuda:
            # gradient accumulators of one backward pass: all parameter gradients plus slack for temporaries
            ops.zero_pool.enable(arena.numel + arena.numel // 2 + (1 << 20), dev)
        for i, p in enumerate(arena.params):
            self.state[p] = {
                "step": torch.tensor(0.0),
                "exp_avg": arena.view_of(self.exp_avg, i),
                "exp_avg_sq": arena.view_of(self.exp_avg_sq, i),
            }

    def zero_grad(self, set_to_none=True):
        super().zero_grad(set_to_none=True)
        ops.zero_pool.reset()
        self.arena.begin_step()

    def enable_device_state(self):
        """Keep lr and the step count in device memory so that step() can be replayed from a hipGraph:
        the captured kernels read {lr, step} from ``self.dyn``; call sync_lr() after scheduler.step()."""
        if self.dyn is None:
            # allocated ONCE for the optimizer's lifetime: every captured graph (one per batch shape) holds this
            # tensor's address in its kernel arguments, so a second allocation would strand the older graphs
            self.dyn = torch.empty(2, device=self.arena.flat.device, dtype=torch.float32)
        self.dyn.copy_(torch.tensor([self.param_groups[0]["lr"], float(self._step)], dtype=torch.float32))
        return self.dyn

    def snapshot(self):
        """Copies of everything a step mutates (weights, moments, counters): GraphedSambertStep restores them after
        its capture warm-up so that warming up a new batch shape does not train on it."""
        return {"flat": self.arena.flat.clone(), "m": self.exp_avg.clone(), "v": self.exp_avg_sq.clone(),
                "step": self._step, "dyn": None if self.dyn is None else self.dyn.clone()}

    def restore(self, snap):
        self.arena.flat.copy_(snap["flat"])
        self.arena.shadow_stale = True
        self.exp_avg.copy_(snap["m"])
        self.exp_avg_sq.copy_(snap["v"])
        self._step = snap["step"]
        if self.dyn is not None and snap["dyn"] is not None:
            self.dyn.copy_(snap["dyn"])

    def sync_lr(self):
        if self.dyn is not None:
            self.dyn[0:1].fill_(float(self.param_groups[0]["lr"]))

    def set_grad_clip(self, max_norm):
        """Fold ``clip_grad_norm_(params, max_norm)`` into the step (norm never leaves the device)."""
        self.max_grad_norm = float(max_norm) if max_norm and max_norm > 0 else 0.0

    @torch.no_grad()
    def step(self, closure=None, packed=False):
        """packed=True: gradients are already packed (and all-reduced) in the arena."""
        if closure is not None:
            raise NotImplementedError("closure")
        group = self.param_groups[0]
        arena = self.arena
        ops.wgrad_overlap.join()  # weight gradients produced on the side stream (no-op unless enabled)
        arena._wn_defer = False
        if arena._direct is not None:
            arena._direct_end()  # (recording -> plan; the plan is out of force until the next zero_grad)
        if packed:
            g = arena.grad
        elif getattr(arena, "overlap", False) and arena._active:
            g = arena.finish_reduce()  # buckets were exchanged while backward was still running
        else:
            g = arena.pack_grads()
            arena.all_reduce_grads()
        gn = None
        if self.max_grad_norm > 0:
            ops.sumsq_into(g, self.gnorm_sq, self._norm_ws)
            gn = self.gnorm_sq
        self._step += 1
        if self.dyn is not None:
            self.dyn[1:2].add_(1.0)
        b1, b2 = group["betas"]
        ops.adam_step(arena.flat, g, self.exp_avg, self.exp_avg_sq, group["lr"], b1, b2, group["eps"],
                      group["weight_decay"], self._step, gnorm_sq=gn, max_norm=self.max_grad_norm, dyn=self.dyn)
        arena.shadow_stale = True  # the bf16 operand images are rebuilt by the next forward (or the next direct use)
        return None  # per-parameter "step" entries are materialised lazily in state_dict()

    def grad_norm(self):
        """Global gradient norm of the last clipped step (device scalar)."""
        return torch.sqrt(self.gnorm_sq)

    def state_dict(self):
        for st in self.state.values():
            st["step"] = torch.tensor(float(self._step))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        steps = []
        for i, p in enumerate(self.arena.params):
            st = self.state[p]
            m, v = self.arena.view_of(self.exp_avg, i), self.arena.view_of(self.exp_avg_sq, i)
            m.copy_(st["exp_avg"])
            v.copy_(st["exp_avg_sq"])
            st["exp_avg"], st["exp_avg_sq"] = m, v
            steps.append(int(float(st["step"])))
        self._step = max(steps) if steps else 0
