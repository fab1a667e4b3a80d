"""Whole-training-step hipGraph capture for the SAM-BERT hot path.

The reference's Sambert_Trainer.train_step (kantts/train/trainer.py:898-1005) issues ~700 tiny
kernels and >= B+12 host syncs per step; on MI355X the step is launch/host bound, not math bound.
Here forward + losses + backward + gradient packing + clip + Adam are captured ONCE into a hipGraph
(torch.cuda.CUDAGraph drives hipStreamBeginCapture) and replayed per step:
  * band width, dropout offset, lr and step count live in device memory (no value is frozen into the
    captured kernel arguments that must change between steps);
  * inputs are static device buffers (`load_batch` copies a new batch of the same shape in place);
  * data-parallel (world_size > 1): RCCL calls are not captured.  The step is captured as a CHAIN OF GRAPH SEGMENTS cut
    inside backward where a gradient bucket becomes complete (the arena's post-accumulate hooks, the same partition the
    eager path uses): segment 0 = forward + backward down to the last bucket (+ its packing), segment k = backward down
    to the next bucket, last segment = average + clip + Adam.  The host replays segment k, issues the asynchronous
    all-reduce of the bucket it packed (it waits for the segment on the collective's own stream) and replays segment
    k + 1 at once -- the exchange of bucket k runs beside the backward of the earlier layers, which is what the
    reference gets from DDP's hooks (kantts/models/__init__.py:71-84,118-121).  KANTTS_DP_SEGMENTS=0 (or any failure of
    the segmented capture) selects the two-graph form: backward | bucketed all-reduce | update, nothing overlapped.
"""
import os

import torch

from kantts._hip import ops, rng_state as _rng_state


class GraphedSambertStep:
    def __init__(self, net, optimizer, scheduler, mel_criterion, prosody_criterion, batch, warmup=3,
                 overlap_wgrad=True, group_wgrads=True, band_width=None, force_wide=False, mas_criteria=None):
        """``band_width``: x_band_width of ``batch`` when the caller already knows it on the host (band_width_of on the
        batch before its upload); None: read from the device batch here, once.  ``force_wide``: capture the wide-band form
        (the five-launch decoder chain, any band width) whatever this batch's band is -- data parallel: the form must be
        the same on every rank, and another rank's batch needs it (train/trainer.py::_agree_on_graph)."""
        self._band_width_arg = band_width
        self._force_wide = bool(force_wide)
        # Monotonic-Alignment-Search step (sambert_16k_MAS*.yaml; reference trainer.py:871-884, 970-983): the two alignment
        # criteria {"AttentionCTCLoss": ..., "AttentionBinarizationLoss": ...} join the objective.  Capturable since round 6:
        # the CTC term is one launch with device-side lengths (csrc/ctc.hip).  The durations come out of the alignment ON THE
        # DEVICE, so the host cannot promise a band class: the wide form is captured.  The KL term's warm-up ratio depends on
        # the epoch: it is a device scalar the host refreshes before a replay (``set_epoch``).
        self.mas_criteria = mas_criteria
        if mas_criteria is not None:
            self._force_wide = True
        # weight gradients leave the critical path: fp32-mode ones as parallel branches of the captured graph
        # (ops._WgradOverlap), bf16-mode ones recorded and issued grouped by shape before the optimizer
        # (kantts._hip.deferred_tn); the variance predictors run as a side branch (ops.side_branch).  These switches only
        # shape what is CAPTURED: they are put back when the constructor returns, so eager code that runs later in the same
        # process (evaluation, tests reading p.grad right after backward()) never sees deferred gradients
        from kantts._hip import deferred_tn

        prev = (ops.wgrad_overlap.enabled, deferred_tn.enabled, ops.side_branch.enabled)
        # Deferred weight gradients hand autograd a buffer that is only filled when the group is flushed.  That is sound
        # as long as every parameter receives exactly ONE gradient per backward into an empty .grad (zero_grad(set_to_none)
        # before each backward, which this class does): a tied parameter would have autograd add a still-empty buffer.
        names = [n for n, _ in net.named_parameters(remove_duplicate=False)]
        if len(names) != len(list(net.parameters())):
            raise NotImplementedError("GraphedSambertStep defers weight gradients; tied parameters are not supported")
        ops.wgrad_overlap.enable(overlap_wgrad, group_wgrads=group_wgrads)
        try:
            self._build(net, optimizer, scheduler, mel_criterion, prosody_criterion, batch, warmup)
        finally:
            ops.wgrad_overlap.join()
            ops.wgrad_overlap.enabled, deferred_tn.enabled, ops.side_branch.enabled = prev

    def _build(self, net, optimizer, scheduler, mel_criterion, prosody_criterion, batch, warmup):
        self.net, self.optimizer, self.scheduler = net, optimizer, scheduler
        self.mel_criterion, self.prosody_criterion = mel_criterion, prosody_criterion
        self.batch = {k: v.clone() for k, v in batch.items()}
        self.device = next(net.parameters()).device
        net.device_band_width = True
        # The band width itself stays in device memory, but the SHAPE of the captured step depends on it: up to 16 the
        # decoder blocks are one launch each (csrc/pnca_block.hip), above that the five-launch chain.  The capture is made
        # for the class this batch is in; load_batch() refuses a batch of the other class (the trainer keys its graph cache
        # on it, train/trainer.py).
        from kantts._hip import ops_bf16
        from kantts.models.sambert.kantts_sambert import band_width_of

        self._r = net.mel_decoder.r
        bw = self._band_width_arg
        if self.mas_criteria is not None:
            bw = ops_bf16.PB_MAX_BAND + 1  # unknown on the host (device-side alignment): the wide form
            self.kl_ratio = torch.zeros((), device=self.device, dtype=torch.float32)
            self.set_epoch(0)
        elif bw is None:
            bw = band_width_of(self.batch["duration_targets"], self.batch["input_lengths"], self._r)
        self.narrow_band = bw <= ops_bf16.PB_MAX_BAND and not self._force_wide
        self._bound = ops_bf16.PB_MAX_BAND if self.narrow_band else None
        self.distributed = optimizer.arena.world_size > 1 or getattr(optimizer.arena, "force_exchange", False)
        # collectives are not captured: the exchange sits between the two graph halves (bucketed, asynchronous), so
        # the hook-driven overlap of the eager path is switched off for this optimizer
        optimizer.arena.overlap = False
        optimizer.enable_device_state()  # idempotent: one [lr, step] tensor shared by every captured shape
        # parameter gradients written straight into the gradient arena: recorded in the first warm-up step, in force
        # from the second (train/optim.py: ParamArena.enable_direct_grads)
        optimizer.arena.enable_direct_grads()
        self.loss = None
        # the warm-up steps exist only to populate allocator pools / lazy kernel state before capture: weights, Adam
        # moments, step counters and the dropout offset are put back afterwards, so a new batch shape costs exactly one
        # optimizer update (the first replay), like the eager path and the reference
        snap = optimizer.snapshot()
        rng_snap = _rng_state(self.device).clone() if self.device.type == "cuda" else None
        # ONE stream for the warm-up and for the capture: autograd replays a parameter's AccumulateGrad node on the stream
        # that was current when the node was created, and nodes created during the warm-up are still alive at capture
        # time (the loss tensor holds the graph).  With a different warm-up stream every gradient accumulation of the
        # captured step sat on a foreign stream, forked from and joined back into the capture (autograd's "AccumulateGrad
        # node's stream does not match" warning) -- and a capture cut in the middle of backward could not end.
        pr = os.environ.get("KANTTS_MAIN_PRIORITY")  # experiment switch: priority of the capture (critical-path) stream
        self._cap_stream = torch.cuda.Stream(priority=int(pr)) if pr else torch.cuda.Stream()
        self._cap_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._cap_stream):
            for _ in range(warmup):
                self._eager_step()
        torch.cuda.current_stream().wait_stream(self._cap_stream)
        torch.cuda.synchronize()
        self.loss = None
        optimizer.restore(snap)
        if rng_snap is not None:
            _rng_state(self.device).copy_(rng_snap)
        self.graph_a = torch.cuda.CUDAGraph()
        self.graph_b = None
        self.segments = None
        self.skip_exchange = False  # measurement only (bench.py: exposed all-reduce time = step with - step without)
        optimizer.zero_grad(set_to_none=True)
        if self.distributed and os.environ.get("KANTTS_DP_SEGMENTS", "1") != "0":
            import sys

            from kantts.train.segments import all_ranks_agree

            try:
                self._capture_segments()
            except Exception as exc:  # the two-graph form below is the proven fallback
                print("[GraphedSambertStep] segmented data-parallel capture failed (%s: %s); capturing the two-graph form"
                      % (type(exc).__name__, str(exc)[:300]), file=sys.stderr)
                self._abort_capture()
                self.segments = None
            # the two forms exchange different message sizes in a different order: one form on EVERY rank
            if not all_ranks_agree(self.segments is not None, self.device) and self.segments is not None:
                print("[GraphedSambertStep] another rank could not capture the segmented form; capturing the two-graph form "
                      "here as well", file=sys.stderr)
                self.segments = self._seg = None
            if self.segments is None:
                optimizer.restore(snap)
                if rng_snap is not None:
                    _rng_state(self.device).copy_(rng_snap)
                optimizer.zero_grad(set_to_none=True)
        if self.segments is not None:
            pass
        elif not self.distributed:
            with torch.cuda.graph(self.graph_a, capture_error_mode="thread_local", stream=self._cap_stream):
                self._forward_backward()
                self._apply()
        else:
            # thread_local: the RCCL watchdog thread polls events while this thread captures
            with torch.cuda.graph(self.graph_a, capture_error_mode="thread_local", stream=self._cap_stream):
                self._forward_backward()
                ops.wgrad_overlap.join()
                optimizer.arena.pack_grads()
            self.graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_b, pool=self.graph_a.pool(), capture_error_mode="thread_local",
                                  stream=self._cap_stream):
                self._apply(packed=True)
        optimizer._step = snap["step"]  # capturing ran step()'s host code once without running its kernels

    # ---- data-parallel: one graph per gradient bucket (train/segments.py) -----------------------------------------
    def _capture_segments(self):
        from kantts.train.segments import SegmentedCapture

        self._seg = SegmentedCapture([self.optimizer.arena], self._cap_stream)
        self._seg.capture(lambda: (self._forward_backward(), self._apply()))
        self.segments = self._seg

    def _abort_capture(self):
        if getattr(self, "_seg", None) is not None:
            self._seg.abort()
            self._seg = None

    def _forward_backward(self):
        b = self.batch
        ops.advance_rng(self.device)
        self.optimizer.zero_grad(set_to_none=True)
        self.net.band_width_bound = self._bound
        try:
            res = self.net(**b)
        finally:
            self.net.band_width_bound = None
        from kantts.train.loss import sambert_loss_sum

        if self.mas_criteria is None:
            self.loss, self.loss_terms = sambert_loss_sum(self.mel_criterion, self.prosody_criterion, b, res)
        else:
            total, terms = sambert_loss_sum(self.mel_criterion, self.prosody_criterion, b, res,
                                            prosody_lengths=res["valid_inter_lengths"])
            ctc = self.mas_criteria["AttentionCTCLoss"](res["attn_logprob"], b["input_lengths"], b["output_lengths"])
            # the criterion at "fully warmed up" (ratio 1) times the device-side ratio of the current epoch
            kl = self.mas_criteria["AttentionBinarizationLoss"](10 ** 9, res["attn_hard"], res["attn_soft"]) * self.kl_ratio
            self.loss = total + ctc + kl
            self.loss_terms = dict(terms, attn_ctc_loss=ctc.detach(), attn_kl_loss=kl.detach())
        self.loss.backward()

    def _apply(self, packed=False):
        self.optimizer.step(packed=packed) if packed else self.optimizer.step()

    def _eager_step(self):
        self._forward_backward()
        self.optimizer.step()

    def set_epoch(self, epoch):
        """MAS step: refresh the device-side warm-up ratio of the attention-binarisation term (reference loss.py:474-478)."""
        if self.mas_criteria is None:
            return
        c = self.mas_criteria["AttentionBinarizationLoss"]
        ratio = 0.0 if epoch < c.start_epoch else min(1.0, (epoch - c.start_epoch) / c.warmup_epoch)
        self.kl_ratio.fill_(float(ratio))

    def load_batch(self, batch, band_width=None):
        """Copy a new batch of the captured shape into the static buffers.  ``band_width``: its x_band_width if the caller
        knows it on the host; else it is read from ``batch`` (a device synchronisation when the batch is on the device)."""
        from kantts._hip import ops_bf16
        from kantts.models.sambert.kantts_sambert import band_width_of

        if band_width is None and not self.narrow_band:
            band_width = 0  # the wide form serves every band width: nothing to check (MAS batches carry no durations)
        if band_width is None:
            band_width = band_width_of(batch["duration_targets"], batch["input_lengths"], self._r)
        if self.narrow_band and band_width > ops_bf16.PB_MAX_BAND:  # (the wide form serves every band width)
            raise ValueError("this step was captured for band widths <= %d, the batch has %d: capture another step for it"
                             % (ops_bf16.PB_MAX_BAND, band_width))
        for k, v in batch.items():
            self.batch[k].copy_(v, non_blocking=True)

    def __call__(self):
        """One training step; returns the (device) loss tensor of this step."""
        if self.segments is not None:
            self.segments.skip_exchange = self.skip_exchange
            self.segments.replay()
        else:
            self.graph_a.replay()
            if self.graph_b is not None:
                if not self.skip_exchange:
                    self.optimizer.arena.all_reduce_grads()
                self.graph_b.replay()
        self.optimizer._step += 1  # the device-side count advances inside the graph; mirror it on the host
        self.scheduler.step()
        self.optimizer.sync_lr()
        return self.loss
