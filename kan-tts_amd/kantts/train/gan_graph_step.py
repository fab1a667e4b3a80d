"""Whole-step hipGraph capture of the HiFi-GAN training step.

A GAN step at batch 32 x 8192 is ~2000 kernel launches and ~1500 small tensor operations issued from Python over up to
eight streams (generator, five period and three scale discriminators, two phases: kantts/train/gan_step.py, reference
kantts/train/trainer.py:469-589).  Once the convolutions run on the bf16 MFMA kernels the host cannot issue the step as
fast as the device executes it.  Here both phases -- generator forward, the discriminators on generated and real audio,
the four losses, both backward passes, the three Adam updates -- are captured ONCE (torch.cuda.CUDAGraph drives
hipStreamBeginCapture; the per-discriminator / per-residual-stack streams of ops.parallel_branches become parallel
branches of the graph) and replayed per step:
  * inputs are static device buffers (``load_batch`` copies a new batch of the same shape in place);
  * learning rates and step counts of the three optimizers live in device memory (ArenaAdam.enable_device_state);
  * nothing in the step reads a value back to the host.
The step must have both phases active (``steps`` past both start thresholds) -- before that the eager
``gan_train_step`` runs.  Data-parallel replicas capture the step as a chain of graph segments cut where a gradient bucket
of the generator's / a discriminator's arena is complete, with the all-reduces issued between the replays
(kantts/train/segments.py): the step stays off the host, and a bucket's exchange runs beside the rest of its backward.
"""
import torch

from kantts.train.gan_step import gan_train_step
from kantts.train.optim import ArenaAdam


class _NoSched:
    def step(self):
        pass


class CaptureRefused(NotImplementedError):
    """The runtime refused to capture the step (an operation that is not permitted while capturing): the one RuntimeError
    of building a GraphedGanStep that the trainer answers with the eager step instead of propagating."""


class GraphedGanStep:
    def __init__(self, model, optimizer, scheduler, criterion, config, y, x, warmup=2, steps=10 ** 9):
        self.model, self.optimizer, self.scheduler, self.criterion, self.config = model, optimizer, scheduler, criterion, config
        self.opts = [optimizer["generator"]] + list(optimizer["discriminator"].values())
        self.scheds = [scheduler["generator"]] + list(scheduler["discriminator"].values())
        if not all(isinstance(o, ArenaAdam) for o in self.opts):
            raise NotImplementedError("GraphedGanStep needs the arena optimizers (hifigan_model_builder on a HIP device, Adam)")
        self.distributed = any(o.arena.world_size > 1 for o in self.opts)
        if steps <= config.get("discriminator_train_start_steps", 0) or steps < config.get("generator_train_start_steps", 0):
            raise ValueError("both phases must be active in a captured GAN step")
        if getattr(model["generator"], "nsf_enable", False):
            raise NotImplementedError("the NSF excitation draws host-seeded random numbers per step: eager step only")
        self.steps = steps
        self.y, self.x = y.clone(), x.clone()
        self._nosched = {"generator": _NoSched(), "discriminator": {k: _NoSched() for k in scheduler["discriminator"]}}
        overlap_before = [getattr(o.arena, "overlap", False) for o in self.opts]
        for o in self.opts:
            o.arena.overlap = False
            o.enable_device_state()
        try:
            self._build(warmup)
        finally:  # the eager step (a new batch shape's warm-up, a step no graph is ready for) keeps its own setting
            for o, ov in zip(self.opts, overlap_before):
                o.arena.overlap = ov

    def _build(self, warmup):
        model, optimizer, scheduler = self.model, self.optimizer, self.scheduler
        # warm-up (allocator pools, lazy kernel attributes) must not train: weights, moments and counters are put back
        snaps = [o.snapshot() for o in self.opts]
        bufs = self._module_buffers()
        buf_snap = [b.clone() for b in bufs]  # e.g. the spectral_norm power-iteration vectors of follow_official_norm
        # one stream for warm-up and capture (AccumulateGrad nodes replay on the stream they were created on)
        self._cap_stream = torch.cuda.Stream()
        self._cap_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._cap_stream):
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(self._cap_stream)
        torch.cuda.synchronize()
        for o, s in zip(self.opts, snaps):
            o.restore(s)
        with torch.no_grad():
            for b, v in zip(bufs, buf_snap):
                b.copy_(v)
        for o in self.opts:
            o.zero_grad(set_to_none=True)
        self.graph, self.segments = None, None
        if self.distributed:
            from kantts.train.segments import SegmentedCapture

            from kantts.train.segments import all_ranks_agree

            self.segments = SegmentedCapture([o.arena for o in self.opts], self._cap_stream)
            err = None
            try:
                self.out = self.segments.capture(self._eager)
            except Exception as exc:
                err = exc
                self.segments.abort()
            # every rank captures the step or none does (a rank replaying segments beside a rank running the eager step
            # would issue different collectives); a refusal reaches the trainer as NotImplementedError -> eager step
            if not all_ranks_agree(err is None, self.y.device):
                for o, s in zip(self.opts, snaps):
                    o.restore(s)
                self.segments = None
                raise NotImplementedError("the data-parallel GAN step could not be captured on every rank (%s)" % (
                    "this rank: %s: %s" % (type(err).__name__, str(err)[:200]) if err is not None else "another rank failed"))
        else:
            self.graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(self.graph, capture_error_mode="thread_local", stream=self._cap_stream):
                    self.out = self._eager()
            except RuntimeError as exc:
                # only what the CAPTURE refused becomes a refusal the trainer may answer with the eager step; an error of
                # the warm-up steps above (out of memory, a kernel's failed check, a shape assertion) is a bug and propagates
                for o, s in zip(self.opts, snaps):
                    o.restore(s)
                self.graph = None
                raise CaptureRefused("the GAN step could not be captured (%s: %s)" % (type(exc).__name__, str(exc)[:300])) from exc
        for o, s in zip(self.opts, snaps):
            o._step = s["step"]  # capture ran step()'s host code once without running its kernels

    def _module_buffers(self):
        mods = [self.model["generator"]] + list(self.model["discriminator"].values())
        return [b for m in mods for b in m.buffers() if b.is_floating_point()]

    def _eager(self):
        return gan_train_step(self.model, self.optimizer, self._nosched, self.criterion, self.config, self.y, self.x,
                              steps=self.steps)

    def load_batch(self, y, x):
        self.y.copy_(y, non_blocking=True)
        self.x.copy_(x, non_blocking=True)

    def __call__(self):
        """One GAN step; returns the dict of (device) loss tensors of this step (overwritten by the next replay)."""
        if self.segments is not None:
            self.segments.replay()
        else:
            self.graph.replay()
        for o, s in zip(self.opts, self.scheds):
            o._step += 1  # the device-side count advances inside the graph; mirror it on the host
            lr = o.param_groups[0]["lr"]
            s.step()
            if o.param_groups[0]["lr"] != lr:
                o.sync_lr()
        return self.out
