"""A training step of a DATA-PARALLEL replica as a chain of hipGraph segments with the gradient exchange between them.

RCCL calls are not captured.  What the reference gets from DistributedDataParallel (kantts/models/__init__.py:71-84,
118-121: gradient buckets all-reduced from autograd hooks while backward is still running) is obtained for a REPLAYED step
by cutting the capture where a bucket of the gradient arena becomes complete (ParamArena's post-accumulate hooks, the
same partition the eager path uses):

    segment 0      forward + backward down to the last-registered bucket, + its packing into the arena
    segment k      backward down to the next bucket, + packing
    last segment   average, clip, Adam (and whatever follows in the step)

``replay()`` launches segment k, issues the asynchronous all-reduce of the bucket it packed -- the collective's own stream
waits for the segment, the launching stream does not -- and launches segment k + 1 at once: bucket k is exchanged beside
the backward of the earlier layers.  Before the segment that averages an arena, its handles are waited for (a stream
wait, not a host wait).  Several arenas (HiFi-GAN: generator, MPD, MSD) cut the same chain.

A cut happens inside ``ParamArena._launch_bucket``, i.e. on autograd's device thread: capture mode is "relaxed" (begin and
end on different host threads; the collective's watchdog polls events meanwhile).  At a cut every helper stream of
kantts._hip.ops is joined into the capture stream, the capture is ended, the next one is begun on the same stream and the
helper streams are forked into it (ops.fork_helper_streams explains why).
"""
import gc

import torch
import torch.distributed as dist

from kantts._hip import ops


def all_ranks_agree(ok, device):
    """The data-parallel forms of a captured step issue DIFFERENT collectives (segments: the arena's param-aligned buckets in
    completion order; two graphs: ``n_buckets`` equal chunks in ascending order), so the choice between them must be the
    same on every rank: a capture that failed on one rank only (memory, a shape-dependent path) makes every rank fall back.
    MIN over the ranks of this rank's success flag; a collective, so every rank must call it at the same point."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() <= 1:
        return bool(ok)
    on_host = dist.get_backend() == "gloo"
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if on_host else device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(int(flag.item()))


class SegmentedCapture:
    def __init__(self, arenas, stream):
        self.arenas, self.stream = list(arenas), stream
        self.segments = []  # (graph, arena | None, bucket | None, last bucket of its arena?)
        self._graph = self._pool = None
        self.skip_exchange = False  # measurement only: replay without the collectives (exposed exchange time)

    # ---- capture ---------------------------------------------------------------------------------------------------
    def _cut(self, arena, bucket):
        with torch.cuda.stream(self.stream):
            ops.join_capturing_side_streams()
            self._graph.capture_end()
            last = all(b["handle"] is not None for b in arena.buckets)  # this one is already marked
            self.segments.append((self._graph, arena, bucket, last))
            if self._pool is None:
                self._pool = self._graph.pool()
            self._graph = torch.cuda.CUDAGraph()
            self._graph.capture_begin(pool=self._pool, capture_error_mode="relaxed")
            ops.fork_helper_streams()

    def capture(self, fn):
        """Run ``fn`` (zero_grad, forward, backward, optimizer steps of the arenas) under capture; returns fn()."""
        self._saved = [(getattr(a, "overlap", False), getattr(a, "_active", False)) for a in self.arenas]
        for a in self.arenas:
            if not getattr(a, "buckets", None):
                a._build_buckets()
            a.overlap = True  # the hooks count the buckets down during the captured backward
            a.capture_stream = self.stream
            a.bucket_ready = (lambda b, a=a: self._cut(a, b))
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.empty_cache()
        try:
            with torch.cuda.stream(self.stream):
                self._graph = torch.cuda.CUDAGraph()
                self._graph.capture_begin(capture_error_mode="relaxed")
                out = fn()
                ops.join_capturing_side_streams()
                self._graph.capture_end()
                self.segments.append((self._graph, None, None, False))
                self._graph = None
        finally:
            self._restore_arenas()
        torch.cuda.current_stream().wait_stream(self.stream)
        want = sum(len(a.buckets) for a in self.arenas) + 1
        if len(self.segments) != want:
            raise RuntimeError("expected %d segments, captured %d" % (want, len(self.segments)))
        return out

    def _restore_arenas(self):
        """The arenas' eager-path settings as they were before the capture: an eager step after it (the warm-up of a new
        batch shape, a step for which no graph is ready) overlaps its exchange with backward exactly as before."""
        saved = getattr(self, "_saved", None) or [(False, False)] * len(self.arenas)
        for a, (overlap, _) in zip(self.arenas, saved):
            a.bucket_ready = a.capture_stream = None
            a.overlap, a._active = overlap, False  # (armed again by the next zero_grad)
        self._saved = None

    def abort(self):
        """After a failed capture: leave no stream in capture mode and no arena armed."""
        from kantts._hip import deferred_tn

        deferred_tn.groups, deferred_tn.copies, deferred_tn.rowsums = {}, [], []
        self._restore_arenas()
        try:
            with torch.cuda.stream(self.stream):
                if self._graph is not None and torch.cuda.is_current_stream_capturing():
                    ops.join_capturing_side_streams()
                    self._graph.capture_end()
        except Exception:
            pass
        self._graph = None
        self.segments = []
        torch.cuda.synchronize()

    # ---- replay ----------------------------------------------------------------------------------------------------
    def replay(self):
        pending = {}
        for g, arena, bucket, last in self.segments:
            g.replay()
            if arena is None or self.skip_exchange:
                continue
            pending.setdefault(id(arena), []).append(arena.exchange(bucket))
            if last:  # the next segment averages this arena's gradients
                for h in pending.pop(id(arena)):
                    h.wait()
