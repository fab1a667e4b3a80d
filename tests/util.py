"""Shared helpers for the test-suite."""
import contextlib
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


class _Patch:
    def __init__(self):
        self.saved = []

    def setattr(self, mod, name, val, raising=True):
        self.saved.append((mod, name, getattr(mod, name)))
        setattr(mod, name, val)

    def undo(self):
        for mod, name, val in reversed(self.saved):
            setattr(mod, name, val)
        self.saved = []


@contextlib.contextmanager
def emulation():
    """Temporarily route the binding to oracle/cabi_numpy (host memory) -- test-only."""
    import conftest

    p = _Patch()
    try:
        yield conftest._emulate(p)
    finally:
        p.undo()


def install_kernel_source(p):
    """Same contract as conftest._emulate (``p``: a monkeypatch-like object), but what is installed is
    tests/hipemu/_build/libkantts_hostsim.so -- the kernel SOURCES of kan-tts_amd/csrc compiled for the host and run by
    the fibre scheduler of tests/hipemu -- loaded by the product's own loader in place of libkantts_hip.so, with host
    tensors.  Test-only (the product refuses host tensors and has no such switch)."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
    try:
        import build as hipemu_build
    finally:
        sys.path.pop(0)
    import kantts._hip as hip
    import kantts._hip.ops as ops
    import kantts._hip.ops_bf16 as ops_bf16
    import kantts.utils.audio_torch as audio_torch

    so = hipemu_build.build()

    def ptr(t, dtype=None):
        if t is None:
            return None
        if dtype is not None and t.dtype != dtype:
            raise TypeError("expected %s, got %s" % (dtype, t.dtype))
        assert not t.is_cuda
        return t.data_ptr()

    p.setattr(hip, "LIB_PATH", so)
    p.setattr(hip, "_lib", _HOSTSIM.get("lib"))
    _HOSTSIM["lib"] = real = hip.lib()  # the product's own loader: same argtypes as for the device library
    for mod in (hip, ops, ops_bf16, audio_torch):
        p.setattr(mod, "lib", lambda: real)
        p.setattr(mod, "ptr", ptr)
        p.setattr(mod, "stream", lambda: None)
    return real


@contextlib.contextmanager
def kernel_source_on_cpu():
    """Temporarily route the binding to the host build of the kernel sources (install_kernel_source) -- test-only."""
    p = _Patch()
    try:
        yield install_kernel_source(p)
    finally:
        p.undo()


_HOSTSIM = {}


def to_dev(x, device):
    if torch.is_tensor(x):
        y = x.detach().to(device)
        if x.requires_grad:
            y.requires_grad_(True)
        return y
    if isinstance(x, (list, tuple)):
        return type(x)(to_dev(t, device) for t in x)
    if isinstance(x, dict):
        return {k: to_dev(v, device) for k, v in x.items()}
    return x


def run_both(fn, *args, grads_of=(), device="cuda"):
    """Run ``fn(*args)`` on the GPU through libkantts_hip.so and on the CPU through the emulated C ABI.
    fn returns a tensor or tuple of tensors; tensors in ``args`` flagged requires_grad get gradients of
    sum(out_k * W_k) with fixed random cotangents.  Returns (gpu_outs, gpu_grads, cpu_outs, cpu_grads).
    ``device="hostsim"``: the first leg runs the kernel sources on the CPU (kernel_source_on_cpu) instead."""
    results = []
    cots = None
    for dev in (device, "cpu"):
        a = [to_dev(t, "cpu" if dev == "hostsim" else dev) for t in args]
        ctx = emulation() if dev == "cpu" else (kernel_source_on_cpu() if dev == "hostsim" else contextlib.nullcontext())
        with ctx:
            # both legs draw their dropout masks from (host seed + device-resident offset): the offsets of the two devices
            # are advanced independently by whatever training steps ran earlier in the session -- start both from zero
            try:
                import kantts._hip as _hip

                _hip.rng_state("cpu" if dev in ("cpu", "hostsim") else dev).zero_()
            except Exception:
                pass
            out = fn(*a)
            outs = [o for o in (out if isinstance(out, (tuple, list)) else (out,)) if torch.is_tensor(o)]
            leaves = [t for t in _flatten(a) if torch.is_tensor(t) and t.requires_grad]
            grads = []
            diff = [o for o in outs if o.requires_grad]
            if leaves and diff:
                if cots is None:
                    g = torch.Generator().manual_seed(99)
                    cots = [torch.randn(o.shape, generator=g) for o in diff]
                loss = sum((o * c.to(o.device)).sum() for o, c in zip(diff, cots))
                grads = torch.autograd.grad(loss, leaves, allow_unused=True)
        results.append(([o.detach().cpu() for o in outs], [None if g is None else g.detach().cpu() for g in grads]))
    return results[0][0], results[0][1], results[1][0], results[1][1]


def _flatten(x):
    for t in x:
        if isinstance(t, (list, tuple)):
            yield from _flatten(t)
        else:
            yield t


def assert_close(a, b, atol, rtol=0.0, what=""):
    assert a.shape == b.shape, "%s shape %s vs %s" % (what, tuple(a.shape), tuple(b.shape))
    err = (a.double() - b.double()).abs()
    tol = atol + rtol * b.double().abs()
    bad = err > tol
    if bad.any():
        i = int(err.argmax())
        raise AssertionError("%s: max abs err %.3e (ref %.3e) at flat %d; %d/%d beyond tol %.1e" % (
            what, float(err.max()), float(b.reshape(-1)[i]), i, int(bad.sum()), err.numel(), atol))


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def assert_grads_close(pairs, tol, what=""):
    """Parameter-gradient parity for networks with LeakyReLU gates.  ``pairs``: iterable of (name, got, ref).
    The whole gradient (all parameters concatenated) must agree to ``tol`` in relative L2.  Individual tensors
    get 10x that: a pre-activation within an ulp of zero can take the other side of the LeakyReLU in two fp32
    summation orders, and ONE such flip moves a 16-element bias gradient by ~1 % while leaving the large
    tensors (and the global norm) untouched."""
    num, den = 0.0, 0.0
    for name, got, ref in pairs:
        got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
        e, r = float((got - ref).pow(2).sum()), float(ref.pow(2).sum())
        num, den = num + e, den + r
        assert (e / (r + 1e-60)) ** 0.5 <= 10 * tol, "%s %s: rel-L2 %.3e" % (what, name, (e / (r + 1e-60)) ** 0.5)
    assert (num / (den + 1e-60)) ** 0.5 <= tol, "%s global rel-L2 %.3e" % (what, (num / (den + 1e-60)) ** 0.5)
