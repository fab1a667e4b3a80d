import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "kan-tts_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu`-marked tests need a HIP device: skip them (instead of failing) on a CPU box.  On a GPU box they always
    run -- a missing libkantts_hip.so must fail loudly there, never skip."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="gpu test: no HIP device in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _global_switches_off():
    """The scheduling switches of the host layer (weight gradients deferred / on a side stream, predictors as a side
    branch) are process-wide; a test that turns them on must not change what the next test sees."""
    yield
    import kantts._hip as hip
    import kantts._hip.ops as ops

    hip.deferred_tn.groups, hip.deferred_tn.copies, hip.deferred_tn.rowsums = {}, [], []  # drop whatever a failed test left
    ops.wgrad_overlap.enabled = False
    hip.deferred_tn.enabled = False
    ops.side_branch.enabled = False
    ops.side_branch.release()


def _emulate(monkeypatch):
    """Route the ctypes binding to oracle/cabi_numpy.EmulatedLib (HOST memory).  Test-only: lets the
    host logic of the product run without a GPU; the product itself has no such switch."""
    import cabi_numpy
    import kantts._hip as hip
    import kantts._hip.ops as ops
    import kantts._hip.ops_bf16 as ops_bf16

    emu = cabi_numpy.EmulatedLib()

    def ptr(t, dtype=None):
        if t is None:
            return None
        if dtype is not None and t.dtype != dtype:
            raise TypeError("expected %s, got %s" % (dtype, t.dtype))
        assert not t.is_cuda
        return t.data_ptr()

    import kantts.utils.audio_torch as audio_torch  # binds lib / ptr / stream by name at import time

    for mod in (hip, ops, ops_bf16, audio_torch):
        monkeypatch.setattr(mod, "lib", lambda: emu, raising=True)
        monkeypatch.setattr(mod, "ptr", ptr, raising=True)
        monkeypatch.setattr(mod, "stream", lambda: None, raising=True)
    return emu


@pytest.fixture
def emulated_cabi(monkeypatch):
    return _emulate(monkeypatch)


@pytest.fixture(scope="session")
def has_gpu():
    import torch

    return torch.cuda.is_available()


# ---- op sweeps on two back ends (tests/test_ops_sweep.py, test_conv_sweep.py): expected values from the oracle / plain torch
class _Bridge:
    """Maps the CPU leaves of a sweep onto the back end under test (identity for the emulated ABI, device twins for the GPU:
    one twin per leaf so that gradients w.r.t. the leaves can be asked for) and results back to the CPU."""

    def __init__(self, device):
        self.device, self._twins = device, {}

    def __call__(self, obj):
        import torch

        if obj is None or self.device is None:
            return obj
        if isinstance(obj, (list, tuple)):
            return type(obj)(self(o) for o in obj)
        if not torch.is_tensor(obj):
            return obj
        key = id(obj)
        if key not in self._twins:
            t = obj.detach().to(self.device)
            self._twins[key] = (obj, t.requires_grad_(True) if obj.requires_grad else t)  # (keeps ``obj`` alive: ids stay unique)
        return self._twins[key][1]

    def back(self, obj):
        import torch

        if obj is None or self.device is None:
            return obj
        if isinstance(obj, (list, tuple)):
            return type(obj)(self.back(o) for o in obj)
        return obj.detach().cpu() if torch.is_tensor(obj) else obj


@pytest.fixture(params=["emulated", pytest.param("gpu", marks=pytest.mark.gpu)])
def dv(request, monkeypatch):
    if request.param == "emulated":
        _emulate(monkeypatch)
        return _Bridge(None)
    import kantts._hip as hip

    hip.lib()
    hip.set_precision("fp32")
    return _Bridge("cuda")


