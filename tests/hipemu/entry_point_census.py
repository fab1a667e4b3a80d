"""pytest plugin (test infrastructure): counts the C-ABI entry points the kernel-source tests actually call.

    PYTHONPATH=tests/hipemu python -m pytest tests/test_kernel_source_on_cpu.py -q -p entry_point_census
    -> profiles/kernel_source_cpu_entry_points.json   (calls per entry point, and the declared ones never called)
"""
import collections
import json
import os
import re

import util

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
COUNT = collections.Counter()
_install = util.install_kernel_source


class _Counting:
    def __init__(self, lib):
        object.__setattr__(self, "_lib", lib)

    def __getattr__(self, name):
        f = getattr(self._lib, name)
        if not name.startswith("kantts_"):
            return f

        def counted(*a, **k):
            COUNT[name] += 1
            return f(*a, **k)

        return counted


def _install_counting(p):
    import kantts._hip as hip
    import kantts._hip.ops as ops
    import kantts._hip.ops_bf16 as ops_bf16
    import kantts.utils.audio_torch as audio_torch

    proxy = _Counting(_install(p))
    for mod in (hip, ops, ops_bf16, audio_torch):
        p.setattr(mod, "lib", lambda: proxy)
    return proxy


util.install_kernel_source = _install_counting


def pytest_sessionfinish(session, exitstatus):
    header = open(os.path.join(ROOT, "include", "kantts_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(kantts_[a-z0-9_]+)\s*\(", header)))
    out = {"declared": len(declared), "called": sum(1 for d in declared if COUNT[d]),
           "never_called": [d for d in declared if not COUNT[d]], "calls": dict(sorted(COUNT.items()))}
    with open(os.path.join(ROOT, "profiles", "kernel_source_cpu_entry_points.json"), "w") as fh:
        json.dump(out, fh, indent=1)
