"""Which kernels wait for their global loads one at a time?  (No GPU needed.)

Compiles every csrc/*.hip to gfx950 assembly and, per kernel, walks the instruction stream: a global / buffer load that is
followed -- within a few vector instructions and before the next load -- by `s_waitcnt vmcnt(0)` has nothing else in flight
beside it.  Kernels where that is the rule are chains of memory round trips whatever their arithmetic: a conversion or a
select right behind a predicated load (`if (ok) v = *p;` ... `use(v)`), a load inside a loop body, an epilogue that reads its
residual row by row.  Round 5 found three that way: the staging of the bf16 weight-gradient contractions (48 loads issued two
at a time: 36 -> 25 us per grouped launch, 6.20 -> 5.90 ms per SAM-BERT step), the FSMN filter-gradient window (72 loads,
one at a time: 21 us per launch whatever the shape) and the FIR kernel's residual rows.

Usage: python scripts/isa_load_bursts.py [min_loads] [name filter ...]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kan-tts_amd", "csrc")


def main():
    min_loads = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
    filt = [a for a in sys.argv[1:] if not a.isdigit()]
    out = tempfile.mkdtemp()
    procs = []
    for f in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        s = os.path.join(out, os.path.basename(f)[:-4] + ".s")
        procs.append((s, subprocess.Popen(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment", "-S",
                                           "--cuda-device-only", f, "-o", s], stderr=subprocess.DEVNULL, cwd=CSRC)))
    rows = []
    for s, p in procs:
        p.wait()
        name, seq = None, []
        for ln in open(s):
            m = re.match(r"^(_Z\w+):", ln)
            if m:
                name, seq = m.group(1), []
                continue
            if name is None:
                continue
            t = ln.strip()
            if t.startswith("s_endpgm"):
                rows.append(score(name, seq, os.path.basename(s)))
                name = None
            elif t.startswith(("global_load", "buffer_load")):
                seq.append("L")
            elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
                seq.append("W0")
            elif t.startswith("s_waitcnt") and "vmcnt" in t:
                seq.append("W")
            elif t.startswith(("v_", "ds_")):
                seq.append("x")
    print("%-8s %-6s %-9s %-9s %s" % ("serial", "loads", "avg burst", "max burst", "kernel"))
    for serial, loads, avg, mx, name, src in sorted(rows, reverse=True):
        if loads >= min_loads and (not filt or any(f in name for f in filt)):
            print("%-8d %-6d %-9.1f %-9d %s [%s]" % (serial, loads, avg, mx, name[:90], src))


def score(name, seq, src):
    serial = loads = 0
    bursts, cur = [], 0
    for i, s in enumerate(seq):
        if s == "L":
            loads += 1
            cur += 1
            j = i + 1
            while j < len(seq) and seq[j] == "x" and j - i < 8:
                j += 1
            if j < len(seq) and seq[j] == "W0":
                serial += 1
        elif s in ("W", "W0"):
            if cur:
                bursts.append(cur)
            cur = 0
    if cur:
        bursts.append(cur)
    return serial, loads, (sum(bursts) / len(bursts)) if bursts else 0.0, max(bursts) if bursts else 0, name, src


if __name__ == "__main__":
    main()
