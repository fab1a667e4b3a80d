"""Which lines of kan-tts_amd/csrc ran in a kernel-source run on the CPU (tests/hipemu/README.md, "Which lines ran").

    HIPEMU_COVERAGE=1 python -m pytest tests/test_kernel_source_on_cpu.py -q      # counters -> tests/hipemu/_build_cov/*.gcda
    python scripts/kernel_coverage.py                    # per-file table + functions that never ran
    python scripts/kernel_coverage.py attn.hip           # the line ranges of one file that never ran

Reads clang's gcov-format counters with the image's gcov; for a template the first record of a line is the sum over its
instantiations, which is what the table counts (a line is "never run" only if no instantiation ran it)."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
COV = os.path.join(ROOT, "tests", "hipemu", "_build_cov")


def gcov_all(out):
    funcs = {}
    for gcda in sorted(glob.glob(os.path.join(COV, "*.gcda"))):
        name = os.path.basename(gcda)[:-5]
        r = subprocess.run(["gcov", "-f", "-o", COV, gcda], cwd=ROOT, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write("gcov failed on %s (skipped)\n" % name)
            continue
        fn = None
        for line in r.stdout.splitlines():
            m = re.match(r"Function '(.*)'", line)
            if m:
                fn = m.group(1)
            elif fn and line.startswith("Lines executed:"):
                funcs.setdefault(name, []).append((fn, line.startswith("Lines executed:0.00%")))
                fn = None
        for g in glob.glob(os.path.join(ROOT, "*.gcov")):
            base = os.path.basename(g)
            if base.endswith((".hip.gcov", ".inc.gcov")) or base == "common.h.gcov":
                os.replace(g, os.path.join(out, base))
            else:
                os.remove(g)
    return funcs


def parse(path):
    seen, src = {}, {}
    for line in open(path, errors="replace"):
        m = re.match(r"\s*([^:]+):\s*(\d+):(.*)", line)
        if not m or m.group(2) == "0":
            continue
        n = int(m.group(2))
        if n in seen:
            continue
        c = m.group(1).strip()
        src[n] = m.group(3)
        seen[n] = None if c == "-" else (0 if c.startswith(("#####", "=====")) else 1)
    return seen, src


def main():
    if not glob.glob(os.path.join(COV, "*.gcda")):
        sys.exit("no counters under %s: run the kernel-source tests with HIPEMU_COVERAGE=1 first" % COV)
    with tempfile.TemporaryDirectory() as out:
        funcs = gcov_all(out)
        if len(sys.argv) > 1:
            seen, src = parse(os.path.join(out, sys.argv[1] + ".gcov"))
            un = sorted(n for n, v in seen.items() if v == 0)
            rngs = []
            for n in un:
                if rngs and n - rngs[-1][1] <= 2:
                    rngs[-1][1] = n
                else:
                    rngs.append([n, n])
            for a, b in rngs:
                print("%d-%d: %s" % (a, b, src[a].strip()[:120]))
            return
        tot = miss = 0
        for f in sorted(glob.glob(os.path.join(out, "*.gcov"))):
            seen, _ = parse(f)
            ex = sum(1 for v in seen.values() if v == 1)
            un = sum(1 for v in seen.values() if v == 0)
            print("%-26s executable lines %5d   never run %4d (%4.1f %%)" % (os.path.basename(f)[:-5], ex + un, un, 100.0 * un / max(1, ex + un)))
            tot += ex + un
            miss += un
        print("%-26s executable lines %5d   never run %4d (%4.1f %%)" % ("total", tot, miss, 100.0 * miss / max(1, tot)))
        names = [(n, fn) for n, lst in sorted(funcs.items()) for fn, never in lst if never]
        dem = subprocess.run(["c++filt"], input="\n".join(fn for _, fn in names), capture_output=True, text=True).stdout.splitlines()
        print("\nkernels / launchers of csrc that never ran:")
        for (n, _), d in zip(names, dem):
            if re.search(r"std::|__gnu|hipemu|operator\(\)|^ato[if]$|make_float|_ZL", d):
                continue
            print("  %-14s %s" % (n, re.sub(r"\(.*", "", d)[:120]))


if __name__ == "__main__":
    main()
