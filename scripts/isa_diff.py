"""Per-kernel comparison of the gfx950 code in two object files (e.g. csrc/x.o before and after an edit that must not touch
the existing kernels): extracts the device code object of each (llvm-objdump --offloading), disassembles it and compares
the instruction streams kernel by kernel.

    python scripts/isa_diff.py /tmp/gemm_bf16.o.before kan-tts_amd/csrc/gemm_bf16.o
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

BIN = "/opt/rocm/lib/llvm/bin"


def kernels(obj):
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(obj, os.path.join(d, "o.o"))
        subprocess.run([os.path.join(BIN, "llvm-objdump"), "--offloading", "o.o"], cwd=d, capture_output=True, check=True)
        dev = [f for f in os.listdir(d) if "amdgcn" in f]
        if not dev:
            raise SystemExit("no device code object in " + obj)
        text = subprocess.run([os.path.join(BIN, "llvm-objdump"), "-d", os.path.join(d, dev[0])], capture_output=True, text=True,
                              check=True).stdout
    out, cur = {}, None
    for ln in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln.strip())
        if m:
            cur = m.group(1)
            out[cur] = []
        elif cur is not None and "\t" in ln and ln.strip() != "...":  # "..." = padding between functions
            out[cur].append(ln.split("//")[0].strip())
    return out


def main(a_path, b_path):
    a, b = kernels(a_path), kernels(b_path)
    same = True
    for k in sorted(set(a) | set(b)):
        if k not in b:
            print("removed   %s" % k)
            same = False
        elif k not in a:
            print("new       %-90s %5d instructions" % (k[:90], len(b[k])))
        elif a[k] != b[k]:
            print("CHANGED   %-90s %5d -> %5d instructions" % (k[:90], len(a[k]), len(b[k])))
            same = False
        else:
            print("identical %-90s %5d instructions" % (k[:90], len(a[k])))
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
