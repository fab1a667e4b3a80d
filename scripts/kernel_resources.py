"""Registers / scratch / LDS of every gfx950 kernel in an object file (from the code object's metadata note).

    python scripts/kernel_resources.py kan-tts_amd/csrc/ffn_pair.o [name-filter]
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

BIN = "/opt/rocm/lib/llvm/bin"


def resources(obj):
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(obj, os.path.join(d, "o.o"))
        subprocess.run([os.path.join(BIN, "llvm-objdump"), "--offloading", "o.o"], cwd=d, capture_output=True, check=True)
        dev = [f for f in os.listdir(d) if "amdgcn" in f]
        if not dev:
            raise SystemExit("no device code object in " + obj)
        text = subprocess.run([os.path.join(BIN, "llvm-readelf"), "--notes", os.path.join(d, dev[0])], capture_output=True,
                              text=True, check=True).stdout
    out, cur = [], {}
    for ln in text.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k in ("agpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "sgpr_count", "vgpr_count",
                 "vgpr_spill_count", "sgpr_spill_count", "max_flat_workgroup_size"):
            cur[k] = int(v)
        elif k == "name" and "vgpr_count" not in cur and v.startswith("_Z") or (k == "name" and cur.get("_want_name")):
            pass
        if k == "symbol":
            cur["symbol"] = v
        if k == "wavefront_size":
            out.append(cur)
            cur = {}
    return out


def demangle(s):
    s = s.replace(".kd", "").strip("'")
    try:
        return subprocess.run([os.path.join(BIN, "llvm-cxxfilt"), s], capture_output=True, text=True).stdout.strip()
    except OSError:
        return s


if __name__ == "__main__":
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    print("%-84s %5s %5s %7s %7s %6s" % ("kernel", "vgpr", "agpr", "scratch", "lds", "spill"))
    for r in resources(sys.argv[1]):
        name = demangle(r.get("symbol", "?"))
        if flt and flt not in name:
            continue
        print("%-84s %5d %5d %7d %7d %6d" % (name[:84], r.get("vgpr_count", -1), r.get("agpr_count", 0),
                                             r.get("private_segment_fixed_size", 0), r.get("group_segment_fixed_size", 0),
                                             r.get("vgpr_spill_count", 0)))
