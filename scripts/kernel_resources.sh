#!/bin/bash
# Per-kernel register / scratch / LDS / occupancy table of the product build (no GPU needed): every csrc/*.hip compiled with
# the Makefile's flags plus -Rpass-analysis=kernel-resource-usage, objects thrown away.
#   bash scripts/kernel_resources.sh > profiles/rNN_kernel_resource_usage.txt
set -e
cd "$(dirname "$0")/../kan-tts_amd/csrc"
out=$(mktemp -d)
for f in *.hip; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -Rpass-analysis=kernel-resource-usage -c "$f" -o "$out/${f%.hip}.o" > "$out/${f%.hip}.txt" 2>&1 &
done
wait
python3 - "$out" <<'PY'
import re, glob, subprocess, sys
rows = []
for f in sorted(glob.glob(sys.argv[1] + "/*.txt")):
    cur = None
    for line in open(f):
        m = re.match(r"(\S+?):(\d+):\d+: remark: Function Name: (\S+)", line)
        if m:
            cur = {"file": m.group(1), "line": int(m.group(2)), "name": m.group(3)}
            rows.append(cur)
            continue
        m = re.match(r"\S+: remark:\s+([A-Za-z ]+?)(?: \[[^\]]+\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    r["dem"] = re.sub(r"\(.*", "", n)[:78]
print("%d kernels (gfx950, -O3)" % len(rows))
print("%-16s %-78s %5s %5s %7s %6s %4s %7s" % ("source:line", "kernel", "VGPR", "AGPR", "scratch", "spills", "occ", "LDS"))
for r in rows:
    print("%-16s %-78s %5d %5d %7d %6d %4d %7d" % ("%s:%d" % (r["file"], r["line"]), r["dem"], r.get("VGPRs", 0), r.get("AGPRs", 0),
          r.get("ScratchSize", 0), r.get("VGPRs Spill", 0), r.get("Occupancy", 0), r.get("LDS Size", 0)))
print()
print("kernels with scratch:", ", ".join(r["dem"] for r in rows if r.get("ScratchSize", 0)) or "none")
PY
rm -rf "$out"
