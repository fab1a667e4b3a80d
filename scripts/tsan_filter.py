"""Sort the ThreadSanitizer reports of a kernel-source run (tests/hipemu/README.md, barrier audit) into the ones whose two
accesses are both in csrc/*.hip and the ones that involve torch's own (uninstrumented) worker threads.

    python scripts/tsan_filter.py /tmp/tsan.*
"""
import re, sys, glob, collections
kernel = collections.Counter(); other = 0; total = 0
for f in (sys.argv[1:] or glob.glob("/tmp/tsan.*")):
    txt = open(f, errors="replace").read()
    for rep in txt.split("==================\n"):
        if "WARNING: ThreadSanitizer" not in rep: continue
        total += 1
        # the two access stacks: from the first line to "Location"/"Thread T"
        head = re.split(r"\n\s+(?:Location is|Thread T\d+ \(|Mutex M)", rep)[0]
        parts = re.split(r"\n\s+Previous ", head)
        if len(parts) != 2: other += 1; continue
        def top_hip(p):
            m = re.search(r"#0 .*?src/([a-z_0-9]+\.hip:\d+)", p)
            return m.group(1) if m else None
        a, b = top_hip(parts[0]), top_hip(parts[1])
        if a and b: kernel[tuple(sorted((a, b)))] += 1
        else: other += 1
print("reports:", total)
print("both accesses in kernel sources:")
for k, v in sorted(kernel.items()): print("  %3d  %s  /  %s" % (v, k[0], k[1]))
print("one or both accesses inside uninstrumented libtorch / OpenMP / OpenBLAS threads:", other)
