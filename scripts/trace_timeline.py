"""Timeline of one captured training step from a rocprofv3 --kernel-trace CSV (optionally .gz, columns Kernel_Name,
Start_Timestamp, End_Timestamp, Queue_Id, Stream_Id): the step is delimited by the Adam launches; prints per kernel name the
count / total / mean, the union busy time, the idle time, the time with exactly one kernel running, and (with --list A B) the
kernels between two offsets in microseconds.  Usage: python scripts/trace_timeline.py trace.csv[.gz] [--list A B]"""
import csv
import gzip
import sys
from collections import defaultdict

path = sys.argv[1]
op = gzip.open if path.endswith(".gz") else open
rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", ""), r.get("Stream_Id", ""))
        for r in csv.DictReader(op(path, "rt"))]
rows.sort(key=lambda r: r[1])
adam = [i for i, r in enumerate(rows) if r[0].startswith("adam_kernel")]
assert len(adam) >= 2, "need two optimizer launches to delimit a step"
a, b = adam[-2], adam[-1]
step = rows[a + 1:b + 1]
t0, t1 = step[0][1], step[-1][2]
print("step: %d kernels, %.3f ms from first start to the optimizer's end" % (len(step), (t1 - t0) / 1e6))
# union / overlap profile
ev = []
for n, s, e, q, st in step:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
busy = {0: 0, 1: 0, 2: 0}
cur, last = 0, t0
for t, d in ev:
    busy[min(cur, 2)] += t - last
    last = t
    cur += d
print("idle %.3f ms, exactly one kernel %.3f ms, two or more %.3f ms" % (busy[0] / 1e6, busy[1] / 1e6, busy[2] / 1e6))
agg = defaultdict(lambda: [0, 0])
for n, s, e, q, st in step:
    agg[n][0] += 1
    agg[n][1] += e - s
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(40)]:
    print("%-62s x%-4d %8.1f us  mean %6.1f" % (n, c, t / 1e3, t / 1e3 / c))
if "--list" in sys.argv:
    i = sys.argv.index("--list")
    lo, hi = float(sys.argv[i + 1]), float(sys.argv[i + 2])
    prev_end = None
    for n, s, e, q, st in step:
        o = (s - t0) / 1e3
        if lo <= o <= hi:
            print("%9.1f us  +%6.1f  dur %6.1f  q%s s%s  %s" % (o, (s - prev_end) / 1e3 if prev_end else 0, (e - s) / 1e3, q, st, n))
        prev_end = e if prev_end is None else max(prev_end, e)
