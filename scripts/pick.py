"""stdin: a bench.py JSON line -> the few numbers an A/B needs (ms per step, forward ms, roofline fraction, loss)."""
import json
import sys

d = json.loads(sys.stdin.read().strip().split("\n")[-1])
r = d.get("roofline") or {}
print("ms_per_step %.3f  forward_ms %s  frac %s  final_loss %.4f" % (
    d["ms_per_step"], r.get("forward_ms"), r.get("frac"), d["config"].get("final_loss", float("nan"))))
