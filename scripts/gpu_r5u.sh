#!/bin/bash
# Round 5 visit u: the autoregressive inference loops as one launch each -- parity on the device, stage breakdown, bench leg.
T=${1:-r5u}
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ar_kernels.py tests/test_decode_graph.py tests/test_config5_inference.py -q -m gpu > gpurun_out/${T}_tests.log 2>&1
echo "tests exit $?"; tail -15 gpurun_out/${T}_tests.log
for m in graph kernel; do
  timeout 300 python scripts/infer_breakdown.py 24 $m > gpurun_out/${T}_infer_breakdown_$m.log 2>&1
  tail -11 gpurun_out/${T}_infer_breakdown_$m.log
done
timeout 900 python bench.py --no-hifigan --no-fp32 --no-roofline --steps 10 --no-forward-only > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench exit $?"; tail -3 gpurun_out/${T}_bench.err
python - $T <<'PY'
import json, sys
d = json.loads(open("gpurun_out/%s_bench.json" % sys.argv[1]).read().strip().splitlines()[-1])
inf = d.get("inference", {})
for k, v in inf.items():
    if isinstance(v, dict) and "utterances_per_s" in v:
        print("%-16s %8.1f utt/s  %12.0f samples/s  %.3f s" % (k, v["utterances_per_s"], v["audio_samples_per_s"], v["seconds"]))
print("value", inf.get("value"), inf.get("unit"))
print("parity", json.dumps(inf.get("parity_error")))
PY
