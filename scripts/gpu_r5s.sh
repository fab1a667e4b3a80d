#!/bin/bash
# Round 5 visit s: direct-gradient test diagnosis (run-to-run noise floor), FSMN filter-gradient kernels (parity + time),
# kernel stats of the captured SAM-BERT steps (non-kantts kernel census).
T=${1:-r5s}
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/diag_direct_grads.py > gpurun_out/${T}_diag.log 2>&1
echo "diag exit $?"; tail -12 gpurun_out/${T}_diag.log
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "fsmn" > gpurun_out/${T}_tests.log 2>&1
echo "tests exit $?"; tail -3 gpurun_out/${T}_tests.log
timeout 300 python scripts/infer_breakdown.py 24 > gpurun_out/${T}_infer_breakdown.log 2>&1
tail -12 gpurun_out/${T}_infer_breakdown.log
ARGS="--no-hifigan --no-inference --no-cpu-baseline --no-fp32 --no-roofline --steps 40"
for i in 1; do
  timeout 300 python bench.py $ARGS > gpurun_out/${T}_bench_$i.json 2> gpurun_out/${T}_bench_$i.err
  python - $i $T <<'PY'
import json, sys
d = json.loads(open("gpurun_out/%s_bench_%s.json" % (sys.argv[2], sys.argv[1])).read().strip().splitlines()[-1])
print("run", sys.argv[1], "ms_per_step", "%.3f" % d["ms_per_step"], "forward_ms", d["roofline"].get("forward_ms"))
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${T}_prof -o sambert -- python $GRAFT_REPO_ROOT/bench.py $ARGS --steps 40 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/${T}_rocprof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/${T}_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -120 "$f" > gpurun_out/${T}_sambert_steps_kernel_stats_top.csv
rm -rf gpurun_out/${T}_prof
head -30 gpurun_out/${T}_sambert_steps_kernel_stats_top.csv | cut -c1-120
