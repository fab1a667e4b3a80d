#!/bin/bash
# Round 5 visit aj: contraction staging without per-chunk branches, FSMN filter-gradient loads issued together.
T=${1:-r5aj}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out; cd $R
timeout 1200 python -m pytest tests/test_gpu_bf16_ops.py tests/test_gpu_ops.py tests/test_gpu_sambert.py tests/test_bench_config_parity.py tests/test_pnca_block.py -q -m gpu > gpurun_out/${T}_tests.log 2>&1; echo "tests exit $?"; tail -4 gpurun_out/${T}_tests.log
ARGS="--no-hifigan --no-inference --no-cpu-baseline --no-fp32 --no-roofline --steps 40"
for i in 1 2; do
  timeout 300 python bench.py $ARGS > gpurun_out/${T}_bench_$i.json 2> gpurun_out/${T}_bench_$i.err
  python - $i $T <<'PY'
import json, sys
for l in open("gpurun_out/%s_bench_%s.json" % (sys.argv[2], sys.argv[1])):
    if l.startswith("{"): d = json.loads(l)
print("run", sys.argv[1], "ms_per_step", "%.3f" % d["ms_per_step"], "forward_ms", d["roofline"].get("forward_ms"))
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof -o p -- python $R/bench.py $ARGS --no-forward-only > $R/gpurun_out/${T}_rocprof.log 2>&1
f=$(find $R/gpurun_out/${T}_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -150 "$f" > $R/gpurun_out/${T}_sambert_steps_kernel_stats_top.csv
rm -rf $R/gpurun_out/${T}_prof
grep -h "bgemm_tn\|bgemm_nt\|fsmn_dw41" $R/gpurun_out/${T}_sambert_steps_kernel_stats_top.csv | cut -c1-60,100-175
