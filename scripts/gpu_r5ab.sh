#!/bin/bash
# Round 5 visit ab: the loss kernel without per-element 64-bit divisions, the fragment-major image kernel with 16-byte
# accesses, four loads in flight in the deterministic gradient norm -- parity tests, step time, kernel durations.
T=${1:-r5ab}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 1200 python -m pytest tests/test_gpu_sambert.py tests/test_bench_config_parity.py tests/test_pnca_block.py tests/test_trainer.py tests/test_gpu_bf16_ops.py tests/test_gpu_ops.py -q -m gpu > gpurun_out/${T}_tests.log 2>&1; echo "tests exit $?"; tail -4 gpurun_out/${T}_tests.log
ARGS="--no-hifigan --no-inference --no-cpu-baseline --no-fp32 --no-roofline --steps 40"
for i in 1 2; do
  timeout 300 python bench.py $ARGS > gpurun_out/${T}_bench_$i.json 2> gpurun_out/${T}_bench_$i.err
  python - $i $T <<'PY'
import json, sys
d = json.loads(open("gpurun_out/%s_bench_%s.json" % (sys.argv[2], sys.argv[1])).read().strip().splitlines()[-1])
print("run", sys.argv[1], "ms_per_step", "%.3f" % d["ms_per_step"], "forward_ms", d["roofline"].get("forward_ms"))
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof -o p -- python $R/bench.py $ARGS --no-forward-only > $R/gpurun_out/${T}_rocprof.log 2>&1
f=$(find $R/gpurun_out/${T}_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -150 "$f" > $R/gpurun_out/${T}_sambert_steps_kernel_stats_top.csv
rm -rf $R/gpurun_out/${T}_prof
grep -h "sumsq_det\|masked_l1_many\|fragmajor\|adam_kernel" $R/gpurun_out/${T}_sambert_steps_kernel_stats_top.csv | cut -c1-40,100-200
