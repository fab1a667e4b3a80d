#!/bin/bash
# Round 5: the backward launch with the ticket-ordered dgamma / dbeta reduction (no contended atomics): parity, per-launch
# time, step A/B, kernel statistics.
T=${1:-r5g}
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest -q -x -m gpu tests/test_pnca_block.py tests/test_bench_config_parity.py -k "not hifigan" > gpurun_out/${T}_tests.log 2>&1; echo "tests exit $?"; tail -n 3 gpurun_out/${T}_tests.log
python scripts/pnca_block_ablate.py 2>&1 | grep KANTTS_PB_DBG
for rep in 1 2; do
  for v in fused chain; do
    unset KANTTS_NO_PNCA_BLOCK
    [ $v = chain ] && export KANTTS_NO_PNCA_BLOCK=1
    timeout 300 python bench.py --no-hifigan --no-inference --no-cpu-baseline --no-fp32 --no-roofline --steps 40 \
      > gpurun_out/${T}_bench_${v}_${rep}.json 2> gpurun_out/${T}_bench_${v}_${rep}.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/${T}_bench_${v}_${rep}.json").read().strip().splitlines()[-1])
print("$v $rep ms_per_step %.3f forward_ms %s" % (d["ms_per_step"], d["roofline"].get("forward_ms")))
PY
  done
done
unset KANTTS_NO_PNCA_BLOCK
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof -o bench -- python $R/bench.py --no-hifigan --no-inference --no-cpu-baseline --no-fp32 --no-roofline --steps 40 > $R/gpurun_out/${T}_rocprof_bench.log 2>&1
f=$(find $R/gpurun_out/${T}_prof -name "*kernel_stats.csv" | head -n 1)
[ -n "$f" ] && head -n 120 "$f" > $R/gpurun_out/${T}_sambert_steps_kernel_stats_top.csv
rm -rf $R/gpurun_out/${T}_prof
python - <<PY
import csv
rows = list(csv.reader(open("$R/gpurun_out/${T}_sambert_steps_kernel_stats_top.csv")))
tot = sum(float(r[2]) for r in rows[1:])
for r in rows[1:16]:
    print("%-70s %6s %9.1f us avg %6.1f  %5.1f%%" % (r[0][:70], r[1], float(r[2]) / 1e3, float(r[3]) / 1e3, 100 * float(r[2]) / tot))
PY
