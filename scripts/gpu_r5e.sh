#!/bin/bash
# Round 5: kernel statistics of the SAM-BERT bench (captured step with the fused decoder blocks), one box.
T=${1:-r5e}
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof -o bench -- python $R/bench.py --no-hifigan --no-inference --no-cpu-baseline --no-fp32 --no-roofline --steps 40 > $R/gpurun_out/${T}_rocprof_bench.log 2>&1
f=$(find $R/gpurun_out/${T}_prof -name "*kernel_stats.csv" | head -n 1)
[ -n "$f" ] && head -n 120 "$f" > $R/gpurun_out/${T}_sambert_steps_kernel_stats_top.csv
rm -rf $R/gpurun_out/${T}_prof
python - <<PY
import csv
rows = list(csv.reader(open("$R/gpurun_out/${T}_sambert_steps_kernel_stats_top.csv")))
tot = sum(float(r[2]) for r in rows[1:])
for r in rows[1:46]:
    print("%-70s %6s %9.1f us avg %6.1f  %5.1f%%" % (r[0][:70], r[1], float(r[2]) / 1e3, float(r[3]) / 1e3, 100 * float(r[2]) / tot))
PY
tail -c 400 $R/gpurun_out/${T}_rocprof_bench.log
