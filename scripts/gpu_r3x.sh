#!/bin/bash
# round-3 visit X: kernel statistics of the SAM-BERT step (training steps only: no forward-only capture, no roofline leg)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3x_prof -o sam -- python $R/bench.py --steps 20 --warmup 3 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline --no-forward-only > $R/gpurun_out/r3x_rocprof.log 2>&1
cd $R
f=$(find gpurun_out/r3x_prof -name "*kernel_stats.csv" | head -n 1)
[ -n "$f" ] && head -n 90 "$f" > gpurun_out/r3x_sambert_kernel_stats_top.csv
rm -rf gpurun_out/r3x_prof
tail -n 3 gpurun_out/r3x_rocprof.log | cut -c1-300
