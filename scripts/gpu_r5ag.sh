#!/bin/bash
# Launches per REPLAY of the captured SAM-BERT step: kernel trace of a 140-step run against the closing visit's 40-step run.
T=${1:-r5ag}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof -o p -- python $R/bench.py --no-hifigan --no-inference --no-cpu-baseline --no-fp32 --no-roofline --no-forward-only --steps 140 --warmup 5 > $R/gpurun_out/${T}_rocprof.log 2>&1
f=$(find $R/gpurun_out/${T}_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $R/gpurun_out/${T}_sambert_140steps_kernel_stats.csv
rm -rf $R/gpurun_out/${T}_prof
wc -l $R/gpurun_out/${T}_sambert_140steps_kernel_stats.csv
