#!/bin/bash
# Round 4, visit AF: kernel statistics of the SAM-BERT training steps alone and of the GAN steps alone (closing code)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4af_prof -o sb -- python $R/bench.py --steps 40 --warmup 5 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline --no-forward-only > /dev/null 2> $R/gpurun_out/r4af_err.log
f=$(find $R/gpurun_out/r4af_prof -name "*kernel_stats.csv" | head -n 1)
[ -n "$f" ] && head -n 70 "$f" > $R/gpurun_out/r4af_sambert_kernel_stats_top.csv
rm -rf $R/gpurun_out/r4af_prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4af_prof2 -o gan -- python $R/scripts/hifigan_bench.py 32 3 bf16 > /dev/null 2> $R/gpurun_out/r4af_err2.log
f=$(find $R/gpurun_out/r4af_prof2 -name "*kernel_stats.csv" | head -n 1)
[ -n "$f" ] && head -n 80 "$f" > $R/gpurun_out/r4af_gan_kernel_stats_top.csv
rm -rf $R/gpurun_out/r4af_prof2
head -5 $R/gpurun_out/r4af_sambert_kernel_stats_top.csv | cut -c1-120
