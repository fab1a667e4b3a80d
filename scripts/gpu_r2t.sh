#!/bin/bash
# round-2 visit T: smoke(), the default bench.py line, rocprofv3 kernel statistics of the same short bench command
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2t_smoke.log 2>&1; tail -3 gpurun_out/r2t_smoke.log
timeout 900 python bench.py > gpurun_out/r2t_bench_full.log 2>&1; tail -c 1500 gpurun_out/r2t_bench_full.log
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2t_prof -o sam -- python $R/bench.py --steps 10 --warmup 3 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference > $R/gpurun_out/r2t_rocprof.log 2>&1 )
f=$(find gpurun_out/r2t_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -80 "$f" > gpurun_out/r2t_sambert_kernel_stats_top.csv && cut -d, -f1-5 gpurun_out/r2t_sambert_kernel_stats_top.csv | sed 's/(.*"/"/' | cut -c1-110 | head -14
rm -rf gpurun_out/r2t_prof
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2t_prof2 -o gan -- python $R/scripts/hifigan_bench.py 32 3 bf16 > $R/gpurun_out/r2t_hifigan.log 2>&1 )
f=$(find gpurun_out/r2t_prof2 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -60 "$f" > gpurun_out/r2t_hifigan_kernel_stats_top.csv
rm -rf gpurun_out/r2t_prof2
grep -o '"gan_step_ms": [0-9.]*' gpurun_out/r2t_hifigan.log
