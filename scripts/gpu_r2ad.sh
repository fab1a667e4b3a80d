#!/bin/bash
# round-2 visit AD: final state (after the discriminator / generator streams and the shared-input projections) -- full GPU suite, smoke(), the default bench.py line, rocprofv3 kernel statistics
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2ad_pytest_gpu_full.log 2>&1; tail -3 gpurun_out/r2ad_pytest_gpu_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2ad_smoke.log 2>&1; tail -1 gpurun_out/r2ad_smoke.log
timeout 900 python bench.py > gpurun_out/r2ad_bench_full.log 2>&1; grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2ad_bench_full.log | head -1
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2ad_prof -o sam -- python $R/bench.py --steps 10 --warmup 3 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference > $R/gpurun_out/r2ad_rocprof.log 2>&1 )
f=$(find gpurun_out/r2ad_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -80 "$f" > gpurun_out/r2ad_sambert_kernel_stats_top.csv
rm -rf gpurun_out/r2ad_prof
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2ad_rocprof.log | head -1
