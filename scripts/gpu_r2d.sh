#!/bin/bash
# round-2 visit D: full GPU test-suite on the new bf16 path + kernel-level profile of the SAM-BERT step
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2d_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2d_pytest.log
tail -6 gpurun_out/r2d_pytest.log | cut -c1-300
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2d_prof -o sam -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-hifigan --no-cpu-baseline --no-fp32 > $GRAFT_REPO_ROOT/gpurun_out/r2d_rocprof.log 2>&1 )
f=$(find gpurun_out/r2d_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -60 "$f" > gpurun_out/r2d_sambert_kernel_stats_top.csv && cut -c1-160 gpurun_out/r2d_sambert_kernel_stats_top.csv | head -45
rm -rf gpurun_out/r2d_prof
