#!/bin/bash
# round-2 visit C: NT / TN kernels with two tiles in flight, TN with 64x128 tiles and an atomics budget
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_bf16_ops.py -x -q > gpurun_out/r2c_ops.log 2>&1
echo "ops rc=$?" >> gpurun_out/r2c_ops.log
tail -8 gpurun_out/r2c_ops.log | cut -c1-400
timeout 300 python scripts/bgemm_bench.py > gpurun_out/r2c_bgemm.log 2>&1
echo "bgemm rc=$?" >> gpurun_out/r2c_bgemm.log
cat gpurun_out/r2c_bgemm.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-hifigan --no-cpu-baseline --no-fp32 > gpurun_out/r2c_bench.log 2> gpurun_out/r2c_bench.err
echo "bench rc=$?" >> gpurun_out/r2c_bench.log
tail -c 3000 gpurun_out/r2c_bench.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2c_prof -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-hifigan --no-cpu-baseline --no-fp32 > $GRAFT_REPO_ROOT/gpurun_out/r2c_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r2c_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -45 "$f" > gpurun_out/r2c_kernel_stats_top.csv && cut -c1-150 gpurun_out/r2c_kernel_stats_top.csv | head -40
rm -rf gpurun_out/r2c_prof
