#!/bin/bash
# round-3 visit C: wgrad with workspace slices, group packing, tile rule; GAN step A/B; rocprof of the GAN step
mkdir -p gpurun_out/r3c
timeout 200 scripts/bench_native/cconv_test 10 quick > gpurun_out/r3c/cconv_native.log 2>&1; tail -2 gpurun_out/r3c/cconv_native.log
timeout 600 python -m pytest tests/test_cconv.py -m gpu -x -q > gpurun_out/r3c/pytest_cconv.log 2>&1; tail -2 gpurun_out/r3c/pytest_cconv.log
timeout 300 python scripts/hifigan_bench.py 32 4 bf16 > gpurun_out/r3c/hifigan_cconv.log 2>&1
echo "cconv: $(grep -o '"generator_forward_ms": [0-9.]*' gpurun_out/r3c/hifigan_cconv.log) $(grep -o '"gan_step_ms": [0-9.]*' gpurun_out/r3c/hifigan_cconv.log)"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3c/prof -o gan -- python $GRAFT_REPO_ROOT/scripts/hifigan_bench.py 32 4 bf16 > $GRAFT_REPO_ROOT/gpurun_out/r3c/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r3c/prof -name "*kernel_stats*" | head -3
f=$(find gpurun_out/r3c/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" > gpurun_out/r3c/gan_kernel_stats_top.csv
find gpurun_out/r3c/prof -name "*.csv" -size +2M -delete; find gpurun_out/r3c/prof -name "*.db" -delete
