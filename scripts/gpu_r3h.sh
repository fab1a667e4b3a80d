#!/bin/bash
# round-3 visit H: all-taps narrow weight gradient, feature-map L1 without copies: tests + GAN step (eager / graph)
mkdir -p gpurun_out/r3h
timeout 600 python -m pytest tests/test_cconv.py -m gpu -x -q > gpurun_out/r3h/pytest_cconv.log 2>&1; tail -n 2 gpurun_out/r3h/pytest_cconv.log
timeout 400 python scripts/hifigan_bench.py 32 4 bf16 > gpurun_out/r3h/hifigan.log 2>&1
echo "$(grep -o '"generator_forward_ms": [0-9.]*' gpurun_out/r3h/hifigan.log) $(grep -o '"gan_step_ms": [0-9.]*' gpurun_out/r3h/hifigan.log) $(grep -o '"gan_step_graph_ms": [0-9.]*' gpurun_out/r3h/hifigan.log)"
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && KANTTS_NO_GAN_GRAPH=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3h/prof -o gan -- python $R/scripts/hifigan_bench.py 32 4 bf16 > $R/gpurun_out/r3h/rocprof.log 2>&1 )
f=$(find gpurun_out/r3h/prof -name "*kernel_stats.csv" | head -n 1); [ -n "$f" ] && head -n 60 "$f" > gpurun_out/r3h/gan_kernel_stats_top.csv
rm -rf gpurun_out/r3h/prof
