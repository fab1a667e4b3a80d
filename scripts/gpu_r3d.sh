#!/bin/bash
# round-3 visit D: image hand-over; GAN step; rocprof kernel statistics of the GAN step (csv)
mkdir -p gpurun_out/r3d
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_cconv.py -m gpu -x -q > gpurun_out/r3d/pytest_cconv.log 2>&1; tail -2 gpurun_out/r3d/pytest_cconv.log
timeout 300 python scripts/hifigan_bench.py 32 4 bf16 > gpurun_out/r3d/hifigan.log 2>&1
echo "cconv: $(grep -o '"generator_forward_ms": [0-9.]*' gpurun_out/r3d/hifigan.log) $(grep -o '"gan_step_ms": [0-9.]*' gpurun_out/r3d/hifigan.log)"
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3d/prof -o gan -- python $R/scripts/hifigan_bench.py 32 4 bf16 > $R/gpurun_out/r3d/rocprof.log 2>&1 )
f=$(find gpurun_out/r3d/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -60 "$f" > gpurun_out/r3d/gan_kernel_stats_top.csv
rm -rf gpurun_out/r3d/prof
timeout 900 python -m pytest tests/test_hifigan.py tests/test_hifigan_nsf.py tests/test_trainer.py tests/test_bench_config_parity.py tests/test_multiband.py -m gpu -x -q -k "hifigan or gan or GAN or nsf or conv or multiband" > gpurun_out/r3d/pytest_hifigan.log 2>&1; tail -2 gpurun_out/r3d/pytest_hifigan.log
