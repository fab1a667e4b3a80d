#!/bin/bash
# round-2 visit S: the full GPU suite on the round's final kernels
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2s_pytest_gpu_full.log 2>&1; tail -12 gpurun_out/r2s_pytest_gpu_full.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference > gpurun_out/r2s_bench.log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2s_bench.log | head -1
