#!/bin/bash
# round-2 visit O: ffn_pair with phase-2 taps (k = 3 backward): parity tests, model tests, step time
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_bf16_ops.py -m gpu -x -q > gpurun_out/r2o_pytest_ops.log 2>&1; tail -4 gpurun_out/r2o_pytest_ops.log
timeout 900 python -m pytest tests/test_gpu_sambert.py "tests/test_bench_config_parity.py::test_sambert_full_b32_matches_oracle" -m gpu -x -q > gpurun_out/r2o_pytest_model.log 2>&1; tail -4 gpurun_out/r2o_pytest_model.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference > gpurun_out/r2o_bench.log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2o_bench.log | head -1
