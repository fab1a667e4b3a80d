#!/bin/bash
# round-2 visit P: dropout RNG with one hash per four elements (all dropout users), ATen census of the step
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16_ops.py tests/test_gpu_sambert.py "tests/test_bench_config_parity.py::test_sambert_full_b32_matches_oracle" -m gpu -x -q > gpurun_out/r2p_pytest.log 2>&1; tail -4 gpurun_out/r2p_pytest.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference > gpurun_out/r2p_bench.log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2p_bench.log | head -1
timeout 300 python scripts/aten_census.py > gpurun_out/r2p_aten_census.log 2>&1; head -80 gpurun_out/r2p_aten_census.log | cut -c1-200
