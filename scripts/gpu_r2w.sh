#!/bin/bash
# round-2 visit W: side branch with the capture-time switches restored; the order that exposed the state leak
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_trainer.py tests/test_ddp_gloo.py tests/test_gpu_sambert.py tests/test_bench_config_parity.py -m gpu -x -q > gpurun_out/r2w_pytest.log 2>&1; tail -4 gpurun_out/r2w_pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference > gpurun_out/r2w_bench.log 2>&1
echo "$(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2w_bench.log | head -1) $(grep -o '"launch": "[a-z]*"' gpurun_out/r2w_bench.log | head -1)"
