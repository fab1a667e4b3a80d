#!/bin/bash
# round-2 visit AC: memory K/V projections of all PNCA blocks through one input-gradient launch: tests + step time
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sambert.py tests/test_trainer.py "tests/test_bench_config_parity.py::test_sambert_full_b32_matches_oracle" tests/test_decode_graph.py -m gpu -x -q > gpurun_out/r2ac_pytest.log 2>&1; tail -3 gpurun_out/r2ac_pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference > gpurun_out/r2ac_bench.log 2>&1
echo "$(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2ac_bench.log | head -1) $(grep -o '"launch": "[a-z]*"' gpurun_out/r2ac_bench.log | head -1)"
