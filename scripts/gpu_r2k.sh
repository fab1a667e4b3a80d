#!/bin/bash
# round-2 visit L: the fused feed-forward pair v2 (fragment-major weights, 8 waves, 4-deep ring): parity, microbenchmark, A/B
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_bf16_ops.py -m gpu -x -q > gpurun_out/r2l_pytest_ops.log 2>&1; tail -4 gpurun_out/r2l_pytest_ops.log
timeout 200 python scripts/ffn_pair_bench.py > gpurun_out/r2l_ffn_pair_bench.log 2>&1; cat gpurun_out/r2l_ffn_pair_bench.log | tail -14
timeout 900 python -m pytest tests/test_gpu_sambert.py "tests/test_bench_config_parity.py::test_sambert_full_b32_matches_oracle" -m gpu -x -q > gpurun_out/r2l_pytest_model.log 2>&1; tail -4 gpurun_out/r2l_pytest_model.log
for v in 1 ""; do
  KANTTS_NO_FFN_PAIR=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference > gpurun_out/r2l_bench_nopair_$v.log 2>&1
  echo "KANTTS_NO_FFN_PAIR='$v': $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2l_bench_nopair_$v.log | head -1)"
done
