#!/bin/bash
# op parity + GEMM probe + bench
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_ops.py tests/test_melspec.py tests/test_hifigan.py -m gpu -q -n 3 --timeout=400 -p no:cacheprovider 2>&1 | tail -25 > $OUT/q_pytest_ops.log; tail -6 $OUT/q_pytest_ops.log
timeout 400 python -m pytest tests/test_gpu_sambert.py -m gpu -q -n 2 --timeout=380 -p no:cacheprovider 2>&1 | tail -12 > $OUT/q_pytest_sambert.log; tail -4 $OUT/q_pytest_sambert.log
timeout 200 python scripts/gemm_probe.py > $OUT/p_probe2.log 2>&1; grep -v -i warn $OUT/p_probe2.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/q_bench_bf16.log 2>&1; tail -1 $OUT/q_bench_bf16.log
echo done
