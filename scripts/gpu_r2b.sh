#!/bin/bash
# round-2 visit B: first run of the bf16-operand kernels (csrc/gemm_bf16.hip): per-op parity vs the emulated ABI,
# microbenchmark vs the round-1 kernel, parity at the benchmarked configs, short bench
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_bf16_ops.py -x -q > gpurun_out/r2b_ops.log 2>&1
echo "ops rc=$?" >> gpurun_out/r2b_ops.log
tail -15 gpurun_out/r2b_ops.log
timeout 300 python scripts/bgemm_bench.py > gpurun_out/r2b_bgemm.log 2>&1
echo "bgemm rc=$?" >> gpurun_out/r2b_bgemm.log
cat gpurun_out/r2b_bgemm.log
timeout 600 python -m pytest tests/test_bench_config_parity.py -x -q -s > gpurun_out/r2b_parity.log 2>&1
echo "parity rc=$?" >> gpurun_out/r2b_parity.log
grep -E "SAM-BERT full|HiFi-GAN V1|passed|failed|rc=" gpurun_out/r2b_parity.log | cut -c1-900
timeout 420 python bench.py --steps 10 --warmup 3 > gpurun_out/r2b_bench.log 2> gpurun_out/r2b_bench.err
echo "bench rc=$?" >> gpurun_out/r2b_bench.log
tail -c 6000 gpurun_out/r2b_bench.log
tail -5 gpurun_out/r2b_bench.err | cut -c1-300
