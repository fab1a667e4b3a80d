#!/bin/bash
# round-3 visit AO: is the batch-32 fp32 HiFi-GAN gradient mismatch of the closing run reproducible?  (alone, repeated, with and without branch streams)
mkdir -p gpurun_out
for i in 1 2 3; do
  timeout 300 python -m pytest tests/test_bench_config_parity.py -m gpu -q -k "batch32 and fp32" 2>&1 | grep -E "passed|failed|G_grad_norm_worst" | cut -c1-260 | tee -a gpurun_out/r3ao_repro.log
done
KANTTS_NO_BRANCH_STREAMS=1 timeout 300 python -m pytest tests/test_bench_config_parity.py -m gpu -q -k "batch32 and fp32" 2>&1 | grep -E "passed|failed|G_grad_norm_worst" | cut -c1-260 | tee -a gpurun_out/r3ao_repro.log
timeout 600 python -m pytest tests/test_bench_config_parity.py -m gpu -q 2>&1 | grep -E "passed|failed|G_grad_norm_worst" | cut -c1-260 | tee -a gpurun_out/r3ao_repro.log
