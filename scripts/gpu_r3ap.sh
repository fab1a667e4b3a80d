#!/bin/bash
# round-3 visit AP: localising the branch-stream race of the fp32 generator backward (allocator reuse vs missing dependency)
mkdir -p gpurun_out
run() { echo "== $*" | tee -a gpurun_out/r3ap_race.log; env "$@" timeout 300 python -m pytest tests/test_bench_config_parity.py -m gpu -q -k "batch32 and fp32" 2>&1 | grep -E "passed|failed|^HiFi-GAN" | cut -c1-230 | tee -a gpurun_out/r3ap_race.log; }
run X=1
run PYTORCH_NO_CUDA_MEMORY_CACHING=1
run PYTORCH_NO_CUDA_MEMORY_CACHING=1
run KANTTS_NO_WEIGHT_IMAGES=1
run KANTTS_NO_WEIGHT_IMAGES=1
run KANTTS_C1_NO_MFMA=1
