#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sambert.py -m gpu -q -x --timeout=600 -p no:cacheprovider 2>&1 | tail -3
export KANTTS_LIB=$PWD/kan-tts_amd/variants/libkantts_GDBG.so
for m in 0 2 11; do KANTTS_GEMM_DBG=$m timeout 120 python scripts/gemm_ablate.py 2>&1 | grep mask; done | tee $OUT/gablate2.log
