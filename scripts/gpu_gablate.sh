#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export KANTTS_LIB=$PWD/kan-tts_amd/variants/libkantts_GDBG.so
for m in 0 1 2 8 3 11; do KANTTS_GEMM_DBG=$m timeout 120 python scripts/gemm_ablate.py 2>&1 | grep mask; done | tee $OUT/gablate.log
