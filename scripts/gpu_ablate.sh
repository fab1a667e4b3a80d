#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export KANTTS_LIB=$PWD/kan-tts_amd/variants/libkantts_CWDBG.so
for m in 0 1 2 4 8 3 12 15; do KANTTS_CW_DBG=$m timeout 120 python scripts/conv_ablate.py 2>&1 | grep mask; done | tee $OUT/ablate.log
