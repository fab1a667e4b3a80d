#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hifigan.py -m gpu -q -x --timeout=600 -p no:cacheprovider 2>&1 | tail -3
export KANTTS_LIB=$PWD/kan-tts_amd/variants/libkantts_CWDBG.so
for m in 0 2 15; do KANTTS_CW_DBG=$m timeout 120 python scripts/conv_ablate.py 2>&1 | grep mask; done | tee $OUT/ablate2.log
