#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hifigan.py -m gpu -q -x --timeout=600 -p no:cacheprovider 2>&1 | tail -5
for cfg in "TB=1" "LDS=78" "LDS=52" "LDS=120"; do
  unset KANTTS_CONV_TB KANTTS_CONV_LDS_KB
  case $cfg in TB=*) export KANTTS_CONV_TB=${cfg#TB=};; LDS=*) export KANTTS_CONV_LDS_KB=${cfg#LDS=};; esac
  echo "== $cfg"; timeout 300 python scripts/hifigan_bench.py 32 2 bf16 2>&1 | grep -v -i warn | tail -1 | cut -c1-700
done | tee $OUT/cw1.log
