#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hifigan.py -m gpu -q -x --timeout=600 -p no:cacheprovider 2>&1 | tail -3
for cfg in "W32 LDS=52" "W16 LDS=52" "W16 LDS=78"; do
  unset KANTTS_CONV_TB KANTTS_CONV_LDS_KB KANTTS_CONV_W32
  for t in $cfg; do case $t in TB=*) export KANTTS_CONV_TB=${t#TB=};; LDS=*) export KANTTS_CONV_LDS_KB=${t#LDS=};; W32) export KANTTS_CONV_W32=1;; esac; done
  echo "== $cfg"; timeout 300 python scripts/hifigan_bench.py 32 2 bf16 2>&1 | grep -v -i warn | tail -1 | cut -c1-560
done | tee $OUT/cw2.log
