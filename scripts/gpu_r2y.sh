#!/bin/bash
# round-2 visit Y: extra flush points inside the decoder / encoder block stacks (A/B over the spacing)
mkdir -p gpurun_out
for cfg in "0 0" "4 0" "6 0" "3 0" "4 4" "2 2"; do
  set -- $cfg
  KANTTS_FLUSH_EVERY_DEC=$1 KANTTS_FLUSH_EVERY_ENC=$2 timeout 200 python bench.py --steps 20 --warmup 5 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference > gpurun_out/r2y_bench_$1_$2.log 2>&1
  echo "dec=$1 enc=$2: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2y_bench_$1_$2.log | head -1) $(grep -o '"launch": "[a-z]*"' gpurun_out/r2y_bench_$1_$2.log | head -1)"
done
