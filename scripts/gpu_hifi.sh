#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 500 python scripts/hifigan_bench.py 32 2 bf16 > $OUT/h_bench.log 2>&1; grep -v -i warn $OUT/h_bench.log | tail -3
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/hprof -o hifi -- python $OLDPWD/scripts/hifigan_bench.py 8 1 bf16 > $OLDPWD/$OUT/h_rocprof.log 2>&1 )
f=$(find $OUT/hprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -22 "$f" | cut -c1-170 > $OUT/h_kernel_stats_top.csv && cat $OUT/h_kernel_stats_top.csv
find $OUT/hprof -name "*trace.csv" -size +6M -delete 2>/dev/null
echo done
