#!/bin/bash
# quick visit: op parity + bench + kernel profile
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_ops.py tests/test_melspec.py tests/test_hifigan.py -m gpu -q -n 3 --timeout=400 -p no:cacheprovider 2>&1 | tail -25 > $OUT/q_pytest_ops.log; tail -6 $OUT/q_pytest_ops.log
timeout 400 python -m pytest tests/test_gpu_sambert.py -m gpu -q -n 2 --timeout=380 -p no:cacheprovider 2>&1 | tail -12 > $OUT/q_pytest_sambert.log; tail -4 $OUT/q_pytest_sambert.log
timeout 300 python scripts/micro.py attn > $OUT/q_micro.log 2>&1; grep -v Warning $OUT/q_micro.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/q_bench_bf16.log 2>&1; tail -1 $OUT/q_bench_bf16.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/qprof -o bench -- python $OLDPWD/bench.py --steps 5 --warmup 2 --mode eager --no-cpu-baseline > $OLDPWD/$OUT/q_rocprof.log 2>&1 )
f=$(find $OUT/qprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -32 "$f" | cut -c1-150 > $OUT/q_kernel_stats_top.csv && cat $OUT/q_kernel_stats_top.csv
find $OUT/qprof -name "*trace.csv" -size +6M -delete 2>/dev/null
echo done
