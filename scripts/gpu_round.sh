#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench (bf16 + fp32), rocprofv3 kernel stats.
# Everything lands under gpurun_out/ (merged back by gpurun).
cd "$(dirname "$0")/.."
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/gpu.txt
nproc >> $OUT/gpu.txt; lscpu | grep "Model name" >> $OUT/gpu.txt
echo "== pytest gpu ops" 
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_melspec.py -m gpu -q -n 2 --timeout=600 -p no:cacheprovider -rA 2>&1 | tail -150 > $OUT/pytest_ops.log
tail -40 $OUT/pytest_ops.log
echo "== pytest gpu sambert"
timeout 600 python -m pytest tests/test_gpu_sambert.py -m gpu -q -n 2 --timeout=900 -p no:cacheprovider -rA 2>&1 | tail -150 > $OUT/pytest_sambert.log
tail -40 $OUT/pytest_sambert.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
echo "== bench"
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_bf16.log 2>&1; tail -2 $OUT/bench_bf16.log
timeout 300 python bench.py --steps 20 --warmup 5 --mode eager --no-cpu-baseline > $OUT/bench_bf16_eager.log 2>&1; tail -1 $OUT/bench_bf16_eager.log
timeout 300 python bench.py --steps 20 --warmup 5 --precision fp32 --no-cpu-baseline > $OUT/bench_fp32.log 2>&1; tail -1 $OUT/bench_fp32.log
echo "== rocprof"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o bench -- python $OLDPWD/bench.py --steps 5 --warmup 2 --mode eager --no-cpu-baseline > $OLDPWD/$OUT/rocprof.log 2>&1 )
find $OUT/prof -name "*kernel_stats*" | head -3
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" > $OUT/kernel_stats_top.csv && cat $OUT/kernel_stats_top.csv | cut -c1-180
# keep the merged-back payload small
find $OUT/prof -name "*.csv" -size +8M -delete 2>/dev/null
echo done
