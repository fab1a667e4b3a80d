#!/bin/bash
# GPU visit: tr-read probe, all parity tests, smoke, default bench (both legs), rocprof of the graph-mode bench.
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 ./scripts/tr_probe > $OUT/tr_probe.log 2>&1; head -20 $OUT/tr_probe.log
echo "== pytest gpu"
timeout 900 python -m pytest tests -m gpu -q -x --timeout=900 -p no:cacheprovider 2>&1 | tail -15 > $OUT/r2_pytest.log; tail -6 $OUT/r2_pytest.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
echo "== bench"
timeout 600 python bench.py > $OUT/r2_bench.log 2>&1; tail -1 $OUT/r2_bench.log | cut -c1-3000
echo "== rocprof"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/r2prof -o bench -- python $OLDPWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hifigan > $OLDPWD/$OUT/r2_rocprof.log 2>&1 )
f=$(find $OUT/r2prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f" | cut -c1-200 > $OUT/r2_kernel_stats_top.csv
find $OUT/r2prof -name "*trace.csv" -size +6M -delete 2>/dev/null
echo done
