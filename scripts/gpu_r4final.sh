#!/bin/bash
# round-4 final visit: full GPU test suite, smoke, the default bench line, kernel statistics of the bench command, the PMC
# passes of the feed-forward block and of the mel-STFT kernel, the 2-rank bench on one device over gloo
T=${1:-r4f}
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -n 3 gpurun_out/${T}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; echo "smoke exit $?"; grep -v Warning gpurun_out/${T}_smoke.log | tail -n 3
timeout 1500 python bench.py > gpurun_out/${T}_bench_full.log 2> gpurun_out/${T}_bench_full.err; echo "bench exit $?"; grep "^\[bench" gpurun_out/${T}_bench_full.err | tail -n 20
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof -o bench -- python $R/bench.py --no-cpu-baseline --no-inference --no-fp32 > $R/gpurun_out/${T}_rocprof_bench.log 2>&1
f=$(find $R/gpurun_out/${T}_prof -name "*kernel_stats.csv" | head -n 1)
[ -n "$f" ] && head -n 140 "$f" > $R/gpurun_out/${T}_bench_kernel_stats_top.csv
rm -rf $R/gpurun_out/${T}_prof
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/${T}_pmc_$c
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${T}_pmc_$c -o pmc -- python $R/scripts/ffn_pmc_probe.py > $R/gpurun_out/${T}_pmc_$c.log 2>&1
  f=$(find $R/gpurun_out/${T}_pmc_$c -name "*counter_collection.csv" | head -n 1)
  [ -n "$f" ] && cp "$f" $R/gpurun_out/${T}_$c.csv
  rm -rf $R/gpurun_out/${T}_pmc_$c
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${T}_melpmc_$c -o pmc -- python $R/scripts/mel_pmc.py > $R/gpurun_out/${T}_melpmc_$c.log 2>&1
  f=$(find $R/gpurun_out/${T}_melpmc_$c -name "*counter_collection.csv" | head -n 1)
  [ -n "$f" ] && python $R/scripts/pmc_summary.py "$f" melspec > $R/gpurun_out/${T}_mel_$c.txt
  rm -rf $R/gpurun_out/${T}_melpmc_$c
done
python $R/scripts/pmc_to_json.py $R/gpurun_out/${T}_FETCH_SIZE.csv $R/gpurun_out/${T}_WRITE_SIZE.csv "round 4 final visit: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python scripts/ffn_pmc_probe.py" $R/gpurun_out/${T}_ffn_block_pmc.json | tail -n 12
rm -f $R/gpurun_out/${T}_FETCH_SIZE.csv $R/gpurun_out/${T}_WRITE_SIZE.csv
cat $R/gpurun_out/${T}_mel_FETCH_SIZE.txt $R/gpurun_out/${T}_mel_WRITE_SIZE.txt
cd $R
timeout 400 python bench.py --gpus 2 --backend gloo --share-device --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${T}_bench_2rank_gloo.log 2>&1; echo "2-rank exit $?"
tail -c 1200 gpurun_out/${T}_bench_full.log
