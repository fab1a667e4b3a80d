#!/bin/bash
# round-3 visit O: memory-side traffic of the feed-forward launches (final kernels), 2-rank bench on one device over gloo, full bench line
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/r3o_pmc_$c
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/r3o_pmc_$c -o pmc -- python $R/scripts/ffn_pmc_probe.py > $R/gpurun_out/r3o_pmc_$c.log 2>&1
  f=$(find $R/gpurun_out/r3o_pmc_$c -name "*counter_collection.csv" | head -n 1)
  [ -n "$f" ] && cp "$f" $R/gpurun_out/r3o_$c.csv && python $R/scripts/pmc_summary.py "$f" bgemm ffn_pair > $R/gpurun_out/r3o_ffn_$c.txt
  rm -rf $R/gpurun_out/r3o_pmc_$c
done
python $R/scripts/pmc_to_json.py $R/gpurun_out/r3o_FETCH_SIZE.csv $R/gpurun_out/r3o_WRITE_SIZE.csv "round 3 visit O: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python scripts/ffn_pmc_probe.py" $R/gpurun_out/r3o_ffn_block_pmc.json | tail -n 40
rm -f $R/gpurun_out/r3o_FETCH_SIZE.csv $R/gpurun_out/r3o_WRITE_SIZE.csv
cd $R
timeout 600 python bench.py --gpus 2 --backend gloo --share-device --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3o_bench_2rank_gloo.log 2>&1
echo "2-rank exit $?"; tail -c 1500 gpurun_out/r3o_bench_2rank_gloo.log
timeout 1200 python bench.py > gpurun_out/r3o_bench_full.log 2>&1
echo "bench exit $?"; tail -c 9000 gpurun_out/r3o_bench_full.log
