#!/bin/bash
# Round 5, fourth visit: fused PNCA block forward + row-local backward on the device: parity, kernel durations (rocprofv3),
# step A/B on one box (fused both ways / fused forward only / chain).
T=${1:-r5d}
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest -q -x -m gpu tests/test_pnca_block.py tests/test_bench_config_parity.py tests/test_gpu_sambert.py \
  tests/test_trainer.py tests/test_decode_graph.py -k "not hifigan" > gpurun_out/${T}_tests.log 2>&1; echo "tests exit $?"; tail -n 4 gpurun_out/${T}_tests.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof -o blk -- python $R/scripts/pnca_block_bench.py 20 > $R/gpurun_out/${T}_block_bench_prof.log 2>&1
f=$(find $R/gpurun_out/${T}_prof -name "*kernel_stats.csv" | head -n 1)
[ -n "$f" ] && head -n 40 "$f" > $R/gpurun_out/${T}_block_bench_kernel_stats_top.csv
rm -rf $R/gpurun_out/${T}_prof
grep blocks $R/gpurun_out/${T}_block_bench_prof.log
cut -d, -f1-4,7-9 $R/gpurun_out/${T}_block_bench_kernel_stats_top.csv | head -n 24
cd $R
for rep in 1 2; do
  for v in fused fwdonly chain; do
    unset KANTTS_NO_PNCA_BLOCK KANTTS_NO_PNCA_BLOCK_BWD
    [ $v = chain ] && export KANTTS_NO_PNCA_BLOCK=1
    [ $v = fwdonly ] && export KANTTS_NO_PNCA_BLOCK_BWD=1
    timeout 300 python bench.py --no-hifigan --no-inference --no-cpu-baseline --no-fp32 --no-roofline --steps 40 \
      > gpurun_out/${T}_bench_${v}_${rep}.json 2> gpurun_out/${T}_bench_${v}_${rep}.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/${T}_bench_${v}_${rep}.json").read().strip().splitlines()[-1])
print("$v $rep ms_per_step %.3f forward_ms %s" % (d["ms_per_step"], d["roofline"].get("forward_ms")))
PY
  done
done
unset KANTTS_NO_PNCA_BLOCK KANTTS_NO_PNCA_BLOCK_BWD
