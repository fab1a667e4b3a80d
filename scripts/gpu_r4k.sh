#!/bin/bash
# Round 4, visit K: 128 x 256 weight-gradient tiles: parity tests, A/B on the step (same box), kernel stats of the TN family
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_bf16_ops.py tests/test_trainer.py tests/test_bench_config_parity.py -m gpu -q -x -k "output_tiles or sambert" 2>&1 | tail -3 | tee gpurun_out/r4k_tests.log
A="--steps 40 --warmup 10 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline --no-forward-only"
for v in "X=1" "KANTTS_TN_TILE=64" "X=2" "KANTTS_TN_TILE=64" "X=3"; do
  env $v timeout 300 python bench.py $A 2> gpurun_out/r4k_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'step %.3f ms  loss %.5f' % (d['ms_per_step'], d['config']['final_loss']))" | tee -a gpurun_out/r4k_step_ab.log
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4k_prof -o sb -- python $R/bench.py --steps 20 --warmup 5 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline --no-forward-only > /dev/null 2> $R/gpurun_out/r4k_prof_err.log
cd $R
f=$(find gpurun_out/r4k_prof -name "*kernel_stats.csv" | head -n 1)
[ -n "$f" ] && head -n 60 "$f" > gpurun_out/r4k_sambert_kernel_stats_top.csv
rm -rf gpurun_out/r4k_prof
grep bgemm_tn gpurun_out/r4k_sambert_kernel_stats_top.csv | cut -c1-160
