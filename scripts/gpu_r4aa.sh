#!/bin/bash
# Round 4, visit AA: bf16 image from the 1-channel first layers: parity, GAN step, act_cast count
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hifigan.py tests/test_trainer.py tests/test_bench_config_parity.py tests/test_conv_sweep.py -m gpu -q -x -k "gan or hifigan or hands_over or one_channel or conv" 2>&1 | tail -3 | tee gpurun_out/r4aa_tests.log
for v in "X=1" "X=2" "X=3"; do
  env $v timeout 300 python scripts/hifigan_bench.py 32 3 bf16 2> gpurun_out/r4aa_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'gan step graph %.2f ms  eager %.2f ms  G fwd %.3f ms' % (d.get('gan_step_graph_ms',-1), d['gan_step_ms'], d['generator_forward_ms']))" | tee -a gpurun_out/r4aa_gan.log
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4aa_prof -o gan -- python $R/scripts/hifigan_bench.py 32 3 bf16 > /dev/null 2> $R/gpurun_out/r4aa_prof_err.log
cd $R
f=$(find gpurun_out/r4aa_prof -name "*kernel_stats.csv" | head -n 1)
[ -n "$f" ] && head -n 70 "$f" > gpurun_out/r4aa_gan_kernel_stats_top.csv
rm -rf gpurun_out/r4aa_prof
grep -E "act_cast|conv_c1_fwd|FillFunctor" gpurun_out/r4aa_gan_kernel_stats_top.csv | cut -d, -f1-5 | cut -c1-60,120-200
