#!/bin/bash
# Round 4, visit AE: one-output-channel convolutions on csrc/conv_n1.hip: parity on the device, GAN step A/B, kernel averages
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hifigan.py tests/test_trainer.py tests/test_bench_config_parity.py tests/test_hifigan_nsf.py tests/test_multiband.py -m gpu -q -x 2>&1 | tail -3 | tee gpurun_out/r4ae_tests.log
for v in "X=1" "KANTTS_NO_CONV_N1=1" "X=2" "KANTTS_NO_CONV_N1=1"; do
  env $v timeout 300 python scripts/hifigan_bench.py 32 3 bf16 2> gpurun_out/r4ae_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'gan step graph %.2f ms  eager %.2f ms  G fwd %.3f ms' % (d.get('gan_step_graph_ms',-1), d['gan_step_ms'], d['generator_forward_ms']))" | tee -a gpurun_out/r4ae_gan_ab.log
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4ae_prof -o gan -- python $R/scripts/hifigan_bench.py 32 3 bf16 > /dev/null 2> $R/gpurun_out/r4ae_prof_err.log
cd $R
f=$(find gpurun_out/r4ae_prof -name "*kernel_stats.csv" | head -n 1)
[ -n "$f" ] && head -n 80 "$f" > gpurun_out/r4ae_gan_kernel_stats_top.csv
rm -rf gpurun_out/r4ae_prof
grep -E "conv_n1|conv_win|conv_direct|conv_wgrad_direct" gpurun_out/r4ae_gan_kernel_stats_top.csv | cut -d, -f1-5 | cut -c1-150
