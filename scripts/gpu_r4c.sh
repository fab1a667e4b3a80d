#!/bin/bash
# Round 4, visit C: data-parallel segment tests again (tolerance form), the table-driven weight norm and the persistent
# 1-channel weight gradient on the device, A/B of both on the GAN step, GAN-only kernel statistics
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ddp_gloo.py -m gpu -q -x -s 2>&1 | grep -v Warning | tail -12 | tee gpurun_out/r4c_ddp.log
timeout 900 python -m pytest tests/test_hifigan.py tests/test_trainer.py -m gpu -q -x -k "weight_norm_table or persistent_grid or graphed_gan or gan_train_step or gan_loss_curve or v1" 2>&1 | tail -5 | tee gpurun_out/r4c_gan_tests.log
for v in "X=1" "KANTTS_NO_WEIGHT_NORM_TABLE=1" "KANTTS_C1_WGRAD_WGS=1000000" "X=2"; do
  env $v timeout 300 python scripts/hifigan_bench.py 32 3 bf16 2> gpurun_out/r4c_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'gan step graph %.2f ms  eager %.2f ms  G fwd %.3f ms  up %.4f ms' % (d.get('gan_step_graph_ms',-1), d['gan_step_ms'], d['generator_forward_ms'], d['upsampling_ms']))" | tee -a gpurun_out/r4c_gan_ab.log
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4c_prof -o gan -- python $R/scripts/hifigan_bench.py 32 5 bf16 > $R/gpurun_out/r4c_rocprof.log 2>&1
cd $R
f=$(find gpurun_out/r4c_prof -name "*kernel_stats.csv" | head -n 1)
[ -n "$f" ] && head -n 100 "$f" > gpurun_out/r4c_gan_kernel_stats_top.csv
rm -rf gpurun_out/r4c_prof
tail -2 gpurun_out/r4c_rocprof.log | cut -c1-600
