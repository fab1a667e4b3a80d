#!/bin/bash
# Round 4, visit AD: accumulator pool re-zeroes only its dirty prefix: parity (graph == eager, loss curves, resume), steps
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_hifigan.py tests/test_trainer.py tests/test_bench_config_parity.py tests/test_gpu_bf16_ops.py tests/test_ddp_gloo.py -m gpu -q -x 2>&1 | tail -3 | tee gpurun_out/r4ad_tests.log
for v in "X=1" "X=2"; do
  env $v timeout 300 python scripts/hifigan_bench.py 32 3 bf16 2> gpurun_out/r4ad_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'gan step graph %.2f ms  eager %.2f ms  G fwd %.3f ms' % (d.get('gan_step_graph_ms',-1), d['gan_step_ms'], d['generator_forward_ms']))" | tee -a gpurun_out/r4ad_gan.log
done
A="--steps 40 --warmup 10 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline --no-forward-only"
for v in "X=1" "X=2"; do
  env $v timeout 300 python bench.py $A 2> gpurun_out/r4ad_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'step %.3f ms  loss %.5f' % (d['ms_per_step'], d['config']['final_loss']))" | tee -a gpurun_out/r4ad_step.log
done
