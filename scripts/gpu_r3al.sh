#!/bin/bash
# round-3 visit AL: embedding-table gradients through LDS images (parity + step time); predictors beside the postnet as default
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sambert.py tests/test_trainer.py -m gpu -q 2>&1 | tail -n 4
A="--steps 20 --warmup 5 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline"
for v in "X=1" "X=2"; do
  env $v timeout 300 python bench.py $A 2> gpurun_out/r3al_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'step %.3f ms  forward %.3f ms' % (d['ms_per_step'], d['roofline']['forward_ms']))" | tee -a gpurun_out/r3al_bench.log
done
