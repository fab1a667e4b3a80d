#!/bin/bash
# Round 4, visit R: embedding-table gradient with an accumulator cache: parity on the device, step time
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_bench_config_parity.py tests/test_trainer.py -m gpu -q -x -k "embed or sambert" 2>&1 | tail -3 | tee gpurun_out/r4r_tests.log
A="--steps 40 --warmup 10 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline --no-forward-only"
for v in "X=1" "X=2" "X=3"; do
  env $v timeout 300 python bench.py $A 2> gpurun_out/r4r_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'step %.3f ms  loss %.5f' % (d['ms_per_step'], d['config']['final_loss']))" | tee -a gpurun_out/r4r_step.log
done
