#!/bin/bash
# Round 4, visit I: the side branch's gradients joined late (predictors' backward beside the decoder's): parity tests + A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_trainer.py tests/test_bench_config_parity.py -m gpu -q -x -k "sambert" 2>&1 | tail -3 | tee gpurun_out/r4i_tests.log
A="--steps 40 --warmup 10 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline --no-forward-only"
for v in "X=1" "KANTTS_NO_LATE_SIDE_GRAD=1" "X=2" "KANTTS_NO_LATE_SIDE_GRAD=1" "X=3"; do
  env $v timeout 300 python bench.py $A 2> gpurun_out/r4i_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'step %.3f ms  loss %.5f' % (d['ms_per_step'], d['config']['final_loss']))" | tee -a gpurun_out/r4i_step_ab.log
done
