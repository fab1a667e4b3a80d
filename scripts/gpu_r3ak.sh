#!/bin/bash
# round-3 visit AK: variance predictors beside the postnet instead of beside the decoder
mkdir -p gpurun_out
A="--steps 20 --warmup 5 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline"
for v in "X=1" "KANTTS_PREDICTORS_LATE=1" "X=2" "KANTTS_PREDICTORS_LATE=1"; do
  env $v timeout 300 python bench.py $A 2> gpurun_out/r3ak_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'step %.3f ms  forward %.3f ms  loss %.5f' % (d['ms_per_step'], d['roofline']['forward_ms'], d['config']['final_loss']))" | tee -a gpurun_out/r3ak_predictors_late.log
done
