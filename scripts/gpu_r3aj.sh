#!/bin/bash
# round-3 visit AJ: extra flush points for the deferred weight gradients, re-measured on the round's final schedule
mkdir -p gpurun_out
A="--steps 20 --warmup 5 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline"
for v in "X=1" "KANTTS_FLUSH_EVERY_ENC=2" "KANTTS_FLUSH_EVERY_ENC=4" "KANTTS_FLUSH_EVERY_DEC=4" "KANTTS_FLUSH_EVERY_DEC=3 KANTTS_FLUSH_EVERY_ENC=2" "KANTTS_NO_EARLY_FLUSH=1"; do
  env $v timeout 300 python bench.py $A 2> gpurun_out/r3aj_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'step %.3f ms  forward %.3f ms' % (d['ms_per_step'], d['roofline']['forward_ms']))" | tee -a gpurun_out/r3aj_flush_points.log
done
