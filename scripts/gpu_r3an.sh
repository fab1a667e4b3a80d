#!/bin/bash
# round-3 visit AN: memory K/V projections beside the decoder's entry projection (A/B)
mkdir -p gpurun_out
A="--steps 20 --warmup 5 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline"
for v in "X=1" "KANTTS_HKV_BESIDE=1" "X=2" "KANTTS_HKV_BESIDE=1"; do
  env $v timeout 300 python bench.py $A 2> gpurun_out/r3an_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'step %.3f ms  forward %.3f ms' % (d['ms_per_step'], d['roofline']['forward_ms']))" | tee -a gpurun_out/r3an_hkv_beside.log
done
