#!/bin/bash
# round-3 visit W: does HIP stream priority keep the side streams (weight gradients, predictor branch) out of the critical path's way?
mkdir -p gpurun_out
python -c "
import torch
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')
for p in (-2,-1,0,1,2):
    try:
        s=torch.cuda.Stream(priority=p); print(p,'->',s.priority)
    except Exception as e: print(p,'error',e)
" 2>&1 | grep -v Warning | tee gpurun_out/r3w_priority.log
A="--steps 20 --warmup 5 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline"
for v in "" "KANTTS_SIDE_PRIORITY=1" "KANTTS_MAIN_PRIORITY=-1" "KANTTS_SIDE_PRIORITY=1 KANTTS_MAIN_PRIORITY=-1"; do
  env $v timeout 300 python bench.py $A 2> gpurun_out/r3w_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'step %.3f ms  forward %.3f ms' % (d['ms_per_step'], d['roofline']['forward_ms']))" | tee -a gpurun_out/r3w_priority.log
done
