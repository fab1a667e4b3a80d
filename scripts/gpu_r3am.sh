#!/bin/bash
# round-3 visit AM: embedding-table gradient forms alone; step time with the token-sliced LDS form and the vectorised masked L1
# (scripts/embed_bwd_probe.py, used by this visit, was removed with the experiment: see profiles/r03_runAM_embed_bwd_forms.log)
mkdir -p gpurun_out
timeout 100 python scripts/embed_bwd_probe.py 2>&1 | grep -v Warning | grep tables | tee gpurun_out/r3am_embed_bwd.log
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "embedding or masked_l1" 2>&1 | tail -n 2
A="--steps 20 --warmup 5 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline"
for v in "X=1" "X=2"; do
  env $v timeout 300 python bench.py $A 2> gpurun_out/r3am_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'step %.3f ms  forward %.3f ms' % (d['ms_per_step'], d['roofline']['forward_ms']))" | tee -a gpurun_out/r3am_bench.log
done
