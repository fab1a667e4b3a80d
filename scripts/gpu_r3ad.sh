#!/bin/bash
# round-3 visit AD: LayerNorm backward with block partials + last-block reduce instead of same-address atomics
mkdir -p gpurun_out
timeout 100 python scripts/ln_bwd_probe.py 2>&1 | grep -v Warning | grep blocks | tee gpurun_out/r3ad_ln_bwd.log
KANTTS_LN_BWD_BLOCKS=128 timeout 100 python scripts/ln_bwd_probe.py 2>&1 | grep -v Warning | grep blocks | tee -a gpurun_out/r3ad_ln_bwd.log
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sambert.py tests/test_gpu_bf16_ops.py tests/test_trainer.py -m gpu -x -q 2>&1 | tail -n 4
A="--steps 20 --warmup 5 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline"
for v in "" "KANTTS_LN_BWD_ATOMICS=1"; do
  env $v timeout 300 python bench.py $A 2> gpurun_out/r3ad_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'step %.3f ms  forward %.3f ms  loss %.6f' % (d['ms_per_step'], d['roofline']['forward_ms'], d['config']['final_loss']))" | tee -a gpurun_out/r3ad_bench.log
done
