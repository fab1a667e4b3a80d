#!/bin/bash
# round-3 visit Y: FSMN filter gradients on the weight-gradient stream (parity + step time); LayerNorm backward vs workgroup cap
mkdir -p gpurun_out
for c in "" 64 192 256 408; do
  env ${c:+KANTTS_LN_BWD_BLOCKS=$c} timeout 100 python scripts/ln_bwd_probe.py 2>&1 | grep -v Warning | grep blocks | tee -a gpurun_out/r3y_ln_bwd_blocks.log
done
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sambert.py tests/test_trainer.py -m gpu -x -q 2>&1 | tail -n 4
A="--steps 20 --warmup 5 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline"
timeout 300 python bench.py $A 2> gpurun_out/r3y_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step %.3f ms  forward %.3f ms' % (d['ms_per_step'], d['roofline']['forward_ms']))" | tee -a gpurun_out/r3y_bench.log
