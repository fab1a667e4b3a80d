#!/bin/bash
# round-3 visit V: query gradient of both PNCA bands in one pass, 32-row tiles for fp32 A operands: parity + step time + kernel statistics
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sambert.py tests/test_gpu_bf16_ops.py tests/test_trainer.py -m gpu -x -q 2>&1 | tail -n 6
A="--steps 20 --warmup 5 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline"
for v in "" "KANTTS_ATTN_NO_DQ2=1"; do
  env $v timeout 300 python bench.py $A 2> gpurun_out/r3v_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'step %.3f ms  forward %.3f ms' % (d['ms_per_step'], d['roofline']['forward_ms']))" | tee -a gpurun_out/r3v_bench.log
done
