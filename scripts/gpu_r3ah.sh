#!/bin/bash
# round-3 visit AH: which gradient-carrying pieces pay beside the encoder (the target-only plan always runs there)
mkdir -p gpurun_out
A="--steps 20 --warmup 5 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline"
for v in "X=1" "KANTTS_BESIDE_PARTS=emb" "KANTTS_BESIDE_PARTS=pe" "KANTTS_BESIDE_PARTS=prenet" "KANTTS_BESIDE_PARTS=emb,pe,prenet" "X=2"; do
  env $v timeout 300 python bench.py $A 2> gpurun_out/r3ah_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'step %.3f ms  forward %.3f ms' % (d['ms_per_step'], d['roofline']['forward_ms']))" | tee -a gpurun_out/r3ah_beside_parts.log
done
timeout 600 python -m pytest tests/test_trainer.py -m gpu -x -q 2>&1 | tail -n 3
