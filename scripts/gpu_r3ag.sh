#!/bin/bash
# round-3 visit AG: emotion / speaker embeddings, pitch / energy embeddings and the decoder prenet beside the encoder as well (parity + step time)
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_sambert.py tests/test_gpu_bf16_ops.py tests/test_trainer.py tests/test_decode_graph.py tests/test_entrypoints.py -m gpu -x -q 2>&1 | tail -n 4
A="--steps 20 --warmup 5 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline"
for v in "" "KANTTS_NO_PLAN_BESIDE=1"; do
  env $v timeout 300 python bench.py $A 2> gpurun_out/r3ag_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'step %.3f ms  forward %.3f ms' % (d['ms_per_step'], d['roofline']['forward_ms']))" | tee -a gpurun_out/r3ag_bench.log
done
