#!/bin/bash
# round-3 visit T: per-node cost of a replayed graph; stacked memory-K/V projection (parity + step time)
mkdir -p gpurun_out
timeout 120 python scripts/graph_gap_probe.py 2>&1 | grep -v Warning | tee gpurun_out/r3t_graph_gap.log
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sambert.py tests/test_gpu_bf16_ops.py tests/test_trainer.py -m gpu -x -q 2>&1 | tail -n 6
A="--steps 20 --warmup 5 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline"
for v in "" "KANTTS_NO_SHARED_ONE_GEMM=1"; do
  env $v timeout 300 python bench.py $A 2> gpurun_out/r3t_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'step %.3f ms  forward %.3f ms' % (d['ms_per_step'], d['roofline']['forward_ms']))" | tee -a gpurun_out/r3t_bench.log
done
