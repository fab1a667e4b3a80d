#!/bin/bash
# round-3 visit AC: 80-byte LDS rows with 16-byte accesses in the attention kernels (microbench, parity, step time)
mkdir -p gpurun_out
timeout 200 python scripts/attn_bench.py 2>&1 | grep -v Warning | grep "^L" | tee -a gpurun_out/r3ac_attn_bench.log
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sambert.py -m gpu -x -q 2>&1 | tail -n 3
A="--steps 20 --warmup 5 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline"
timeout 300 python bench.py $A 2> gpurun_out/r3ac_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step %.3f ms  forward %.3f ms' % (d['ms_per_step'], d['roofline']['forward_ms']))" | tee -a gpurun_out/r3ac_bench.log
