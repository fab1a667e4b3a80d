#!/bin/bash
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sambert.py -m gpu -q -x --timeout=900 -p no:cacheprovider 2>&1 | grep -E "^E|passed|failed|FAILED" | cut -c1-300 | head -20
timeout 400 python bench.py --no-cpu-baseline > $OUT/q7_bench.log 2>&1; tail -1 $OUT/q7_bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; h=d['hifigan']
print('sambert ms/step %.2f'%d['ms_per_step'], 'frac %.3f'%r['frac'], r['launch_us'])
print('hifigan gan_step_ms %.1f'%h['gan_step_ms'], 'G fwd ms %.2f'%h['generator_forward_ms'], 'upsampling', h['upsampling']['ms'], h['upsampling']['frac'], h['upsampling'].get('stage_us'))"
