#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
run() { echo "== $1"; env $1 timeout 300 python bench.py --no-cpu-baseline --no-hifigan 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('ms/step %.2f'%d['ms_per_step'], 'gemm_ms_eager %.2f'%r['gemm_ms_per_step_eager_events'], r['launch_us'])"; }
run "X=1"
run "KANTTS_GEMM_BIGK=1"
run "KANTTS_GEMM_BIGK=0"
run "KANTTS_GEMM_BM=3"
run "KANTTS_GEMM_BM=6"
