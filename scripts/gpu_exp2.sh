#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_sambert.py -m gpu -q -x --timeout=600 -p no:cacheprovider -k wgrad 2>&1 | grep -E "^E|passed|failed|FAILED" | cut -c1-300 | head
run() { echo "== $1"; timeout 300 python bench.py --no-cpu-baseline --no-hifigan $1 2>$OUT/exp2_err.log | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('ms/step %.2f'%d['ms_per_step'], d['config']['launch'], 'loss', d['config']['final_loss'])"; grep -i "capture failed" $OUT/exp2_err.log | head -2; }
run ""
run ""
run "--mode eager"
run "--mode eager --no-wgrad-overlap"
