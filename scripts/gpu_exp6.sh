#!/bin/bash
cd "$(dirname "$0")/.."
timeout 400 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout=400 -p no:cacheprovider 2>&1 | grep -E "^E|passed|failed|FAILED" | cut -c1-300 | head
for v in "X=1" "KANTTS_GEMM_NOONE=1"; do
  echo "== $v"; env $v timeout 120 python scripts/gemm_probe.py 2>&1 | grep -v -i "warn\|amdgpu.ids" | grep -v "wgrad\|matmul\|copy"
  env $v timeout 300 python bench.py --no-cpu-baseline --no-hifigan 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('ms/step %.2f'%d['ms_per_step'], 'gemm_ms_eager %.2f'%r['gemm_ms_per_step_eager_events'])"
done
