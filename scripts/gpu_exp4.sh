#!/bin/bash
cd "$(dirname "$0")/.."
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_hifigan.py -m gpu -q -x --timeout=300 -p no:cacheprovider -k "linear or conv_ops or hifigan_gpu" 2>&1 | grep -E "^E|passed|failed|FAILED" | cut -c1-300 | head
for v in "X=1" "KANTTS_GEMM_BM=1"; do
  echo "== $v"; env $v timeout 120 python scripts/gemm_probe.py 2>&1 | grep -v -i "warn\|amdgpu.ids" | grep -v "wgrad\|matmul\|copy"
done
python - <<'PY'
import sys, os
sys.path.insert(0, 'kan-tts_amd')
import torch, bench, kantts._hip as hip
r = bench.hifigan_leg(hip, 'bf16', steps=1)
print('upsampling', r['upsampling']['ms'], r['upsampling']['stage_us'], 'gan', r['gan_step_ms'])
PY
