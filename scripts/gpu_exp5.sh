#!/bin/bash
cd "$(dirname "$0")/.."
timeout 400 python -m pytest tests/test_hifigan.py -m gpu -q -x --timeout=400 -p no:cacheprovider 2>&1 | grep -E "^E|passed|failed|FAILED" | cut -c1-300 | head
python - <<'PY'
import sys, os
sys.path.insert(0, 'kan-tts_amd')
import torch, bench, kantts._hip as hip
r = bench.hifigan_leg(hip, 'bf16', steps=2)
print('upsampling', r['upsampling']['ms'], r['upsampling']['frac'], r['upsampling']['tflops'], r['upsampling']['stage_us'], 'gan', r['gan_step_ms'], 'G fwd', r['generator_forward_ms'])
PY
