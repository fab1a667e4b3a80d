#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hifigan.py tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -8
bash scripts/gpu_hifi.sh
