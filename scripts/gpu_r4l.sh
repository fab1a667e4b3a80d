#!/bin/bash
mkdir -p gpurun_out
timeout 900 python scripts/tn_sweep.py 2>&1 | grep -v Warn | tee gpurun_out/r4l_tn_sweep.log
timeout 300 python -m pytest tests/test_gpu_bf16_ops.py -m gpu -q -x -k "output_tiles" 2>&1 | tail -3
