#!/bin/bash
# round-3 visit AE: LayerNorm backward block partials through agent-scope atomic stores / loads (no fences)
mkdir -p gpurun_out
timeout 100 python scripts/ln_bwd_probe.py 2>&1 | grep -v Warning | grep blocks | tee gpurun_out/r3ae_ln_bwd.log
KANTTS_LN_BWD_BLOCKS=128 timeout 100 python scripts/ln_bwd_probe.py 2>&1 | grep -v Warning | grep blocks | tee -a gpurun_out/r3ae_ln_bwd.log
timeout 300 python -m pytest tests/test_gpu_bf16_ops.py -m gpu -x -q -k "layer_norm" 2>&1 | tail -n 2
