#!/bin/bash
# round-3 visit I: MFMA forms of the one-channel conv gradients, fused residual stack: tests + GAN step
mkdir -p gpurun_out/r3i
timeout 900 python -m pytest tests/test_cconv.py tests/test_hifigan.py -m gpu -x -q > gpurun_out/r3i/pytest.log 2>&1; tail -n 2 gpurun_out/r3i/pytest.log
timeout 400 python scripts/hifigan_bench.py 32 4 bf16 > gpurun_out/r3i/hifigan.log 2>&1
echo "$(grep -o '"generator_forward_ms": [0-9.]*' gpurun_out/r3i/hifigan.log) $(grep -o '"gan_step_ms": [0-9.]*' gpurun_out/r3i/hifigan.log) $(grep -o '"gan_step_graph_ms": [0-9.]*' gpurun_out/r3i/hifigan.log)"
KANTTS_C1_NO_MFMA=1 KANTTS_NO_RES_STACK=1 timeout 400 python scripts/hifigan_bench.py 32 4 bf16 > gpurun_out/r3i/hifigan_ab.log 2>&1
echo "A/B old c1 + no res stack: $(grep -o '"generator_forward_ms": [0-9.]*' gpurun_out/r3i/hifigan_ab.log) $(grep -o '"gan_step_ms": [0-9.]*' gpurun_out/r3i/hifigan_ab.log) $(grep -o '"gan_step_graph_ms": [0-9.]*' gpurun_out/r3i/hifigan_ab.log)"
