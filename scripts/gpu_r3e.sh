#!/bin/bash
# round-3 visit E: hipGraph capture of the GAN step
mkdir -p gpurun_out/r3e
timeout 600 python -m pytest tests/test_cconv.py tests/test_hifigan.py -m gpu -x -q -k "handover or graphed" > gpurun_out/r3e/pytest.log 2>&1; tail -2 gpurun_out/r3e/pytest.log
timeout 400 python scripts/hifigan_bench.py 32 4 bf16 > gpurun_out/r3e/hifigan.log 2>&1
echo "$(grep -o '"generator_forward_ms": [0-9.]*' gpurun_out/r3e/hifigan.log) $(grep -o '"gan_step_ms": [0-9.]*' gpurun_out/r3e/hifigan.log) $(grep -o '"gan_step_graph_ms": [0-9.]*' gpurun_out/r3e/hifigan.log) $(grep -o '"graph_capture_s": [0-9.]*' gpurun_out/r3e/hifigan.log)"
tail -3 gpurun_out/r3e/hifigan.log | cut -c1-600
