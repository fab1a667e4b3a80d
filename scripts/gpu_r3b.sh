#!/bin/bash
# round-3 visit B: cconv path through the Python wrappers -- parity tests, GAN step A/B (cconv on / off), per-shape table
mkdir -p gpurun_out/r3b
timeout 600 python -m pytest tests/test_cconv.py -m gpu -x -q > gpurun_out/r3b/pytest_cconv.log 2>&1; tail -3 gpurun_out/r3b/pytest_cconv.log
timeout 300 python scripts/hifigan_bench.py 32 4 bf16 > gpurun_out/r3b/hifigan_cconv.log 2>&1
echo "cconv: $(grep -o '"generator_forward_ms": [0-9.]*' gpurun_out/r3b/hifigan_cconv.log) $(grep -o '"gan_step_ms": [0-9.]*' gpurun_out/r3b/hifigan_cconv.log)"
KANTTS_NO_CCONV=1 timeout 300 python scripts/hifigan_bench.py 32 4 bf16 > gpurun_out/r3b/hifigan_nocconv.log 2>&1
echo "no cconv: $(grep -o '"generator_forward_ms": [0-9.]*' gpurun_out/r3b/hifigan_nocconv.log) $(grep -o '"gan_step_ms": [0-9.]*' gpurun_out/r3b/hifigan_nocconv.log)"
timeout 300 python scripts/conv_shape_bench.py 32 > gpurun_out/r3b/conv_shapes.log 2>&1; grep "conv launches total" gpurun_out/r3b/conv_shapes.log
timeout 900 python -m pytest tests/test_hifigan.py tests/test_hifigan_nsf.py tests/test_trainer.py tests/test_bench_config_parity.py -m gpu -x -q -k "hifigan or gan or GAN or nsf or conv" > gpurun_out/r3b/pytest_hifigan.log 2>&1; tail -3 gpurun_out/r3b/pytest_hifigan.log
