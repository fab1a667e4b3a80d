#!/bin/bash
# round-2 visit AB: the residual stacks of a generator stage as parallel branches: parity tests, generator forward + GAN step
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hifigan.py tests/test_hifigan_nsf.py tests/test_trainer.py tests/test_bench_config_parity.py -m gpu -x -q -k "hifigan or gan or GAN or nsf" > gpurun_out/r2ab_pytest.log 2>&1; tail -3 gpurun_out/r2ab_pytest.log
timeout 300 python scripts/hifigan_bench.py 32 4 bf16 > gpurun_out/r2ab_hifigan.log 2>&1
echo "$(grep -o '"generator_forward_ms": [0-9.]*' gpurun_out/r2ab_hifigan.log) $(grep -o '"gan_step_ms": [0-9.]*' gpurun_out/r2ab_hifigan.log)"
