#!/bin/bash
# Round 4, visit U: the GAN criteria in one launch each: parity on the device, GAN step A/B (same box)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_hifigan.py tests/test_bench_config_parity.py tests/test_trainer.py tests/test_multiband.py -m gpu -q -x -k "gan or hifigan or generator or criteria or loss" 2>&1 | tail -3 | tee gpurun_out/r4u_tests.log
for v in "X=1" "KANTTS_NO_FUSED_GAN_LOSS=1" "X=2" "KANTTS_NO_FUSED_GAN_LOSS=1" "X=3"; do
  env $v timeout 300 python scripts/hifigan_bench.py 32 3 bf16 2> gpurun_out/r4u_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'gan step graph %.2f ms  eager %.2f ms  G fwd %.3f ms' % (d.get('gan_step_graph_ms',-1), d['gan_step_ms'], d['generator_forward_ms']))" | tee -a gpurun_out/r4u_gan_ab.log
done
