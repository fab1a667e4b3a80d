#!/bin/bash
# Round 4, visit Z: discriminator weight images beside the generator forward, discriminator updates on parallel streams:
# parity (graph == eager, loss curves, DP), GAN step A/B on one box
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hifigan.py tests/test_trainer.py tests/test_bench_config_parity.py tests/test_multiband.py tests/test_ddp_gloo.py -m gpu -q -x -k "gan or hifigan or multiband or weight_norm" 2>&1 | tail -3 | tee gpurun_out/r4z_tests.log
for v in "X=1" "KANTTS_NO_IMAGES_BESIDE=1" "KANTTS_NO_PARALLEL_DSTEP=1" "X=2" "KANTTS_NO_IMAGES_BESIDE=1" "KANTTS_NO_PARALLEL_DSTEP=1" "X=3"; do
  env $v timeout 300 python scripts/hifigan_bench.py 32 3 bf16 2> gpurun_out/r4z_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'gan step graph %.2f ms  eager %.2f ms  G fwd %.3f ms' % (d.get('gan_step_graph_ms',-1), d['gan_step_ms'], d['generator_forward_ms']))" | tee -a gpurun_out/r4z_gan_ab.log
done
