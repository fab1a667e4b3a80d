#!/bin/bash
# Round 4, visit S: the multi-receptive-field mean as one launch: parity on the device, GAN step A/B (same box)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_hifigan.py tests/test_bench_config_parity.py tests/test_trainer.py -m gpu -q -x -k "mean_many or hifigan or generator or gan" 2>&1 | tail -3 | tee gpurun_out/r4s_tests.log
for v in "X=1" "KANTTS_NO_MEAN_MANY=1" "X=2" "KANTTS_NO_MEAN_MANY=1"; do
  env $v timeout 300 python scripts/hifigan_bench.py 32 3 bf16 2> gpurun_out/r4s_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'gan step graph %.2f ms  eager %.2f ms  G fwd %.3f ms' % (d.get('gan_step_graph_ms',-1), d['gan_step_ms'], d['generator_forward_ms']))" | tee -a gpurun_out/r4s_gan_ab.log
done
