#!/bin/bash
# Round 4, visit J: table-driven weight-norm backward on the device; half-batch concurrency probe
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hifigan.py tests/test_trainer.py -m gpu -q -x -k "weight_norm_table or graphed_gan or gan_loss_curve or gan_train_step" 2>&1 | tail -3 | tee gpurun_out/r4j_tests.log
timeout 400 python scripts/half_batch_probe.py 2>&1 | grep -E "chain|Error|error" | tee gpurun_out/r4j_half_batch_probe.log
for v in "X=1" "X=2"; do
  env $v timeout 300 python scripts/hifigan_bench.py 32 3 bf16 2> gpurun_out/r4j_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'gan step graph %.2f ms  eager %.2f ms  G fwd %.3f ms' % (d.get('gan_step_graph_ms',-1), d['gan_step_ms'], d['generator_forward_ms']))" | tee -a gpurun_out/r4j_gan.log
done
