#!/bin/bash
# Round 4, visit F: the bucket-packing order fix on the captured data-parallel GAN step; same-box A/B of the first transposed
# convolution's tile rule; stream kernel with the early activation fetch; GAN step
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ddp_gloo.py -m gpu -q -x -s 2>&1 | grep -v Warning | grep -E "captured vs|after 3 steps|passed|failed|Error|assert" | cut -c1-400 | tee gpurun_out/r4f_ddp.log
for rep in 1 2 3; do timeout 200 python scripts/up_tile_sweep.py ab 2>&1 | grep stage | tee -a gpurun_out/r4f_up_ab.log; done
timeout 120 python scripts/up_bench.py 2>&1 | grep stage_us | tee gpurun_out/r4f_up_chain.log
timeout 120 python scripts/up_tile_sweep.py narrow 2>&1 | grep stage | tee gpurun_out/r4f_up_narrow.log
for v in "X=1" "X=2"; do
  env $v timeout 300 python scripts/hifigan_bench.py 32 3 bf16 2> gpurun_out/r4f_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'gan step graph %.2f ms  eager %.2f ms  G fwd %.3f ms' % (d.get('gan_step_graph_ms',-1), d['gan_step_ms'], d['generator_forward_ms']))" | tee -a gpurun_out/r4f_gan.log
done
