#!/bin/bash
# Round 4, visit D: upsampling tile / ring-depth sweep, stream-kernel workgroup sweep, the DP form test with its noise floor,
# the tiled weight-norm table kernel on the GAN step, new cconv tiles against the numpy model
mkdir -p gpurun_out
timeout 300 python scripts/up_tile_sweep.py 2>&1 | grep -v Warn | tee gpurun_out/r4d_up_tiles.log
for pc in 1 2 3 4; do KANTTS_UPSTREAM_WG_PER_CU=$pc timeout 120 python scripts/up_tile_sweep.py narrow 2>&1 | grep stage | tee -a gpurun_out/r4d_up_narrow.log; done
timeout 600 python -m pytest tests/test_ddp_gloo.py -m gpu -q -x -s 2>&1 | grep -v Warning | grep -E "after 3 steps|passed|failed|Error|assert" | tee gpurun_out/r4d_ddp.log
timeout 600 python -m pytest tests/test_cconv.py tests/test_hifigan.py -m gpu -q -x -k "every_tile or weight_norm_table" 2>&1 | tail -3 | tee gpurun_out/r4d_tiles_test.log
for v in "X=1" "X=2"; do
  env $v timeout 300 python scripts/hifigan_bench.py 32 3 bf16 2> gpurun_out/r4d_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'gan step graph %.2f ms  eager %.2f ms  G fwd %.3f ms' % (d.get('gan_step_graph_ms',-1), d['gan_step_ms'], d['generator_forward_ms']))" | tee -a gpurun_out/r4d_gan.log
done
