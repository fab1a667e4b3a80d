#!/bin/bash
# Round 4, visit E: fused dual-path stage, register-weight upsampling kernel, stage-0 tile rule
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hifigan.py tests/test_hifigan_nsf.py tests/test_bench_config_parity.py -m gpu -q -x -k "not sambert" 2>&1 | tail -4 | tee gpurun_out/r4e_hifigan_tests.log
timeout 600 python -m pytest tests/test_ddp_gloo.py -m gpu -q -x -s -k gan 2>&1 | grep -v Warning | grep -E "captured|eager|passed|failed|Error|assert" | tee gpurun_out/r4e_ddp_gan.log
for v in "X=default" "KANTTS_UPSTREAM_TPW=1" "KANTTS_UPSTREAM_TPW=2" "KANTTS_UPSTREAM_TPW=4" "KANTTS_UPSTREAM_TPW=16" "KANTTS_UPSTREAM_LDS=1"; do
  echo "$v" | tee -a gpurun_out/r4e_up_narrow.log
  env $v timeout 120 python scripts/up_tile_sweep.py narrow 2>&1 | grep stage | tee -a gpurun_out/r4e_up_narrow.log
done
timeout 120 python scripts/up_bench.py 2>&1 | grep stage_us | tee gpurun_out/r4e_up_chain.log
for v in "X=1" "KANTTS_NO_DUAL_FUSE=1" "X=2" "KANTTS_NO_DUAL_FUSE=1"; do
  env $v timeout 300 python scripts/hifigan_bench.py 32 3 bf16 2> gpurun_out/r4e_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'gan step graph %.2f ms  eager %.2f ms  G fwd %.3f ms' % (d.get('gan_step_graph_ms',-1), d['gan_step_ms'], d['generator_forward_ms']))" | tee -a gpurun_out/r4e_gan.log
done
