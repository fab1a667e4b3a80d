#!/bin/bash
# First GPU visit of the next round: the work of round 3 that has only run on the kernel-source CPU build.
#   1. the LayerNorm-backward epilogue kernels against the two-launch forms and the numpy model ON THE DEVICE
#   2. the same training-step parity tests with the switch on (graph == eager, loss curve)
#   3. A / B of the step time, switch off / on, twice each on this box
#   4. per-launch times of the two fused launches against the two-launch forms (scripts/lnbwd_bench.py)
#   5. the native attention harness (scripts/bench_native/attn_test.cpp): first run
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_r4_first.sh'
mkdir -p gpurun_out
export KANTTS_LN_BWD_EPILOGUE=1 KANTTS_RELU_GATE_EPILOGUE=1
timeout 300 python -m pytest tests/test_gpu_bf16_ops.py -m gpu -q -x -k "layernorm_backward" 2>&1 | tail -5 | tee gpurun_out/r4a_lnbwd_ops.log
timeout 400 python -m pytest tests/test_trainer.py tests/test_bench_config_parity.py -m gpu -q -x -k "sambert" 2>&1 | tail -5 | tee gpurun_out/r4a_lnbwd_model.log
unset KANTTS_LN_BWD_EPILOGUE KANTTS_RELU_GATE_EPILOGUE
A="--steps 20 --warmup 5 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline"
for v in "X=1" "KANTTS_LN_BWD_EPILOGUE=1" "KANTTS_RELU_GATE_EPILOGUE=1" "X=2" "KANTTS_LN_BWD_EPILOGUE=1" "KANTTS_RELU_GATE_EPILOGUE=1"; do
  env $v timeout 300 python bench.py $A 2> gpurun_out/r4a_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'step %.3f ms  forward %.3f ms  loss %.5f' % (d['ms_per_step'], d['roofline']['forward_ms'], d['config']['final_loss']))" | tee -a gpurun_out/r4a_lnbwd_step_ab.log
done
timeout 200 python scripts/lnbwd_bench.py 2>&1 | tee gpurun_out/r4a_lnbwd_per_launch.log
hipcc --offload-arch=gfx950 -O2 -c scripts/bench_native/attn_test.cpp -o /tmp/attn_test.o && \
  hipcc --offload-arch=gfx950 /tmp/attn_test.o kan-tts_amd/csrc/attn.o -o /tmp/attn_test && \
  timeout 120 /tmp/attn_test 50 2>&1 | tee gpurun_out/r4a_attn_native.log
