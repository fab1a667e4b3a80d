#!/bin/bash
# Round 4, visit M: fused SAM-BERT objective (parity on the device, A/B on the step) and the bgemm_tn slice rule A/B, same box
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16_ops.py tests/test_trainer.py tests/test_bench_config_parity.py -m gpu -q -x -k "fused_sambert or masked_l1 or output_tiles or sambert" 2>&1 | tail -3 | tee gpurun_out/r4m_tests.log
A="--steps 40 --warmup 10 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline --no-forward-only"
for v in "X=1" "KANTTS_NO_FUSED_LOSS=1" "KANTTS_TN_SLICE_RULE_R2=1" "X=2" "KANTTS_NO_FUSED_LOSS=1" "KANTTS_TN_SLICE_RULE_R2=1" "X=3"; do
  env $v timeout 300 python bench.py $A 2> gpurun_out/r4m_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'step %.3f ms  loss %.5f' % (d['ms_per_step'], d['config']['final_loss']))" | tee -a gpurun_out/r4m_step_ab.log
done
