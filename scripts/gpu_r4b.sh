#!/bin/bash
# Round 4, visit B: (1) the data-parallel captured steps (segments / two-graph / GAN) with two processes on the one GPU over
# gloo; (2) bench's 2-rank path incl. the exposed-exchange measurement; (3) the new reference-recorded fixtures on the device;
# (4) A/B of the LayerNorm-backward epilogue default and the ReLU-gate hand-over, 40 steps each, interleaved twice
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ddp_gloo.py -m gpu -q -x 2>&1 | tail -15 | tee gpurun_out/r4b_ddp.log
timeout 300 python -m pytest tests/test_dsp_reference_fixture.py tests/test_thirdparty_kat.py tests/test_gpu_bf16_ops.py -m gpu -q 2>&1 | tail -5 | tee gpurun_out/r4b_pins.log
A="--steps 40 --warmup 10 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline --no-forward-only"
for v in "X=1" "KANTTS_NO_LN_BWD_EPILOGUE=1" "KANTTS_RELU_GATE_EPILOGUE=1" "X=2" "KANTTS_NO_LN_BWD_EPILOGUE=1" "KANTTS_RELU_GATE_EPILOGUE=1"; do
  env $v timeout 300 python bench.py $A 2> gpurun_out/r4b_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'step %.3f ms  loss %.5f' % (d['ms_per_step'], d['config']['final_loss']))" | tee -a gpurun_out/r4b_step_ab.log
done
timeout 400 python bench.py --gpus 2 --backend gloo --share-device --steps 3 --warmup 1 2> gpurun_out/r4b_2rank_err.log | tail -1 > gpurun_out/r4b_bench_2rank.json
tail -5 gpurun_out/r4b_2rank_err.log
python -c "
import json; d=json.load(open('gpurun_out/r4b_bench_2rank.json')); print(d['ms_per_step'], d['config']['data_parallel'])"
