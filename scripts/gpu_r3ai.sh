#!/bin/bash
# round-3 visit AI: default pieces beside the encoder (plan + pitch / energy embeddings + decoder prenet); the GAN branch-stream test in detail
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_trainer.py -m gpu -x -q --tb=short -k "gan_step_branch_streams" 2>&1 | grep -v Warning | tail -n 25
timeout 900 python -m pytest tests/test_trainer.py tests/test_gpu_sambert.py -m gpu -q 2>&1 | tail -n 4
A="--steps 20 --warmup 5 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline"
timeout 300 python bench.py $A 2> gpurun_out/r3ai_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step %.3f ms  forward %.3f ms' % (d['ms_per_step'], d['roofline']['forward_ms']))" | tee -a gpurun_out/r3ai_bench.log
