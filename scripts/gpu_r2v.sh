#!/bin/bash
# round-2 visit V: the variance predictors as a side branch (second stream / parallel graph branch): graph-vs-eager tests, A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_trainer.py tests/test_ddp_gloo.py tests/test_gpu_sambert.py -m gpu -x -q > gpurun_out/r2v_pytest.log 2>&1; tail -4 gpurun_out/r2v_pytest.log
for v in 1 ""; do
  KANTTS_NO_SIDE_BRANCH=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference > gpurun_out/r2v_bench_noside_$v.log 2>&1
  echo "KANTTS_NO_SIDE_BRANCH='$v': $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2v_bench_noside_$v.log | head -1) $(grep -o '"final_loss": [0-9.]*' gpurun_out/r2v_bench_noside_$v.log | head -1) $(grep -o 'capture failed[^"]*' gpurun_out/r2v_bench_noside_$v.log | head -1)"
done
