#!/bin/bash
# round-2 visit X: deferred weight gradients issued early on the side stream (flush points at the encoder output and the
# postnet input): graph-vs-eager tests, parity at the bench config, A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_trainer.py tests/test_ddp_gloo.py tests/test_gpu_sambert.py "tests/test_bench_config_parity.py::test_sambert_full_b32_matches_oracle" -m gpu -x -q > gpurun_out/r2x_pytest.log 2>&1; tail -4 gpurun_out/r2x_pytest.log
for v in 1 ""; do
  KANTTS_NO_EARLY_FLUSH=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference > gpurun_out/r2x_bench_noearly_$v.log 2>&1
  echo "KANTTS_NO_EARLY_FLUSH='$v': $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2x_bench_noearly_$v.log | head -1) $(grep -o '"final_loss": [0-9.]*' gpurun_out/r2x_bench_noearly_$v.log | head -1) $(grep -o '"launch": "[a-z]*"' gpurun_out/r2x_bench_noearly_$v.log | head -1)"
done
