#!/bin/bash
# round-2 visit R: quad-layout LSTM with shared sigmoid/tanh code: correctness + timings
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sambert.py -m gpu -x -q -k "lstm or LSTM or sambert or tiny or full" > gpurun_out/r2r_pytest.log 2>&1; tail -3 gpurun_out/r2r_pytest.log
timeout 120 python scripts/lstm_bench.py > gpurun_out/r2r_lstm_bench.log 2>&1; cat gpurun_out/r2r_lstm_bench.log | tail -7
timeout 300 python bench.py --steps 10 --warmup 3 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference > gpurun_out/r2r_bench.log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2r_bench.log | head -1
