#!/bin/bash
# round-2 visit I: validate the chunked LSTM prefetch (correctness first), the restored NT tile rule; profile the step
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sambert.py tests/test_gpu_bf16_ops.py "tests/test_bench_config_parity.py::test_sambert_full_b32_matches_oracle" -m gpu -x -q > gpurun_out/r2i_pytest.log 2>&1
tail -5 gpurun_out/r2i_pytest.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference > gpurun_out/r2i_bench.log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2i_bench.log | head -2
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2i_prof -o sam -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference > $GRAFT_REPO_ROOT/gpurun_out/r2i_rocprof.log 2>&1 )
f=$(find gpurun_out/r2i_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -70 "$f" > gpurun_out/r2i_sambert_kernel_stats_top.csv && cut -c1-140 gpurun_out/r2i_sambert_kernel_stats_top.csv | head -24
rm -rf gpurun_out/r2i_prof
