#!/bin/bash
# round-2 visit Q: quad-layout LSTM kernels (one barrier per step); GAN step with the discriminators frozen in the
# generator phase
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sambert.py "tests/test_bench_config_parity.py::test_sambert_full_b32_matches_oracle" -m gpu -x -q > gpurun_out/r2q_pytest.log 2>&1; tail -4 gpurun_out/r2q_pytest.log
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2q_prof -o sam -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference > $GRAFT_REPO_ROOT/gpurun_out/r2q_bench.log 2>&1 )
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2q_bench.log | head -1
f=$(find gpurun_out/r2q_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -70 "$f" > gpurun_out/r2q_sambert_kernel_stats_top.csv && grep lstm gpurun_out/r2q_sambert_kernel_stats_top.csv | sed 's/(float const[^"]*"/"/' | cut -c1-120
rm -rf gpurun_out/r2q_prof
timeout 600 python -m pytest tests/test_hifigan.py tests/test_trainer.py -m gpu -x -q > gpurun_out/r2q_pytest_gan.log 2>&1; tail -3 gpurun_out/r2q_pytest_gan.log
timeout 300 python scripts/hifigan_bench.py 32 3 bf16 > gpurun_out/r2q_hifigan.log 2>&1; grep -o '"gan_step_ms": [0-9.]*' gpurun_out/r2q_hifigan.log
