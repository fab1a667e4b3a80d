#!/bin/bash
# round-3 final visit: full GPU test suite, smoke, the default bench line, kernel statistics of the bench command
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3h_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -n 3 gpurun_out/r3h_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3h_smoke.log 2>&1; echo "smoke exit $?"; grep -v Warning gpurun_out/r3h_smoke.log | tail -n 3
timeout 1500 python bench.py > gpurun_out/r3h_bench_full.log 2> gpurun_out/r3h_bench_full.err; echo "bench exit $?"; grep "^\[bench" gpurun_out/r3h_bench_full.err | tail -n 20
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3h_prof -o bench -- python $R/bench.py --no-cpu-baseline --no-inference --no-fp32 > $R/gpurun_out/r3h_rocprof_bench.log 2>&1
cd $R
f=$(find gpurun_out/r3h_prof -name "*kernel_stats.csv" | head -n 1)
[ -n "$f" ] && head -n 120 "$f" > gpurun_out/r3h_bench_kernel_stats_top.csv
rm -rf gpurun_out/r3h_prof
timeout 400 python bench.py --gpus 2 --backend gloo --share-device --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3h_bench_2rank_gloo.log 2>&1; echo "2-rank exit $?"
tail -c 1500 gpurun_out/r3h_bench_full.log
