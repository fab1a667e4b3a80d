#!/bin/bash
# Round 5 closing visit: full GPU suite, smoke(), the default bench line, rocprofv3 kernel statistics of (a) the bench
# command, (b) the SAM-BERT steps alone, (c) the forward-only graph, (d) the inference leg (one-launch loops).
T=${1:-r5final}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/${T}_pytest_gpu.log 2>&1; echo "pytest -m gpu exit $?"; tail -n 6 gpurun_out/${T}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; echo "smoke exit $?"; tail -n 2 gpurun_out/${T}_smoke.log
timeout 900 python bench.py > gpurun_out/${T}_bench_full.log 2> gpurun_out/${T}_bench_full.err; echo "bench exit $?"; tail -n 1 gpurun_out/${T}_bench_full.log | cut -c1-1500
cd /tmp && export TMPDIR=/tmp
stats() {  # name, command...
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof_$name -o p -- "$@" > $R/gpurun_out/${T}_rocprof_$name.log 2>&1
  local f=$(find $R/gpurun_out/${T}_prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -150 "$f" > $R/gpurun_out/${T}_${name}_kernel_stats_top.csv
  rm -rf $R/gpurun_out/${T}_prof_$name
  echo "== $name"; head -12 $R/gpurun_out/${T}_${name}_kernel_stats_top.csv | cut -c1-140
}
stats sambert_steps python $R/bench.py --no-hifigan --no-inference --no-cpu-baseline --no-fp32 --no-roofline --no-forward-only --steps 40 --warmup 5
stats forward_only python $R/bench.py --no-hifigan --no-inference --no-cpu-baseline --no-fp32 --no-roofline --steps 1 --warmup 1 --forward-replays 300
stats inference python $R/scripts/infer_breakdown.py 48 kernel
stats bench_command python $R/bench.py --no-cpu-baseline
cd $R
