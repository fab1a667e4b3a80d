#!/bin/bash
# round-2 visit M: memory-side traffic of the fused feed-forward launches, kernel statistics of the step, full bench line
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/r2m_pmc_$c
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/r2m_pmc_$c -o pmc -- python $R/scripts/ffn_pmc_probe.py > $R/gpurun_out/r2m_pmc_$c.log 2>&1
  f=$(find $R/gpurun_out/r2m_pmc_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_summary.py "$f" bgemm ffn_pair > $R/gpurun_out/r2m_ffn_$c.txt
  rm -rf $R/gpurun_out/r2m_pmc_$c
done
cat $R/gpurun_out/r2m_ffn_FETCH_SIZE.txt $R/gpurun_out/r2m_ffn_WRITE_SIZE.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2m_prof -o sam -- python $R/bench.py --steps 10 --warmup 3 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference > $R/gpurun_out/r2m_rocprof.log 2>&1
cd $R
f=$(find gpurun_out/r2m_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -70 "$f" > gpurun_out/r2m_sambert_kernel_stats_top.csv && cut -d, -f1-5 gpurun_out/r2m_sambert_kernel_stats_top.csv | sed 's/(.*"/"/' | cut -c1-120 | head -30
rm -rf gpurun_out/r2m_prof
timeout 900 python bench.py > gpurun_out/r2m_bench_full.log 2>&1
tail -c 6000 gpurun_out/r2m_bench_full.log
