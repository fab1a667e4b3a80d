#!/bin/bash
# round-2 visit J: memory-side traffic of the bf16 feed-forward contractions (FETCH_SIZE / WRITE_SIZE in separate passes),
# kernel statistics of the HiFi-GAN V1 GAN step at batch 32 x 8192
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/r2j_pmc_$c
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/r2j_pmc_$c -o pmc -- python $R/scripts/ffn_pmc_probe.py > $R/gpurun_out/r2j_pmc_$c.log 2>&1
  f=$(find $R/gpurun_out/r2j_pmc_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_summary.py "$f" bgemm > $R/gpurun_out/r2j_ffn_$c.txt
  rm -rf $R/gpurun_out/r2j_pmc_$c
done
cat $R/gpurun_out/r2j_ffn_FETCH_SIZE.txt $R/gpurun_out/r2j_ffn_WRITE_SIZE.txt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2j_prof -o gan -- python $R/scripts/hifigan_bench.py 32 3 bf16 > $R/gpurun_out/r2j_hifigan.log 2>&1
cd $R
f=$(find gpurun_out/r2j_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -60 "$f" > gpurun_out/r2j_hifigan_kernel_stats_top.csv && cut -d, -f1-5 gpurun_out/r2j_hifigan_kernel_stats_top.csv | sed 's/(.*"/"/' | cut -c1-120 | head -30
rm -rf gpurun_out/r2j_prof
tail -2 gpurun_out/r2j_hifigan.log | cut -c1-600
