#!/bin/bash
# round-3 visit U: which ATen launches are left (census), postnet GEMM tile heights, step time with the bf16 FSMN hidden layer
mkdir -p gpurun_out
timeout 200 python scripts/aten_census.py 2>&1 | grep -v Warning | cut -c1-200 > gpurun_out/r3u_aten_census.log; head -n 60 gpurun_out/r3u_aten_census.log
for bm in "" 128 32; do
  env ${bm:+KANTTS_BGEMM_BM=$bm} timeout 120 python scripts/postnet_gemm_bench.py 2>&1 | grep -v Warning | tee -a gpurun_out/r3u_postnet_gemm.log
done
A="--steps 20 --warmup 5 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline"
timeout 300 python bench.py $A 2> gpurun_out/r3u_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step %.3f ms  forward %.3f ms' % (d['ms_per_step'], d['roofline']['forward_ms']))" | tee -a gpurun_out/r3u_bench.log
