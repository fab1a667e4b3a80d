#!/bin/bash
# round-3 visit R: LayerNorm in the producers' epilogues (parity), SAM-BERT step time, kernel statistics of the step
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sambert.py tests/test_gpu_bf16_ops.py tests/test_trainer.py tests/test_decode_graph.py -m gpu -x -q 2>&1 | tail -n 8
A="--steps 20 --warmup 5 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline"
for v in "" "KANTTS_NO_PRENORM=1"; do
  env $v timeout 300 python bench.py $A 2> gpurun_out/r3r_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'step %.3f ms  forward %.3f ms' % (d['ms_per_step'], d['roofline']['forward_ms']))" | tee -a gpurun_out/r3r_bench.log
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3r_prof -o sam -- python $R/bench.py --steps 10 --warmup 3 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline > $R/gpurun_out/r3r_rocprof.log 2>&1
cd $R
f=$(find gpurun_out/r3r_prof -name "*kernel_stats.csv" | head -n 1)
[ -n "$f" ] && head -n 80 "$f" > gpurun_out/r3r_sambert_kernel_stats_top.csv && cut -d, -f1-5 gpurun_out/r3r_sambert_kernel_stats_top.csv | sed 's/(.*"/"/' | cut -c1-110 | head -n 45
rm -rf gpurun_out/r3r_prof
