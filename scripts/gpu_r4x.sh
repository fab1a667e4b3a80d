#!/bin/bash
# Round 4, visit X: 1-channel weight gradient: rotated atomic flush, workgroup caps 512 / 256 / 128 (kernel averages)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hifigan.py tests/test_conv_sweep.py -m gpu -q -x -k "one_channel or conv_variants or c1 or gan_step" 2>&1 | tail -2 | tee gpurun_out/r4x_tests.log
cd /tmp && export TMPDIR=/tmp
for cap in 512 256 128; do
  KANTTS_C1_WGRAD_WGS=$cap timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4x_prof_$cap -o gan -- python $R/scripts/hifigan_bench.py 32 3 bf16 > $R/gpurun_out/r4x_bench_$cap.json 2> $R/gpurun_out/r4x_prof_err.log
  f=$(find $R/gpurun_out/r4x_prof_$cap -name "*kernel_stats.csv" | head -n 1)
  echo "cap $cap: $(grep conv_c1_wgrad_mfma $f | cut -d, -f1-4 | tr '\n' ' ')" | tee -a $R/gpurun_out/r4x_c1.log
  python -c "
import json,sys
d=json.loads(open('$R/gpurun_out/r4x_bench_$cap.json').read().strip().splitlines()[-1]); print('cap $cap gan step graph %.2f ms (under the profiler)' % d.get('gan_step_graph_ms',-1))" | tee -a $R/gpurun_out/r4x_c1.log
  rm -rf $R/gpurun_out/r4x_prof_$cap
done
