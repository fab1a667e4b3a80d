#!/bin/bash
# Round 5: deferred, batched row sums + reference discipline of the fused backward: tests, 3-way step A/B, kernel trace.
T=${1:-r5j}
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest -q -x -m gpu tests/test_pnca_block.py tests/test_bench_config_parity.py tests/test_gpu_sambert.py \
  tests/test_trainer.py -k "not hifigan" > gpurun_out/${T}_tests.log 2>&1; echo "tests exit $?"; tail -n 3 gpurun_out/${T}_tests.log
python scripts/pnca_block_ablate.py 2>&1 | grep KANTTS_PB_DBG
for rep in 1 2; do
  for v in fused fwdonly chain; do
    unset KANTTS_NO_PNCA_BLOCK KANTTS_NO_PNCA_BLOCK_BWD
    [ $v = chain ] && export KANTTS_NO_PNCA_BLOCK=1
    [ $v = fwdonly ] && export KANTTS_NO_PNCA_BLOCK_BWD=1
    timeout 300 python bench.py --no-hifigan --no-inference --no-cpu-baseline --no-fp32 --no-roofline --steps 40 \
      > gpurun_out/${T}_bench_${v}_${rep}.json 2> gpurun_out/${T}_bench_${v}_${rep}.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/${T}_bench_${v}_${rep}.json").read().strip().splitlines()[-1])
print("$v $rep ms_per_step %.3f forward_ms %s" % (d["ms_per_step"], d["roofline"].get("forward_ms")))
PY
  done
done
unset KANTTS_NO_PNCA_BLOCK KANTTS_NO_PNCA_BLOCK_BWD
bash scripts/gpu_r5i.sh ${T}
