#!/bin/bash
# Round 5: kernel B (attention backward + QKV input gradient + LayerNorm backward in one launch) on the device: parity, step
# A/B on one box, then the rest of the GPU suite that the previous visit's -x cut off.
T=${1:-r5r}
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest -q -x -m gpu tests/test_pnca_block.py tests/test_bench_config_parity.py tests/test_gpu_sambert.py \
  tests/test_trainer.py tests/test_device_batching.py -k "not hifigan" > gpurun_out/${T}_tests.log 2>&1; echo "tests exit $?"; tail -n 3 gpurun_out/${T}_tests.log
for rep in 1 2; do
  for v in all noattn; do
    unset KANTTS_NO_PNCA_ATTN_BWD
    [ $v = noattn ] && export KANTTS_NO_PNCA_ATTN_BWD=1
    timeout 300 python bench.py --no-hifigan --no-inference --no-cpu-baseline --no-fp32 --no-roofline --steps 40 \
      > gpurun_out/${T}_bench_${v}_${rep}.json 2> gpurun_out/${T}_bench_${v}_${rep}.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/${T}_bench_${v}_${rep}.json").read().strip().splitlines()[-1])
print("$v $rep ms_per_step %.3f forward_ms %s" % (d["ms_per_step"], d["roofline"].get("forward_ms")))
PY
  done
done
unset KANTTS_NO_PNCA_ATTN_BWD
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_bench_config_parity.py --deselect tests/test_pnca_block.py > gpurun_out/${T}_pytest_gpu_rest.log 2>&1; echo "rest of the suite exit $?"; tail -n 5 gpurun_out/${T}_pytest_gpu_rest.log
