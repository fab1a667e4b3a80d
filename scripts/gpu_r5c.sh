#!/bin/bash
# Round 5, third visit: the one-launch PNCA decoder block (csrc/pnca_block.hip) on the device -- parity against the chain and
# the oracle, per-block timing, step A/B on one box.
T=${1:-r5c}
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest -q -x -m gpu tests/test_pnca_block.py tests/test_bench_config_parity.py tests/test_gpu_sambert.py \
  tests/test_trainer.py tests/test_decode_graph.py -k "not hifigan" > gpurun_out/${T}_tests.log 2>&1; echo "tests exit $?"; tail -n 6 gpurun_out/${T}_tests.log
timeout 300 python scripts/pnca_block_bench.py > gpurun_out/${T}_pnca_block_bench.log 2>&1; echo "block bench exit $?"; grep blocks gpurun_out/${T}_pnca_block_bench.log
for rep in 1 2; do
  for v in fused chain; do
    if [ $v = chain ]; then export KANTTS_NO_PNCA_BLOCK=1; else unset KANTTS_NO_PNCA_BLOCK; fi
    timeout 300 python bench.py --no-hifigan --no-inference --no-cpu-baseline --no-fp32 --no-roofline --steps 40 \
      > gpurun_out/${T}_bench_${v}_${rep}.json 2> gpurun_out/${T}_bench_${v}_${rep}.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/${T}_bench_${v}_${rep}.json").read().strip().splitlines()[-1])
print("$v $rep ms_per_step %.3f forward_ms %s" % (d["ms_per_step"], d["roofline"].get("forward_ms")))
PY
  done
done
unset KANTTS_NO_PNCA_BLOCK
