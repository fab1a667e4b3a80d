#!/bin/bash
# Round 5: does forking the predictor branch where its inputs became ready (beside the decoder) help?  A/B on one box.
T=${1:-r5m}
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in early late; do
    unset KANTTS_NO_EARLY_FORK
    [ $v = late ] && export KANTTS_NO_EARLY_FORK=1
    timeout 300 python bench.py --no-hifigan --no-inference --no-cpu-baseline --no-fp32 --no-roofline --steps 40 \
      > gpurun_out/${T}_bench_${v}_${rep}.json 2> gpurun_out/${T}_bench_${v}_${rep}.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/${T}_bench_${v}_${rep}.json").read().strip().splitlines()[-1])
print("$v $rep ms_per_step %.3f forward_ms %s" % (d["ms_per_step"], d["roofline"].get("forward_ms")))
PY
  done
done
