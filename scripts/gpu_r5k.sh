#!/bin/bash
# Round 5: predictors issued after the postnet (their backward beside the postnet LSTM's) -- step A/B on one box + trace.
T=${1:-r5k}
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2; do
  for v in after before; do
    unset KANTTS_PREDICTORS_BEFORE_POSTNET
    [ $v = before ] && export KANTTS_PREDICTORS_BEFORE_POSTNET=1
    timeout 300 python bench.py --no-hifigan --no-inference --no-cpu-baseline --no-fp32 --no-roofline --steps 40 \
      > gpurun_out/${T}_bench_${v}_${rep}.json 2> gpurun_out/${T}_bench_${v}_${rep}.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/${T}_bench_${v}_${rep}.json").read().strip().splitlines()[-1])
print("$v $rep ms_per_step %.3f forward_ms %s loss %s" % (d["ms_per_step"], d["roofline"].get("forward_ms"), d["config"]["final_loss"]))
PY
  done
done
unset KANTTS_PREDICTORS_BEFORE_POSTNET
timeout 600 python -m pytest -q -x -m gpu tests/test_bench_config_parity.py tests/test_trainer.py tests/test_gpu_sambert.py -k "not hifigan" > gpurun_out/${T}_tests.log 2>&1; echo "tests exit $?"; tail -n 3 gpurun_out/${T}_tests.log
bash scripts/gpu_r5i.sh ${T}
