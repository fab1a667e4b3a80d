#!/bin/bash
# Round 5: the target-only plan as one launch -- parity, step / forward A/B on one box, kernel census of the step.
T=${1:-r5n}
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest -q -x -m gpu tests/test_teacher_plan.py tests/test_melspec.py tests/test_bench_config_parity.py tests/test_gpu_sambert.py \
  tests/test_trainer.py -k "not hifigan" > gpurun_out/${T}_tests.log 2>&1; echo "tests exit $?"; tail -n 3 gpurun_out/${T}_tests.log
for rep in 1 2; do
  for v in plan noplan; do
    unset KANTTS_NO_PLAN_KERNEL
    [ $v = noplan ] && export KANTTS_NO_PLAN_KERNEL=1
    timeout 300 python bench.py --no-hifigan --no-inference --no-cpu-baseline --no-fp32 --no-roofline --steps 40 \
      > gpurun_out/${T}_bench_${v}_${rep}.json 2> gpurun_out/${T}_bench_${v}_${rep}.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/${T}_bench_${v}_${rep}.json").read().strip().splitlines()[-1])
print("$v $rep ms_per_step %.3f forward_ms %s" % (d["ms_per_step"], d["roofline"].get("forward_ms")))
PY
  done
done
unset KANTTS_NO_PLAN_KERNEL
bash scripts/gpu_r5i.sh ${T}
