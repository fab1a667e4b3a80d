#!/bin/bash
# Round 5: sweep of the weight-gradient flush points with the fused decoder blocks (the deferred launches cost 0.7 ms of the
# step's critical path: profiles/r05_runO_family_ablation.log).
T=${1:-r5p}
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
ARGS="--no-hifigan --no-inference --no-cpu-baseline --no-fp32 --no-roofline --no-forward-only --steps 40"
for cfg in "0 4" "3 4" "4 4" "6 4" "0 2" "4 2" "0 0" "2 2" "0 4"; do
  set -- $cfg
  KANTTS_FLUSH_EVERY_DEC=$1 KANTTS_FLUSH_EVERY_ENC=$2 timeout 200 python bench.py $ARGS > gpurun_out/${T}_b.json 2> gpurun_out/${T}_b.err
  python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r5p_b.json").read().strip().splitlines()[-1])
print("flush every dec %s enc %s: ms_per_step %.3f" % (sys.argv[1], sys.argv[2], d["ms_per_step"]))
PY
done
