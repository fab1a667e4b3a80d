#!/bin/bash
# Timing experiment: the captured SAM-BERT step with the weight-gradient kernel's atomics replaced by stores (upper bound).
T=${1:-r5ah}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out; cd $R
ARGS="--no-hifigan --no-inference --no-cpu-baseline --no-fp32 --no-roofline --no-forward-only --steps 40"
for rep in 1 2; do
  for v in product noatomics; do
    unset KANTTS_LIB
    [ $v = noatomics ] && export KANTTS_LIB=$R/kan-tts_amd/variants/libkantts_TNPROBE.so
    timeout 300 python bench.py $ARGS > gpurun_out/${T}_${v}_$rep.json 2> gpurun_out/${T}_${v}_$rep.err
    python - <<PY
import json
for l in open("gpurun_out/${T}_${v}_$rep.json"):
    if l.startswith("{"): d = json.loads(l)
print("$v $rep ms_per_step %.3f" % d["ms_per_step"])
PY
  done
done
