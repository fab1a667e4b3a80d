#!/bin/bash
# Last check of the round: FSMN op tests + the SAM-BERT leg after the FIR revert.
T=${1:-r5ak}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out; cd $R
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "fsmn" > gpurun_out/${T}_tests.log 2>&1; echo "tests exit $?"; tail -2 gpurun_out/${T}_tests.log
timeout 300 python bench.py --no-hifigan --no-inference --no-cpu-baseline --no-fp32 --no-roofline --steps 40 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python - $T <<'PY'
import json, sys
for l in open("gpurun_out/%s_bench.json" % sys.argv[1]):
    if l.startswith("{"): d = json.loads(l)
print("ms_per_step %.3f forward_ms %.3f" % (d["ms_per_step"], d["roofline"]["forward_ms"]))
PY
