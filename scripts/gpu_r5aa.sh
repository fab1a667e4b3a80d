#!/bin/bash
T=${1:-r5aa}
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ar_kernels.py tests/test_config5_inference.py -q -m gpu > gpurun_out/${T}_tests.log 2>&1; echo "tests exit $?"; tail -3 gpurun_out/${T}_tests.log
KANTTS_LIB=$GRAFT_REPO_ROOT/kan-tts_amd/variants/libkantts_ARPROF.so timeout 300 python scripts/decode_kernel_bench.py 1 96 > gpurun_out/${T}_decode_prof.log 2>&1
tail -12 gpurun_out/${T}_decode_prof.log
timeout 300 python scripts/infer_breakdown.py 24 kernel > gpurun_out/${T}_infer_breakdown_kernel.log 2>&1
tail -11 gpurun_out/${T}_infer_breakdown_kernel.log
timeout 900 python bench.py --no-hifigan --no-fp32 --no-roofline --steps 10 --no-forward-only --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench exit $?"; tail -2 gpurun_out/${T}_bench.err
python - $T <<'PY'
import json, sys
d = json.loads(open("gpurun_out/%s_bench.json" % sys.argv[1]).read().strip().splitlines()[-1])
inf = d.get("inference", {})
for k, v in inf.items():
    if isinstance(v, dict) and "utterances_per_s" in v:
        print("%-16s %8.1f utt/s  %12.0f samples/s  %.3f s" % (k, v["utterances_per_s"], v["audio_samples_per_s"], v["seconds"]))
print("value", inf.get("value"), inf.get("unit"))
PY
