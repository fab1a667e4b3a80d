#!/bin/bash
# round-3 visit K: PNCA x / h attention on two streams (forward and backward), forward-only timing in bench.py
mkdir -p gpurun_out/r3k
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sambert.py tests/test_trainer.py tests/test_device_batching.py -m gpu -x -q > gpurun_out/r3k/pytest.log 2>&1; tail -n 2 gpurun_out/r3k/pytest.log
timeout 600 python bench.py --no-hifigan --no-cpu-baseline --no-fp32 --no-inference > gpurun_out/r3k/bench_sambert.log 2>gpurun_out/r3k/bench_sambert.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r3k/bench_sambert.log").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("streams: ms/step %.3f  fwd_ms %s  fwd_mfma %s" % (d["ms_per_step"], r.get("forward_ms"), r.get("forward_mfma_frac")))
except Exception as e:
    print("parse failed", e)
PY
KANTTS_NO_ATTN_STREAMS=1 timeout 600 python bench.py --no-hifigan --no-cpu-baseline --no-fp32 --no-inference > gpurun_out/r3k/bench_sambert_noattn.log 2>/dev/null
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r3k/bench_sambert_noattn.log").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("no attn streams: ms/step %.3f  fwd_ms %s" % (d["ms_per_step"], r.get("forward_ms")))
except Exception as e:
    print("parse failed", e)
PY
tail -n 3 gpurun_out/r3k/bench_sambert.err
