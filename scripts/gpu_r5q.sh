#!/bin/bash
# Round 5: full GPU suite + smoke + full default bench (new roofline object, copy roof, inference parity + CPU baseline).
T=${1:-r5q}
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -n 4 gpurun_out/${T}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; echo "smoke exit $?"; grep -v Warning gpurun_out/${T}_smoke.log | tail -n 2
timeout 1500 python bench.py > gpurun_out/${T}_bench_full.log 2> gpurun_out/${T}_bench_full.err; echo "bench exit $?"; grep "^\[bench" gpurun_out/${T}_bench_full.err | tail -n 12
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5q_bench_full.log").read().strip().splitlines()[-1])
r = d["roofline"]
print({k: d[k] for k in ("value", "ms_per_step")})
print({k: r.get(k) for k in ("bound", "achieved", "peak", "frac", "forward_ms")})
print("families", {k: (round(v["mean_launch_us"], 1), round(v["mfma_frac"], 4), round(v["hbm_frac"], 3)) for k, v in r.get("families", {}).items()})
h = d.get("hifigan", {})
print("hifigan", {k: h.get(k) for k in ("value", "ms_per_step")}, h.get("upsampling"))
print("inference", {k: d.get("inference", {}).get(k) for k in ("value", "parity_error", "cpu_baseline", "batch1_graph")})
print("melspec", d.get("melspec"))
PY
