#!/bin/bash
# Round 5, first visit: the parity holes the round-4 review named, as -m gpu tests; RCCL executed once (world size 1);
# baseline bench of the round's starting code on this box.
T=${1:-r5a}
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest -q -x -m gpu \
  tests/test_melspec.py tests/test_gpu_ops.py::test_attention_longer_than_a_workgroup_gpu \
  tests/test_config5_inference.py tests/test_bench_config_parity.py -k "not hifigan" \
  > gpurun_out/${T}_new_tests.log 2>&1; echo "new tests exit $?"; tail -n 8 gpurun_out/${T}_new_tests.log
timeout 120 python __graft_entry__.py smoke > gpurun_out/${T}_smoke.log 2>&1; echo "smoke exit $?"; tail -n 2 gpurun_out/${T}_smoke.log
timeout 600 python bench.py --rccl-world1 --no-hifigan --no-inference --no-cpu-baseline --no-fp32 --no-roofline \
  > gpurun_out/${T}_bench_rccl_world1.json 2> gpurun_out/${T}_bench_rccl_world1.err; echo "rccl bench exit $?"
tail -c 1500 gpurun_out/${T}_bench_rccl_world1.json; tail -n 5 gpurun_out/${T}_bench_rccl_world1.err
timeout 900 python bench.py > gpurun_out/${T}_bench_full.log 2> gpurun_out/${T}_bench_full.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5a_bench_full.log").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"].get("forward_ms"), d.get("inference", {}).get("parity_error"),
      d.get("inference", {}).get("cpu_baseline"))
PY
