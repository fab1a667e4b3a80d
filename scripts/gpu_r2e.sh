#!/bin/bash
# round-2 visit E: grouped deferred weight gradients, LSTM projections on the bf16 kernels, LayerNorm backward with fewer
# same-address atomics: A/B of the grouping, parity at the benchmarked config, kernel profile
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_bf16_ops.py tests/test_multiband.py "tests/test_bench_config_parity.py::test_sambert_full_b32_matches_oracle" tests/test_gpu_sambert.py tests/test_ddp_gloo.py -m gpu -x -q > gpurun_out/r2e_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2e_pytest.log
tail -4 gpurun_out/r2e_pytest.log | cut -c1-300
for flag in "" "--no-wgrad-group"; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-hifigan --no-cpu-baseline --no-fp32 $flag > gpurun_out/r2e_bench$flag.log 2> gpurun_out/r2e_bench$flag.err
  echo "bench $flag rc=$?"
  python - <<PY
import json
for ln in open("gpurun_out/r2e_bench$flag.log"):
    if ln.startswith("{"):
        d = json.loads(ln); r = d["roofline"]
        print("$flag", d["ms_per_step"], d["value"], r["gemm_launches_per_step"], r["gemm_ms_per_step_eager_events"])
PY
done
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2e_prof -o sam -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-hifigan --no-cpu-baseline --no-fp32 > $GRAFT_REPO_ROOT/gpurun_out/r2e_rocprof.log 2>&1 )
f=$(find gpurun_out/r2e_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -60 "$f" > gpurun_out/r2e_sambert_kernel_stats_top.csv && cut -c1-150 gpurun_out/r2e_sambert_kernel_stats_top.csv | head -32
rm -rf gpurun_out/r2e_prof
