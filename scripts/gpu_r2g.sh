#!/bin/bash
# round-2 visit G: streaming upsampling kernels, LSTM prefetch rings, LN backward with fused residual gradient,
# conv_wgrad atomics budget A/B (per-shape table), bench
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hifigan.py tests/test_gpu_bf16_ops.py tests/test_gpu_ops.py "tests/test_bench_config_parity.py" tests/test_gpu_sambert.py -m gpu -x -q > gpurun_out/r2g_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2g_pytest.log
tail -4 gpurun_out/r2g_pytest.log | cut -c1-400
for cap in 0 3 12; do
  KANTTS_WGRAD_ATOMICS=$cap timeout 200 python scripts/conv_shape_bench.py 32 > gpurun_out/r2g_conv_shapes_cap$cap.log 2>&1
  echo "cap=$cap: $(grep 'conv launches total' gpurun_out/r2g_conv_shapes_cap$cap.log)"
  grep -E "conv_wgrad" gpurun_out/r2g_conv_shapes_cap$cap.log | awk '{s+=$1} END {print "  conv_wgrad ms in top-60:", s}'
done
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference > gpurun_out/r2g_bench.log 2> gpurun_out/r2g_bench.err
echo "bench rc=$?"
python - <<'PY'
import json
for ln in open("gpurun_out/r2g_bench.log"):
    if ln.startswith("{"):
        d = json.loads(ln)
        print("sambert", d["ms_per_step"], d["value"], d["roofline"]["launch_us"], d["roofline"]["frac"])
        h = d.get("hifigan", {})
        print("hifigan", {k: h.get(k) for k in ("gan_step_ms", "generator_forward_ms", "error")})
        print("up bf16", h.get("upsampling"))
        print("up fp32", h.get("upsampling_fp32_storage"))
PY
tail -3 gpurun_out/r2g_bench.err | cut -c1-300
