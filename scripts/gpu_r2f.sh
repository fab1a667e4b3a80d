#!/bin/bash
# round-2 visit F: graph-replayed free-running decode (config 5), deterministic clip norm (two-process captured DP step),
# conv_wgrad atomics budget (GAN step), full bench line with every leg
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_decode_graph.py tests/test_ddp_gloo.py tests/test_multiband.py tests/test_hifigan.py tests/test_trainer.py -m gpu -x -q > gpurun_out/r2f_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2f_pytest.log
tail -5 gpurun_out/r2f_pytest.log | cut -c1-300
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2f_bench.log 2> gpurun_out/r2f_bench.err
echo "bench rc=$?"
python - <<'PY'
import json
for ln in open("gpurun_out/r2f_bench.log"):
    if ln.startswith("{"):
        d = json.loads(ln)
        print("sambert", d["ms_per_step"], d["value"])
        print("fp32", d.get("fp32_path"))
        h = d.get("hifigan", {})
        print("hifigan", {k: h.get(k) for k in ("gan_step_ms", "generator_forward_ms", "value", "error")}, h.get("upsampling", {}).get("stage_us"))
        print("inference", json.dumps(d.get("inference"), indent=0)[:2500])
        print("parity", d.get("parity_error"))
        print("cpu", d.get("cpu_baseline"))
PY
tail -3 gpurun_out/r2f_bench.err | cut -c1-300
