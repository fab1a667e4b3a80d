#!/bin/bash
# round-2 visit N: contraction census of the training step by launch shape; HiFi-GAN V1 parity values at the bench config
mkdir -p gpurun_out
timeout 400 python scripts/gemm_census.py > gpurun_out/r2n_gemm_census.log 2>&1; tail -50 gpurun_out/r2n_gemm_census.log
timeout 900 python -m pytest "tests/test_bench_config_parity.py" -m gpu -x -q -k hifigan > gpurun_out/r2n_pytest_hifi_parity.log 2>&1; tail -3 gpurun_out/r2n_pytest_hifi_parity.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/parity_at_bench_configs.json"))
for k, v in d.items():
    if "hifigan" in k:
        print(k, json.dumps(v))
PY
