#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_hifigan.py tests/test_gpu_sambert.py -m gpu -q -x --timeout=900 -p no:cacheprovider 2>&1 | grep -E "^E|passed|failed|FAILED" | cut -c1-300 | head -20
timeout 200 python scripts/gemm_probe.py 2>&1 | grep -v -i warn > $OUT/q4_probe.log; cat $OUT/q4_probe.log
timeout 400 python bench.py --no-cpu-baseline > $OUT/q4_bench.log 2>&1; tail -1 $OUT/q4_bench.log | cut -c1-2500
