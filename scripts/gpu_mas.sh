#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_mas.py -m gpu -q -x --timeout=600 -p no:cacheprovider 2>&1 | tail -25 > $OUT/mas_pytest.log; tail -25 $OUT/mas_pytest.log
timeout 200 python scripts/mas_bench.py > $OUT/mas_bench.log 2>&1; cat $OUT/mas_bench.log
