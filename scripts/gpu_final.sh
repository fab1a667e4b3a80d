#!/bin/bash
# Last visit of a round when GPU minutes are short: all GPU tests, a short bench (no CPU baseline / HiFi-GAN leg), smoke.
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
timeout 400 python -m pytest tests -m gpu -q -x --timeout=400 -p no:cacheprovider 2>&1 | tail -6 > $OUT/final_pytest.log; tail -4 $OUT/final_pytest.log
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hifigan > $OUT/final_bench.log 2>&1; tail -1 $OUT/final_bench.log | cut -c1-700
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/final_smoke.log 2>&1; tail -1 $OUT/final_smoke.log
