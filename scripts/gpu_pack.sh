#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hifigan.py -m gpu -q -x --timeout=600 -p no:cacheprovider 2>&1 | tail -3
timeout 300 python scripts/hifigan_bench.py 32 3 bf16 2>&1 | grep -v -i warn | tail -1 | cut -c1-600 | tee $OUT/pack_bench.log
timeout 300 python scripts/conv_shape_bench.py 32 2>&1 | grep -v -i warn > $OUT/conv_shapes2.log; head -24 $OUT/conv_shapes2.log
