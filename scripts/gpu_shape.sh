#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
timeout 400 python scripts/conv_shape_bench.py 32 2>&1 | grep -v -i warn > $OUT/conv_shapes.log; head -70 $OUT/conv_shapes.log
