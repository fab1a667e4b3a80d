#!/bin/bash
# round-3 visit Z: attention launches on their own (band 5 / 20, dropout on / off)
mkdir -p gpurun_out
timeout 200 python scripts/attn_bench.py 2>&1 | grep -v Warning | tee gpurun_out/r3z_attn_bench.log
