#!/bin/bash
T=${1:-r5v}
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/decode_kernel_bench.py 1 96 > gpurun_out/${T}_decode_bench.log 2>&1
timeout 300 python scripts/decode_kernel_bench.py 32 96 >> gpurun_out/${T}_decode_bench.log 2>&1
grep "layers" gpurun_out/${T}_decode_bench.log
