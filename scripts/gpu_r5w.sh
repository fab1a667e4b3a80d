#!/bin/bash
T=${1:-r5w}
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ar_kernels.py -q -m gpu > gpurun_out/${T}_tests.log 2>&1; echo "tests exit $?"; tail -3 gpurun_out/${T}_tests.log
KANTTS_LIB=$GRAFT_REPO_ROOT/kan-tts_amd/variants/libkantts_ARPROF.so timeout 300 python scripts/decode_kernel_bench.py 1 96 > gpurun_out/${T}_decode_prof.log 2>&1
tail -12 gpurun_out/${T}_decode_prof.log
timeout 300 python scripts/decode_kernel_bench.py 32 96 > gpurun_out/${T}_decode_bench.log 2>&1
grep layers gpurun_out/${T}_decode_bench.log
