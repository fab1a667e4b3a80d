#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 700 python -m pytest tests/test_melspec.py tests/test_hifigan.py -m gpu -q -n 3 --timeout=600 -p no:cacheprovider 2>&1 | tail -30 > $OUT/q3_pytest.log; tail -12 $OUT/q3_pytest.log
echo done
