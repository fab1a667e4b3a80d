#!/bin/bash
# round-3 visit AQ: branch-stream race of the fp32 generator backward: inside the backward pass, or at its end?
mkdir -p gpurun_out
run() { echo "== $*" | tee -a gpurun_out/r3aq_race.log; for i in 1 2 3 4 5 6; do env "$@" timeout 300 python -m pytest tests/test_bench_config_parity.py -m gpu -q -k "batch32 and fp32" 2>&1 | grep -E "passed|failed" | cut -c1-80; done | sort | uniq -c | tee -a gpurun_out/r3aq_race.log; }
run X=1
run KANTTS_TEST_SYNC_AFTER_BACKWARD=1
run KANTTS_NO_BRANCH_STREAMS=1
