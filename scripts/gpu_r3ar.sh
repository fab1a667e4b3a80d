#!/bin/bash
# round-3 visit AR: branch-private gradient copies (ops._BranchExit): the batch-32 fp32 generator test eight times, then the HiFi-GAN GPU tests
mkdir -p gpurun_out
echo "== with _BranchExit" | tee gpurun_out/r3ar_race.log
for i in 1 2 3 4 5 6 7 8; do timeout 300 python -m pytest tests/test_bench_config_parity.py -m gpu -q -k "batch32 and fp32" 2>&1 | grep -E "passed|failed" | cut -c1-40; done | sort | uniq -c | tee -a gpurun_out/r3ar_race.log
timeout 600 python -m pytest tests/test_hifigan.py tests/test_trainer.py -m gpu -q -k "gan or hifigan or GAN" 2>&1 | tail -n 2
