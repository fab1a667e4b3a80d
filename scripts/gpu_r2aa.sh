#!/bin/bash
# round-2 visit AA: the sub-discriminators of MPD / MSD on their own streams: parity tests, GAN step A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hifigan.py tests/test_hifigan_nsf.py tests/test_multiband.py tests/test_trainer.py tests/test_bench_config_parity.py -m gpu -x -q -k "hifigan or gan or GAN or multiband or nsf or spec or pqmf" > gpurun_out/r2aa_pytest.log 2>&1; tail -3 gpurun_out/r2aa_pytest.log
for v in 1 ""; do
  KANTTS_NO_BRANCH_STREAMS=$v timeout 300 python scripts/hifigan_bench.py 32 4 bf16 > gpurun_out/r2aa_hifigan_nostreams_$v.log 2>&1
  echo "KANTTS_NO_BRANCH_STREAMS='$v': $(grep -o '"gan_step_ms": [0-9.]*' gpurun_out/r2aa_hifigan_nostreams_$v.log) $(grep -o '"max_mem_GB": [0-9.]*' gpurun_out/r2aa_hifigan_nostreams_$v.log)"
done
