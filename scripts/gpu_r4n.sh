#!/bin/bash
# Round 4, visit N: register-resident mel-STFT kernel: parity on the device, A/B against the radix-2 kernel, grid sweep
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_melspec.py tests/test_multiband.py tests/test_dsp_reference_fixture.py tests/test_hifigan.py -m gpu -q -x -k "melspec or register or mrstft or multispec or dsp or gan_step" 2>&1 | tail -5 | tee gpurun_out/r4n_tests.log
MEL_SWEEP=1 timeout 600 python scripts/mel_bench.py 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r4n_mel_bench.log
