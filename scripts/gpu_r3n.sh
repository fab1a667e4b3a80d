#!/bin/bash
# round-3 visit N: window-form narrow convolution kernel; A/B of the round's switches on the captured GAN step
mkdir -p gpurun_out/r3n
timeout 900 python -m pytest tests/test_cconv.py tests/test_hifigan.py -m gpu -x -q > gpurun_out/r3n/pytest.log 2>&1; tail -n 2 gpurun_out/r3n/pytest.log
run() { tag=$1; shift; env "$@" timeout 400 python scripts/hifigan_bench.py 32 4 bf16 > gpurun_out/r3n/hifigan_$tag.log 2>&1; echo "$tag: $(grep -o '"generator_forward_ms": [0-9.]*' gpurun_out/r3n/hifigan_$tag.log) $(grep -o '"gan_step_ms": [0-9.]*' gpurun_out/r3n/hifigan_$tag.log) $(grep -o '"gan_step_graph_ms": [0-9.]*' gpurun_out/r3n/hifigan_$tag.log)"; }
run all X=1
run no_narrow KANTTS_NO_CCONV_NARROW=1
run no_wimages KANTTS_NO_WEIGHT_IMAGES=1
run no_resstack KANTTS_NO_RES_STACK=1
run no_c1mfma KANTTS_C1_NO_MFMA=1
run no_taps KANTTS_NO_WGRAD_TAPS=1
run all2 X=1
