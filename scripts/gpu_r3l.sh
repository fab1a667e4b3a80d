#!/bin/bash
# round-3 visit L: weight-norm kernel with bf16 images; schedule parity tests (SAM-BERT graph == eager at the full
# config, GAN branch streams), HiFi-GAN batch-32 parity against the reference fixture; GAN bench
mkdir -p gpurun_out/r3l
timeout 1200 python -m pytest tests/test_trainer.py tests/test_bench_config_parity.py tests/test_cconv.py tests/test_hifigan.py -m gpu -x -q -k "schedule or branch_streams or batch32 or hifigan_v1_512 or cconv or fused or graphed or handover" > gpurun_out/r3l/pytest.log 2>&1; tail -n 3 gpurun_out/r3l/pytest.log
timeout 400 python scripts/hifigan_bench.py 32 4 bf16 > gpurun_out/r3l/hifigan.log 2>&1
echo "$(grep -o '"generator_forward_ms": [0-9.]*' gpurun_out/r3l/hifigan.log) $(grep -o '"gan_step_ms": [0-9.]*' gpurun_out/r3l/hifigan.log) $(grep -o '"gan_step_graph_ms": [0-9.]*' gpurun_out/r3l/hifigan.log)"
