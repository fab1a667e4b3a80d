#!/bin/bash
# round-2 visit AE: hardware queue count (HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues, default 4) with the
# eight discriminator streams of the GAN step and the three streams of the SAM-BERT step
mkdir -p gpurun_out
for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q timeout 200 python scripts/hifigan_bench.py 32 4 bf16 > gpurun_out/r2ae_hifigan_q$q.log 2>&1
  echo "GPU_MAX_HW_QUEUES=$q: $(grep -o '"gan_step_ms": [0-9.]*' gpurun_out/r2ae_hifigan_q$q.log)"
done
GPU_MAX_HW_QUEUES=8 timeout 200 python bench.py --steps 20 --warmup 5 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference > gpurun_out/r2ae_bench_q8.log 2>&1
echo "SAM-BERT q=8: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2ae_bench_q8.log | head -1)"
