#!/bin/bash
# round-2 visit A: new parity tests at the benchmarked configs + the new bench contract (before the bf16-storage work)
mkdir -p gpurun_out
python -m pytest tests/test_bench_config_parity.py tests/test_trainer.py tests/test_hifigan.py::test_hifigan_v1_gpu_matches_reference_fixture -m gpu -x -q -s > gpurun_out/r2a_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2a_pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r2a_bench.log 2> gpurun_out/r2a_bench.err
echo "bench rc=$?" >> gpurun_out/r2a_bench.log
tail -5 gpurun_out/r2a_pytest.log
tail -c 3000 gpurun_out/r2a_bench.log
