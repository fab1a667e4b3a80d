#!/bin/bash
# Round 5, second visit: config-5 parity after the position-term fix, the repeated-layer weight-norm backward on the device,
# RCCL with the exchange path forced on (graph segments + the all-reduce between their replays, world size 1).
T=${1:-r5b}
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest -q -m gpu tests/test_config5_inference.py \
  tests/test_hifigan.py::test_weight_norm_table_backward_with_a_layer_applied_twice_gpu \
  > gpurun_out/${T}_new_tests.log 2>&1; echo "new tests exit $?"; grep -E "^(fp32|bf16) \{|passed|failed" gpurun_out/${T}_new_tests.log | cut -c 1-900
timeout 600 python bench.py --rccl-world1 --no-hifigan --no-inference --no-cpu-baseline --no-fp32 --no-roofline \
  > gpurun_out/${T}_bench_rccl_world1.json 2> gpurun_out/${T}_bench_rccl_world1.err; echo "rccl bench exit $?"
tail -c 1100 gpurun_out/${T}_bench_rccl_world1.json; grep -v "^\[W\|amdgpu.ids" gpurun_out/${T}_bench_rccl_world1.err | tail -n 8
