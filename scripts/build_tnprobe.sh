#!/bin/bash
# Timing-only build of the weight-gradient kernel with plain stores instead of its fp32 atomics (-DTN_NO_ATOMICS: WRONG
# gradients): the upper bound of what an atomics-free combination of the token slices could gain.
# KANTTS_LIB=kan-tts_amd/variants/libkantts_TNPROBE.so
cd "$(dirname "$0")/../kan-tts_amd/csrc"
make -s
mkdir -p ../variants
OTHERS=$(ls *.o | grep -v gemm_bf16.o)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -DTN_NO_ATOMICS -c gemm_bf16.hip -o /tmp/gemm_bf16_tnprobe.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libkantts_TNPROBE.so $OTHERS /tmp/gemm_bf16_tnprobe.o
ls -la ../variants/libkantts_TNPROBE.so
