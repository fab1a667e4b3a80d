#!/bin/bash
# Experiment builds of libkantts_hip.so: same sources, one macro each (gemm_fast.hip F_VARIANT_*).
cd "$(dirname "$0")/../kan-tts_amd/csrc"
make -s
mkdir -p ../variants
OTHERS=$(ls *.o | grep -v gemm_fast.o)
for v in BK64; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -DF_VARIANT_$v -c gemm_fast.hip -o /tmp/gemm_fast_$v.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libkantts_$v.so $OTHERS /tmp/gemm_fast_$v.o
done
ls -la ../variants
