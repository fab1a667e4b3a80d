#!/bin/bash
# Phase-timing build of the one-launch inference loops (csrc/ar_infer.hip, -DAR_PROFILE): selected with
# KANTTS_LIB=kan-tts_amd/variants/libkantts_ARPROF.so; timing only, never the product (scripts/decode_kernel_bench.py).
cd "$(dirname "$0")/../kan-tts_amd/csrc"
make -s
mkdir -p ../variants
OTHERS=$(ls *.o | grep -v ar_infer.o)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -DAR_PROFILE -c ar_infer.hip -o /tmp/ar_infer_prof.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libkantts_ARPROF.so $OTHERS /tmp/ar_infer_prof.o
ls -la ../variants/libkantts_ARPROF.so
