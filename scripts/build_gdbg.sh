#!/bin/bash
# Ablation build of gemm_fast (KANTTS_GEMM_DBG mask: 1 no operand loads, 2 no output stores, 8 no MFMA).
cd "$(dirname "$0")/../kan-tts_amd/csrc"
make -s
mkdir -p ../variants
OTHERS=$(ls *.o | grep -v gemm_fast.o)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -DF_DEBUG -c gemm_fast.hip -o /tmp/gemm_fast_dbg.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libkantts_GDBG.so $OTHERS /tmp/gemm_fast_dbg.o
ls -la ../variants/libkantts_GDBG.so
