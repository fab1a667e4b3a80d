#!/bin/bash
# Ablation build of the fused PNCA block kernels (csrc/pnca_block.hip, -DPB_DEBUG): KANTTS_PB_DBG masks phases off (see the
# header of the source).  Selected with KANTTS_LIB=kan-tts_amd/variants/libkantts_PBDBG.so; timing only, never the product.
cd "$(dirname "$0")/../kan-tts_amd/csrc"
make -s
mkdir -p ../variants
OTHERS=$(ls *.o | grep -v pnca_block.o)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -DPB_DEBUG -c pnca_block.hip -o /tmp/pnca_block_dbg.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libkantts_PBDBG.so $OTHERS /tmp/pnca_block_dbg.o
ls -la ../variants/libkantts_PBDBG.so
