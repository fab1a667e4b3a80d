#!/bin/bash
# Ablation build of conv_win (KANTTS_CW_DBG mask: 1 no window loads, 2 no output stores, 4 no weight loads, 8 no MFMA).
cd "$(dirname "$0")/../kan-tts_amd/csrc"
make -s
mkdir -p ../variants
OTHERS=$(ls *.o | grep -v conv_win.o)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -DCW_DEBUG -c conv_win.hip -o /tmp/conv_win_dbg.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libkantts_CWDBG.so $OTHERS /tmp/conv_win_dbg.o
ls -la ../variants/libkantts_CWDBG.so
