#!/bin/bash
# Experiment builds of libkantts_hip.so with one part of lstm_fwd_pair_kernel's step masked (lstm.hip: LSTM_ABL); results
# are wrong, only the time of scripts/lstm_bench.py is read.  -> kan-tts_amd/variants/libkantts_lstmabl{1,2,4,8,15}.so
cd "$(dirname "$0")/../kan-tts_amd/csrc"
make -s
mkdir -p ../variants
OTHERS=$(ls *.o | grep -v "^lstm.o")
for v in 1 2 4 8 15; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -DLSTM_ABL=$v -c lstm.hip -o /tmp/lstm_abl$v.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libkantts_lstmabl$v.so $OTHERS /tmp/lstm_abl$v.o
done
ls ../variants
