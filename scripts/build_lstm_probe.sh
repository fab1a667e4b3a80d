#!/bin/bash
# Builds the MFMA-LSTM measurement prototype (scripts/lstm_mfma_probe.hip) -> kan-tts_amd/variants/liblstm_mfma_probe.so
cd "$(dirname "$0")/.."
mkdir -p kan-tts_amd/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Rpass-analysis=kernel-resource-usage scripts/lstm_mfma_probe.hip \
  -o kan-tts_amd/variants/liblstm_mfma_probe.so 2>&1 | grep -E "error|VGPRs:|Scratch|LDS Size"
ls -la kan-tts_amd/variants/liblstm_mfma_probe.so
