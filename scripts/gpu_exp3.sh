#!/bin/bash
cd "$(dirname "$0")/.."
for v in "" BK64 NT DIRECT_EPI; do
  echo "== variant ${v:-base}"
  if [ -n "$v" ]; then export KANTTS_LIB=$PWD/kan-tts_amd/variants/libkantts_$v.so; else unset KANTTS_LIB; fi
  timeout 120 python scripts/gemm_probe.py 2>&1 | grep -v -i "warn\|amdgpu.ids" | grep -v "wgrad\|matmul\|copy"
done
