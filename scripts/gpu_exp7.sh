#!/bin/bash
cd "$(dirname "$0")/.."
bash scripts/gpu_exp5.sh
bash scripts/gpu_pmc2.sh > /dev/null 2>&1
grep -E "conv_win_kernel<true, 4, 2>|conv_wgrad_kernel<true, 64>" gpurun_out/pmc4_summary.txt | grep -E "INSTS_VALU|INSTS_SALU|INSTS_MFMA|WAVE_CYCLES|WAIT_ANY|MFMA_BUSY|ACTIVE_INST_ANY"
