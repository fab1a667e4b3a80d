#!/bin/bash
# Round 5: ablation timings of the fused PNCA block kernels (experiment build -DPB_DEBUG, masks price the phases).
T=${1:-r5f}
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
python scripts/pnca_block_ablate.py > gpurun_out/${T}_ablate.log 2>&1
export KANTTS_LIB=$GRAFT_REPO_ROOT/kan-tts_amd/variants/libkantts_PBDBG.so
for m in 0 1 2 4 8 16 3 7 6 24 31; do
  KANTTS_PB_DBG=$m python scripts/pnca_block_ablate.py >> gpurun_out/${T}_ablate.log 2>&1
done
grep KANTTS_PB_DBG gpurun_out/${T}_ablate.log; grep -i "error\|Traceback" gpurun_out/${T}_ablate.log | head
