#!/bin/bash
T=${1:-r5ac}
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/diag_side_stream.py > gpurun_out/${T}_diag_side.log 2>&1
grep "rep" gpurun_out/${T}_diag_side.log | cut -c1-400
