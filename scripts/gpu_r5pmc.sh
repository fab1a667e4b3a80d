#!/bin/bash
# Round 5: PMC passes (FETCH_SIZE and WRITE_SIZE each in its own pass, kernel trace only -- gpurun refuses more) over the
# round's kernels: two eager SAM-BERT steps and one run of the one-launch decoder loop.  Summaries: scripts/pmc_summary.py.
T=${1:-r5pmc}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/${T}_step_$C -o p -- python $R/bench.py --mode eager --steps 2 --warmup 1 --no-hifigan --no-inference --no-cpu-baseline --no-fp32 --no-roofline --no-forward-only > $R/gpurun_out/${T}_step_$C.log 2>&1
  f=$(find $R/gpurun_out/${T}_step_$C -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_summary.py "$f" pnca_ ffn_pair lstm_ bgemm_tn bgemm_nt fsmn_ masked_l1 adam sumsq embed_sum rows_sum teacher_plan > $R/gpurun_out/${T}_step_$C.txt
  rm -rf $R/gpurun_out/${T}_step_$C
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/${T}_dec_$C -o p -- python $R/scripts/decode_kernel_bench.py 1 96 > $R/gpurun_out/${T}_dec_$C.log 2>&1
  f=$(find $R/gpurun_out/${T}_dec_$C -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_summary.py "$f" pnca_decode_run dur_ar > $R/gpurun_out/${T}_dec_$C.txt
  rm -rf $R/gpurun_out/${T}_dec_$C
done
grep -h "pnca_\|ffn_pair" $R/gpurun_out/${T}_step_*.txt | head -20
grep -h "pnca_decode" $R/gpurun_out/${T}_dec_*.txt | tail -8
