#!/bin/bash
# Round 5: kernel TRACE (start / end per kernel) of the captured SAM-BERT step, fused decoder blocks both ways vs forward only:
# where does the step's time go around the fused backward launches?
T=${1:-r5i}
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in fused; do
  unset KANTTS_NO_PNCA_BLOCK_BWD
  [ $v = fwdonly ] && export KANTTS_NO_PNCA_BLOCK_BWD=1
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${T}_prof_$v -o bench -- python $R/bench.py --no-hifigan --no-inference --no-cpu-baseline --no-fp32 --no-roofline --no-forward-only --steps 10 --warmup 2 > $R/gpurun_out/${T}_rocprof_$v.log 2>&1
  f=$(find $R/gpurun_out/${T}_prof_$v -name "*kernel_trace.csv" | head -n 1)
  python - "$f" "$R/gpurun_out/${T}_trace_tail_$v.csv.gz" <<'PY'
import csv, sys, gzip
rows = list(csv.DictReader(open(sys.argv[1])))
keep = rows[-1500:]
w = csv.DictWriter(gzip.open(sys.argv[2], 'wt'), fieldnames=['Kernel_Name', 'Start_Timestamp', 'End_Timestamp', 'Queue_Id', 'Stream_Id'], extrasaction='ignore')
w.writeheader()
for r in keep:
    r['Kernel_Name'] = r['Kernel_Name'][:60]
    w.writerow(r)
print(len(rows), "kernels traced")
PY
  rm -rf $R/gpurun_out/${T}_prof_$v
  tail -c 300 $R/gpurun_out/${T}_rocprof_$v.log | grep -o '"ms_per_step": [0-9.]*'
done
