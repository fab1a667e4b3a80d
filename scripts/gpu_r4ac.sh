#!/bin/bash
# Round 4, visit AC (same as H, after the round-4 changes): kernel TRACE (timestamps per dispatch) of the captured SAM-BERT step, to read the critical path off it
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r4ac_prof -o sb -- python $R/bench.py --steps 12 --warmup 4 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline --no-forward-only > $R/gpurun_out/r4ac_bench.json 2> $R/gpurun_out/r4ac_err.log
cd $R
f=$(find gpurun_out/r4ac_prof -name "*kernel_trace.csv" | head -n 1)
ls -la $f
python - "$f" <<'PY'
import csv, sys, gzip
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), rows[0].keys())
# keep the last ~3 steps' worth: the tail of the trace
keep = rows[-2600:]
w = csv.DictWriter(gzip.open('gpurun_out/r4ac_trace_tail.csv.gz', 'wt'), fieldnames=['Kernel_Name', 'Start_Timestamp', 'End_Timestamp', 'Queue_Id', 'Stream_Id', 'Grid_Size_X', 'Workgroup_Size_X', 'LDS_Block_Size', 'VGPR_Count'], extrasaction='ignore')
w.writeheader()
for r in keep:
    w.writerow(r)
PY
rm -rf gpurun_out/r4ac_prof
tail -c 400 gpurun_out/r4ac_bench.json
