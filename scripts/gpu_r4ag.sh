#!/bin/bash
# Round 4, visit AG (same as V, closing code): kernel TRACE of the captured GAN step (timestamps per dispatch) to read its critical path
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r4ag_prof -o gan -- python $R/scripts/hifigan_bench.py 32 3 bf16 > $R/gpurun_out/r4ag_bench.json 2> $R/gpurun_out/r4ag_err.log
cd $R
f=$(find gpurun_out/r4ag_prof -name "*kernel_trace.csv" | head -n 1)
ls -la $f
python - "$f" <<'PY'
import csv, sys, gzip
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows))
keep = rows[-9000:]
w = csv.DictWriter(gzip.open('gpurun_out/r4ag_trace_tail.csv.gz', 'wt'), fieldnames=['Kernel_Name', 'Start_Timestamp', 'End_Timestamp', 'Queue_Id', 'Grid_Size_X', 'Workgroup_Size_X'], extrasaction='ignore')
w.writeheader()
for r in keep:
    w.writerow(r)
PY
rm -rf gpurun_out/r4ag_prof
tail -c 300 gpurun_out/r4ag_bench.json
