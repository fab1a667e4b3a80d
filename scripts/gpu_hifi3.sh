#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/hprof -o hifi -- python $OLDPWD/scripts/hifigan_bench.py 32 2 bf16 > $OLDPWD/$OUT/h_rocprof.log 2>&1 )
grep -v -i warn $OUT/h_rocprof.log | tail -1 | cut -c1-600
f=$(find $OUT/hprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" | cut -c1-200 > $OUT/h_kernel_stats_top.csv
python - <<'PY'
import csv, collections
rows=list(csv.DictReader(open('gpurun_out/hprof/hifi_kernel_trace.csv')))
agg=collections.defaultdict(lambda:[0,0])
for r in rows:
    n=r['Kernel_Name']
    key=(n[:70], r['Grid_Size_X'],r['Grid_Size_Y'],r['Grid_Size_Z'])
    d=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    agg[key][0]+=d; agg[key][1]+=1
tot=sum(v[0] for v in agg.values())
print('total kernel ms',tot/1e6, 'launches', len(rows))
with open('gpurun_out/h_by_grid.txt','w') as f:
    for k,v in sorted(agg.items(), key=lambda kv:-kv[1][0])[:70]:
        f.write('%-72s grid=%s,%s,%s n=%d tot=%.2fms avg=%.1fus\n'%(k[0],k[1],k[2],k[3],v[1],v[0]/1e6,v[0]/v[1]/1e3))
PY
rm -f $OUT/hprof/hifi_kernel_trace.csv
head -30 $OUT/h_by_grid.txt
