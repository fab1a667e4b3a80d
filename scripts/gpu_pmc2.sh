#!/bin/bash
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $OUT/pmc4_$tag
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc4_$tag -o pmc -- python $OLDPWD/scripts/hifigan_bench.py 16 1 bf16 > $OUT/p4_pmc_$tag.log 2>&1
done
cd $OLDPWD
python - <<'PY'
import csv, glob, collections
out = open('gpurun_out/pmc4_summary.txt', 'w')
for f in sorted(glob.glob('gpurun_out/pmc4_*/*counter_collection.csv')):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        n = r['Kernel_Name']
        if 'conv_win_kernel' in n or 'conv_wgrad_kernel' in n:
            k = (n[:58], r['Counter_Name'])
            agg[k][0] += 1; agg[k][1] += float(r['Counter_Value'])
    for k, (n, s) in sorted(agg.items()):
        out.write('%-60s %-28s total %.4g  n %d\n' % (k[0], k[1], s, n))
out.close()
print(open('gpurun_out/pmc4_summary.txt').read())
PY
find $OUT/pmc4_* -name "*.csv" -size +3M -delete 2>/dev/null
