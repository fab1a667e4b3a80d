#!/bin/bash
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT" "SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $OUT/pmc2_$tag
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc2_$tag -o pmc -- python $OLDPWD/scripts/gemm_probe.py fwd > $OUT/p2_pmc_$tag.log 2>&1
done
cd $OLDPWD
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc2_*/*counter_collection.csv')):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: [0, 0.0])
    meta = {}
    for r in rows:
        if 'gemm_fast' in r['Kernel_Name']:
            k = (r['Kernel_Name'][:60], r['Grid_Size'], r['Counter_Name'])
            agg[k][0] += 1
            agg[k][1] += float(r['Counter_Value'])
            meta[r['Kernel_Name'][:60]] = (r.get('VGPR_Count'), r.get('Accum_VGPR_Count'), r.get('LDS_Block_Size'), r.get('SGPR_Count'))
    for k, (n, s) in sorted(agg.items()):
        print(k, 'avg', s / n, 'n', n)
    print(meta)
PY
