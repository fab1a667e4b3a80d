#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 200 python scripts/gemm_probe.py; KANTTS_GEMM_BM=32 timeout 200 python scripts/gemm_probe.py; KANTTS_GEMM_BM=64 timeout 200 python scripts/gemm_probe.py ) > $OUT/p_probe.log 2>&1
grep -v -i warn $OUT/p_probe.log
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OLDPWD/$OUT/pmc_$tag -o pmc -- python $OLDPWD/scripts/gemm_probe.py fwd128 > $OLDPWD/$OUT/p_pmc_$tag.log 2>&1
done
cd $OLDPWD
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_*/*counter_collection.csv')):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        if 'gemm_fast' in r['Kernel_Name']:
            k = r['Counter_Name']
            agg[k][0] += 1
            agg[k][1] += float(r['Counter_Value'])
    for k, (n, s) in agg.items():
        print(f.split('/')[1], k, 'per-dispatch avg', s / n, 'n', n)
PY
echo done
