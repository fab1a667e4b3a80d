#!/bin/bash
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_hifigan.py tests/test_gpu_sambert.py -m gpu -q -x --timeout=900 -p no:cacheprovider 2>&1 | grep -E "^E|passed|failed|FAILED" | cut -c1-300 | head -20
timeout 200 python scripts/gemm_probe.py 2>&1 | grep -v -i warn | grep -v "splitk= [124] \|splitk=32" > $OUT/q6_probe.log; cat $OUT/q6_probe.log
timeout 400 python bench.py --no-cpu-baseline > $OUT/q6_bench.log 2>&1; tail -1 $OUT/q6_bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; h=d['hifigan']
print('sambert ms/step %.2f'%d['ms_per_step'], 'frac %.3f'%r['frac'], r['launch_us'])
print('hifigan gan_step_ms %.1f'%h['gan_step_ms'], 'G fwd ms %.2f'%h['generator_forward_ms'], 'upsampling', h['upsampling']['ms'], h['upsampling']['frac'])"
cd /tmp
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf $OUT/pmc3_$set
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc3_$set -o pmc -- python $OLDPWD/scripts/gemm_probe.py fwd > $OUT/p3_pmc_$set.log 2>&1
done
cd $OLDPWD
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc3_*/*counter_collection.csv')):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        if 'gemm_fast' in r['Kernel_Name']:
            k = (r['Kernel_Name'][:62], r['Grid_Size'], r['Counter_Name'])
            agg[k][0] += 1; agg[k][1] += float(r['Counter_Value'])
    for k, (n, s) in sorted(agg.items()): print(k, 'avg', s / n, 'n', n)
PY
