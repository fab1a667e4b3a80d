#!/bin/bash
# Round 4, visit O: mel-STFT register kernel: where the wave cycles go (SQ counters, two passes), then the timing
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/r4o_pmc_$tag -o mel -- python $R/scripts/mel_pmc.py > /dev/null 2> $R/gpurun_out/r4o_pmc_err.log
done
cd $R
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r4o_pmc_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if "melspec" in row["Kernel_Name"]:
                a = acc[row["Counter_Name"]]
                a[0] += float(row["Counter_Value"]); a[1] += 1
        for k, (v, n) in sorted(acc.items()):
            print("%-24s %.4g per launch (%d records)" % (k, v / max(n, 1) * (n / 3 if n % 3 == 0 else 1), n))
PY
MEL_SWEEP=1 timeout 600 python scripts/mel_bench.py 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r4o_mel_bench.log
