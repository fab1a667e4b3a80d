#!/bin/bash
# Round 4, visit Q: the PMC passes of the feed-forward block again (the json step of the final visit did not know the new
# bgemm_tn template arguments)
T=r4q
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/${T}_pmc_$c
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${T}_pmc_$c -o pmc -- python $R/scripts/ffn_pmc_probe.py > $R/gpurun_out/${T}_pmc_$c.log 2>&1
  f=$(find $R/gpurun_out/${T}_pmc_$c -name "*counter_collection.csv" | head -n 1)
  [ -n "$f" ] && cp "$f" $R/gpurun_out/${T}_$c.csv
  rm -rf $R/gpurun_out/${T}_pmc_$c
done
python $R/scripts/pmc_to_json.py $R/gpurun_out/${T}_FETCH_SIZE.csv $R/gpurun_out/${T}_WRITE_SIZE.csv "round 4 visit Q: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python scripts/ffn_pmc_probe.py" $R/gpurun_out/${T}_ffn_block_pmc.json | tail -n 40
