# The two counter passes of scripts/ffn_pmc_probe.py (FETCH_SIZE, WRITE_SIZE: separate runs, kernel trace only) and their
# summary as JSON (scripts/pmc_to_json.py).  Usage (GPU box, from the repo root): bash scripts/ffn_pmc_passes.sh TAG [env...]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; T=${1:?tag}; shift
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && env "$@" timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${T}_pmc_$c -o p -- python $R/scripts/ffn_pmc_probe.py > $O/${T}_pmc_$c.log 2>&1 )
done
f=$(find $O/${T}_pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
w=$(find $O/${T}_pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python $R/scripts/pmc_to_json.py "$f" "$w" "round 6 ($T): rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python scripts/ffn_pmc_probe.py" $O/${T}_ffn_block_pmc.json
rm -rf $O/${T}_pmc_FETCH_SIZE $O/${T}_pmc_WRITE_SIZE
python -c "
import json,sys
d=json.load(open('$O/${T}_ffn_block_pmc.json'))['bf16']
for k,v in d['launches'].items(): print('%-45s traffic %6.1f MB  algorithmic %6.1f MB  x%.2f' % (k, v['traffic_bytes']/1e6, v['algorithmic_bytes']/1e6, v['traffic_bytes']/v['algorithmic_bytes']))
"
