#!/bin/bash
# One parameterised GPU visit (replaces the per-visit gpu_r5*.sh files of round 5, which stay in git history).
#   gpurun --timeout S -- 'bash scripts/gpu_visit.sh TAG step [step ...]'
# Steps (run in the order given; every output lands in gpurun_out/TAG_*):
#   suite            python -m pytest tests -m gpu
#   tests:EXPR       python -m pytest tests -m gpu -k EXPR
#   smoke            __graft_entry__.smoke()
#   bench            python bench.py (the driver's default line)
#   bench:ARGS       python bench.py ARGS            (commas in ARGS become spaces)
#   stats:NAME:CMD   rocprofv3 --kernel-trace --stats of CMD (commas become spaces; $R = repo root) -> TAG_NAME_kernel_stats_top.csv
#   pmc:NAME:CTRS:CMD  rocprofv3 --pmc CTRS (plus-separated) --kernel-trace of CMD -> TAG_NAME_pmc.csv (own pass, no other trace domain)
#   run:NAME:CMD     any command, log to TAG_NAME.log
T=${1:?tag}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp R
cd $R
for step in "$@"; do
  kind=${step%%:*}; rest=${step#*:}
  case $kind in
    suite) timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/${T}_pytest_gpu.log 2>&1; echo "pytest -m gpu exit $?"; tail -n 6 $O/${T}_pytest_gpu.log ;;
    tests) timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider -k "$rest" > $O/${T}_pytest_k.log 2>&1; echo "pytest -k exit $?"; tail -n 15 $O/${T}_pytest_k.log ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${T}_smoke.log 2>&1; echo "smoke exit $?"; tail -n 2 $O/${T}_smoke.log ;;
    bench) if [ "$rest" = "bench" ]; then a=""; n=full; else a=${rest//,/ }; n=$(echo "$rest" | tr -c 'a-zA-Z0-9\n' '_' | cut -c1-40); fi
           timeout 1200 python bench.py $a > $O/${T}_bench_$n.log 2> $O/${T}_bench_$n.err; echo "bench $a exit $?"; tail -n 1 $O/${T}_bench_$n.log | cut -c1-2500 ;;
    stats) name=${rest%%:*}; cmd=${rest#*:}; cmd=${cmd//,/ }
           ( cd /tmp && eval timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_$name -o p -- $cmd > $O/${T}_rocprof_$name.log 2>&1 )
           f=$(find $O/${T}_prof_$name -name "*kernel_stats.csv" | head -1)
           [ -n "$f" ] && head -150 "$f" > $O/${T}_${name}_kernel_stats_top.csv
           rm -rf $O/${T}_prof_$name
           echo "== $name"; head -14 $O/${T}_${name}_kernel_stats_top.csv | cut -c1-150 ;;
    pmc)   name=${rest%%:*}; r2=${rest#*:}; ctrs=${r2%%:*}; cmd=${r2#*:}; cmd=${cmd//,/ }
           ( cd /tmp && eval timeout 1200 rocprofv3 --pmc ${ctrs//+/ } --kernel-trace --output-format csv -d $O/${T}_pmc_$name -o p -- $cmd > $O/${T}_rocprof_pmc_$name.log 2>&1 )
           f=$(find $O/${T}_pmc_$name -name "*counter_collection.csv" | head -1)
           [ -n "$f" ] && python $R/scripts/pmc_summary.py "$f" > $O/${T}_${name}_pmc.txt 2>&1
           rm -rf $O/${T}_pmc_$name
           echo "== pmc $name"; head -30 $O/${T}_${name}_pmc.txt | cut -c1-160 ;;
    run)   name=${rest%%:*}; cmd=${rest#*:}
           timeout 1800 bash -c "$cmd" > $O/${T}_$name.log 2>&1; echo "$name exit $?"; tail -n 25 $O/${T}_$name.log | cut -c1-300 ;;
    *) echo "unknown step $step" ;;
  esac
done
