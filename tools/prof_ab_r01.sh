#!/bin/bash
# needs the round-1 tree next to this one:  git worktree add -f build/r01_tree 6312f20 && (cd build/r01_tree && python __graft_entry__.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export GJX_BENCH_FILTERS=16
for tree in build/r01_tree .; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_ab_$(basename $(realpath $tree)); rm -rf $OUT; mkdir -p $OUT
  (cd $tree && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload smc_lgssm --no-cpu-baseline --steps 2 --warmup 1 > $OUT/trace.log 2>&1)
  echo "== $tree"
  find $OUT -name "*kernel_stats.csv" -exec cat {} \; | python3 -c "
import csv,sys
for r in csv.DictReader(sys.stdin):
    if float(r['Percentage'])>1: print(f\"{r['Name'][:80]:80s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us\")"
  find $OUT -name "*.csv" -size +1M -delete
done
