#!/bin/bash
# kernel-trace of the bootstrap-SMC step kernels: bash tools/prof_smc1.sh [filters] (default 1 = the literal BASELINE config)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
F=${1:-1}
OUT=gpurun_out/prof_smc_f$F; rm -rf $OUT; mkdir -p $OUT
export GJX_BENCH_FILTERS=$F
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload smc_lgssm --no-cpu-baseline --steps 2 --warmup 1 > $OUT/trace.log 2>&1 || exit 1
find $OUT -name "*kernel_stats.csv" -exec cat {} \; > $OUT/kernel_stats.txt
find $OUT -name "*.csv" -size +1M -delete
python3 - $OUT/kernel_stats.txt <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}")
PY
