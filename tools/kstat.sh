#!/bin/bash
# kernel-trace stats of a command on the GPU box: bash tools/kstat.sh <tag> <program and args...>   (through gpurun)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/kstat_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o k -- "$@" > "$OUT/run.log" 2>&1 || { tail -5 "$OUT/run.log"; exit 1; }
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:9.2f} min {float(r["MinNs"])/1e3:8.2f} max {float(r["MaxNs"])/1e3:8.2f} pct {r["Percentage"]}')
PY
find "$OUT" -name "*.csv" -size +1M -delete
