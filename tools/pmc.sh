#!/bin/bash
# counters of a command on the GPU box: bash tools/pmc.sh <tag> "<counters>" <program and args...>   (through gpurun)
TAG=$1; shift; CTRS=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --pmc $CTRS --output-format csv -d "$OUT" -o p -- "$@" > "$OUT/run.log" 2>&1 || { tail -5 "$OUT/run.log"; exit 1; }
f=$(find "$OUT" -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    n = max(len(v) for v in d.values())
    if n < 20: continue
    print(k, "launches", n)
    for c, v in sorted(d.items()):
        v = sorted(v); print(f"   {c:28s} median {v[len(v)//2]:14.0f}")
PY
find "$OUT" -name "*.csv" -size +1M -delete
