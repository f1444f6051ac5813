#!/bin/bash
# SQ / L2 counters of every k_resample launch of the collapsing-weights run: bash tools/pmc_collapse.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/pmc_collapse; rm -rf $OUT; mkdir -p $OUT
B="python3 tools/prof_collapse.py"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM --output-format csv -d $OUT/sq1 -- $B > $OUT/sq1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/sq2 -- $B > $OUT/sq2.log 2>&1 || exit 1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
for d in ("sq1", "sq2"):
    rows = collections.defaultdict(dict)
    for f in glob.glob(f"{sys.argv[1]}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_resample" in r["Kernel_Name"]:
                rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(rows)[-16:]
    for i in ids:
        print(d, i, "  ".join(f"{k}={v:.4g}" for k, v in sorted(rows[i].items())))
PY
find $OUT -name "*.csv" -size +1M -delete
