#!/bin/bash
# SQ / LDS counters and kernel trace of the ONE-filter bootstrap-SMC step (the literal BASELINE configs 3 and 5):
#   gpurun --timeout 900 -- 'bash profiles/collect_smc1.sh r02a'
# Counters in their own passes (no trace domains with --pmc); summaries: tools/pmc_by_kernel.py.
set -o pipefail
TAG=${1:-r02}
OUT=gpurun_out/smc1_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf "$OUT" && mkdir -p "$OUT"
export GJX_BENCH_FILTERS=1
for w in smc_lgssm smc_hmm; do
  B="python3 bench.py --workload $w --no-cpu-baseline --steps 2 --warmup 1"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${w}_trace" -- $B > "$OUT/${w}_trace.log" 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/${w}_sq1" -- $B > "$OUT/${w}_sq1.log" 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA --output-format csv -d "$OUT/${w}_sq2" -- $B > "$OUT/${w}_sq2.log" 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM --output-format csv -d "$OUT/${w}_sq3" -- $B > "$OUT/${w}_sq3.log" 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d "$OUT/${w}_grbm" -- $B > "$OUT/${w}_grbm.log" 2>&1 || exit 1
done
for d in "$OUT"/*_sq1 "$OUT"/*_sq2 "$OUT"/*_sq3 "$OUT"/*_grbm; do echo "== $d"; python3 tools/pmc_by_kernel.py "$d"; done > "$OUT/pmc_summary.txt" 2>&1
find "$OUT" -name "*kernel_stats.csv" -exec sh -c 'echo "== $1"; cat "$1"' _ {} \; > "$OUT/kernel_stats.txt"
# keep only summaries (raw per-dispatch CSVs are large)
find "$OUT" -name "*.csv" -size +2M -delete
tail -n 40 "$OUT/pmc_summary.txt"
