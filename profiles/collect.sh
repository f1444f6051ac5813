#!/bin/bash
# Collect the round's rocprof evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash profiles/collect.sh r01'
# Writes raw output under gpurun_out/profiles_<tag>/; profiles/summarize.py turns it into the
# committed summaries.  Counters are collected in their own passes (no trace domains with --pmc).
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/profiles_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf "$OUT" && mkdir -p "$OUT"
# Timing runs use bench.py's defaults (2048-pass clock ramp + 64 warm-up + 1024 timed passes, 8 passes per launch):
# the timed region is the LAST 128 importance launches of the trace.  Counter runs need no steady clocks: 48 timed + 8
# warm-up passes without the ramp, so every launch covers exactly 8 passes and a run makes 56 passes.
BENCH="python3 bench.py"
PMCBENCH="python3 bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-extra"
export GJX_BENCH_RAMP=2048
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/bench_under_rocprof.log" 2>&1 || exit 1
export GJX_BENCH_RAMP=0
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $PMCBENCH > "$OUT/fetch.log" 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $PMCBENCH > "$OUT/write.log" 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/sq" -- $PMCBENCH > "$OUT/sq.log" 2>&1 || exit 1
export GJX_BENCH_RAMP=2048
# HBM traffic of the SMC step kernels (8 filters of 1e6 particles per launch), one counter per pass
for w in smc_lgssm smc_hmm; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d "$OUT/smc_${w}_$c" -- python3 bench.py --workload $w --no-cpu-baseline --steps 1 --warmup 1 > "$OUT/smc_${w}_$c.log" 2>&1 || exit 1
  done
done
timeout -k 10 300 $BENCH > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.err" || exit 1
tail -c 600 "$OUT/bench_plain.json"
