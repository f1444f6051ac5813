#!/bin/bash
# Round-4 rocprof evidence on the GPU box (through gpurun from the repo root):
#   gpurun --timeout 1100 -- 'bash profiles/collect_r04.sh r04'
# Raw output under gpurun_out/profiles_<tag>/; profiles/summarize_r04.py turns it into the committed summaries.
# Counters are collected in their own passes (--pmc never together with a trace domain), the program directly after `--`.
set -o pipefail
TAG=${1:-r04}
OUT=gpurun_out/profiles_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf "$OUT" && mkdir -p "$OUT"
# 1. the driver's command under the kernel trace (every kernel of the bench line, extras included)
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err" || exit 1
echo "trace done"
# 1b. r04 (VERDICT r03 hygiene): a kernel trace PER WORKLOAD, so that every `configs` entry has its own rows (the one-filter LGSSM
# row of the whole-run trace mixes the normal and the collapsing-weights runs)
for w in smc_lgssm smc_hmm scan_hmm; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$w" -- python3 bench.py --workload $w --no-cpu-baseline --steps 2 --warmup 1 > "$OUT/trace_$w.json" 2> "$OUT/trace_$w.err" || exit 1
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_sharded_rank0" -- python3 tools/time_sharded_rank0.py > "$OUT/trace_sharded_rank0.json" 2> "$OUT/trace_sharded_rank0.err" || exit 1
echo "per-workload traces done"
# 2. counters per workload: HBM traffic (FETCH_SIZE, WRITE_SIZE: one per pass) and two SQ sets
export GJX_BENCH_RAMP=0
declare -A CMD
CMD[importance]="python3 bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-extra"
CMD[importance_fast]="python3 bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-extra --fast-math"
CMD[smc_lgssm]="python3 bench.py --workload smc_lgssm --no-cpu-baseline --steps 2 --warmup 1"
CMD[smc_hmm]="python3 bench.py --workload smc_hmm --no-cpu-baseline --steps 2 --warmup 1"
CMD[scan_lgssm]="python3 bench.py --workload scan_lgssm --no-cpu-baseline"
CMD[scan_hmm]="python3 bench.py --workload scan_hmm --no-cpu-baseline"
CMD[importance_threefry]="python3 bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-extra --rng threefry"
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM"
SQ2="SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32"
# r04: where the SMC step's waves are parked: s_waitcnt / barrier parking, LDS traffic, lane utilisation
SQ3="SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES"
# (counter passes: 8 importance passes per launch, so that every launch of these runs covers the same work)
export GJX_BENCH_FILTERS=1 GJX_BENCH_MIN_S=0.01 GJX_BENCH_LAUNCH=8
for w in smc_lgssm smc_hmm importance scan_hmm scan_lgssm importance_fast; do
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${w}_fetch" -- ${CMD[$w]} > "$OUT/${w}_fetch.log" 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${w}_write" -- ${CMD[$w]} > "$OUT/${w}_write.log" 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --pmc $SQ1 --output-format csv -d "$OUT/${w}_sq1" -- ${CMD[$w]} > "$OUT/${w}_sq1.log" 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --pmc $SQ2 --output-format csv -d "$OUT/${w}_sq2" -- ${CMD[$w]} > "$OUT/${w}_sq2.log" 2>&1 || exit 1
  case $w in smc_*) timeout -k 10 200 rocprofv3 --pmc $SQ3 --output-format csv -d "$OUT/${w}_sq3" -- ${CMD[$w]} > "$OUT/${w}_sq3.log" 2>&1 || echo "$w: no SQ3 counters";; esac
  echo "$w counters done"
done
# L2 behaviour of the scans (the HMM scan reads scattered table lines): hits, misses, requests per launch
L2="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
for w in scan_lgssm scan_hmm; do
  timeout -k 10 200 rocprofv3 --pmc $L2 --output-format csv -d "$OUT/${w}_l2" -- ${CMD[$w]} > "$OUT/${w}_l2.log" 2>&1 || echo "$w: no L2 counters (see ${w}_l2.log)"
done
# r04: the rejection samplers (lane utilisation) and the phase times of the FINAL step kernel (profiling build)
bash tools/pmc_samplers.sh > "$OUT/pmc_samplers.txt" 2>&1 || echo "pmc_samplers failed"
[ -f genjax-chi_amd/lib/libgjx_hip_prof.so ] && bash tools/phases_smc.sh > "$OUT/phases_smc.txt" 2>&1
# ... and the instructions per wave behind those phase times (VALU / SALU / LDS / VMEM, cumulative by phase)
[ -f genjax-chi_amd/lib/libgjx_hip_prof.so ] && bash tools/pmc_phases.sh lgssm > "$OUT/pmc_phases.txt" 2>&1
unset GJX_BENCH_RAMP GJX_BENCH_MIN_S GJX_BENCH_LAUNCH
# 3. the plain bench line of the same build
GJX_BENCH_DETAIL="$OUT/bench_detail.json" timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.err" || exit 1
python3 profiles/summarize_r04.py "$TAG" "$OUT" > "$OUT/summarize.log" 2>&1 || { tail -20 "$OUT/summarize.log"; exit 1; }
# keep the summaries; drop the raw per-dispatch CSVs (large)
mkdir -p "$OUT/summary" && cp profiles/${TAG}_* "$OUT/summary/" 2>/dev/null
[ -f "$OUT/pmc_phases.txt" ] && cp "$OUT/pmc_phases.txt" "$OUT/summary/${TAG}_phase_counters.txt"
[ -f "$OUT/phases_smc.txt" ] && cp "$OUT/phases_smc.txt" "$OUT/summary/${TAG}_phase_times.txt"
find "$OUT" -name "*.csv" -size +1M -delete
tail -5 "$OUT/summarize.log"
