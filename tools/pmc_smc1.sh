#!/bin/bash
# SQ counters of the one-filter LGSSM step kernels: bash tools/pmc_smc1.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/pmc_smc1_${1:-x}; rm -rf $OUT; mkdir -p $OUT
export GJX_BENCH_FILTERS=${2:-1}
B="python3 bench.py --workload smc_lgssm --no-cpu-baseline --steps 2 --warmup 1"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/sq1 -- $B > $OUT/sq1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $OUT/sq2 -- $B > $OUT/sq2.log 2>&1 || exit 1
for d in $OUT/sq1 $OUT/sq2; do python3 tools/pmc_by_kernel.py $d; done | grep -A1 -E "k_resample|k_tile_sums" 
find $OUT -name "*.csv" -size +1M -delete
