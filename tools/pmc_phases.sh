#!/bin/bash
# VALU / SALU / LDS / VMEM instructions per wave of the one-filter LGSSM step, cumulative by phase (the early exits of the
# profiling build, as tools/phases_smc.sh): python -c "import __graft_entry__ as g; g.build_profile()" first, then
#   gpurun -- 'bash tools/pmc_phases.sh [lgssm|hmm]'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export GJX_HIP_LIB="$GRAFT_REPO_ROOT/genjax-chi_amd/lib/libgjx_hip_prof.so"
export GJX_SMC_DEBUG_FIXED=1 GJX_PHASE_MODEL=${1:-lgssm}
OUT=gpurun_out/pmc_phases; rm -rf $OUT; mkdir -p $OUT
for st in 15 14 1 2 3 4 5 0; do
  export GJX_SMC_DEBUG_STOP=$st
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/s$st -- python3 tools/pmc_phases_run.py > $OUT/s$st.log 2>&1 || { tail -5 $OUT/s$st.log; exit 1; }
  echo "stop $st: $(python3 tools/pmc_by_kernel.py $OUT/s$st | grep -A1 k_resample | tail -1)"
  find $OUT/s$st -name "*.csv" -size +1M -delete
done
