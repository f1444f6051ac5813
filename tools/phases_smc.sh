# cumulative per-step time of the one-filter LGSSM step by phase (early exits on a fixed input population); needs the
# profiling build of the library (python -c "import __graft_entry__ as g; g.build_profile()" before gpurun):
#   gpurun -- 'bash tools/phases_smc.sh'
# 15: launch + dispatch only; 14: + record loads and the policy's prefetch; 1: + merge; 2: + locate; 3: + window scan;
# 4: + max-scan; 5: + gather and compute; 0: the whole step
export GJX_HIP_LIB="$GRAFT_REPO_ROOT/genjax-chi_amd/lib/libgjx_hip_prof.so"
for st in 15 14 1 2 3 4 5 0; do
  echo "stop $st: $(GJX_SMC_DEBUG_FIXED=1 GJX_SMC_DEBUG_STOP=$st python tools/time_lgssm1.py 2>&1 | grep lgssm)"
done
