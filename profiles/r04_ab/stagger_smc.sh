# experiment: delay a subset of the step's workgroups at kernel start (profiling build): bash tools/stagger_smc.sh
export GJX_HIP_LIB="$GRAFT_REPO_ROOT/genjax-chi_amd/lib/libgjx_hip_prof.so"
for sel in 0 1 2 3; do for d in 0 4 8 16 32; do
  echo "sel $sel delay $((d*80)) ns: $(GJX_SMC_DEBUG_STOP=$(( (d<<8) | (sel<<4) )) python tools/time_lgssm1.py 2>&1 | grep -E 'lgssm|hmm' | tr '\n' ' ')"
done; done
