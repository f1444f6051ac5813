cd "$GRAFT_REPO_ROOT"
L=$GRAFT_REPO_ROOT/genjax-chi_amd/lib
for i in 1 2; do
  for v in full np na; do
    if [ $v = full ]; then unset GJX_HIP_LIB; else export GJX_HIP_LIB=$L/libgjx_hip_$v.so; fi
    a=$(python tools/time_lgssm1.py 2>&1 | tail -2 | tr '\n' ' ')
    b=$(python tools/time_smc_step.py 65536 100 2>&1 | grep "F=1" | tr '\n' ' ')
    echo "$v: $a | $b"
  done
done
