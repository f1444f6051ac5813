for st in 1 2 3 4 5 0; do
  echo "stop $st: $(GJX_SMC_DEBUG_FIXED=1 GJX_SMC_DEBUG_STOP=$st python tools/time_lgssm1.py 2>&1 | grep lgssm)"
done
