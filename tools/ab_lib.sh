#!/bin/bash
# same-box A/B of two builds of the library (through gpurun): bash tools/ab_lib.sh <other libgjx_hip.so>
# alternates the tree's library and the other one over the importance, scan and one-filter SMC workloads of bench.py
OTHER=$1
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2; do
  for w in importance scan_lgssm smc_lgssm smc_hmm; do
    for which in new old; do
      if [ $which = old ]; then export GJX_HIP_LIB="$OTHER"; else unset GJX_HIP_LIB; fi
      if [ $w = importance ]; then args="--no-extra"; else args="--workload $w"; fi
      python bench.py $args --no-cpu-baseline > gpurun_out/ab_${w}_${which}_$i.json 2> gpurun_out/ab.err || { tail -5 gpurun_out/ab.err; exit 1; }
      python - "$w" "$which" "gpurun_out/ab_${w}_${which}_$i.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], "value %.4g" % d["value"], "ms %.5f" % d.get("ms_per_step", d.get("ms_per_run", 0)), "frac %.3f" % d["roofline"]["frac"], "kernel_ms %.5f" % d["roofline"].get("kernel_ms", 0))
PY
    done
  done
done
