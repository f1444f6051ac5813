#!/bin/bash
# ImportanceK at ONE pass per launch (the literal single call) by kernel form and waves-per-SIMD hint:
#   gpurun -- 'bash tools/ab_onepass.sh'      (prints particles/s, ms per call, kernel ms, fraction of the 48-B line)
for cfg in "quad 0" "pair 6" "pair 8" "quad 8" "pair 7"; do
  set -- $cfg
  out=$(GJX_JIT_FORM=$1 GJX_JIT_MIN_WAVES=$2 GJX_BENCH_LAUNCH=1 GJX_PLAN_JIT_VERBOSE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'), d['roofline']['frac'])")
  echo "form $1 min_waves $2: $out"
done
