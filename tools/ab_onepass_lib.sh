#!/bin/bash
# same-box A/B of two builds of the library at ONE importance pass per launch (and the headline): bash tools/ab_onepass_lib.sh <other.so>
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2 3; do
  for which in new old; do
    if [ $which = old ]; then export GJX_HIP_LIB="$1"; else unset GJX_HIP_LIB; fi
    a=$(GJX_BENCH_LAUNCH=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1 pass: %.4g p/s, kernel %.2f us' % (d['value'], d['roofline']['kernel_ms']*1e3))")
    b=$(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('20 passes: %.4g p/s, frac %.3f' % (d['value'], d['roofline']['frac']))")
    echo "$which: $a | $b"
  done
done
