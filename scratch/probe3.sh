for mw in 0 8 6 4; do echo "== GJX_JIT_MIN_WAVES=$mw"; GJX_JIT_MIN_WAVES=$mw timeout -k 10 100 python scratch/probe2.py 2>&1 | grep -v amdgpu | grep -E "n   1000000|n  16000000"; done
