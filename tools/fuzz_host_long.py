import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "genjax-chi_amd")
import fuzz_models
for impl in (0, 1):
    print("models", impl, fuzz_models.run(120, 900 + impl, impl), flush=True)
    print("scans", impl, fuzz_models.run_scans(90, 950 + impl, impl), flush=True)
