"""Time the plan-driven (generated-policy) bootstrap filter on the LGSSM written as a user model, next to the
hand-written LGSSM filter: python tools/time_user_smc.py [philox|threefry]  (GPU box)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genjax-chi_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from genjax._amd import prng, workloads as W  # noqa: E402
from genjax._amd.runtime import load_hip_ops  # noqa: E402
from test_gpu_parity_abi import _smc_plans  # noqa: E402

impl = 0 if len(sys.argv) > 1 and sys.argv[1] == "threefry" else 1
ops = load_hip_ops()
T, n = 100, 1_000_000
y = W.lgssm_data(T)
sk, rk = W.smc_key_schedule(prng.key(1, impl), T)
plan, _ = _smc_plans(ops)
for name, fn in (("generated", lambda: ops.smc_run_plan(plan, impl, n, sk, rk, y)),
                 ("hand-written", lambda: ops.smc_run_lgssm(impl, n, sk, rk, W.lgssm_model(), y))):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name:13s} {dt * 1e3:7.3f} ms per T={T} run of {n} particles = {n * T / dt:.3e} particle-steps/s")
