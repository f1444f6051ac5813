"""Bayesian linear regression `y_i ~ normal(w * x_i + b, s)` with 50 data points, 1e6 particles, through the host API: the
fused kernel (site arguments over two traced values = postfix programs, gjx.h GJX_ARG_EXPR) against the per-site column
path the same model took before (one log-density launch and two torch kernels per observation):
python tools/time_regression.py"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genjax-chi_amd"))
import torch  # noqa: E402

import genjax  # noqa: E402
from genjax import ChoiceMapBuilder as C, Target, gen, normal  # noqa: E402
from genjax._amd import plan as P  # noqa: E402
from genjax.inference.smc import ImportanceK  # noqa: E402

torch.manual_seed(0)
xs = torch.linspace(-2, 2, 50).tolist()
ys = [1.5 * x - 0.5 + 0.3 * float(torch.randn(())) for x in xs]


@gen
def regression():
    w = normal(0.0, 2.0) @ "w"
    b = normal(0.0, 2.0) @ "b"
    for i, x in enumerate(xs):
        normal(w * x + b, 0.3) @ ("y", i)


chm = C.n()
for i, y in enumerate(ys):
    chm = chm | C["y", i].set(y)
alg = ImportanceK(Target(regression, (), chm), k_particles=1_000_000)


def timed(label, reps, estimate=None):
    estimate = estimate or alg.log_marginal_likelihood_estimate
    for r in range(3):
        float(estimate(genjax.random.key(r, "philox")))
    ts = []
    for r in range(reps):
        t0 = time.perf_counter()
        z = float(estimate(genjax.random.key(10 + r, "philox")))
        ts.append(time.perf_counter() - t0)
    z = float(estimate(genjax.random.key(777, "philox")))
    print(f"{label}: median {statistics.median(ts) * 1e3:.3f} ms per estimate (log Z at key 777: {z:.6f})")
    return z


from genjax._amd import inference as I  # noqa: E402

general = lambda key: I.SMCAlgorithm.log_marginal_likelihood_estimate(alg, key)  # (run_smc: the trace is materialised)
zf = timed("estimate-only fused kernel, one library call (52 sites)", 30)
zg = timed("general route (fused kernel with its trace columns)", 30, general)
orig = P.try_fused_generate
P.try_fused_generate = lambda *a, **k: None  # the per-site column path
try:
    ze = timed("per-site launches", 5, general)
finally:
    P.try_fused_generate = orig
print("same estimate:", zf == ze == zg)
