"""Host-API overhead of one ImportanceK estimate (tracing, lowering, launch, fold) against the kernel's ~18 us:
python tools/time_host_api.py"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genjax-chi_amd"))
import torch  # noqa: E402

import genjax  # noqa: E402
from genjax import ChoiceMapBuilder as C, Target, gen, normal  # noqa: E402
from genjax.inference.smc import ImportanceK  # noqa: E402
from genjax._amd import workloads as W  # noqa: E402

y = W.gaussian10_data()


@gen
def model():
    for i in range(10):
        z = normal(0.0, 1.0) @ f"z{i}"
        _ = normal(z, 0.5) @ f"y{i}"


chm = C.n()
for i in range(10):
    chm = chm | C[f"y{i}"].set(float(y[i]))
alg = ImportanceK(Target(model, (), chm), k_particles=1_000_000)
for rep in range(3):
    alg.log_marginal_likelihood_estimate(genjax.random.key(rep, "philox"))
torch.cuda.synchronize()
ts = []
for rep in range(40):
    t0 = time.perf_counter()
    z = alg.log_marginal_likelihood_estimate(genjax.random.key(100 + rep, "philox"))
    float(z)
    ts.append(time.perf_counter() - t0)
print("per call (us):", " ".join(f"{t * 1e6:.0f}" for t in ts))
print(f"ImportanceK.log_marginal_likelihood_estimate (1e6 particles, 20 sites): median {statistics.median(ts) * 1e6:.0f} us per call, min {min(ts) * 1e6:.0f} us")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for rep in range(20):
    float(alg.log_marginal_likelihood_estimate(genjax.random.key(200 + rep)))
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
