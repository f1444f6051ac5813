import os, sys, time, statistics
sys.path.insert(0, "genjax-chi_amd")
import torch, genjax
from genjax import ChoiceMapBuilder as C, Target, gen, normal
from genjax.inference.smc import ImportanceK
from genjax._amd import workloads as W, plan as P
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
for rep in range(5): float(alg.log_marginal_likelihood_estimate(genjax.random.key(rep)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for rep in range(200):
    float(alg.log_marginal_likelihood_estimate(genjax.random.key(200 + rep)))
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
