import sys,time
sys.path.insert(0,'genjax-chi_amd')
import torch, genjax
from genjax import gen, normal, ChoiceMapBuilder as C
from genjax.inference.smc import BootstrapSMC, LinearGaussianSSM, StateSpaceModel
from genjax._amd import workloads as W
@gen
def init():
    x = normal(0.0, 1.0) @ "x"
    normal(x, 0.5) @ "y"
    return x
@gen
def step(x):
    x2 = normal(0.9 * x, 1.0) @ "x"
    normal(x2, 0.5) @ "y"
    return x2
y=W.lgssm_data(100); n=1_000_000
key=genjax.random.key(1,"philox")
for name,smc in (("generated",BootstrapSMC(StateSpaceModel(init,step), C["y"].set(torch.tensor(y)), n)),("hand-written",BootstrapSMC(LinearGaussianSSM(), y, n))):
    r=smc.run(key); torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(5): r=smc.run(key)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/5
    print(name, "ms/run", dt*1e3, "us/step", dt*1e4, "logZ", r.log_marginal_likelihood, W.lgssm_exact_log_z(y))
