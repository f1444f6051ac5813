import sys,time
sys.path.insert(0,'genjax-chi_amd')
import torch
from genjax._amd.runtime import load_hip_ops
from genjax._amd import workloads as W
ops=load_hip_ops()
for n in (1_000_000, 8_000_000, 32_000_000):
    w=W.LgssmSMC(ops,1,1,n,20)
    out=w.run(); torch.cuda.synchronize()
    t=time.perf_counter(); out=w.run(); torch.cuda.synchronize(); dt=time.perf_counter()-t
    r=w.result(out)
    print(n, 'us/step', dt/20*1e6, 'Gps/s', n*20/dt/1e9, r['log_z'], r['log_z_exact'])
    g=W.gaussian10_importance(ops,1,0,n)
    print('  imp', g['log_z'], g['log_z_rows'], g['log_z_exact'])
    del w,out,r,g
    torch.cuda.empty_cache()
