import sys, time
sys.path.insert(0,'genjax-chi_amd')
import torch
from genjax._amd import workloads as W
from genjax._amd.runtime import load_hip_ops
ops=load_hip_ops()
for n,T,mode in ((100_000,100,1),(1_000_000,50,1),(100_000,20,0))[:int(sys.argv[1]) if len(sys.argv) > 1 else 3]:
    w=W.HmmScan(ops,1,4,n,T,cat_mode=mode)
    w.run(); torch.cuda.synchronize()
    t0=time.perf_counter(); w.run(); torch.cuda.synchronize(); dt=time.perf_counter()-t0
    r=w.result()
    print(f"n {n} T {T} mode {mode}: {dt*1e3:.2f} ms -> {n*T/dt:.3e} particle-steps/s  logz {r['log_z']:.3f}")
