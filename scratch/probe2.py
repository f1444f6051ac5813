import sys,time,os
sys.path.insert(0,'genjax-chi_amd')
import torch
from genjax._amd.runtime import load_hip_ops
from genjax._amd import workloads as W
ops=load_hip_ops()
for impl in (1,0):
  for n in (250_000, 1_000_000, 4_000_000, 16_000_000):
    wl=W.Gaussian10(ops,impl,0,n)
    for i in range(3): ops.importance_run(wl.plan,wl.keys,n,[],[torch.float32]*10)
    torch.cuda.synchronize()
    evs=[]
    for i in range(20):
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record(); ops.importance_run(wl.plan,wl.keys,n,[],[torch.float32]*10); e1.record(); evs.append((e0,e1))
    torch.cuda.synchronize()
    ms=sorted(a.elapsed_time(b) for a,b in evs)[len(evs)//2]
    print(f"impl {impl} n {n:>9d} kernel {ms*1e3:8.1f} us  {n/ms/1e6:8.2f} Gparticles/s  {48*n/ms/1e6:8.1f} GB/s")
    del wl
