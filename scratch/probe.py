import sys,time,os
sys.path.insert(0,'genjax-chi_amd')
import torch
from genjax._amd.runtime import load_hip_ops
from genjax._amd import workloads as W
ops=load_hip_ops()
n=1_000_000
wl=W.Gaussian10(ops,1,0,n)
def step(ev):
    if ev:
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    vals,score,logw,mp=ops.importance_run(wl.plan,wl.keys,n,[],[torch.float32]*10)
    if ev: e1.record()
    lse,m,q=ops.logsumexp(logw,max_partials=mp)
    return lse
for i in range(5): step(False)
torch.cuda.synchronize()
ts=[]
for i in range(12):
    t=time.perf_counter(); step(i>=4); ts.append(time.perf_counter()-t)
torch.cuda.synchronize()
print(['%.3f'%(x*1e3) for x in ts])
t=time.perf_counter()
for i in range(50): step(True)
torch.cuda.synchronize()
print('50 steps ms/step', (time.perf_counter()-t)/50*1e3)
t=time.perf_counter()
for i in range(50): step(False)
torch.cuda.synchronize()
print('50 steps no events ms/step', (time.perf_counter()-t)/50*1e3)
