import sys
sys.path.insert(0,'genjax-chi_amd')
import torch
from genjax._amd.runtime import load_hip_ops
from genjax._amd.ops import KeyBatch
ops=load_hip_ops()
n=1_000_000
lw=torch.randn(n,device='cuda')
for i in range(20):
    a,m,q=ops.resample('systematic',KeyBatch(1,2,parent=(1,i)),lw)
torch.cuda.synchronize()
