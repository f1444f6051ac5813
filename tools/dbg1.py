import sys
sys.path.insert(0,'genjax-chi_amd'); sys.path.insert(0,'tests')
import numpy as np, torch
from genjax._amd.abi import GjxLib
from genjax._amd import abi, prng, workloads as W
from genjax._amd.ops import Ops, KeyBatch
from genjax._amd.runtime import load_hip_ops
hip=load_hip_ops()
ora=Ops(GjxLib('oracle/libgjx_oracle.so','cpu'))
n=70000
y=np.array([0.1,float('nan'),0.3,0.2,-0.4],dtype=np.float32)
for T in (2,3):
    sk,rk=W.smc_key_schedule(prng.key(11,0),5)
    sk,rk=sk[:T],rk[:T]
    h=hip.smc_run_lgssm(0,n,sk,rk,abi.Lgssm(0.0,1.0,0.9,1.0,0.5),y[:T],True)
    o=ora.smc_run_lgssm(0,n,sk,rk,abi.Lgssm(0.0,1.0,0.9,1.0,0.5),y[:T],True)
    print("T",T,"e",h[0].cpu().tolist(),o[0].tolist(),"q",h[1].cpu().tolist(),o[1].tolist())
    for t in range(T):
        d=(h[4][t].cpu()!=o[4][t]).nonzero().flatten()
        print(" anc t",t,"ndiff",d.numel(),d[:5].tolist(), h[4][t].cpu()[d[:5]].tolist(), o[4][t][d[:5]].tolist())
    sd=(h[2].cpu()!=o[2]).nonzero().flatten()
    print(" state ndiff",sd.numel(),sd[:5].tolist(), h[2].cpu()[sd[:5]].tolist(), o[2][sd[:5]].tolist())
    ld=~((h[3].cpu()==o[3])|(h[3].cpu().isnan()&o[3].isnan()))
    print(" logw ndiff",int(ld.sum()))
lw=torch.full((5000,),float('-inf'))
a,e,q=hip.resample("systematic",KeyBatch(0,2,parent=(5,1)),lw.cuda(),5000)
print("all -inf identity:", bool(torch.equal(a.cpu(),torch.arange(5000,dtype=torch.int32))), int(e), int(q))
