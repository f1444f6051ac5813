"""Where the HMM-as-a-scan kernel's time goes: the full step (latent categorical + observed categorical) against the latent
site alone: python tools/hmm_scan_split.py   (through gpurun)"""
import sys
sys.path.insert(0, "genjax-chi_amd")
import torch
from genjax._amd import abi, workloads as W
from genjax._amd.runtime import load_hip_ops

ops = load_hip_ops()
n, T = 1_000_000, 500
full = W.HmmScan(ops, 1, 4, n, T)
lat = W.HmmScan(ops, 1, 4, n, T)
z = abi.Site()
z.dist, z.observed, z.out_col = abi.DIST_CATEGORICAL, 0, 0
z.n_cat, z.n_rows, z.cat_mode = lat.k, lat.k, 1
z.arg[0] = abi.Arg(abi.ARG_STATE, 0, 1.0, 0.0, None)
z.logits = lat.trans.data_ptr()
lat.plan = ops.scan_plan_create([z], [abi.Arg(abi.ARG_SITE, 0, 1.0, 0.0, None)], 1)
for name, w in (("latent + observed", full), ("latent only", lat)):
    w.run(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); w.run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f"{name}: {ts[2]:.3f} ms per pass = {n * T / ts[2] / 1e6:.2f}e9 particle-steps/s", flush=True)
