"""One-filter LGSSM / HMM runs for rocprofv3 --kernel-trace --stats: python tools/prof_smc_one.py [reps]"""
import sys
sys.path.insert(0, "genjax-chi_amd")
import torch
from genjax._amd import workloads as W
from genjax._amd.runtime import load_hip_ops

ops = load_hip_ops()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for w in (W.LgssmSMC(ops, 1, 5, 1_000_000, 100), W.HmmSMC(ops, 1, 5, 1_000_000, 100, n_states=256)):
    for _ in range(reps):
        w.run()
    torch.cuda.synchronize()
