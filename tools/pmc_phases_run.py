"""The launches tools/pmc_phases.sh counts: a few runs of the one-filter filter of 1e6 particles (GJX_PHASE_MODEL=lgssm|hmm)."""
import os, sys
sys.path.insert(0, "genjax-chi_amd")
import torch
from genjax._amd import workloads as W
from genjax._amd.runtime import load_hip_ops
ops = load_hip_ops()
n, T = 1_000_000, 50
w = W.HmmSMC(ops, 1, 5, n, T, n_states=256) if os.environ.get("GJX_PHASE_MODEL") == "hmm" else W.LgssmSMC(ops, 1, 5, n, T)
for _ in range(3):
    w.run()
torch.cuda.synchronize()
