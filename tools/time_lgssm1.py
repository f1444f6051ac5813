import sys
sys.path.insert(0, "genjax-chi_amd")
import torch
from genjax._amd import workloads as W
from genjax._amd.runtime import load_hip_ops
ops = load_hip_ops()
n, T = 1_000_000, 100
for name, mk in (("lgssm", lambda: W.LgssmSMC(ops, 1, 5, n, T)), ("hmm", lambda: W.HmmSMC(ops, 1, 5, n, T, n_states=256))):
    w = mk()
    for _ in range(5):
        w.run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); w.run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f"{name}: median {ts[10] * 1e3 / T:.2f} us/step, min {ts[0] * 1e3 / T:.2f}", flush=True)
