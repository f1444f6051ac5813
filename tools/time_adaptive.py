"""Per-step time of the ESS-adaptive one-filter bootstrap filters (threshold 0.5): python tools/time_adaptive.py [n] [T]"""
import sys
sys.path.insert(0, "genjax-chi_amd")
import torch
from genjax._amd import workloads as W
from genjax._amd.runtime import load_hip_ops

ops = load_hip_ops()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
for name, mk in (("lgssm", lambda: W.LgssmSMC(ops, 1, 5, n, T, ess_threshold=0.5)),
                 ("hmm", lambda: W.HmmSMC(ops, 1, 5, n, T, n_states=256, ess_threshold=0.5))):
    w = mk()
    for _ in range(3):
        w.run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); w.run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f"{name} adaptive 0.5: {ts[len(ts) // 2] * 1e3 / T:.2f} us/step  min {ts[0] * 1e3 / T:.2f}", flush=True)
