"""Per-step time of the one-filter bootstrap filters (HIP events around whole runs): python tools/time_smc_step.py [n] [T]"""
import sys, time
sys.path.insert(0, "genjax-chi_amd")
import torch
from genjax._amd import workloads as W
from genjax._amd.runtime import load_hip_ops

ops = load_hip_ops()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
for name, mk in (("lgssm", lambda F: W.LgssmSMC(ops, 1, 5, n, T, filters=F)),
                 ("hmm", lambda F: W.HmmSMC(ops, 1, 5, n, 5 * T, n_states=256, filters=F))):
    for F in (1, 16):
        w = mk(F)
        steps = T if name == "lgssm" else 5 * T
        for _ in range(3):
            w.run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); w.run(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        med = ts[len(ts) // 2]
        print(f"{name} F={F}: {med * 1e3 / steps:.2f} us/step  ({n * F * steps / med / 1e6:.1f}e9 particle-steps/s)  min {ts[0] * 1e3 / steps:.2f}", flush=True)
