"""The one-filter LGSSM step by population size around the LDS-prefix / precomputed-prefix switch (1024 tiles):
   python tools/time_route.py   (through gpurun)"""
import sys
sys.path.insert(0, "genjax-chi_amd")
import torch
from genjax._amd import workloads as W
from genjax._amd.runtime import load_hip_ops

ops = load_hip_ops()
T = 60
for n in (1000448, 1048576, 1049600, 2000896, 4001792, 8003584):
    w = W.LgssmSMC(ops, 1, 5, n, T)
    for _ in range(3):
        w.run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); w.run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    med = ts[len(ts) // 2]
    print(f"n={n} tiles={n // 1024}: {med * 1e3 / T:.2f} us/step = {med * 1e3 / T / (n / 1e6):.2f} us per 1e6 particles", flush=True)
