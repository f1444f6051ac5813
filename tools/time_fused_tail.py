"""Kernel time of the importance launch with and without the fused log-sum-exp tail (run under tools/kstat.sh)."""
import sys
sys.path.insert(0, ".")
import torch
import bench
from genjax._amd import workloads as W
from genjax._amd.runtime import get_ops

ops = get_ops()
for impl, n in (("threefry", 1_000_000), ("philox", 1_000_000), ("philox", 100_000), ("philox", 10_000), ("philox", 4_000_000)):
    wl = W.Gaussian10(ops, impl, seed=4, n_local=n)
    prep = wl.prepare()
    for name, fn in (("plain", prep.launch_importance), ("rows", lambda: (prep.launch_importance(), prep.launch_lse_rows())),
                     ("fused", prep.launch_fused)):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s = torch.cuda.current_stream()
        ev0.record(s)
        for _ in range(200):
            fn()
        ev1.record(s)
        torch.cuda.synchronize()
        print(impl, n, name, f"{ev0.elapsed_time(ev1) * 1e3 / 200:.2f} us per call (stream time)")
