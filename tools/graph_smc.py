"""Does a hipGraph of the whole one-filter SMC run beat the stream of launches?  python tools/graph_smc.py
(capture LgssmSMC.run() = 200 kernel launches into a graph through torch.cuda.CUDAGraph, replay, compare per-step time)"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genjax-chi_amd"))
import torch  # noqa: E402

from genjax._amd import workloads as W  # noqa: E402
from genjax._amd.runtime import load_hip_ops  # noqa: E402

ops = load_hip_ops()
for kind in ("lgssm", "hmm"):
    T = 100 if kind == "lgssm" else 500
    wl = W.LgssmSMC(ops, 1, 1, 1_000_000, T) if kind == "lgssm" else W.HmmSMC(ops, 1, 2, 1_000_000, T)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            out = wl.run()
    torch.cuda.synchronize()
    ref = wl.result(out)

    def timeit(fn, reps=30):
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts)

    with torch.cuda.stream(side):
        t_stream = timeit(lambda: wl.run())
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=side):
            out_g = wl.run()
        t_graph = timeit(lambda: g.replay())
        res = wl.result(out_g)
        same = res["log_z"] == ref["log_z"]
        print(f"{kind}: stream {t_stream / T * 1e6:.2f} us/step   graph {t_graph / T * 1e6:.2f} us/step   same log Z: {same}")
    except Exception as e:
        print(f"{kind}: stream {t_stream / T * 1e6:.2f} us/step   graph capture failed: {type(e).__name__}: {e}")
