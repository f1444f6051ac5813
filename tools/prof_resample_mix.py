"""Worst cases of the resampling kernel through the generic entry point (one MI355X):  python tools/prof_resample_mix.py
gjx_resample_systematic on 1e6 weights: uniform, one particle with ALL the mass, one particle with 60 % of the mass
and the rest spread evenly (no idle tile to delegate to), 32 particles with 3 % each."""
import math
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genjax-chi_amd"))
import torch  # noqa: E402

from genjax._amd.ops import HipEvent, KeyBatch  # noqa: E402
from genjax._amd.runtime import load_hip_ops  # noqa: E402

ops = load_hip_ops()
n = 1_000_000
dev = ops.device()
cases = {}
cases["uniform"] = torch.zeros(n, device=dev)
lw = torch.full((n,), -200.0, device=dev); lw[123_456] = 0.0
cases["one particle owns all"] = lw
lw = torch.zeros(n, device=dev); lw[123_456] = math.log(1.5 * n)
cases["one particle 60%, rest even"] = lw
lw = torch.zeros(n, device=dev); lw[torch.arange(32, device=dev) * 31_013 + 7] = math.log(0.03 / 0.04 * n / 1.0)
cases["32 particles ~1.4% each... rest even"] = lw
kb = KeyBatch(1, 2, parent=(5, 6))
for name, w in cases.items():
    ts = []
    for r in range(12):
        e0, e1 = HipEvent(), HipEvent()
        e0.record(ops.stream())
        a, m, q = ops.resample("systematic", kb, w)
        e1.record(ops.stream())
        torch.cuda.synchronize()
        ts.append(e0.elapsed_ms(e1) * 1e3)
    cnt = torch.bincount(a.long(), minlength=n)
    print(f"{name:40s} median {statistics.median(ts[2:]):8.1f} us   max offspring {int(cnt.max())}   distinct ancestors {int((cnt > 0).sum())}")
