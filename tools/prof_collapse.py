"""Per-step device time of the collapsing-weights LGSSM filter (one MI355X):  python tools/prof_collapse.py [n]
Runs the bench's collapse sequence step by step through the step-level entry points with HIP events around every
k_resample launch, and prints each step's time next to what the step's weights looked like (heavy tiles, idle tiles)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genjax-chi_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from genjax._amd import abi, prng, workloads as W  # noqa: E402
from genjax._amd.ops import HipEvent  # noqa: E402
from genjax._amd.runtime import load_hip_ops  # noqa: E402

ops = load_hip_ops()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
yc = np.tile(np.array([0.1, 25.0, -40.0, -39.5, 60.0, 60.2, 0.0, 3.0], dtype=np.float32), 2)
T = len(yc)
model = abi.Lgssm(0.0, 1.0, 0.9, 1.0, 0.05)
sk, rk = W.smc_key_schedule(prng.key(1, 1), T)
cfg = ops.smc_config(1, n, 0, n, sk, rk)
nt = ops.num_tiles(n)
dev = ops.device()
state = [torch.zeros(n, device=dev), torch.zeros(n, device=dev)]
logw = [torch.zeros(n, device=dev), torch.zeros(n, device=dev)]
tile_sums = torch.zeros(nt, dtype=torch.int64, device=dev)
mp = torch.empty(nt, device=dev)
out_max = torch.empty(T, device=dev)
out_q = torch.zeros(T, dtype=torch.int64, device=dev)
for rep in range(2):
    rows = []
    for t in range(T):
        cur, prv = t & 1, (t & 1) ^ 1
        prev = (state[prv], logw[prv], out_max[t - 1:t], tile_sums, out_q[t - 1:t]) if t else (None,) * 5
        info = ""
        if t:
            ts = tile_sums.cpu().numpy().astype(np.float64)
            slots = ts / ts.sum() * n
            info = f"heavy {int((slots > 4088).sum()):4d}  idle {int((ts == 0).sum()):4d}  max-share {slots.max() / n:.3f}"
        e0, e1, e2 = HipEvent(), HipEvent(), HipEvent()
        e0.record(ops.stream())
        ops.smc_lgssm_step_a(cfg, model, t, float(yc[t]), *prev, state[cur], logw[cur], mp, None)
        e1.record(ops.stream())
        ops.smc_step_b(cfg, logw[cur], mp, out_max[t:t + 1], tile_sums)
        e2.record(ops.stream())
        torch.cuda.synchronize()
        rows.append((t, e0.elapsed_ms(e1) * 1e3, e1.elapsed_ms(e2) * 1e3, info))
    if rep:
        for t, a, b, info in rows:
            print(f"step {t:2d}: step_a {a:8.1f} us  step_b {b:6.1f} us   {info}")
