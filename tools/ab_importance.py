"""A/B of the importance-kernel forms in ONE process, interleaved rounds (one MI355X):
  python tools/ab_importance.py [rounds]
forms: one | pair | quad particles per lane  x  exact | fast math  x  1 | 8 passes per launch.
Prints the median / min time per 1e6-particle pass of every variant and, for the fast plans, the largest
deviation from the exact plan (same counters)."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genjax-chi_amd"))
import torch  # noqa: E402

from genjax._amd import workloads as W  # noqa: E402
from genjax._amd.ops import HipEvent  # noqa: E402
from genjax._amd.runtime import load_hip_ops  # noqa: E402

ops = load_hip_ops()
N = 1_000_000
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
variants = []
for form in ("pair", "quad", "one"):
    for fast in (False, True):
        for L in (1, 8):
            os.environ["GJX_JIT_FORM"] = form
            wl = W.Gaussian10(ops, 1, seed=0, n_local=N, fast_math=fast)
            prep = wl.prepare(fold_batch=L, passes=L)
            prep.launch_passes(0, L)  # builds the kernel of this form now
            torch.cuda.synchronize()
            variants.append((f"{form:4s} {'fast ' if fast else 'exact'} L={L}", form, wl, prep, L))
times = {v[0]: [] for v in variants}
for _ in range(20):  # clock ramp
    for name, form, wl, prep, L in variants:
        os.environ["GJX_JIT_FORM"] = form
        prep.launch_passes(0, L)
torch.cuda.synchronize()
for r in range(rounds):
    for name, form, wl, prep, L in variants:
        os.environ["GJX_JIT_FORM"] = form
        reps = 16 if L == 1 else 4
        for _ in range(3):
            prep.launch_passes(0, L)
        a, b = HipEvent(), HipEvent()
        a.record(ops.stream())
        for _ in range(reps):
            prep.launch_passes(0, L)
        b.record(ops.stream())
        times[name].append(a.elapsed_ms(b) * 1e3 / (reps * L))
for name, ts in times.items():
    print(f"{name}: median {statistics.median(ts):7.2f} us/pass   min {min(ts):7.2f}   ({48e6 / (statistics.median(ts) * 1e-6) / 8e12:.3f} of 8 TB/s)")
# deviations of the fast plans
os.environ["GJX_JIT_FORM"] = "pair"
ex = W.gaussian10_importance(ops, 1, seed=5, n=N)
for form in ("pair", "quad", "one"):
    os.environ["GJX_JIT_FORM"] = form
    fa = W.gaussian10_importance(ops, 1, seed=5, n=N, fast_math=True)
    lw = ((fa["logw"].double() - ex["logw"].double()).abs() / ex["logw"].double().abs()).max().item()
    va = max((a.double() - b.double()).abs().max().item() for a, b in zip(fa["values"], ex["values"]))
    print(f"fast {form}: max rel dev logw {lw:.3g}, max abs dev values {va:.3g}, log Z {fa['log_z_rows']:.7f} vs {ex['log_z_rows']:.7f}")
